mkdir -p gpurun_out/r3_sorted
A="--no-cpu --no-recall --queries 0 --steps 3 --warmup 1"
for cfg in c5 c4s c2; do
  case $cfg in c5) args="--items 5000000 --dim 1024 --metric hamming --ef 64";; c4s) args="--items 4000000 --dim 128";; c2) args="";; esac
  for v in plain sorted plain sorted; do
    extra=""; [ $v = sorted ] && extra="--sort-by-cluster"
    timeout -k 10 400 python bench.py $A $args $extra --out gpurun_out/r3_sorted/${cfg}_$v.json > gpurun_out/r3_sorted/${cfg}_$v.log 2>&1 || { echo fail; tail -3 gpurun_out/r3_sorted/${cfg}_$v.log; }
    python3 -c "
import json; j=json.load(open('gpurun_out/r3_sorted/${cfg}_$v.json')); b=j['build']
print('$cfg $v ms', j['ms_per_step'], 'walk', b['t_walk_kernels_s'], 'prune', b['t_prune_kernels_s'], 'apply', b['t_apply_kernels_s'], 'evals', b['evals_walk'])"
  done
done
