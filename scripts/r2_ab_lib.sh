# same-box A/B of two builds of the library: A = hannoy_amd/libhannoy_amd.so, B = hannoy_amd/libhannoy_amd_ab.so
# (python -m hannoy_amd.buildlib --out hannoy_amd/libhannoy_amd_ab.so on a modified tree); alternating runs
mkdir -p gpurun_out
A="--no-cpu --no-recall --queries 0 --alt-data none --steps 2 --warmup 1"
CFGS=${CFGS:-"c5 c4s"}
for name in $CFGS; do
  case $name in
    c5) args="--items 5000000 --dim 1024 --metric hamming --ef 64";;
    c4s) args="--items 4000000 --dim 128";;
    c2) args="";;
    c3) args="--metric euclidean --M 32 --ef 200";;
  esac
  for v in A B A B A B; do
    if [ $v = B ]; then export HNY_LIB=$PWD/hannoy_amd/libhannoy_amd_ab.so; else unset HNY_LIB; fi
    timeout -k 10 300 python bench.py $A $args > gpurun_out/r2_ablib_${name}_$v.json 2> gpurun_out/r2_ablib_${name}_$v.err || exit 1
    python - <<PY
import json
j=json.loads(open("gpurun_out/r2_ablib_${name}_$v.json").read().strip().splitlines()[-1]); b=j["build"]
print("$name $v", "ms", j["ms_per_step"], "walk", b["t_walk_kernels_s"], "prune", b["t_prune_kernels_s"])
PY
  done
done
