# same-box A/B of the XCD-tiled work queue of k_walk (HNY_XCD_TILE = members per tile, 0 = one counter)
mkdir -p gpurun_out
A="--no-cpu --no-recall --queries 0 --alt-data none --steps 2 --warmup 1"
for cfg in "c2" "c5 --items 5000000 --dim 1024 --metric hamming --ef 64" "c4s --items 4000000 --dim 128"; do
  set -- $cfg; name=$1; shift
  for t in 0 64 0 16 32 128 512; do
    HNY_XCD_TILE=$t timeout -k 10 300 python bench.py $A "$@" > gpurun_out/r2_xcd_${name}_$t.json 2> gpurun_out/r2_xcd_${name}_$t.err || exit 1
    python - <<PY
import json
j=json.loads(open("gpurun_out/r2_xcd_${name}_$t.json").read().strip().splitlines()[-1]); b=j["build"]
print("$name tile=$t", "ms", j["ms_per_step"], "walk", b["t_walk_kernels_s"])
PY
  done
done
