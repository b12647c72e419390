# tile sweep of the XCD-tiled work queue on the HBM-bound configs (C2, C3), same box
mkdir -p gpurun_out
A="--no-cpu --no-recall --queries 0 --alt-data none --steps 2 --warmup 1"
for cfg in "c2" "c3 --metric euclidean --M 32 --ef 200"; do
  set -- $cfg; name=$1; shift
  for t in 0 256 512 1024 2048 4096 8192 0; do
    HNY_XCD_TILE=$t timeout -k 10 300 python bench.py $A "$@" > gpurun_out/r2_xcd2_${name}_$t.json 2> gpurun_out/r2_xcd2_${name}_$t.err || exit 1
    python - <<PY
import json
j=json.loads(open("gpurun_out/r2_xcd2_${name}_$t.json").read().strip().splitlines()[-1]); b=j["build"]
print("$name tile=$t", "ms", j["ms_per_step"], "walk", b["t_walk_kernels_s"])
PY
  done
done
