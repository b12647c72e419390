# same-box A/B of the speculative neighbour-list fetch (HNY_SPEC_LIST=0/1; the switch lived in the experiment only, DESIGN.md 5 "Short rows"): C5, 4M x 128, C2 walk seconds
mkdir -p gpurun_out
A="--no-cpu --no-recall --queries 0 --alt-data none --steps 2 --warmup 1"
for cfg in "c5 --items 5000000 --dim 1024 --metric hamming --ef 64" "c4s --items 4000000 --dim 128" "c2"; do
  set -- $cfg; name=$1; shift
  for sp in 0 1 0 1; do
    HNY_SPEC_LIST=$sp timeout -k 10 300 python bench.py $A "$@" > gpurun_out/r2_ab_${name}_$sp.json 2> gpurun_out/r2_ab_${name}_$sp.err || exit 1
    python - <<PY
import json
j=json.loads(open("gpurun_out/r2_ab_${name}_$sp.json").read().strip().splitlines()[-1]); b=j["build"]
print("$name spec=$sp", "ms", j["ms_per_step"], "walk", b["t_walk_kernels_s"])
PY
  done
done
