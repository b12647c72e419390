# same-box A/B of the XCD-tiled work queue in the Reader's batched search (C2 index, 32 768 queries, ef_search 100)
mkdir -p gpurun_out
for t in 0 512 0 512; do
  HNY_XCD_TILE=$t timeout -k 10 300 python bench.py --no-cpu --alt-data none --steps 1 --warmup 0 > gpurun_out/r2_sx_$t.json 2> gpurun_out/r2_sx_$t.err || exit 1
  python - <<PY
import json
j=json.loads(open("gpurun_out/r2_sx_$t.json").read().strip().splitlines()[-1])
print("tile $t", "build ms", j["ms_per_step"], "search", j["search"], "recall", j["recall_at_10"])
PY
done
