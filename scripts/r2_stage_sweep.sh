# C2 prune time against the LDS staging budget of the workgroup prune (HNY_STAGE_BYTES)
mkdir -p gpurun_out
A="--no-cpu --no-recall --alt-data none --steps 2 --warmup 1"
for sb in 12288 18432 24576 36864 49152; do
  HNY_STAGE_BYTES=$sb timeout -k 10 300 python bench.py $A > gpurun_out/r2_stage_$sb.json 2> gpurun_out/r2_stage_$sb.err || exit 1
  python - <<PY
import json
j=json.loads(open("gpurun_out/r2_stage_$sb.json").read().strip().splitlines()[-1]); b=j["build"]
print("stage_bytes", $sb, "ms", j["ms_per_step"], "prune", b["t_prune_kernels_s"], "apply", b["t_apply_kernels_s"])
PY
done
