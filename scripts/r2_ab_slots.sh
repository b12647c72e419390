# same-box sweep: resident walk waves (HNY_WALK_SLOTS) x XCD tile on C2
mkdir -p gpurun_out
A="--no-cpu --no-recall --queries 0 --alt-data none --steps 2 --warmup 1"
for cfg in "4096 512" "3072 384" "3584 448" "4096 256" "2048 256" "4096 512"; do
  set -- $cfg
  HNY_WALK_SLOTS=$1 HNY_XCD_TILE=$2 timeout -k 10 300 python bench.py $A > gpurun_out/r2_slots_$1_$2.json 2> gpurun_out/r2_slots_$1_$2.err || exit 1
  python - <<PY
import json
j=json.loads(open("gpurun_out/r2_slots_$1_$2.json").read().strip().splitlines()[-1]); b=j["build"]
print("slots $1 tile $2", "ms", j["ms_per_step"], "walk", b["t_walk_kernels_s"])
PY
done
