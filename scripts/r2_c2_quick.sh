# C2 (and optionally C3) step time with the per-phase kernel times: bash scripts/r2_c2_quick.sh <tag> [c3]
mkdir -p gpurun_out
tag=${1:-x}
A="--no-cpu --no-recall --alt-data none --steps 3 --warmup 1"
timeout -k 10 300 python bench.py $A > gpurun_out/r2_c2q_${tag}_c2.json 2> gpurun_out/r2_c2q_${tag}_c2.err
if [ "$2" = "c3" ]; then
  timeout -k 10 300 python bench.py $A --metric euclidean --M 32 --ef 200 > gpurun_out/r2_c2q_${tag}_c3.json 2> gpurun_out/r2_c2q_${tag}_c3.err
fi
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_c2q_${tag}_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); b=j["build"]
        print(f.split("_")[-1][:-5].ljust(4), int(j["value"]), "ms", j["ms_per_step"], "walk", b["t_walk_kernels_s"], "prune", b["t_prune_kernels_s"], "sort", b["t_sort_kernels_s"], "apply", b["t_apply_kernels_s"], "export", b["t_export_s"], "frac", j["roofline"]["frac"])
    except Exception as e:
        print(f, "ERR", e)
PY
