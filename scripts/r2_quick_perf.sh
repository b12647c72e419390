# walk seconds / vec per s of the configs that matter: C5, C4-like (4M x 128), C2, C3
mkdir -p gpurun_out
tag=${1:-x}
A="--no-cpu --no-recall --alt-data none --steps 2 --warmup 1"
timeout -k 10 300 python bench.py $A --items 5000000 --dim 1024 --metric hamming --ef 64 > gpurun_out/r2_q_${tag}_c5.json 2> gpurun_out/r2_q_${tag}_c5.err
timeout -k 10 300 python bench.py $A --items 4000000 --dim 128 > gpurun_out/r2_q_${tag}_c4s.json 2> gpurun_out/r2_q_${tag}_c4s.err
timeout -k 10 300 python bench.py $A > gpurun_out/r2_q_${tag}_c2.json 2> gpurun_out/r2_q_${tag}_c2.err
timeout -k 10 300 python bench.py $A --metric euclidean --M 32 --ef 200 > gpurun_out/r2_q_${tag}_c3.json 2> gpurun_out/r2_q_${tag}_c3.err
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_q_${tag}_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); b=j["build"]
        print(f.split("_")[-1][:-5].ljust(4), int(j["value"]), "ms", j["ms_per_step"], "walk", b["t_walk_kernels_s"], "prune", b["t_prune_kernels_s"], "sort", b["t_sort_kernels_s"], "apply", b["t_apply_kernels_s"], "export", b["t_export_s"], "frac", j["roofline"]["frac"])
    except Exception as e:
        print(f, "ERR", e)
PY
