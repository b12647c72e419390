# bench lines of the other BASELINE configs at FULL size on one MI355X (profiles/r02_c3_*, r02_c4_*, r02_c5_*)
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --no-cpu --alt-data none --metric euclidean --M 32 --ef 200 > gpurun_out/r02_c3.json 2> gpurun_out/r02_c3.err
timeout -k 10 500 python bench.py --no-cpu --alt-data none --items 10000000 --dim 128 --steps 2 > gpurun_out/r02_c4.json 2> gpurun_out/r02_c4.err
timeout -k 10 400 python bench.py --no-cpu --alt-data none --items 5000000 --dim 1024 --metric hamming --ef 64 > gpurun_out/r02_c5.json 2> gpurun_out/r02_c5.err
timeout -k 10 400 python bench.py --no-cpu --alt-data none --data uniform --steps 1 > gpurun_out/r02_c2_uniform.json 2> gpurun_out/r02_c2_uniform.err
python - <<'PY'
import json
for f in ["c3","c4","c5","c2_uniform"]:
    try:
        j=json.loads(open(f"gpurun_out/r02_{f}.json").read().strip().splitlines()[-1]); b=j["build"]
        print(f, int(j["value"]), "ms", j["ms_per_step"], "walk", b["t_walk_kernels_s"], "prune", b["t_prune_kernels_s"], "frac", j["roofline"]["frac"], "recall", j.get("recall_at_10"))
    except Exception as e:
        print(f, "ERR", e)
PY
