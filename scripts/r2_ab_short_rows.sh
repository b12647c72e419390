set -x
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sub_wave" > gpurun_out/r2_c_sub.log 2>&1 || { tail -30 gpurun_out/r2_c_sub.log; exit 1; }
tail -3 gpurun_out/r2_c_sub.log
HNY_SUB=1 HNY_DEBUG_SUB=1 timeout -k 10 500 python bench.py --no-cpu --steps 2 --warmup 1 --items 5000000 --dim 1024 --metric hamming --ef 64 > gpurun_out/r2_c_c5_sub.json 2> gpurun_out/r2_c_c5_sub.err || { tail -5 gpurun_out/r2_c_c5_sub.err; exit 1; }
HNY_SUB=0 timeout -k 10 500 python bench.py --no-cpu --no-recall --steps 2 --warmup 1 --items 5000000 --dim 1024 --metric hamming --ef 64 > gpurun_out/r2_c_c5_nosub.json 2> gpurun_out/r2_c_c5_nosub.err
HNY_SUB=1 HNY_DEBUG_SUB=1 timeout -k 10 500 python bench.py --no-cpu --steps 2 --warmup 1 --items 4000000 --dim 128 > gpurun_out/r2_c_c4s_sub.json 2> gpurun_out/r2_c_c4s_sub.err
HNY_SUB=0 timeout -k 10 500 python bench.py --no-cpu --no-recall --steps 2 --warmup 1 --items 4000000 --dim 128 > gpurun_out/r2_c_c4s_nosub.json 2> gpurun_out/r2_c_c4s_nosub.err
python - <<'PY'
import json
for f in ["c5_sub","c5_nosub","c4s_sub","c4s_nosub"]:
    try:
        j=json.loads(open(f"gpurun_out/r2_c_{f}.json").read().strip().splitlines()[-1])
        print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"], j["build"]["t_walk_kernels_s"], j["build"]["t_prune_kernels_s"], j["build"]["t_sort_kernels_s"], j["build"]["t_apply_kernels_s"], j.get("recall_at_10"), j["build"]["sub_wave_walks"], j["build"]["sub_wave_handed_over"])
    except Exception as e:
        print(f, "ERR", e)
PY
