# resident walk waves (one wave per query) on C5: latency-bound or throughput-bound?
mkdir -p gpurun_out
for sl in 1280 2560 3840 5120; do
  HNY_WALK_SLOTS=$sl timeout -k 10 300 python bench.py --no-cpu --no-recall --alt-data none --steps 1 --warmup 1 --items 5000000 --dim 1024 --metric hamming --ef 64 > gpurun_out/r2_sl_c5_$sl.json 2> gpurun_out/r2_sl_c5_$sl.err
  HNY_WALK_SLOTS=$sl timeout -k 10 300 python bench.py --no-cpu --no-recall --alt-data none --steps 1 --warmup 1 --items 4000000 --dim 128 > gpurun_out/r2_sl_c4s_$sl.json 2> gpurun_out/r2_sl_c4s_$sl.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_sl_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); b=j["build"]
        print(f, j["value"], "walk", b["t_walk_kernels_s"], "frac", j["roofline"]["frac"])
    except Exception as e:
        print(f, "ERR", e)
PY
