# (historic: ran on the kernels of commit e1860fe, where HNY_VIS_SLOTS still reached the short-row kernels)
# one-wave-per-query walk with an LDS visited table in front of the HBM bitset, rows <= 1 KB (C5 / C4-like)
mkdir -p gpurun_out
for vs in 0 768 1280 1792 2304; do
  HNY_VIS_SLOTS=$vs timeout -k 10 300 python bench.py --no-cpu --no-recall --steps 1 --warmup 1 --items 5000000 --dim 1024 --metric hamming --ef 64 > gpurun_out/r2_vs_c5_$vs.json 2> gpurun_out/r2_vs_c5_$vs.err
  HNY_VIS_SLOTS=$vs timeout -k 10 300 python bench.py --no-cpu --no-recall --steps 1 --warmup 1 --items 4000000 --dim 128 > gpurun_out/r2_vs_c4s_$vs.json 2> gpurun_out/r2_vs_c4s_$vs.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_vs_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); b=j["build"]
        print(f, j["value"], "walk", b["t_walk_kernels_s"], "frac", j["roofline"]["frac"])
    except Exception as e:
        print(f, "ERR", e)
PY
