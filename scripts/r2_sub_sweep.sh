# sweep of the sub-wave walk's resident blocks (4 queries each) x visited-table slots on C5 / C4-like data
mkdir -p gpurun_out
for bl in 1024 1536 2048 3072; do
 for vs in 2048 4096; do
  HNY_SUB=1 HNY_SUB_BLOCKS=$bl HNY_SUB_VSLOTS=$vs timeout -k 10 300 python bench.py --no-cpu --no-recall --steps 1 --warmup 1 --items 5000000 --dim 1024 --metric hamming --ef 64 > gpurun_out/r2_sw_c5_${bl}_$vs.json 2> gpurun_out/r2_sw_c5_${bl}_$vs.err
  HNY_SUB=1 HNY_SUB_BLOCKS=$bl HNY_SUB_VSLOTS=$vs timeout -k 10 300 python bench.py --no-cpu --no-recall --steps 1 --warmup 1 --items 4000000 --dim 128 > gpurun_out/r2_sw_c4s_${bl}_$vs.json 2> gpurun_out/r2_sw_c4s_${bl}_$vs.err
 done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_sw_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); b=j["build"]
        print(f, j["value"], "walk", b["t_walk_kernels_s"], "frac", j["roofline"]["frac"], "sub", b["sub_wave_walks"], "handed", b["sub_wave_handed_over"])
    except Exception as e:
        print(f, "ERR", e)
PY
