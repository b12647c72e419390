#!/bin/bash
# round-3 starting point on one box: C2 (default line), C5, C4
export TMPDIR=/tmp
out=gpurun_out/r3_base
rm -rf $out && mkdir -p $out
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --out $out/c2.json > $out/c2.log 2>&1; echo "c2 rc=$?"
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --items 5000000 --dim 1024 --metric hamming --ef 64 --out $out/c5.json > $out/c5.log 2>&1; echo "c5 rc=$?"
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu --items 10000000 --dim 128 --out $out/c4.json > $out/c4.log 2>&1; echo "c4 rc=$?"
python3 - <<'PY'
import json
for n in ("c2","c5","c4"):
    try:
        j=json.load(open(f"gpurun_out/r3_base/{n}.json"))
        b=j["build"]; r=j["roofline"]
        print(n, "value", j["value"], "ms", j["ms_per_step"], "walk", b["t_walk_kernels_s"], "prune", b["t_prune_kernels_s"], "sort", b["t_sort_kernels_s"], "apply", b["t_apply_kernels_s"], "export", b["t_export_s"], "frac", r["frac"], "alg", r["frac_algorithmic"], "recall", j.get("recall_at_10"), "cpu", j.get("cpu_baseline",{}).get("value"))
    except Exception as e: print(n, "failed", e)
PY
