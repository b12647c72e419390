#!/bin/bash
# the bench lines committed under profiles/r04_*: run on one MI355X box after the PMC passes (scripts/r3_pmc.sh),
# so that every line finds its workload's counted traffic
export TMPDIR=/tmp
out=gpurun_out/r4_final
rm -rf $out && mkdir -p $out
run() { name=$1; shift; timeout -k 10 600 python bench.py "$@" --out $out/$name.json > $out/$name.log 2>&1; echo "$name rc=$?"; }
run c2_bench_default_run
run c2_bench_overlap_data --steps 3 --warmup 1 --no-cpu --data overlap --alt-data none
run c3_1Mx768_euclidean_M32_ef200_bench --steps 3 --warmup 1 --no-cpu --metric euclidean --M 32 --ef 200
run c5_5Mx1024bit_hamming_bench --steps 3 --warmup 1 --no-cpu --items 5000000 --dim 1024 --metric hamming --ef 64
run c4_10Mx128_cosine_bench --steps 2 --warmup 1 --no-cpu --items 10000000 --dim 128
run bench_gpus2_gloo_shared_gpu --gpus 2 --backend gloo --items 200000 --steps 2 --warmup 1 --no-cpu
HNY_MGPU_SHIM=1 HNY_MGPU_VERIFY=1 run bench_gpus2_native_shim --gpus 2 --native --items 200000 --steps 2 --warmup 1 --no-cpu
run bench_gpus1_native_rccl --gpus 1 --native --items 200000 --steps 2 --warmup 1 --no-cpu
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_final/*.json")):
    j = json.load(open(f)); b = j["build"]; r = j["roofline"]
    print(f.split("/")[-1][:-5], "| value", j["value"], "ms", j["ms_per_step"], "| walk", b["t_walk_kernels_s"], "prune", b["t_prune_kernels_s"],
          "sort", b["t_sort_kernels_s"], "apply", b["t_apply_kernels_s"], "export", b["t_export_s"], "| frac", r["frac"], "alg", r["frac_algorithmic"],
          "reuse", r.get("reuse"), "wasted", r.get("wasted_traffic_ratio"), "stale", r.get("traffic_stale"), "| recall", j.get("recall_at_10"),
          "cpu", (j.get("cpu_baseline") or {}).get("value"), "alt", j.get("value_alt"), "issue", (j.get("roofline_issue") or {}).get("frac"), "ranks", j.get("ranks_seen"), flush=True)
PY
