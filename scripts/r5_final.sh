#!/bin/bash
# the bench lines committed under profiles/r05_*: run on one MI355X box after the PMC passes (scripts/r5_profiles.sh),
# so that every line finds its workload's counted traffic and L2 hit rate.  C4 / C5 on `overlap` data: the distribution
# their recall parity was measured on (recall@10 >= 0.9 at ef_search 100).
export TMPDIR=/tmp
out=gpurun_out/r5_final
rm -rf $out && mkdir -p $out
run() { name=$1; shift; timeout -k 10 900 python bench.py "$@" --out $out/$name.json > $out/$name.log 2>&1; echo "$name rc=$?"; }
case "${1:-all}" in
all|c2)
run r05_c2_bench_default_run ;;&
all|rest)
run r05_c3_1Mx768_euclidean_M32_ef200_bench --steps 3 --warmup 1 --no-cpu --metric euclidean --M 32 --ef 200
run r05_c5_5Mx1024bit_hamming_bench --steps 3 --warmup 1 --no-cpu --items 5000000 --dim 1024 --metric hamming --ef 64 --data overlap --queries 5000
run r05_c4_10Mx128_cosine_bench --steps 2 --warmup 1 --no-cpu --items 10000000 --dim 128 --data overlap --queries 5000
run r05_bench_gpus2_gloo_shared_gpu --gpus 2 --backend gloo --items 200000 --steps 2 --warmup 1 --no-cpu
HNY_MGPU_SHIM=1 HNY_MGPU_VERIFY=1 run r05_bench_gpus2_native_shim --gpus 2 --native --items 200000 --steps 2 --warmup 1 --no-cpu
run r05_bench_gpus1_native_rccl --gpus 1 --native --items 200000 --steps 2 --warmup 1 --no-cpu ;;
esac
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5_final/*.json")):
    j = json.load(open(f)); b = j["build"]; r = j["roofline"] or {}
    print(f.split("/")[-1][:-5], "| value", j["value"], "ms", j["ms_per_step"], "first", j.get("first_build_ms"), "| walk", b["t_walk_kernels_s"], "prune", b["t_prune_kernels_s"],
          "sort", b["t_sort_kernels_s"], "apply", b["t_apply_kernels_s"], "export", b["t_export_s"], "| frac", r.get("frac"), "alg", r.get("frac_algorithmic"),
          "l2", r.get("l2_hit_rate"), "stale", r.get("traffic_stale"), "| recall", j.get("recall_at_10"),
          "cpu", (j.get("cpu_baseline") or {}).get("value"), (j.get("cpu_baseline") or {}).get("size"), "alt", j.get("value_alt"),
          "issue", (j.get("roofline_issue") or {}).get("frac"), "ranks", j.get("ranks_seen"), flush=True)
PY
