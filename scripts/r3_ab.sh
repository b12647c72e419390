#!/bin/bash
# same-box A/B, alternating runs: A = hannoy_amd/libhannoy_amd.so with ENV_A, B = ${LIB_B:-the same library} with ENV_B
#   gpurun -- 'CFGS="c5 c4s" ENV_B="HNY_WALK_SLOTS=6144" bash scripts/r3_ab.sh'
#   gpurun -- 'CFGS="c5" LIB_B=hannoy_amd/libhannoy_amd_ab.so bash scripts/r3_ab.sh'   (python -m hannoy_amd.buildlib --out ...)
mkdir -p gpurun_out/r3_ab
A="--no-cpu --no-recall --queries 0 --steps ${STEPS:-3} --warmup 1"
CFGS=${CFGS:-"c5 c4s"}
ORDER=${ORDER:-"A B A B"}
for name in $CFGS; do
  case $name in
    c5) args="--items 5000000 --dim 1024 --metric hamming --ef 64";;
    c5s) args="--items 2000000 --dim 1024 --metric hamming --ef 64";;
    c4) args="--items 10000000 --dim 128";;
    c4s) args="--items 4000000 --dim 128";;
    c2) args="";;
    c3) args="--metric euclidean --M 32 --ef 200";;
  esac
  i=0
  for v in $ORDER; do
    i=$((i+1))
    if [ $v = B ]; then envs="$ENV_B"; lib=${LIB_B:+$PWD/$LIB_B}; else envs="$ENV_A"; lib=""; fi
    env $envs ${lib:+HNY_LIB=$lib} timeout -k 10 400 python bench.py $A $args --out gpurun_out/r3_ab/${name}_${v}_$i.json > gpurun_out/r3_ab/${name}_${v}_$i.log 2>&1 || { echo "$name $v failed"; tail -3 gpurun_out/r3_ab/${name}_${v}_$i.log; exit 1; }
    python3 - <<PY
import json
j=json.load(open("gpurun_out/r3_ab/${name}_${v}_$i.json")); b=j["build"]
print("$name $v", "ms", j["ms_per_step"], "walk", b["t_walk_kernels_s"], "prune", b["t_prune_kernels_s"], "sort", b["t_sort_kernels_s"], "apply", b["t_apply_kernels_s"], "export", b["t_export_s"], "evals", b["evals_walk"], flush=True)
PY
  done
done
