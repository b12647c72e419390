#!/bin/bash
# same-box comparison of several variants (library builds and/or environment knobs), ROUNDS alternating passes:
#   VARIANTS='base||;nb512|HNY_VIS_BUCKETS=512|;w7|HNY_WALK_SLOTS=7168|hannoy_amd/libhannoy_amd_w7.so' CFGS="c5 c4s" bash scripts/r4_variants.sh
# each variant: name|environment assignments|library (empty = the in-tree build)
mkdir -p gpurun_out/r4_var
A="--no-cpu --no-recall --queries 0 --alt-data none --steps ${STEPS:-2} --warmup 1"
CFGS=${CFGS:-"c5"}
ROUNDS=${ROUNDS:-2}
IFS=';' read -ra VARS <<< "$VARIANTS"
for name in $CFGS; do
  case $name in
    c5) args="--items 5000000 --dim 1024 --metric hamming --ef 64";;
    c5s) args="--items 2000000 --dim 1024 --metric hamming --ef 64";;
    c4) args="--items 10000000 --dim 128";;
    c4s) args="--items 4000000 --dim 128";;
    c2) args="";;
    c3) args="--metric euclidean --M 32 --ef 200";;
  esac
  for r in $(seq 1 $ROUNDS); do
    for v in "${VARS[@]}"; do
      IFS='|' read -r vn envs lib <<< "$v"
      f=gpurun_out/r4_var/${name}_${vn}_$r
      env $envs ${lib:+HNY_LIB=$PWD/$lib} timeout -k 10 400 python bench.py $A $args --out $f.json > $f.log 2>&1 || { echo "$name $vn failed"; tail -5 $f.log; exit 1; }
      python3 - <<PY
import json
j=json.load(open("$f.json")); b=j["build"]
print("$name %-12s" % "$vn", "ms", j["ms_per_step"], "walk", b["t_walk_kernels_s"], "prune", b["t_prune_kernels_s"], "sort", b["t_sort_kernels_s"], "apply", b["t_apply_kernels_s"], "evals", b["evals_walk"], "links", b.get("n_links_added"), flush=True)
PY
    done
  done
done
