#!/bin/bash
# wave cycles per walk phase + event counts (library built with HNY_CFLAGS=-DHNY_PHASE_CLOCKS as LIB):
#   python -m hannoy_amd.buildlib --out hannoy_amd/libhannoy_amd_ph.so   (with HNY_CFLAGS=-DHNY_PHASE_CLOCKS)
#   gpurun -- 'LIB=hannoy_amd/libhannoy_amd_ph.so CFGS="c5 c4s" bash scripts/r4_phase_clocks.sh'
mkdir -p gpurun_out/r4_ph
A="--no-cpu --no-recall --queries 0 --alt-data none --steps 1 --warmup 0"
export HNY_LIB=$PWD/${LIB:-hannoy_amd/libhannoy_amd_ph.so}
for name in ${CFGS:-c5}; do
  case $name in
    c5) args="--items 5000000 --dim 1024 --metric hamming --ef 64";;
    c4) args="--items 10000000 --dim 128";;
    c4s) args="--items 4000000 --dim 128";;
    c2) args="";;
  esac
  timeout -k 10 400 python bench.py $A $args --out gpurun_out/r4_ph/$name.json > gpurun_out/r4_ph/$name.log 2> gpurun_out/r4_ph/$name.err || { tail -5 gpurun_out/r4_ph/$name.err; exit 1; }
  python3 - <<PY
import json, re
j = json.load(open("gpurun_out/r4_ph/$name.json")); b = j["build"]
err = open("gpurun_out/r4_ph/$name.err").read()
tot = [0] * 6; ext = [0] * 6
for m in re.finditer(r"walk wave cycles: pop (\d+) list\+visited (\d+) distances (\d+) insert (\d+) \| expansions (\d+) \| whole kernel (\d+)", err):
    for i in range(6): tot[i] += int(m.group(i + 1))
for m in re.finditer(r"short walk: visited wait (\d+) \| lanes asked (\d+) accepted (\d+) expansions that accepted (\d+) pool scans (\d+) expansions with nothing new (\d+)", err):
    for i in range(6): ext[i] += int(m.group(i + 1))
e = tot[4] or 1
print("== $name: walk", b["t_walk_kernels_s"], "s,", b["evals_walk"], "evaluations,", e, "expansions,", round(b["evals_walk"] / e, 2), "evaluations per expansion")
print("   cycles per expansion: pop %.0f  list fetch %.0f  visited %.0f  distances %.0f  insert %.0f  | whole kernel per expansion %.0f"
      % (tot[0] / e, tot[1] / e, ext[0] / e, tot[2] / e, tot[3] / e, tot[5] / e))
print("   per expansion: lanes that asked the visited set %.2f, accepted keys %.3f, expansions that accepted any %.3f, pool scans %.3f, nothing new %.3f"
      % (ext[1] / e, ext[2] / e, ext[3] / e, ext[4] / e, ext[5] / e))
PY
done
