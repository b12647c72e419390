#!/bin/bash
# Round 5: where do the walk kernel's cycles go, by instruction — rocprofv3 PC sampling of one C5-shape build
# (library built with -gline-tables-only: HNY_CFLAGS=-gline-tables-only python -m hannoy_amd.buildlib --out hannoy_amd/libhannoy_amd_g.so).
#   gpurun --timeout 900 -- 'bash scripts/r5_pcsamp.sh'
# -> gpurun_out/r5_pcs/{avail.txt, summary_<method>.txt}   (scripts/r5_pcsamp_summary.py folds the samples by source line)
export TMPDIR=/tmp
out=gpurun_out/r5_pcs
mkdir -p $out
ARGS=${ARGS:-"--items 5000000 --dim 1024 --metric hamming --ef 64 --data overlap"}
LIBG=$PWD/hannoy_amd/libhannoy_amd_g.so
[ -f $LIBG ] && export HNY_LIB=$LIBG
rocprofv3 -L > $out/avail_full.txt 2>&1
grep -i -A12 "pc.sampl" $out/avail_full.txt | head -60 > $out/avail.txt
run() { # method unit interval
  rm -rf $out/run_$1
  timeout -k 10 400 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 \
     --kernel-trace --output-format csv -d $out/run_$1 -- python3 bench.py --no-cpu --no-recall --queries 0 --steps 1 --warmup 0 --alt-data none $ARGS > $out/run_$1.log 2>&1
  rc=$?
  echo "== $1 rc $rc"; tail -3 $out/run_$1.log | cut -c1-300
  f=$(find $out/run_$1 -name "*pc_sampling*.csv" | head -1)
  [ -n "$f" ] && python3 scripts/r5_pcsamp_summary.py $f > $out/summary_$1.txt 2>&1 && head -60 $out/summary_$1.txt
  rm -rf $out/run_$1
  return $rc
}
run stochastic cycles ${INTERVAL:-1048576} || run host_trap time ${INTERVAL_US:-256}
