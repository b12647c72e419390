#!/bin/bash
# HBM-side traffic of one bench workload: rocprofv3 --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in
# separate --pmc passes (TCC slots: FETCH 3 + WRITE 2 > 4), summarised per kernel family by scripts/pmc_summary.py.
#   gpurun --timeout 1100 -- 'NAME=c5 ARGS="--items 5000000 --dim 1024 --metric hamming --ef 64" bash scripts/r3_pmc.sh'
# writes gpurun_out/r3_pmc/<NAME>_{pmc_hbm.json,kernel_stats.csv,bench.json}; copy them to profiles/ as
#   r03_pmc_hbm_<n>x<dim>_<metric>_M<M>_ef<ef>_<data>.json (bench.py looks the workload up by that name),
#   r03_<NAME>_kernel_stats.csv, r03_<NAME>_bench_under_rocprof.json
export TMPDIR=/tmp
out=gpurun_out/r3_pmc
mkdir -p $out
NAME=${NAME:-c2}
ARGS=${ARGS:-""}
COMMON="--no-cpu --no-recall --queries 0 --steps 1 --alt-data none"
rm -rf $out/${NAME}_s $out/${NAME}_f $out/${NAME}_w
if [ -z "$SKIP_STATS" ]; then
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${NAME}_s -- python3 bench.py $COMMON --warmup 1 $ARGS > $out/${NAME}_s.log 2>&1
cp $(find $out/${NAME}_s -name "*kernel_stats.csv") $out/${NAME}_kernel_stats.csv
grep -a '"metric"' $out/${NAME}_s.log | tail -1 > $out/${NAME}_bench.json
fi
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${NAME}_f -- python3 bench.py $COMMON --warmup 0 $ARGS > $out/${NAME}_f.log 2>&1 &&
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${NAME}_w -- python3 bench.py $COMMON --warmup 0 $ARGS > $out/${NAME}_w.log 2>&1 &&
python3 scripts/pmc_summary.py $(find $out/${NAME}_f -name "*counter_collection.csv") $(find $out/${NAME}_w -name "*counter_collection.csv") $out/${NAME}_pmc_hbm.json > $out/${NAME}_pmc.txt 2>&1
rm -rf $out/${NAME}_s $out/${NAME}_f $out/${NAME}_w
echo "== $NAME"; cat $out/${NAME}_pmc.txt; head -8 $out/${NAME}_kernel_stats.csv 2>/dev/null
