#!/bin/bash
# Round 5: HBM-side traffic AND L2 hit rate of one bench workload: rocprofv3 --kernel-trace --stats, then FETCH_SIZE,
# WRITE_SIZE and TCC_HIT_sum + TCC_MISS_sum in separate --pmc passes, summarised per kernel family (scripts/pmc_summary.py).
#   gpurun --timeout 1100 -- 'NAME=c2 KEY=1000000x768_cosine_M16_ef100_clustered ARGS="" bash scripts/r5_pmc.sh'
# writes gpurun_out/r5_pmc/{r05_pmc_hbm_<KEY>.json, r05_<NAME>_kernel_stats.csv, r05_<NAME>_bench_under_rocprof.json}
# (-> profiles/: bench.py looks the workload up by KEY and carries l2_hit_rate in `roofline`)
export TMPDIR=/tmp
out=gpurun_out/r5_pmc
mkdir -p $out
NAME=${NAME:-c2}
KEY=${KEY:-1000000x768_cosine_M16_ef100_clustered}
ARGS=${ARGS:-""}
COMMON="--no-cpu --no-recall --queries 0 --steps 1 --alt-data none"
rm -rf $out/${NAME}_s $out/${NAME}_f $out/${NAME}_w $out/${NAME}_h
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${NAME}_s -- python3 bench.py $COMMON --warmup 1 $ARGS > $out/${NAME}_s.log 2>&1 &&
cp $(find $out/${NAME}_s -name "*kernel_stats.csv") $out/r05_${NAME}_kernel_stats.csv &&
grep -a '"metric"' $out/${NAME}_s.log | tail -1 > $out/r05_${NAME}_bench_under_rocprof.json &&
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${NAME}_f -- python3 bench.py $COMMON --warmup 0 $ARGS > $out/${NAME}_f.log 2>&1 &&
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${NAME}_w -- python3 bench.py $COMMON --warmup 0 $ARGS > $out/${NAME}_w.log 2>&1 &&
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/${NAME}_h -- python3 bench.py $COMMON --warmup 0 $ARGS > $out/${NAME}_h.log 2>&1 &&
python3 scripts/pmc_summary.py $(find $out/${NAME}_f -name "*counter_collection.csv") $(find $out/${NAME}_w -name "*counter_collection.csv") $out/r05_pmc_hbm_${KEY}.json $(find $out/${NAME}_h -name "*counter_collection.csv") > $out/r05_${NAME}_pmc.txt 2>&1
rc=$?
rm -rf $out/${NAME}_s $out/${NAME}_f $out/${NAME}_w $out/${NAME}_h
echo "== $NAME (rc $rc)"; cat $out/r05_${NAME}_pmc.txt; head -8 $out/r05_${NAME}_kernel_stats.csv 2>/dev/null
exit $rc
