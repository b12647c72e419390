#!/bin/bash
# rocprofv3 passes behind profiles/r02_c2_*: run on the MI355X box from the repo root
#   gpurun --timeout 1100 -- 'bash scripts/profile_c2.sh'
export TMPDIR=/tmp
out=gpurun_out/r02g
rm -rf $out && mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 1 --warmup 1 --no-cpu --alt-data none > $out/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-recall --queries 0 --alt-data none > $out/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-recall --queries 0 --alt-data none > $out/write.log 2>&1
python3 scripts/pmc_summary.py $(find $out/fetch -name "*counter_collection.csv") $(find $out/write -name "*counter_collection.csv") $out/pmc_hbm.json > $out/pmc_summary.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv") $out/kernel_stats.csv
grep -a '"metric"' $out/stats.log | tail -1 > $out/bench_under_rocprof.json
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -size +20M -delete
du -sh $out; cat $out/pmc_summary.log
