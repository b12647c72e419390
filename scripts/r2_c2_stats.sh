# kernel-level breakdown of one C2 build (rocprofv3 --kernel-trace --stats) + the bench line of the same run
export TMPDIR=/tmp
out=gpurun_out/r2_c2
rm -rf $out && mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 1 --warmup 1 --no-cpu --alt-data none > $out/stats.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv") $out/kernel_stats.csv
grep -a '"metric"' $out/stats.log | tail -1 > $out/bench_under_rocprof.json
find $out -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r2_c2/kernel_stats.csv")))
for r in rows[:22]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(6), f'{float(r["TotalDurationNs"])/1e6:9.2f} ms', f'{float(r["AverageNs"])/1e3:9.1f} us')
PY
