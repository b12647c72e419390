#!/bin/bash
# how the k_walk time of a level-0 batch splits into the descent launch and the layer-0 launch (kernel trace)
export TMPDIR=/tmp
out=gpurun_out/r3_split
rm -rf $out && mkdir -p $out
ARGS=${ARGS:-"--items 5000000 --dim 1024 --metric hamming --ef 64"}
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 bench.py --no-cpu --no-recall --queries 0 --steps 1 --warmup 0 --alt-data none $ARGS > $out/run.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r3_split/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_walk<" in r["Kernel_Name"] and "heap" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print("k_walk dispatches", len(d), "total ms", round(sum(d), 1))
big = [(i, x) for i, x in enumerate(d) if x > 5]
# the last batches: descent, layer-0 alternate
tail = d[-12:]
print("last 12 dispatches (ms):", [round(x, 1) for x in tail])
ev = sum(d[-12::2]); od = sum(d[-11::2])
print("of the last 12: even positions", round(ev, 1), "odd positions", round(od, 1))
PY
find $out -name "*.csv" -delete
