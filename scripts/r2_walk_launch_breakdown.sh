# per-dispatch durations of one C2 build: how much of k_walk's time are the greedy-descent launches?
export TMPDIR=/tmp
out=gpurun_out/r2_trace
rm -rf $out && mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-recall --queries 0 --alt-data none > $out/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r2_trace/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the build: from the first k_walk to the last
idx = [i for i, r in enumerate(rows) if "k_walk" in r["Kernel_Name"]]
rows = rows[idx[0]:idx[-1] + 1]
t0 = int(rows[0]["Start_Timestamp"])
import collections
agg = collections.OrderedDict()
seq = []
for r in rows:
    n = r["Kernel_Name"]
    fam = next((k for k in ("k_walk", "k_prune_wg", "k_apply_wg", "k_apply_append", "k_emit", "k_segments", "k_finalize") if k in n), "sort/other")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    seq.append((fam, (int(r["Start_Timestamp"]) - t0) / 1e3, d))
walks = [s for s in seq if s[0] == "k_walk"]
print("k_walk dispatches", len(walks), "total ms", sum(w[2] for w in walks) / 1e3)
small = [w for w in walks if w[2] < 1500]
print("dispatches < 1.5 ms (descents + ramp-up):", len(small), "total ms", sum(w[2] for w in small) / 1e3)
print("last 12 k_walk dispatches (start us, dur us):", [(round(w[1]), round(w[2])) for w in walks[-12:]])
# one steady-state batch: everything between two consecutive large walks near the end
tail = seq[-40:]
for s in tail: print(s[0].ljust(16), round(s[1]), round(s[2], 1))
PY
rm -rf $out/t
