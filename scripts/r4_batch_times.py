#!/usr/bin/env python3
"""Per-batch wall time of one build with a synchronisation after the search and after the link phase of every batch
(what the ramp-up costs, what a multi-GPU build can shard): python scripts/r4_batch_times.py C2|C3|C4|C5 [items]
-> gpurun_out/r04_<cfg>_batch_times.txt (+ .json with the same rows)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import hannoy_amd as H  # noqa: E402
from bench import gen_data  # noqa: E402

CFG = {"C2": ("cosine", 1_000_000, 768, 16, 100), "C3": ("euclidean", 1_000_000, 768, 32, 200),
       "C4": ("cosine", 10_000_000, 128, 16, 100), "C5": ("hamming", 5_000_000, 1024, 16, 64)}
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
mname, n, dim, M, ef = CFG[name]
if len(sys.argv) > 2:
    n = int(sys.argv[2])
metric = {"cosine": H.COSINE, "euclidean": H.EUCLIDEAN, "hamming": H.HAMMING}[mname]
dev = torch.device("cuda", 0)
x = gen_data(torch, n, dim, "clustered", 42, dev).cpu().numpy()
items = H.ItemSet.from_f32(metric, x)
del x
b = H.Builder(items, M=M, M0=2 * M, ef_construction=ef, seed=42)
for rep in range(2):
    b.reset()
    b.sync()
    rows = []
    t_all = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        r = b.next_batch()
        if r.count == 0:
            break
        b.search(0, r.count)
        b.sync()
        t1 = time.perf_counter()
        b.apply()
        b.sync()
        rows.append((int(r.count), int(r.level), time.perf_counter() - t0, t1 - t0))
    tot = time.perf_counter() - t_all
lines = ["%s: %d x %d %s M=%d efC=%d; total (with two syncs per batch) %.1f ms, %d batches" % (name, n, dim, mname, M, ef, tot * 1e3, len(rows))]
acc = 0.0
for c, lvl, t, ts in rows:
    acc += t
    lines.append("batch of %7d (level %d): %8.3f ms (search %8.3f, apply %7.3f)  cumulative %8.2f ms" % (c, lvl, t * 1e3, ts * 1e3, (t - ts) * 1e3, acc * 1e3))
print("\n".join(lines))
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(out, exist_ok=True)
open(os.path.join(out, f"r04_{name.lower()}_batch_times.txt"), "w").write("\n".join(lines) + "\n")
json.dump({"config": name, "n": n, "dim": dim, "metric": mname, "M": M, "ef": ef, "total_ms": tot * 1e3,
           "batches": [{"count": c, "level": lvl, "ms": t * 1e3, "search_ms": ts * 1e3, "apply_ms": (t - ts) * 1e3}
                       for c, lvl, t, ts in rows]}, open(os.path.join(out, f"r04_{name.lower()}_batch_times.json"), "w"))
