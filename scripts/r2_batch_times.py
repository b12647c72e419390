"""per-batch wall time of a C2 build (next_batch / search / apply / sync), to see what the ramp-up costs"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hannoy_amd as H
from bench import gen_data
dev = torch.device("cuda", 0)
x = gen_data(torch, 1_000_000, 768, "clustered", 42, dev).cpu().numpy()
items = H.ItemSet.from_f32(H.COSINE, x)
b = H.Builder(items, M=16, M0=32, ef_construction=100, seed=42)
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
        rows.append((r.count, time.perf_counter() - t0, t1 - t0))
    tot = time.perf_counter() - t_all
print("total (with a sync per batch) %.1f ms, %d batches" % (tot * 1e3, len(rows)))
acc = 0.0
for n, t, ts in rows:
    acc += t
    print("batch of %6d: %7.3f ms (search %7.3f, apply %7.3f)  cumulative %7.2f ms" % (n, t * 1e3, ts * 1e3, (t - ts) * 1e3, acc * 1e3))
