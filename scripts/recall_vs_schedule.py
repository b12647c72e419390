#!/usr/bin/env python3
"""Recall@10 of the batch-synchronous GPU schedule against the SEQUENTIAL reference semantics
(oracle, 1 thread) and the rayon-like CPU build, same data / levels, N = 200k x 768 cosine."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import hannoy_amd as H
from oracle import orc

n, dim, nq = int(os.environ.get("N", 200000)), 768, 1000
dev = torch.device("cuda", 0)
x_dev = bench.gen_data(torch, 1_000_000, dim, "clustered", 42, dev)[:n].contiguous()
g = torch.Generator(device=dev); g.manual_seed(42)
centres = torch.rand((1024, dim), generator=g, device=dev) * 2 - 1
g2 = torch.Generator(device=dev); g2.manual_seed(1042)
which = torch.randint(0, 1024, (nq,), generator=g2, device=dev)
q_dev = centres[which] + 0.15 * torch.randn((nq, dim), generator=g2, device=dev)
truth = bench.brute_force_topk(torch, "cosine", x_dev, q_dev, 10)
items = H.ItemSet.from_f32(H.COSINE, x_dev.cpu().numpy())
levels = H.draw_levels(42, 16, n)
items.levels = levels
qc, qh = H.encode_vectors(H.COSINE, q_dev.cpu().numpy())
ds = orc.Dataset(orc.COSINE, dim, items.ids, items.codes, items.headers, levels)
cores = os.cpu_count()
out = {"n": n}
def rec(gr):
    ids, _, cnt = orc.search(ds, gr, qc, qh, k=10, ef_search=100, threads=cores)
    return round(bench.recall_at_k(ids, cnt, truth), 4)
for frac, bmax in ((0.02, 4096), (0.05, 8192), (0.1, 8192), (0.25, 32768), (1.0, 32768)):
    t0 = time.time()
    gg = H.build(items, M=16, M0=32, ef_construction=100, batch_frac=frac, batch_max=bmax)
    out[f"gpu_frac{frac}_bmax{bmax}"] = {"recall": rec(gg), "s": round(time.time() - t0, 2)}
    print(json.dumps(out), flush=True)
t0 = time.time(); og = orc.build(ds, M=16, M0=32, ef=100, threads=cores)
out["cpu_rayon_like"] = {"recall": rec(og), "s": round(time.time() - t0, 1), "threads": cores}
print(json.dumps(out), flush=True)
t0 = time.time(); og = orc.build(ds, M=16, M0=32, ef=100, threads=1)
out["cpu_sequential_1_thread"] = {"recall": rec(og), "s": round(time.time() - t0, 1)}
print(json.dumps(out), flush=True)
