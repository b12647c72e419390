#!/usr/bin/env python3
"""Is the recall drop at 10M x 128 (10k points per Gaussian cluster) inherent to the data?  Same
per-cluster density at a size the CPU oracle can build: GPU-built vs CPU-built recall@10."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import hannoy_amd as H
from oracle import orc

n, dim, ncl, nq = 400_000, 128, 40, 1000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
centres = torch.rand((ncl, dim), generator=g, device=dev) * 2 - 1
x_dev = centres[torch.randint(0, ncl, (n,), generator=g, device=dev)] + 0.15 * torch.randn((n, dim), generator=g, device=dev)
q_dev = centres[torch.randint(0, ncl, (nq,), generator=g, device=dev)] + 0.15 * torch.randn((nq, dim), generator=g, device=dev)
truth = bench.brute_force_topk(torch, "cosine", x_dev, q_dev, 10)
items = H.ItemSet.from_f32(H.COSINE, x_dev.cpu().numpy())
qc, qh = H.encode_vectors(H.COSINE, q_dev.cpu().numpy())
out = {"n": n, "dim": dim, "points_per_cluster": n // ncl}
with H.Builder(items, M=16, M0=32, ef_construction=100, seed=42) as b:
    b.run(); gg = b.finish()
    for ef in (100, 400):
        ids, _, cnt = b.search_knn(qc, qh, k=10, ef_search=ef)
        out[f"gpu_built_ef{ef}"] = bench.recall_at_k(ids, cnt, truth)
lv = np.zeros(n, np.uint8); np.maximum.at(lv, gg.rec_item, gg.rec_layer)
ds = orc.Dataset(orc.COSINE, dim, items.ids, items.codes, items.headers, lv)
t0 = time.time(); og = orc.build(ds, M=16, M0=32, ef=100, threads=os.cpu_count()); out["cpu_build_s"] = time.time() - t0
for ef in (100, 400):
    ids, _, cnt = orc.search(ds, og, qc, qh, k=10, ef_search=ef, threads=os.cpu_count())
    out[f"cpu_built_ef{ef}"] = bench.recall_at_k(ids, cnt, truth)
print(json.dumps(out))
