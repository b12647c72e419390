#!/usr/bin/env python3
"""Retry / heap paths at scale (no oracle: error-free completion + recall against brute force):
  (1) 2M x 64-bit Hamming codes: build (tie pools overflow all the time -> k_walk_heap) + 10 000 searches at ef_search 300
  (2) 1M x 768 cosine: filtered search, 30 % of the items as candidates, k = 10 at ef_search 5 000 (k_nns_heap) and
      unfiltered at ef_search 5 000 (result sets in HBM), recall@10 against exact top-10 among the candidates / all items
    python scripts/r3_stress_scale.py        (on the MI355X box)"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import hannoy_amd as H

dev = torch.device("cuda", 0)
out = {}
# (1)
n, dim = 2_000_000, 64
x = bench.gen_data(torch, n, dim, "uniform", 7, dev).cpu().numpy()
items = H.ItemSet.from_f32(H.HAMMING, x)
q = bench.gen_data(torch, 10_000, dim, "uniform", 7, dev, queries=True).cpu().numpy()
qc, qh = H.encode_vectors(H.HAMMING, q)
with H.Builder(items, M=16, M0=32, ef_construction=64) as b:
    t0 = time.perf_counter(); b.run(); g = b.finish(); t1 = time.perf_counter()
    ids, d, cnt = b.search_knn(qc, qh, k=10, ef_search=300)
    t2 = time.perf_counter()
out["hamming64_2M"] = dict(build_s=round(t1 - t0, 3), search_10k_ef300_s=round(t2 - t1, 3), batches=int(g.n_batches),
                           walk_evals=int(g.n_evals_walk), all_queries_answered=bool((cnt == 10).all()),
                           tie_pool_overflow_reported=int(g.n_tie_pool_overflow))
print(json.dumps(out), flush=True)
del items, x
# (2)
n, dim = 1_000_000, 768
xd = bench.gen_data(torch, n, dim, "clustered", 42, dev)
qd = bench.gen_data(torch, 200, dim, "clustered", 42, dev, queries=True)
x = xd.cpu().numpy()
items = H.ItemSet.from_f32(H.COSINE, x)
qc, qh = H.encode_vectors(H.COSINE, qd.cpu().numpy())
rng = np.random.default_rng(3)
cand = np.flatnonzero(rng.random(n) < 0.3).astype(np.uint32)
truth_all = bench.brute_force_topk(torch, "cosine", xd, qd, 10)
truth_c = cand[np.asarray(bench.brute_force_topk(torch, "cosine", xd[torch.from_numpy(cand.astype(np.int64)).to(dev)], qd, 10))]
with H.Builder(items, M=16, M0=32, ef_construction=100) as b:
    b.run(); b.finish()
    t0 = time.perf_counter()
    ids, d, cnt = b.search_knn(qc, qh, k=10, ef_search=5000)
    t1 = time.perf_counter()
    fi, fd, fc = b.nns(qc, qh, k=10, ef_search=5000, candidates=cand)
    t2 = time.perf_counter()
cs = set(cand.tolist())
out["cosine768_1M"] = dict(unfiltered_ef5000_s=round(t1 - t0, 3), recall_unfiltered=round(bench.recall_at_k(ids, cnt, truth_all), 4),
                           filtered_ef5000_s=round(t2 - t1, 3), recall_filtered=round(bench.recall_at_k(fi, fc, truth_c), 4),
                           filtered_hits_all_candidates=bool(all(int(v) in cs for r in range(len(fc)) for v in fi[r, :fc[r]])))
print(json.dumps(out), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r3_stress_scale.json"), "w"), indent=1)
