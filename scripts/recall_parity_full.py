#!/usr/bin/env python3
"""One-off: recall@10 of the GPU-built index vs the CPU(oracle)-built index at the FULL C2 size.
The CPU build is the oracle in rayon-like mode on every host core (reference x86 summation order);
both graphs are searched with the restated Reader (oracle, CPU) and the GPU graph also with
hny_builder_search_knn.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--items", type=int, default=1_000_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--data", default="clustered")
    p.add_argument("--queries", type=int, default=1000)
    a = p.parse_args()
    import torch
    import bench
    import hannoy_amd as H
    from oracle import orc
    dev = torch.device("cuda", 0)
    x_dev = bench.gen_data(torch, a.items, a.dim, a.data, 42, dev)
    g = torch.Generator(device=dev); g.manual_seed(42)
    centres = torch.rand((1024, a.dim), generator=g, device=dev, dtype=torch.float32) * 2 - 1
    g2 = torch.Generator(device=dev); g2.manual_seed(1042)
    which = torch.randint(0, 1024, (a.queries,), generator=g2, device=dev)
    q_dev = centres[which] + 0.15 * torch.randn((a.queries, a.dim), generator=g2, device=dev)
    if a.data == "uniform":
        q_dev = bench.gen_data(torch, a.queries, a.dim, "uniform", 1042, dev)
    truth = bench.brute_force_topk(torch, "cosine", x_dev, q_dev, 10)
    x = x_dev.cpu().numpy()
    items = H.ItemSet.from_f32(H.COSINE, x)
    qc, qh = H.encode_vectors(H.COSINE, q_dev.cpu().numpy())
    out = {"n": a.items, "dim": a.dim, "data": a.data, "cores": os.cpu_count()}
    with H.Builder(items, M=16, M0=32, ef_construction=100, seed=42) as b:
        t0 = time.perf_counter(); b.run(); gg = b.finish(); out["gpu_build_s"] = time.perf_counter() - t0
        ids, _, cnt = b.search_knn(qc, qh, k=10, ef_search=100)
        out["recall_gpu_built_gpu_search"] = bench.recall_at_k(ids, cnt, truth)
    # same levels for the CPU build: recover them from the GPU graph's records
    lv = np.zeros(a.items, np.uint8)
    np.maximum.at(lv, gg.rec_item, gg.rec_layer)
    ds = orc.Dataset(orc.COSINE, a.dim, items.ids, items.codes, items.headers, lv)
    cores = os.cpu_count() or 1
    ids, _, cnt = orc.search(ds, gg, qc, qh, k=10, ef_search=100, threads=cores)
    out["recall_gpu_built_cpu_search"] = bench.recall_at_k(ids, cnt, truth)
    t0 = time.perf_counter()
    og = orc.build(ds, M=16, M0=32, ef=100, order=orc.ORDER_X86, threads=cores)
    out["cpu_build_s"] = time.perf_counter() - t0
    out["cpu_vec_per_s"] = a.items / out["cpu_build_s"]
    ids, _, cnt = orc.search(ds, og, qc, qh, k=10, ef_search=100, threads=cores)
    out["recall_cpu_built_cpu_search"] = bench.recall_at_k(ids, cnt, truth)
    out["cpu_links"] = int(len(og.nbrs)); out["gpu_links"] = int(len(gg.nbrs))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
