#!/usr/bin/env python3
"""recall@10 of the GPU-built index vs the CPU(oracle)-built index at the FULL size of a BASELINE config
(north_star: "Reader recall@10 on the produced index matches CPU-built recall +-0.5 %").

  python scripts/recall_parity_full.py --config C2|C3|C4|C5 [--out profiles/r03_c4_full_scale_recall_parity.json]

Same synthetic data as bench.py, identical levels on both sides (drawn like the reference draws them:
StdRng::seed_from_u64(42) -> ChaCha12 -> WeightedIndex), GPU build with the DEFAULT schedule (batch caps of
hny_default_batch_max), CPU build = the oracle in rayon-like mode on every host core in the reference's x86
summation order.  Both graphs are searched by the same searcher (hny_builder_search_knn == the restated
Reader, bit for bit: tests/test_gpu_nns.py) for 1 000 held-out queries against exact ground truth, at
ef_search = 100 and at the first ef_search of 200 / 400 / 800 / 1600 (--deep: … 12 800) where the CPU-built index reaches 0.9
(dense synthetic clusters sit well below that at 100).  Prints / writes one JSON object."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {  # BASELINE.json configs[1..4]
    "C2": dict(n=1_000_000, dim=768, metric="cosine", M=16, ef=100),
    "C3": dict(n=1_000_000, dim=768, metric="euclidean", M=32, ef=200),
    "C4": dict(n=10_000_000, dim=128, metric="cosine", M=16, ef=100),
    "C5": dict(n=5_000_000, dim=1024, metric="hamming", M=16, ef=64),
}


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    p.add_argument("--items", type=int, default=0, help="override n (smaller rehearsal)")
    p.add_argument("--data", default="clustered")
    p.add_argument("--queries", type=int, default=1000)
    p.add_argument("--batch-max", type=int, default=0)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--out", default=None)
    p.add_argument("--deep", action="store_true", help="also ef_search 3 200 / 6 400 / 12 800 (dense synthetic clusters)")
    a = p.parse_args()
    c = dict(CONFIGS[a.config])
    if a.items:
        c["n"] = a.items
    import torch
    import bench
    import hannoy_amd as H
    from oracle import orc
    stop = threading.Event()
    phase = ["start"]

    def beat():  # the CPU build prints nothing for minutes: gpurun takes a silent command for hung
        t0 = time.time()
        while not stop.wait(45):
            print(f"[recall_parity] {phase[0]} ... {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    threading.Thread(target=beat, daemon=True).start()

    dev = torch.device("cuda", 0)
    metric = {"cosine": H.COSINE, "euclidean": H.EUCLIDEAN, "hamming": H.HAMMING}[c["metric"]]
    M, M0, ef = c["M"], 2 * c["M"], c["ef"]
    phase[0] = "data"
    x_dev = bench.gen_data(torch, c["n"], c["dim"], a.data, a.seed, dev)
    q_dev = bench.gen_data(torch, a.queries, c["dim"], a.data, a.seed, dev, queries=True)
    truth = bench.brute_force_topk(torch, c["metric"], x_dev, q_dev, 10)
    x = x_dev.cpu().numpy()
    del x_dev
    levels = H.draw_levels(a.seed, M, c["n"])
    items = H.ItemSet.from_f32(metric, x, levels=levels)
    del x
    qc, qh = H.encode_vectors(metric, q_dev.cpu().numpy())
    cores = os.cpu_count() or 1
    out = {"config": a.config, **c, "M0": M0, "data": a.data, "queries": a.queries, "cores": cores,
           "batch_max": a.batch_max or H.default_batch_max(c["n"]), "levels": "StdRng::seed_from_u64(%d)" % a.seed}
    efs = [100, 200, 400, 800, 1600] + ([3200, 6400, 12800] if a.deep else [])

    def recalls(builder):
        r = {}
        for e in efs:
            ids, _, cnt = builder.search_knn(qc, qh, k=10, ef_search=e)
            r[e] = round(bench.recall_at_k(ids, cnt, truth), 4)
        return r
    phase[0] = "GPU build"
    with H.Builder(items, M=M, M0=M0, ef_construction=ef, batch_max=a.batch_max) as b:
        b.run()                      # warm-up (first launches, allocations)
        b.finish()
        b.reset()
        t0 = time.perf_counter()
        b.run()
        gg = b.finish()
        out["gpu_build_s"] = round(time.perf_counter() - t0, 4)
        out["gpu_vec_per_s"] = round(c["n"] / out["gpu_build_s"], 1)
        rg = recalls(b)
    out["gpu_links"] = int(len(gg.nbrs))
    del gg
    phase[0] = f"CPU build ({cores} threads)"
    ds = orc.Dataset(metric, c["dim"], items.ids, items.codes, items.headers, levels)
    t0 = time.perf_counter()
    og = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_X86, threads=cores)
    out["cpu_build_s"] = round(time.perf_counter() - t0, 2)
    out["cpu_vec_per_s"] = round(c["n"] / out["cpu_build_s"], 1)
    out["cpu_links"] = int(len(og.nbrs))
    phase[0] = "search of the CPU-built graph"
    with H.Builder(items, prev=og, load=True, M=M, M0=M0, ef_construction=ef) as b:
        rc = recalls(b)
    first = next((e for e in efs if rc[e] >= 0.9), None)  # None: never, up to the largest ef_search tried
    out["recall_at_10"] = {str(e): {"gpu_built": rg[e], "cpu_built": rc[e], "diff": round(rg[e] - rc[e], 4)}
                           for e in efs if e <= max(first or efs[-1], 100)}
    out["ef_search_where_cpu_built_reaches_0.9"] = first
    out["within_half_percent"] = all(abs(v["diff"]) <= 0.005 for v in out["recall_at_10"].values())
    stop.set()
    line = json.dumps(out)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")
    print(line, flush=True)


if __name__ == "__main__":
    main()
