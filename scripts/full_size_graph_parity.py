#!/usr/bin/env python3
"""BASELINE configs at FULL size: the GPU-built graph against the oracle's, edge for edge.

The regular -m gpu suite checks the full-size builds through size-independent properties (the oracle
needs minutes per config on the box's host cores); this script does the whole comparison once per
round and writes gpurun_out/r05_full_size_graph_parity.json (-> profiles/):  same data as bench.py, same levels, same
batch schedule, oracle in the wave summation order on every host core -> identical records
(rec_item, rec_layer, offsets, neighbours), entry points, link count and walk-evaluation count.

  python scripts/full_size_graph_parity.py [C2 C3 C5 C4_4M C4]      (DATA=overlap|clustered overrides the data kind)

Round 5: C4 at its real 10M (the oracle's wave-order reduction in AVX2, prefetch hints, a visited bitset and a
link phase spread over the threads by target brought its build inside one gpurun call); C4 / C5 on `overlap` data,
the distribution their bench lines are taken on.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import hannoy_amd as H  # noqa: E402
from bench import gen_data  # noqa: E402
from oracle import orc  # noqa: E402

CFG = {"C2": ("cosine", 1_000_000, 768, 16, 100), "C3": ("euclidean", 1_000_000, 768, 32, 200),
       "C4": ("cosine", 10_000_000, 128, 16, 100), "C5": ("hamming", 5_000_000, 1024, 16, 64),
       # C4's shape at 40 % of its size (round 4: all the oracle finished inside one gpurun call)
       "C4_4M": ("cosine", 4_000_000, 128, 16, 100)}


def heartbeat():
    import threading

    def beat():
        t0 = time.time()
        while True:
            time.sleep(60)
            print(f"[{int(time.time() - t0)} s] oracle still building ...", flush=True)
    threading.Thread(target=beat, daemon=True).start()


def main():
    which = sys.argv[1:] or ["C2", "C3", "C5"]
    heartbeat()
    dev = torch.device("cuda", 0)
    cores = orc.host_threads()  # what the cgroup grants (16 on the GPU box), not the 256 logical CPUs it shows
    # On the GPU box only gpurun_out/ travels back, and it starts empty there: a run of SOME configs starts from
    # the committed file (profiles/ travels with the snapshot) and adds to it, so copying the result back to
    # profiles/ never drops the entries of an earlier call (round 3 lost three of four that way).
    name = "r05_full_size_graph_parity.json"
    out_path = os.environ.get("OUT", os.path.join(ROOT, "gpurun_out", name))
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    out = {}
    for src in (os.path.join(ROOT, "profiles", name), out_path):
        if os.path.exists(src):
            out.update(json.load(open(src)))
    import hashlib
    ksha = hashlib.sha1(open(os.path.join(ROOT, "hannoy_amd", "csrc", "hny_kernels.hip"), "rb").read()).hexdigest()
    for name in which:
        mname, n, dim, M, ef = CFG[name]
        metric = {"cosine": H.COSINE, "euclidean": H.EUCLIDEAN, "hamming": H.HAMMING}[mname]
        kind = os.environ.get("DATA") or ("overlap" if name in ("C4", "C5", "C4_4M") else "clustered")
        x = gen_data(torch, n, dim, kind, 42, dev).cpu().numpy()
        items = H.ItemSet.from_f32(metric, x)
        del x
        levels = H.draw_levels(42, M, n)  # what hny_build draws from StdRng::seed_from_u64(42)
        items.levels = levels
        t0 = time.perf_counter()
        bmax = H.default_batch_max(n)  # what batch_max = 0 selects: 65 536 at 1M, 262 144 at 5M
        g = H.build(items, M=M, M0=2 * M, ef_construction=ef, seed=42)
        t_gpu = time.perf_counter() - t0
        ds = orc.Dataset(metric, dim, items.ids, items.codes, items.headers, levels)
        t0 = time.perf_counter()
        o = orc.build(ds, M=M, M0=2 * M, ef=ef, order=orc.ORDER_WAVE, threads=cores, batch_frac=1.0,
                      batch_max=bmax)
        t_cpu = time.perf_counter() - t0
        same = (np.array_equal(g.rec_item, o.rec_item) and np.array_equal(g.rec_layer, o.rec_layer)
                and np.array_equal(g.offsets, o.offsets) and np.array_equal(g.nbrs, o.nbrs)
                and g.entry_points.tolist() == o.entry_points.tolist() and g.max_level == o.max_level)
        res = {"config": f"{name}: {n} x {dim} {mname}, M={M} M0={2 * M} efC={ef}, {kind} synthetic data, seed 42, default schedule (batch_max {bmax})",
               "graphs_identical": bool(same), "records": int(len(g.rec_item)), "links": int(len(g.nbrs)),
               "n_links_added": [int(g.n_links_added), int(o.n_links_added)],
               "n_evals_walk": [int(g.n_evals_walk), int(o.n_evals_walk)],
               "gpu_build_incl_upload_s": round(t_gpu, 2), "oracle_build_s": round(t_cpu, 1), "oracle_threads": cores,
               "kernel_source_sha1": ksha}
        print(json.dumps(res), flush=True)
        out[name] = res
        json.dump(out, open(out_path, "w"), indent=1)
        assert same and g.n_links_added == o.n_links_added and g.n_evals_walk == o.n_evals_walk, name
        del g, o, ds, items


if __name__ == "__main__":
    main()
