#!/usr/bin/env python3
"""CPU baseline honesty check: the oracle's rayon-like build (the port of the reference's parallel loop,
src/hnsw.rs:172-185) on the first 200 000 vectors of the C2 workload, swept over thread counts on the GPU box's host.
The box shows 256 logical CPUs but its cgroup grants 16 CPUs' worth of time (cpu.max), so "256 threads" is 256
threads sharing 16 cores.  Writes gpurun_out/r04_cpu_baseline_thread_sweep.json (-> profiles/); bench.py's
cpu_baseline uses `best_threads`.

  gpurun --timeout 900 -- 'python scripts/r4_cpu_thread_sweep.py [n=200000]'
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


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    dim, M, ef = 768, 16, 100
    x = gen_data(torch, n, dim, "clustered", 42, torch.device("cuda", 0)).cpu().numpy()
    items = H.ItemSet.from_f32(H.COSINE, x)
    del x
    levels = H.draw_levels(42, M, n)
    ds = orc.Dataset(H.COSINE, dim, items.ids, items.codes, items.headers, levels)
    quota = orc.host_threads()
    try:
        cpu_max = open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        cpu_max = None
    out = {"workload": f"first {n} vectors of C2 (1M x 768 cosine, clustered, seed 42), M={M} M0={2 * M} efC={ef}, "
                       f"x86 summation order (AVX2+FMA), rayon-like parallel insert, vectors in RAM",
           "logical_cpus": os.cpu_count(), "cgroup_cpu_max": cpu_max, "host_threads": quota, "runs": []}
    for t in [8, 16, 32, 64, 128, 256]:
        if t > (os.cpu_count() or 1):
            continue
        t0 = time.perf_counter()
        g = orc.build(ds, M=M, M0=2 * M, ef=ef, order=orc.ORDER_X86, threads=t)
        dt = time.perf_counter() - t0
        r = {"threads": t, "seconds": round(dt, 2), "vectors_per_s": round(n / dt, 1), "links": int(len(g.nbrs)),
             "walk_evals": int(g.n_evals_walk)}
        print(json.dumps(r), flush=True)
        out["runs"].append(r)
        del g
    best = max(out["runs"], key=lambda r: r["vectors_per_s"])
    out["best_threads"] = best["threads"]
    out["best_vectors_per_s"] = best["vectors_per_s"]
    # 2 flop per dimension and evaluation (fma), evaluations of the best run
    out["best_gflops"] = round(best["walk_evals"] * 2 * dim / best["seconds"] / 1e9, 1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_cpu_baseline_thread_sweep.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "runs"}))


if __name__ == "__main__":
    main()
