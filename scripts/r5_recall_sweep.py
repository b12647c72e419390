#!/usr/bin/env python3
"""Round 5: the batch_max of the batch-synchronous schedule against recall, at FULL size, with the comparator's
spread measured at full size too (VERDICT r4 "weak 1": C4 built with 524 288-member batches scored 0.66 - 0.79 pt
below ONE CPU build at ef_search 100).

For every --data kind: ground truth once, then
  * one GPU build per --batch-max value (0 = hny_default_batch_max) -> build seconds + recall@10 at --ef-search;
  * --cpu-builds N rayon-like CPU builds (the oracle in the x86 order on the box's CPU quota; thread interleaving
    differs from run to run) -> the comparator's median / min / max at the same ef_search values.

  gpurun --timeout 1200 -- 'python scripts/r5_recall_sweep.py --config C4 --data lat16 --batch-max 65536,0 --cpu-builds 3'
writes gpurun_out/r05_<config>_recall_sweep_<data>.json after every build (-> profiles/)."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {"C2": dict(n=1_000_000, dim=768, metric="cosine", M=16, ef=100),
           "C3": dict(n=1_000_000, dim=768, metric="euclidean", M=32, ef=200),
           "C4": dict(n=10_000_000, dim=128, metric="cosine", M=16, ef=100),
           "C5": dict(n=5_000_000, dim=1024, metric="hamming", M=16, ef=64)}


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--config", default="C4", choices=sorted(CONFIGS))
    p.add_argument("--data", default="overlap", help="comma-separated gen_data kinds")
    p.add_argument("--items", type=int, default=0)
    p.add_argument("--batch-max", default="65536,131072,262144,524288")
    p.add_argument("--batch-frac", default="1.0")
    p.add_argument("--ef-search", default="100,200,400")
    p.add_argument("--cpu-builds", type=int, default=0)
    p.add_argument("--queries", type=int, default=1000)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--tag", default="")
    p.add_argument("--level-seeds", default="",
                   help="comma-separated seeds of the level draw (default: --seed): every GPU / CPU build is repeated per "
                        "seed on the SAME data and queries — the build-to-build spread of either side at fixed data")
    a = p.parse_args()
    c = dict(CONFIGS[a.config])
    if a.items:
        c["n"] = a.items
    import torch
    import bench
    import hannoy_amd as H
    from oracle import orc
    stop, phase = threading.Event(), ["start"]

    def beat():
        t0 = time.time()
        while not stop.wait(45):
            print(f"[r5_recall_sweep] {phase[0]} ... {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    threading.Thread(target=beat, daemon=True).start()
    dev = torch.device("cuda", 0)
    metric = {"cosine": H.COSINE, "euclidean": H.EUCLIDEAN, "hamming": H.HAMMING}[c["metric"]]
    M, M0, ef = c["M"], 2 * c["M"], c["ef"]
    threads = orc.host_threads()
    efs = [int(e) for e in a.ef_search.split(",")]
    bmaxes = [int(b) for b in a.batch_max.split(",") if b != ""]
    fracs = [float(f) for f in a.batch_frac.split(",")]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)

    for kind in a.data.split(","):
        out_path = os.path.join(ROOT, "gpurun_out", f"r05_{a.config.lower()}_recall_sweep_{kind}{a.tag}.json")
        out = {"config": a.config, **c, "M0": M0, "data": kind, "queries": a.queries,
               "host": {"logical_cpus": os.cpu_count(), "cpu_quota": threads},
               "levels": "StdRng::seed_from_u64(%d)" % a.seed, "ef_search": efs, "gpu_builds": [], "cpu_builds": []}

        def save():
            json.dump(out, open(out_path, "w"), indent=1)

        phase[0] = f"{kind}: data"
        x_dev = bench.gen_data(torch, c["n"], c["dim"], kind, a.seed, dev)
        q_dev = bench.gen_data(torch, a.queries, c["dim"], kind, a.seed, dev, queries=True)
        qc, qh = H.encode_vectors(metric, q_dev.cpu().numpy())
        phase[0] = f"{kind}: ground truth"
        truth = bench.brute_force_topk(torch, c["metric"], x_dev, q_dev, 10)
        kth = None
        if c["metric"] == "hamming":  # tie-aware recall: a hit is an item no farther than the true 10th neighbour
            ks = []
            db = (x_dev > 0).half()
            for q0 in range(0, q_dev.shape[0], 128):
                qb = (q_dev[q0:q0 + 128] > 0).half()
                d = qb @ (1 - db).T + (1 - qb) @ db.T
                ks.append(torch.topk(d.float(), 10, dim=1, largest=False).values[:, 9])
            kth = torch.cat(ks).round().cpu().numpy().astype(np.int64)
            del db
        x = x_dev.cpu().numpy()
        del x_dev
        torch.cuda.empty_cache()
        lseeds = [int(v) for v in a.level_seeds.split(",") if v != ""] or [a.seed]
        levels = H.draw_levels(lseeds[0], M, c["n"])
        phase[0] = f"{kind}: encode"
        items = H.ItemSet.from_f32(metric, x, levels=levels)
        del x
        out["level_seeds"] = lseeds

        def recalls(builder):
            r = {}
            for e in efs:
                ids, dists, cnt = builder.search_knn(qc, qh, k=10, ef_search=e)
                r[str(e)] = round(bench.recall_at_k(ids, cnt, truth), 4)
                if kth is not None:
                    bits = np.rint(dists.astype(np.float64) * c["dim"]).astype(np.int64)
                    ok = (bits <= kth[:, None]) & (np.arange(10)[None, :] < cnt[:, None])
                    r[str(e) + "_tie_aware"] = round(float(ok.sum()) / (10 * len(kth)), 4)
            return r

        for lsd in lseeds:
          levels = H.draw_levels(lsd, M, c["n"])
          items.levels = levels
          for fr in fracs:
            for bm in bmaxes:
                phase[0] = f"{kind}: GPU build batch_max {bm} frac {fr} level seed {lsd}"
                kw = dict(M=M, M0=M0, ef_construction=ef, batch_frac=fr)
                if bm:
                    kw["batch_max"] = bm
                with H.Builder(items, **kw) as b:
                    b.run()
                    b.finish()
                    b.reset()
                    t0 = time.perf_counter()
                    b.run()
                    g = b.finish()
                    dt = time.perf_counter() - t0
                    rec = {"batch_max": bm or H.default_batch_max(c["n"]), "level_seed": lsd,
                           "default": bm == 0, "batch_frac": fr, "build_s": round(dt, 4),
                           "vec_per_s": round(c["n"] / dt, 1), "n_batches": int(g.n_batches),
                           "links": int(len(g.nbrs)), "evals_walk": int(g.n_evals_walk),
                           "recall_at_10": recalls(b)}
                    del g
                out["gpu_builds"].append(rec)
                print(json.dumps({"data": kind, **rec}), flush=True)
                save()
        if a.cpu_builds:
          for lsd in lseeds:
            levels = H.draw_levels(lsd, M, c["n"])
            items.levels = levels
            ds = orc.Dataset(metric, c["dim"], items.ids, items.codes, items.headers, levels)
            for r_ in range(a.cpu_builds):
                phase[0] = f"{kind}: CPU build {r_ + 1} of {a.cpu_builds} ({threads} threads), level seed {lsd}"
                t0 = time.perf_counter()
                og = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_X86, threads=threads)
                dt = time.perf_counter() - t0
                with H.Builder(items, prev=og, load=True, M=M, M0=M0, ef_construction=ef) as b:
                    rec = {"run": r_ + 1, "level_seed": lsd, "threads": threads, "build_s": round(dt, 2),
                           "vec_per_s": round(c["n"] / dt, 1), "links": int(len(og.nbrs)), "recall_at_10": recalls(b)}
                del og
                out["cpu_builds"].append(rec)
                print(json.dumps({"data": kind, "cpu": rec}), flush=True)
                save()
            key = (lambda e: str(e) + "_tie_aware") if kth is not None else str
            summ = {}
            for e in efs:
                cpu = sorted(b_["recall_at_10"][key(e)] for b_ in out["cpu_builds"])
                med = cpu[len(cpu) // 2] if len(cpu) % 2 else 0.5 * (cpu[len(cpu) // 2 - 1] + cpu[len(cpu) // 2])
                summ[str(e)] = {"cpu_median": round(med, 4), "cpu_min": cpu[0], "cpu_max": cpu[-1],
                                "gpu": [{"batch_max": g_["batch_max"], "batch_frac": g_["batch_frac"],
                                         "recall": g_["recall_at_10"][key(e)],
                                         "minus_cpu_median": round(g_["recall_at_10"][key(e)] - med, 4),
                                         "within_half_point_of_median_or_inside_range":
                                             bool(abs(g_["recall_at_10"][key(e)] - med) <= 0.005 or
                                                  cpu[0] <= g_["recall_at_10"][key(e)] <= cpu[-1])}
                                        for g_ in out["gpu_builds"]]}
            out["summary"] = summ
            save()
        del items
    stop.set()


if __name__ == "__main__":
    main()
