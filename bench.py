#!/usr/bin/env python3
"""bench.py — headline benchmark: HNSW build throughput (vectors indexed / s) + recall@10.

Workload (BASELINE.json configs[1], "C2"): 1M x 768 f32 Cosine, M=16 (M0=32), ef_construction=100,
one MI355X, vectors resident in HBM before the timed region.  One "step" = one complete build of
the index (graph reset -> every batch searched, pruned, linked -> records exported to the host).

  python bench.py --gpus N --steps K --warmup W         (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (see the contract in the task statement) carrying `roofline`
(dominant kernel = k_walk, HIP events on the build stream) and `cpu_baseline` (the CPU oracle, i.e.
a port of the reference algorithm — the real hannoy crate cannot be built here: no Rust, no LMDB).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--items", "--n", dest="n", type=int, default=1_000_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--metric", default="cosine", choices=["cosine", "euclidean", "manhattan", "hamming"])
    p.add_argument("--M", type=int, default=16)
    p.add_argument("--M0", type=int, default=0)
    p.add_argument("--ef", type=int, default=100)
    p.add_argument("--data", default="clustered", choices=["clustered", "overlap", "uniform"])
    p.add_argument("--alt-data", default=None, choices=["overlap", "uniform", "clustered", "none"],
                   help="second distribution measured in the same run (value_alt / roofline_alt); default: "
                        "`overlap` for the default C2 run on one GPU, none otherwise")
    p.add_argument("--alt-steps", type=int, default=2)
    p.add_argument("--cpu-full", action="store_true",
                   help="time the CPU baseline on ALL vectors (about 80-100 s at C2) instead of a bounded sample")
    p.add_argument("--batch-frac", type=float, default=0.0)
    p.add_argument("--batch-max", type=int, default=0)
    p.add_argument("--queries", type=int, default=1000)
    p.add_argument("--ef-search", type=int, default=100)
    p.add_argument("--cpu-sample", type=int, default=0,
                   help="items in the CPU baseline sample; 0 = calibrate for about --cpu-seconds of wall time")
    p.add_argument("--cpu-seconds", type=float, default=15.0)
    p.add_argument("--no-cpu", action="store_true")
    p.add_argument("--no-recall", action="store_true")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--x86-order", action="store_true",
                   help="strict mode: f32 distances in the reference's x86 summation order")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo: test mode, ranks may share one GPU, exchange staged through the host")
    return p.parse_args()


def gen_data(torch, n, dim, kind, seed, device, queries=False):
    """Synthetic vectors, generated on the GPU (counter-based Philox -> identical on every rank).
    queries=True: held-out points of the SAME distribution (same centres / basis, fresh noise)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    g2 = torch.Generator(device=device)
    g2.manual_seed(seed + 1000)
    gn = g2 if queries else g  # the structure (centres, basis) always comes from `g`
    if kind == "uniform":  # reference-style U(-1,1) (src/tests/mod.rs:133-136)
        return torch.rand((n, dim), generator=gn, device=device, dtype=torch.float32) * 2 - 1
    if kind == "overlap":
        # embedding-like: 1 024 OVERLAPPING clusters on a 32-d manifold embedded in `dim` dimensions
        # (latent centres N(0, 1), points centre + 0.5 N(0, 1): the spread inside a cluster is half the
        # spread of the centres, so neighbourhoods cross cluster borders) + 0.05 isotropic noise
        k = min(32, dim)
        centres = torch.randn((1024, k), generator=g, device=device, dtype=torch.float32)
        basis = torch.linalg.qr(torch.randn((dim, k), generator=g, device=device, dtype=torch.float32))[0].T
        which = torch.randint(0, 1024, (n,), generator=gn, device=device)
        out = torch.empty((n, dim), device=device, dtype=torch.float32)
        for lo in range(0, n, 1 << 18):  # in slices: bounded temporaries
            hi = min(n, lo + (1 << 18))
            z = centres[which[lo:hi]] + 0.5 * torch.randn((hi - lo, k), generator=gn, device=device)
            out[lo:hi] = z @ basis + 0.05 * torch.randn((hi - lo, dim), generator=gn, device=device)
        return out
    # 1024-centre Gaussian mixture, centres U(-1,1), sigma 0.15: well separated
    centres = torch.rand((1024, dim), generator=g, device=device, dtype=torch.float32) * 2 - 1
    which = torch.randint(0, 1024, (n,), generator=gn, device=device)
    return centres[which] + 0.15 * torch.randn((n, dim), generator=gn, device=device, dtype=torch.float32)


def brute_force_topk(torch, metric, data, queries, k):
    """Exact top-k under the reference's metric definition (plumbing, torch on the GPU)."""
    out = []
    for q0 in range(0, queries.shape[0], 256):
        q = queries[q0:q0 + 256]
        if metric == "cosine":
            s = (q / q.norm(dim=1, keepdim=True)) @ (data / data.norm(dim=1, keepdim=True)).T
            out.append(torch.topk(s, k, dim=1).indices)
        elif metric == "euclidean":
            d = torch.cdist(q, data)
            out.append(torch.topk(d, k, dim=1, largest=False).indices)
        elif metric == "manhattan":
            d = torch.cdist(q, data, p=1)
            out.append(torch.topk(d, k, dim=1, largest=False).indices)
        else:  # hamming on the Binary codec bits (x > 0)
            qb, db = (q > 0).float(), (data > 0).float()
            d = qb @ (1 - db).T + (1 - qb) @ db.T
            out.append(torch.topk(d, k, dim=1, largest=False).indices)
    return torch.cat(out).cpu().numpy()


def recall_at_k(found, counts, truth):
    hit = 0
    for i in range(truth.shape[0]):
        hit += len(set(found[i, :counts[i]].tolist()) & set(truth[i].tolist()))
    return hit / truth.size


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    import hannoy_amd as H
    from hannoy_amd import multigpu
    H.load_library()
    metric = {"cosine": H.COSINE, "euclidean": H.EUCLIDEAN, "manhattan": H.MANHATTAN,
              "hamming": H.HAMMING}[a.metric]
    M0 = a.M0 or 2 * a.M

    # ---- synthetic data (+ held-out queries from the same distribution) ----
    x_dev = gen_data(torch, a.n, a.dim, a.data, a.seed, dev)
    q_dev = gen_data(torch, a.queries, a.dim, a.data, a.seed, dev, queries=True) if a.queries else None
    x = x_dev.cpu().numpy()
    items = H.ItemSet.from_f32(metric, x)
    row_bytes = items.codes.shape[1]
    bytes_per_eval = row_bytes + items.headers.shape[1]  # row + header

    def timed_builds(items_, steps, warmup):
        """W untimed + K timed full builds (graph reset -> every batch -> records exported)."""
        b = H.Builder(items_, M=a.M, M0=M0, ef_construction=a.ef, seed=a.seed,
                      batch_frac=a.batch_frac, batch_max=a.batch_max, device=local_rank,
                      x86_order=a.x86_order)
        b.set_profiling(True)
        drv = multigpu.Driver(b, torch, dist if world > 1 else None, rank, world, dev,
                              host_staged=(a.backend == "gloo"))

        def step():
            b.reset()
            drv.run()
            return b.finish()

        for _ in range(warmup):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g = None
        for _ in range(steps):
            g = step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt_], device=dev if a.backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ = float(tt.item())
        if g is None:
            g = step()
        return b, drv, g, dt_

    def roofline_of(g, dt_, steps):
        """dominant kernel (k_walk): algorithmic bytes / device time, HIP events on the builder's
        stream around every k_walk dispatch (a level-0 batch: descent dispatch + key sort + layer-0
        dispatch under one pair); `launches` = k_walk dispatches, what rocprofv3 counts"""
        wb = g.n_evals_walk * bytes_per_eval
        if g.t_walk_kernels_s <= 0:
            return None
        ach = wb / g.t_walk_kernels_s / 1e9
        r = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "k_walk",
             "launches": int(g.n_walk_launches),
             "avg_launch_ms": round(1e3 * g.t_walk_kernels_s / max(1, g.n_walk_launches), 4),
             "algorithmic_bytes_per_launch": int(wb / max(1, g.n_walk_launches)),
             "bytes_per_eval": bytes_per_eval}
        if steps:  # the same bytes over the WHOLE step (prune, link ops, export included)
            r["frac_whole_build"] = round(wb / (dt_ / steps) / 1e9 / HBM_PEAK_GBS, 4)
        if ach > HBM_PEAK_GBS:  # SURVEY 8(d)'s numerator counts every evaluated row, wherever it came from
            r["note"] = ("achieved = algorithmic bytes / kernel time exceeds the HBM peak because part of the row "
                         "reads are L2 hits (XCD-tiled work queue: neighbouring queries share an L2); "
                         "l2_to_fabric_gbs is the counted L2->fabric traffic over the same time")
        return r

    def build_stats(g):
        return {"n_batches": int(g.n_batches), "n_distance_evals": int(g.n_distance_evals),
                "evals_walk": int(g.n_evals_walk), "evals_prune": int(g.n_evals_prune),
                "evals_apply": int(g.n_evals_apply), "links_added": int(g.n_links_added),
                "t_build_s": round(g.t_build_s, 3), "t_export_s": round(g.t_export_s, 3),
                "t_upload_s": round(g.t_upload_s, 3),
                "t_walk_kernels_s": round(g.t_walk_kernels_s, 3),
                "t_prune_kernels_s": round(g.t_prune_kernels_s, 3),
                "t_sort_kernels_s": round(g.t_sort_kernels_s, 3),
                "t_apply_kernels_s": round(g.t_apply_kernels_s, 3),
                "tie_pool_overflow": int(g.n_tie_pool_overflow),
                "sub_wave_walks": int(g.n_sub_walks), "sub_wave_handed_over": int(g.n_sub_retries)}

    builder, driver, graph, dt = timed_builds(items, a.steps, a.warmup)
    value = a.n * a.steps / dt if a.steps else 0.0
    roof = roofline_of(graph, dt, a.steps)

    # HBM-side traffic of the dominant kernel: PMC counters cannot be read from inside this process,
    # so the figure comes from the committed rocprofv3 --pmc passes of this same command
    # (scripts/profile_c2.sh; FETCH_SIZE x2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes), per launch.
    # The profile records the hash of the kernel source it was taken on: a mismatch is flagged.
    default_c2 = (a.n == 1_000_000 and a.dim == 768 and a.metric == "cosine" and a.M == 16
                  and a.ef == 100 and a.data == "clustered" and not a.batch_frac and not a.batch_max
                  and world == 1 and not a.x86_order)
    pmc_name = next((f for f in ("r02_c2_pmc_hbm.json", "r01_c2_pmc_hbm_specialised.json")
                     if os.path.exists(os.path.join(ROOT, "profiles", f))), None)
    if roof and default_c2 and pmc_name:
        with open(os.path.join(ROOT, "profiles", pmc_name)) as f:
            pj = json.load(f)
        pk = pj.get("k_walk")
        if pk:
            total = pk["hbm_read_bytes_corrected_x2"] + pk["hbm_write_bytes"]
            roof["traffic"] = int(total / max(1, pk["launches"]))
            # FETCH_SIZE counts what the Infinity Cache serves too: this is L2-to-fabric traffic (an upper
            # bound on HBM), and `achieved` above is algorithmic bytes (it also credits L2 hits)
            roof["l2_to_fabric_gbs"] = round(roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9, 1)
            import hashlib
            with open(os.path.join(ROOT, "hannoy_amd", "csrc", "hny_kernels.hip"), "rb") as f:
                sha = hashlib.sha1(f.read()).hexdigest()
            roof["traffic_source"] = (f"profiles/{pmc_name}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                      f"command (scripts/profile_c2.sh), 2x FETCH + WRITE summed over the "
                                      f"{pk['launches']} k_walk dispatches of one build, per dispatch")
            roof["traffic_stale"] = pj.get("kernel_source_sha1") != sha  # kernels changed since the PMC passes

    known = {(1_000_000, 768, "cosine", 16, 100): "C2", (1_000_000, 768, "euclidean", 32, 200): "C3",
             (10_000_000, 128, "cosine", 16, 100): "C4 (on %d GPU)" % world,
             (5_000_000, 1024, "hamming", 16, 64): "C5 (on %d GPU)" % world}
    cfg_name = known.get((a.n, a.dim, a.metric, a.M, a.ef), "custom")
    shape = f"{a.n // 1_000_000}M" if a.n % 1_000_000 == 0 and a.n else str(a.n)
    out = {
        "metric": f"vectors indexed/sec (build) + recall@10, {shape} x {a.dim} {a.metric.capitalize()} "
                  f"M={a.M} efC={a.ef}",
        "value": round(value, 1), "unit": "vectors/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(1e3 * dt / max(1, a.steps), 2),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if metric < H.HAMMING else "u64-popcount", "data": "synthetic",
        "config": {"workload": f"{cfg_name}: {a.n} x {a.dim} {a.metric}, M={a.M} M0={M0} efC={a.ef}, "
                               f"{a.data} synthetic vectors resident in HBM, 1 step = 1 full build",
                   "n": a.n, "dim": a.dim, "M": a.M, "M0": M0, "ef_construction": a.ef,
                   "batch_frac": builder.opts.batch_frac or 1.0,
                   "batch_max": builder.opts.batch_max or H.default_batch_max(a.n),
                   "parallelism": f"item-sharded search x{world}, replicated graph",
                   "distance_order": "x86 (strict)" if a.x86_order else "wave"},
        "roofline": roof,
        "build": build_stats(graph),
    }

    if rank == 0 and not a.no_recall and a.queries:
        truth = brute_force_topk(torch, a.metric, x_dev, q_dev, 10)
        qc, qh = H.encode_vectors(metric, q_dev.cpu().numpy())
        t1 = time.perf_counter()
        ids, dists, counts = builder.search_knn(qc, qh, k=10, ef_search=a.ef_search)
        ts = time.perf_counter() - t1
        out["recall_at_10"] = round(recall_at_k(ids, counts, truth), 4)
        out["search"] = {"queries": a.queries, "ef_search": a.ef_search,
                         "qps_incl_transfers": round(a.queries / ts, 1)}
        # throughput of the batched searcher on a large query set (same distribution, no ground truth)
        nqs = 32768
        qb = gen_data(torch, nqs, a.dim, a.data, a.seed, dev, queries=True).cpu().numpy()
        qbc, qbh = H.encode_vectors(metric, qb)
        builder.search_knn(qbc[:4096], qbh[:4096], k=10, ef_search=a.ef_search)  # warm-up
        t1 = time.perf_counter()
        builder.search_knn(qbc, qbh, k=10, ef_search=a.ef_search)
        out["search"]["qps_batch_32768_incl_transfers"] = round(nqs / (time.perf_counter() - t1), 1)

    # ---- CPU baseline (rank 0, N=1 only): the oracle = port of the reference algorithm ----
    if rank == 0 and world == 1 and not a.no_cpu:
        from oracle import orc
        from tests.conftest import draw_levels
        cores = os.cpu_count() or 1
        ns = a.n if a.cpu_full else min(a.cpu_sample, a.n)
        if ns <= 0:  # calibrate on 4000 items, then size the sample for ~cpu_seconds of wall time
            nc = min(4000, a.n)
            dsc = orc.Dataset(metric, a.dim, np.arange(nc, dtype=np.uint32), items.codes[:nc],
                              items.headers[:nc], draw_levels(nc, a.M, a.seed))
            t1 = time.perf_counter()
            orc.build(dsc, M=a.M, M0=M0, ef=a.ef, order=orc.ORDER_X86, threads=cores)
            rate = nc / max(time.perf_counter() - t1, 1e-3)
            # the per-insert cost grows with the index: assume half the calibrated rate
            ns = int(min(a.n, max(10000, min(200000, 0.5 * rate * a.cpu_seconds))))
        lv = draw_levels(ns, a.M, a.seed)
        ds = orc.Dataset(metric, a.dim, np.arange(ns, dtype=np.uint32), items.codes[:ns],
                         items.headers[:ns], lv)
        t1 = time.perf_counter()
        og = orc.build(ds, M=a.M, M0=M0, ef=a.ef, order=orc.ORDER_X86, threads=cores)
        tc = time.perf_counter() - t1
        out["cpu_baseline"] = {
            "value": round(ns / tc, 1), "unit": "vectors/s", "cores": cores, "kind": "port",
            "size": "port-full" if ns == a.n else "port-sample",
            "sample": (f"all {a.n} vectors" if ns == a.n else
                       f"first {ns} of the {a.n} vectors (a smaller index is cheaper per insert than the "
                       f"full one, so this flatters the CPU; --cpu-full times all of them)") +
                      f", same params, vectors in RAM, {cores} threads (rayon-like), AVX2+FMA kernels",
            "seconds": round(tc, 2)}
        full = os.path.join(ROOT, "profiles", "r01_c2_full_scale_recall_parity_final.json")
        if default_c2 and ns != a.n and os.path.exists(full):  # one-off measurement of the same baseline at full size
            with open(full) as f:
                fj = json.load(f)
            out["cpu_baseline"]["full_size_run"] = {
                "vectors_per_s": round(fj["cpu_vec_per_s"], 1), "seconds": round(fj["cpu_build_s"], 1),
                "cores": fj["cores"], "recall_at_10": fj["recall_cpu_built_cpu_search"],
                "source": "profiles/r01_c2_full_scale_recall_parity_final.json (scripts/recall_parity_full.py)"}
        # recall parity on the sample: CPU-built vs GPU-built graph, both searched by the oracle
        if not a.no_recall and a.queries:
            sub = H.ItemSet(metric, a.dim, ds.ids, ds.codes, ds.headers, lv)
            gg = H.build(sub, M=a.M, M0=M0, ef_construction=a.ef, batch_frac=a.batch_frac,
                         batch_max=a.batch_max, device=local_rank)
            truth_s = brute_force_topk(torch, a.metric, x_dev[:ns], q_dev, 10)
            r_cpu = recall_at_k(*orc.search(ds, og, qc, qh, k=10, ef_search=a.ef_search,
                                            threads=cores)[0::2], truth_s)
            r_gpu = recall_at_k(*orc.search(ds, gg, qc, qh, k=10, ef_search=a.ef_search,
                                            threads=cores)[0::2], truth_s)
            out["recall_parity_on_sample"] = {"n": ns, "cpu_built": round(r_cpu, 4),
                                              "gpu_built": round(r_gpu, 4)}
    # ---- a second distribution in the same line: the headline data (well-separated clusters) is the
    # friendly case for the memory system; `overlap` (overlapping clusters on a 32-d manifold) is what
    # embedding collections look like.  value_alt / roofline_alt / recall_at_10_alt, same parameters.
    alt = a.alt_data or ("overlap" if default_c2 else "none")
    if world == 1 and alt != "none" and alt != a.data:
        builder.close()
        del x_dev, x, items
        xa_dev = gen_data(torch, a.n, a.dim, alt, a.seed, dev)
        qa_dev = gen_data(torch, a.queries, a.dim, alt, a.seed, dev, queries=True) if a.queries else None
        items_a = H.ItemSet.from_f32(metric, xa_dev.cpu().numpy())
        builder, driver, ga, dta = timed_builds(items_a, a.alt_steps, 1)
        out["value_alt"] = round(a.n * a.alt_steps / dta, 1) if a.alt_steps else 0.0
        out["alt"] = {"data": f"{alt} synthetic vectors (bench.py gen_data), same n / dim / M / efC",
                      "steps": a.alt_steps, "ms_per_step": round(1e3 * dta / max(1, a.alt_steps), 2),
                      "build": build_stats(ga)}
        out["roofline_alt"] = roofline_of(ga, dta, a.alt_steps)
        if not a.no_recall and a.queries:
            truth_a = brute_force_topk(torch, a.metric, xa_dev, qa_dev, 10)
            qca, qha = H.encode_vectors(metric, qa_dev.cpu().numpy())
            ida, _, cna = builder.search_knn(qca, qha, k=10, ef_search=a.ef_search)
            out["recall_at_10_alt"] = round(recall_at_k(ida, cna, truth_a), 4)
    if world > 1:  # replicas must be bit-identical: compare a checksum of the exported graph
        import zlib
        cs = zlib.crc32(graph.nbrs.tobytes()) ^ zlib.crc32(graph.offsets.tobytes())
        tc = torch.tensor([cs, -cs], dtype=torch.int64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        out["replicas_identical"] = bool(tc[0].item() == cs and -tc[1].item() == cs)
        out["n_collectives"] = driver.n_collectives
    if rank == 0:
        print(json.dumps(out), flush=True)
    builder.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
