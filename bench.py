#!/usr/bin/env python3
"""bench.py — headline benchmark: HNSW build throughput (vectors indexed / s) + recall@10.

Workload (BASELINE.json configs[1], "C2"): 1M x 768 f32 Cosine, M=16 (M0=32), ef_construction=100,
vectors resident in HBM before the timed region.  One "step" = one complete build of the index
(graph reset -> every batch searched, pruned, linked -> records exported to the host).

  python bench.py --gpus N --steps K --warmup W

N > 1 runs the item-sharded multi-GPU build, one rank per GPU over RCCL:
  * under torch.distributed.run (RANK / WORLD_SIZE in the environment) every process is one rank;
  * started plainly (`python bench.py --gpus N`), this process launches the N ranks itself — fresh child
    processes of this script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, before
    anything here touches a GPU — relays rank 0's JSON line and fails if a rank fails;
  * `--native`: ONE process, hny_multi_builder (one host thread + one replica per GPU, hny_multi.cpp),
    i.e. what a Rust Writer::build would call.
`--backend gloo` is the test mode: ranks may share one GPU, the exchange is staged through the host.

Prints ONE JSON line on rank 0 (see the contract in the task statement) carrying `roofline`
(dominant kernel = k_walk, HIP events on the build stream) and `cpu_baseline` (the CPU oracle, i.e.
a port of the reference algorithm — the real hannoy crate cannot be built here: no Rust, no LMDB).
`--alt-data overlap` measures a second distribution in the same run (three more builds; off by default).
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SORT_BY_CLUSTER = False  # --sort-by-cluster


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
    p.add_argument("--data", default="clustered",
                   help="clustered | overlap | manifold | uniform | lat<k>[i<iso>][s<spread>] (gen_data)")
    p.add_argument("--alt-data", default=None, choices=["overlap", "uniform", "clustered", "none"],
                   help="second distribution measured in the same run (value_alt / roofline_alt / recall_at_10_alt): "
                        "one warm-up + --alt-steps builds.  Default: overlap for the headline C2 workload, else none")
    p.add_argument("--alt-steps", type=int, default=1)
    p.add_argument("--cpu-full", action="store_true",
                   help="time the CPU baseline on ALL vectors whatever it takes (C4: ~4 minutes)")
    p.add_argument("--cpu-sample-only", action="store_true",
                   help="never time the full index on the CPU, even where it fits --cpu-full-budget")
    p.add_argument("--cpu-full-budget", type=float, default=120.0,
                   help="the CPU baseline is timed on ALL vectors when the (pessimistic) calibration predicts at most "
                        "this many seconds (C2: 20 s measured on 16 cores, predicted 30 - 50), else on a bounded sample "
                        "of about --cpu-seconds")
    p.add_argument("--batch-frac", type=float, default=0.0)
    p.add_argument("--batch-max", type=int, default=0)
    p.add_argument("--queries", type=int, default=1000)
    p.add_argument("--ef-search", type=int, default=100)
    p.add_argument("--cpu-sample", type=int, default=0,
                   help="items in the CPU baseline sample; 0 = calibrate for about --cpu-seconds of wall time")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    p.add_argument("--cpu-threads", type=int, default=0,
                   help="threads of the CPU baseline; 0 = the best of the committed sweep "
                        "(profiles/r04_cpu_baseline_thread_sweep.json), else what the cgroup's CPU quota grants")
    p.add_argument("--graph-cache-gb", type=float, default=8.0,
                   help="hny_set_graph_cache: released export arrays are recycled by the next build of the loop (what "
                        "a resident service would set; 0 = the library's default: off)")
    p.add_argument("--no-cpu", action="store_true")
    p.add_argument("--no-recall", action="store_true")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--x86-order", action="store_true",
                   help="strict mode: f32 distances in the reference's x86 summation order")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo: test mode, ranks may share one GPU, exchange staged through the host")
    p.add_argument("--native", action="store_true",
                   help="--gpus N inside ONE process: hny_multi_builder (one host thread + replica per GPU, "
                        "RCCL all-gathers on the builders' streams) instead of one process per GPU")
    p.add_argument("--launch-timeout", type=float, default=3000.0,
                   help="self-launched ranks (--gpus N without torchrun) are stopped after this many seconds")
    p.add_argument("--out", default=None, help="also write the JSON line to this file")
    p.add_argument("--sort-by-cluster", action="store_true",
                   help="experiment: clustered data with ids in cluster order (neighbours in the graph then have "
                        "neighbouring ids: what a locality-ordered id space would give the visited bitsets)")
    p.add_argument("--rendezvous-only", action="store_true",
                   help="launch check without a GPU: the ranks meet over gloo and rank 0 reports who came; no build")
    return p.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes of this very
    script (the environment torch.distributed.run would give them), relay rank 0's JSON line, fail if any
    rank fails or hangs.  Nothing in this process has touched a GPU (torch is not even imported)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus),
                   LOCAL_WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HNY_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL between processes)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    deadline = time.time() + a.launch_timeout
    failed = None
    pending = set(range(a.gpus))
    import threading
    out0 = []
    rd = threading.Thread(target=lambda: out0.extend(procs[0].stdout.read().decode("utf-8", "replace").splitlines()))
    rd.start()
    while pending and failed is None:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is not None:
                pending.discard(r)
                if rc != 0:
                    failed = f"rank {r} exited with code {rc}"
        if time.time() > deadline:
            failed = f"ranks {sorted(pending)} still running after {a.launch_timeout:.0f} s"
        time.sleep(0.05)
    if failed:
        for p_ in procs:  # the exact processes started above
            if p_.poll() is None:
                p_.kill()
    for p_ in procs:
        p_.wait()
    rd.join()
    lines = [ln for ln in out0 if ln.startswith("{") and '"metric"' in ln]
    for ln in out0:
        if ln not in lines:
            print(ln, file=sys.stderr)
    if failed or not lines:
        print(f"bench.py --gpus {a.gpus}: {failed or 'rank 0 printed no result'}", file=sys.stderr)
        raise SystemExit(1)
    print(lines[-1], flush=True)


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
    m_lat = re.fullmatch(r"lat(\d+)(?:i([0-9.]+))?(?:s([0-9.]+))?", kind)
    if kind in ("overlap", "manifold") or m_lat:
        # embedding-like: 1 024 OVERLAPPING clusters on a 32-d manifold embedded in `dim` dimensions
        # (latent centres N(0, 1), points centre + 0.5 N(0, 1): the spread inside a cluster is half the
        # spread of the centres, so neighbourhoods cross cluster borders) + 0.05 isotropic noise
        # ("manifold": 0.01 — at 5M / 10M points the manifold is populated so densely that 0.05 of isotropic
        # noise per component, not the manifold, decides who the ten nearest neighbours are)
        # "lat<k>[i<iso>][s<spread>]": the same family with a k-d manifold (the local intrinsic dimension of real
        # embedding sets is 10 - 20, not 32), iso noise (default 0.01) and cluster spread (default 0.5) as given
        iso = 0.05 if kind == "overlap" else 0.01
        k = min(32, dim)
        spread = 0.5
        if m_lat:
            k = min(int(m_lat.group(1)), dim)
            iso = float(m_lat.group(2)) if m_lat.group(2) else 0.01
            spread = float(m_lat.group(3)) if m_lat.group(3) else 0.5
        centres = torch.randn((1024, k), generator=g, device=device, dtype=torch.float32)
        basis = torch.linalg.qr(torch.randn((dim, k), generator=g, device=device, dtype=torch.float32))[0].T
        which = torch.randint(0, 1024, (n,), generator=gn, device=device)
        out = torch.empty((n, dim), device=device, dtype=torch.float32)
        for lo in range(0, n, 1 << 18):  # in slices: bounded temporaries
            hi = min(n, lo + (1 << 18))
            z = centres[which[lo:hi]] + spread * torch.randn((hi - lo, k), generator=gn, device=device)
            out[lo:hi] = z @ basis + iso * torch.randn((hi - lo, dim), generator=gn, device=device)
        return out
    # 1024-centre Gaussian mixture, centres U(-1,1), sigma 0.15: well separated
    centres = torch.rand((1024, dim), generator=g, device=device, dtype=torch.float32) * 2 - 1
    which = torch.randint(0, 1024, (n,), generator=gn, device=device)
    if SORT_BY_CLUSTER and not queries:
        which = torch.sort(which).values
    return centres[which] + 0.15 * torch.randn((n, dim), generator=gn, device=device, dtype=torch.float32)


def brute_force_topk(torch, metric, data, queries, k, chunk=1 << 20):
    """Exact top-k under the reference's metric definition (plumbing, torch on the GPU).

    The items are scored in pieces of `chunk` rows and the per-piece top-(k+32) merged, then re-ranked in f64.
    Rounds 1-4 scored all items with ONE f32 GEMM per 256 queries: beyond 2^32 bytes of scores (n > 4.19 M: C4's
    10M x 128, C5's 5M codes) that GEMM / top-k returns garbage for part of the matrix on this stack, and the
    "truth" held ~12 % wrong neighbours — the recall plateau of 0.84 - 0.88 "whatever the beam width" that rounds
    3-4 reported for every 10M index, GPU- or CPU-built, was this function (scripts/r5_truth_diag.py: the same
    index scores 1.0 against an exact f64 truth and 0.8799 against the old one)."""
    nq, n = queries.shape[0], data.shape[0]
    kk = min(n, k + 32)
    dev = data.device
    if metric == "cosine":
        qn = queries / queries.norm(dim=1, keepdim=True)
    elif metric == "hamming":
        qn = (queries > 0).float()
    else:
        qn = queries
    best_v = torch.full((nq, 0), 0.0, device=dev)
    best_i = torch.zeros((nq, 0), dtype=torch.int64, device=dev)
    for lo in range(0, n, chunk):
        xb = data[lo:lo + chunk]
        if metric == "cosine":
            sc = -(qn @ (xb / xb.norm(dim=1, keepdim=True)).T)  # ascending = closer, like the distances below
        elif metric == "euclidean":
            sc = torch.cdist(qn, xb)
        elif metric == "manhattan":
            sc = torch.cdist(qn, xb, p=1)
        else:  # hamming on the Binary codec bits (x > 0)
            db = (xb > 0).float()
            sc = qn @ (1 - db).T + (1 - qn) @ db.T
        v, i = torch.topk(sc, min(kk, xb.shape[0]), dim=1, largest=False)
        cv, ci = torch.cat([best_v, v], 1), torch.cat([best_i, i + lo], 1)
        o = torch.topk(cv, min(kk, cv.shape[1]), dim=1, largest=False).indices
        best_v, best_i = torch.gather(cv, 1, o), torch.gather(ci, 1, o)
    if metric == "hamming":  # integer distances: exact already; ties stay in (distance, id) order
        key = best_v.double() * float(n) + best_i.double()
        o = torch.argsort(key, dim=1)[:, :k]
        return torch.gather(best_i, 1, o).cpu().numpy()
    # exact re-rank of the kk survivors in f64 (f32 scores of neighbours 10 and 11 can tie at 10M items)
    out = []
    for q0 in range(0, nq, 64):
        cand = best_i[q0:q0 + 64]
        xd = data[cand.reshape(-1)].double().reshape(cand.shape[0], cand.shape[1], -1)
        qd = queries[q0:q0 + 64].double()[:, None, :]
        if metric == "cosine":
            d = -(xd * qd).sum(2) / (xd.norm(dim=2) * qd.norm(dim=2))
        elif metric == "euclidean":
            d = ((xd - qd) ** 2).sum(2)
        else:
            d = (xd - qd).abs().sum(2)
        o = torch.argsort(d, dim=1, stable=True)[:, :k]
        out.append(torch.gather(cand, 1, o))
    return torch.cat(out).cpu().numpy()


def recall_at_k(found, counts, truth):
    hit = 0
    for i in range(truth.shape[0]):
        hit += len(set(found[i, :counts[i]].tolist()) & set(truth[i].tolist()))
    return hit / truth.size


def main():
    global SORT_BY_CLUSTER
    a = parse()
    SORT_BY_CLUSTER = a.sort_by_cluster
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and not a.native and "RANK" not in os.environ:
        return self_launch(a)  # before anything touches a GPU
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.native and world > 1:
        raise SystemExit("--native is one process for all GPUs: start it without torchrun")
    if not a.native and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree")
    launcher = ("native" if a.native else "single" if world == 1 else
                "self" if os.environ.get("HNY_BENCH_SELF_LAUNCHED") else "torchrun")
    import torch
    import torch.distributed as dist
    if a.rendezvous_only:  # the launch path alone (tests/test_bench_launch.py): no GPU, no build
        ranks = [rank]
        if world > 1:
            dist.init_process_group("gloo")
            ranks = [None] * world
            dist.all_gather_object(ranks, rank)
        if rank == 0:
            print(json.dumps({"metric": "rendezvous only (no build)", "n_gpus": a.gpus, "launcher": launcher,
                              "ranks_seen": dist.get_world_size() if world > 1 else 1, "ranks": ranks}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    n_dev = torch.cuda.device_count()
    if world > 1 and a.backend == "nccl" and world > n_dev:
        raise SystemExit(f"--gpus {world} over RCCL needs {world} GPUs, {n_dev} visible (--backend gloo: test mode)")
    local_rank %= n_dev
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

    class NativeDriver:  # --native: the resident multi-builder behind the C ABI
        def __init__(self, mb):
            self.mb = mb

        @property
        def n_collectives(self):
            return self.mb.n_collectives

    def timed_builds(items_, steps, warmup):
        """W untimed + K timed full builds (graph reset -> every batch -> records exported)."""
        kw = dict(M=a.M, M0=M0, ef_construction=a.ef, seed=a.seed, batch_frac=a.batch_frac,
                  batch_max=a.batch_max, x86_order=a.x86_order)
        if a.native:
            # HNY_MGPU_SHIM=1 (test mode): the ranks share GPU 0, copies instead of RCCL
            devs = [0] * a.gpus if os.environ.get("HNY_MGPU_SHIM") else list(range(a.gpus))
            mb = H.MultiBuilder(items_, devices=devs, **kw)
            mb.set_profiling(True)
            b, drv = mb.replica(0), NativeDriver(mb)
            b._mb = mb  # keeps the replicas alive as long as the search handle
            step = mb.run
        else:
            b = H.Builder(items_, device=local_rank, **kw)
            b.set_profiling(True)
            drv = multigpu.Driver(b, torch, dist if world > 1 else None, rank, world, dev,
                                  host_staged=(a.backend == "gloo"))

            def step():
                b.reset()
                drv.run()
                return b.finish()

        first_ms = None
        for w_ in range(warmup):
            tw = time.perf_counter()
            step()
            if w_ == 0:  # the first build on a fresh builder: what a one-off hny_build pays beyond the upload
                first_ms = round(1e3 * (time.perf_counter() - tw), 2)
        timed_builds.first_build_ms = first_ms
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g = None
        for _ in range(steps):
            g = None  # the previous build's records are released before the next export (hny_graph_free)
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

    def roofline_of(g, dt_, steps, data_kind):
        """dominant kernel (k_walk).  Two byte counts over ONE time (HIP events on the builder's stream
        around every k_walk dispatch; `launches` = k_walk dispatches, what rocprofv3 counts):
          algorithmic = walk distance evaluations x (row + header) bytes — credits a row that
                        an L2 hit served, so it can exceed the HBM peak;
          counted     = L2->fabric bytes of the k_walk dispatches from the committed rocprofv3 --pmc passes
                        of this same workload (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md HBM) — an
                        upper bound on HBM traffic (Infinity-Cache hits are counted), includes the bytes
                        the walk moves beyond the rows (lists, visited sets).
        achieved = min(algorithmic, counted) / time: useful bytes that really crossed the fabric, never
        above what the memory system delivered; frac = achieved / 8 TB/s."""
        # N > 1: one GPU's share — rank 0's own walks over rank 0's k_walk time (one process per GPU: the
        # counters of rank 0's graph ARE its shard; --native sums the shards, so the average share is used)
        wb = g.n_evals_walk * bytes_per_eval / (a.gpus if a.native else 1)
        if g.t_walk_kernels_s <= 0:
            return None
        launches = max(1, int(g.n_walk_launches))
        alg = wb / g.t_walk_kernels_s / 1e9
        r = {"bound": "hbm", "achieved": round(alg, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(alg / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "k_walk",
             "achieved_basis": "algorithmic bytes / k_walk time (no PMC profile committed for this workload)",
             "achieved_algorithmic": round(alg, 1), "frac_algorithmic": round(alg / HBM_PEAK_GBS, 4),
             "launches": launches, "avg_launch_ms": round(1e3 * g.t_walk_kernels_s / launches, 4),
             "algorithmic_bytes_per_launch": int(wb / launches), "bytes_per_eval": bytes_per_eval}
        if a.gpus > 1:
            r["scope"] = "one GPU's shard of the walks (rank 0)"
        if steps:  # the same bytes over the WHOLE step (prune, link ops, export included)
            r["frac_whole_build"] = round(wb / (dt_ / steps) / 1e9 / HBM_PEAK_GBS, 4)
        pmc_name, pj = pmc_profile(data_kind)
        pk = pj.get("k_walk") if pj else None
        if pk and world == 1 and not a.native:
            import hashlib
            total = pk["hbm_read_bytes_corrected_x2"] + pk["hbm_write_bytes"]
            # the PMC pass ran one build of this same workload: its bytes over ITS dispatches
            r["traffic"] = int(total / max(1, pk["launches"]))
            counted = r["traffic"] * launches / g.t_walk_kernels_s / 1e9
            r["achieved_counted"] = round(counted, 1)
            r["achieved"] = round(min(alg, counted), 1)
            r["frac"] = round(min(alg, counted) / HBM_PEAK_GBS, 4)
            r["achieved_basis"] = ("min(algorithmic, counted L2->fabric) bytes / k_walk time; counted = rocprofv3 --pmc "
                                   "2 x FETCH_SIZE + WRITE_SIZE of the k_walk dispatches (Infinity-Cache hits included)")
            if alg >= counted:
                r["reuse"] = round(alg / counted, 3)  # evaluated rows served by L2 instead of the fabric
            else:
                r["wasted_traffic_ratio"] = round(counted / alg, 3)  # bytes moved beyond the rows
            if pk.get("tcc_hit") is not None and pk.get("tcc_miss") is not None and pk["tcc_hit"] + pk["tcc_miss"] > 0:
                # the same round's TCC_HIT_sum / TCC_MISS_sum pass over the k_walk dispatches: why algorithmic
                # bytes can exceed what crossed the fabric
                r["l2_hit_rate"] = round(pk["tcc_hit"] / (pk["tcc_hit"] + pk["tcc_miss"]), 4)
            r["traffic_read_bytes"] = int(pk["hbm_read_bytes_corrected_x2"] / max(1, pk["launches"]))
            r["traffic_write_bytes"] = int(pk["hbm_write_bytes"] / max(1, pk["launches"]))
            with open(os.path.join(ROOT, "hannoy_amd", "csrc", "hny_kernels.hip"), "rb") as f:
                sha = hashlib.sha1(f.read()).hexdigest()
            r["traffic_source"] = (f"profiles/{pmc_name}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                   f"workload (scripts/r3_pmc.sh), summed over the {pk['launches']} k_walk "
                                   f"dispatches of one build, per dispatch")
            r["traffic_stale"] = pj.get("kernel_source_sha1") != sha  # kernels changed since the PMC passes
        return r

    def roofline_issue_of(g, data_kind):
        """Rows of at most 512 B: the walk is bound by instruction issue, not by bytes.  The record
        prices it against the issue ports: one vector and one scalar instruction per SIMD every 4 cycles, 1 024
        SIMDs, 2.4 GHz.  Instructions per evaluation come from the committed rocprofv3 --pmc SQ_* passes of
        this workload (scripts/r4_sq.sh -> profiles/r04_sq_<workload>.json), the rate from THIS run's k_walk time."""
        if row_bytes > 512 or g.t_walk_kernels_s <= 0 or world != 1 or a.native:
            return None
        if a.batch_frac or a.batch_max or a.x86_order:
            return None
        key = f"{a.n}x{a.dim}_{a.metric}_M{a.M}_ef{a.ef}_{data_kind}"
        import glob
        names = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_sq_{key}.json")), reverse=True)
        if not names:
            return None
        with open(names[0]) as f:
            sj = json.load(f)
        kw_ = sj.get("k_walk")
        if not kw_ or not kw_.get("evals"):
            return None
        import hashlib
        with open(os.path.join(ROOT, "hannoy_amd", "csrc", "hny_kernels.hip"), "rb") as f:
            sha = hashlib.sha1(f.read()).hexdigest()
        v_pe, s_pe = kw_["SQ_INSTS_VALU"] / kw_["evals"], kw_["SQ_INSTS_SALU"] / kw_["evals"]
        peak = 1024 * 2.4e9 / 4 / 1e9  # G instructions / s per issue port
        ach_v = v_pe * g.n_evals_walk / g.t_walk_kernels_s / 1e9
        ach_s = s_pe * g.n_evals_walk / g.t_walk_kernels_s / 1e9
        return {"bound": "valu-issue", "kernel": "k_walk", "insts_per_eval": round(v_pe, 2),
                "scalar_insts_per_eval": round(s_pe, 2), "achieved": round(ach_v, 1), "peak": round(peak, 1),
                "unit": "G vector instructions/s", "frac": round(ach_v / peak, 4),
                "frac_scalar_port": round(ach_s / peak, 4),
                "wave_cycles_waiting": round(kw_.get("SQ_WAIT_ANY", 0) / max(1.0, kw_.get("SQ_WAVE_CYCLES_p2", 1.0)), 3),
                "wave_cycles_issue_stalled": round(kw_.get("SQ_WAIT_INST_ANY", 0) / max(1.0, kw_.get("SQ_WAVE_CYCLES_p2", 1.0)), 3),
                "source": f"profiles/{os.path.basename(names[0])} (rocprofv3 --pmc SQ_*, scripts/r4_sq.sh)",
                "stale": sj.get("kernel_source_sha1") != sha}

    def pmc_profile(data_kind):
        """the committed PMC summary of this exact workload (newest round first), or (None, None)"""
        if a.batch_frac or a.batch_max or a.x86_order:
            return None, None
        key = f"{a.n}x{a.dim}_{a.metric}_M{a.M}_ef{a.ef}_{data_kind}"
        import glob
        names = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_hbm_{key}.json")), reverse=True)
        if not names and key == "1000000x768_cosine_M16_ef100_clustered":
            names = [os.path.join(ROOT, "profiles", "r02_c2_pmc_hbm.json")]
        for nm in names:
            if os.path.exists(nm):
                with open(nm) as f:
                    return os.path.basename(nm), json.load(f)
        return None, None

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
                "tie_pool_overflow": int(g.n_tie_pool_overflow)}

    H.set_graph_cache(int(a.graph_cache_gb * (1 << 30)))
    builder, driver, graph, dt = timed_builds(items, a.steps, a.warmup)
    value = a.n * a.steps / dt if a.steps else 0.0
    roof = roofline_of(graph, dt, a.steps, a.data)
    roof_issue = roofline_issue_of(graph, a.data)
    default_c2 = (a.n == 1_000_000 and a.dim == 768 and a.metric == "cosine" and a.M == 16
                  and a.ef == 100 and a.data == "clustered" and not a.batch_frac and not a.batch_max
                  and world == 1 and not a.native and not a.x86_order)

    known = {(1_000_000, 768, "cosine", 16, 100): "C2", (1_000_000, 768, "euclidean", 32, 200): "C3",
             (10_000_000, 128, "cosine", 16, 100): "C4 (on %d GPU)" % a.gpus,
             (5_000_000, 1024, "hamming", 16, 64): "C5 (on %d GPU)" % a.gpus}
    cfg_name = known.get((a.n, a.dim, a.metric, a.M, a.ef), "custom")
    shape = f"{a.n // 1_000_000}M" if a.n % 1_000_000 == 0 and a.n else str(a.n)
    out = {
        "metric": f"vectors indexed/sec (build) + recall@10, {shape} x {a.dim} {a.metric.capitalize()} "
                  f"M={a.M} efC={a.ef}",
        "value": round(value, 1), "unit": "vectors/s", "n_gpus": a.gpus, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(1e3 * dt / max(1, a.steps), 2),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if metric < H.HAMMING else "u64-popcount", "data": "synthetic",
        "config": {"workload": f"{cfg_name}: {a.n} x {a.dim} {a.metric}, M={a.M} M0={M0} efC={a.ef}, "
                               f"{a.data} synthetic vectors resident in HBM, 1 step = 1 full build",
                   "n": a.n, "dim": a.dim, "M": a.M, "M0": M0, "ef_construction": a.ef,
                   "batch_frac": builder.opts.batch_frac or 1.0,
                   "batch_max": builder.opts.batch_max or H.default_batch_max(a.n),
                   "parallelism": f"item-sharded search x{a.gpus}, replicated graph",
                   "distance_order": "x86 (strict)" if a.x86_order else "wave"},
        # untimed warm-up 1: a fresh builder's first build, nothing to recycle yet — what a one-off build takes
        # beyond the upload; the timed steps recycle the export arrays of the graph released before them
        "first_build_ms": getattr(timed_builds, "first_build_ms", None),
        "graph_cache_bytes": int(a.graph_cache_gb * (1 << 30)),
        "roofline": roof,
        "build": build_stats(graph),
        # who ran: ranks that really took part (dist.get_world_size() / replicas of the multi-builder),
        # the GPU each one used, and how they were started
        "ranks_seen": (builder._mb.world if a.native else dist.get_world_size() if world > 1 else 1),
        "launcher": launcher,
    }
    if roof_issue:
        out["roofline_issue"] = roof_issue
    if a.native:
        out["devices"] = [int(d) for d in builder._mb.opts.devices[:a.gpus]]
    elif world > 1:
        dv = [None] * world
        dist.all_gather_object(dv, int(local_rank))
        out["devices"] = dv
    else:
        out["devices"] = [int(local_rank)]
    if a.gpus > 1 and a.backend == "gloo" and not a.native:
        out["note_backend"] = "gloo test mode: ranks may share a GPU and exchange through the host — not a scaling measurement"

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
    if rank == 0 and a.gpus == 1 and not a.no_cpu:
        from oracle import orc

        def draw_levels(n_, M_, seed_):  # the levels hny_build itself would draw (rand 0.8.5 StdRng, hnsw.rs:113-119)
            return H.draw_levels(seed_, M_, n_)
        # threads: the best of the committed sweep on this kind of box, else what the CPU quota grants
        # (a GPU box shows 256 logical CPUs and grants 16 CPUs' worth of time)
        cores = a.cpu_threads or orc.host_threads()
        sweep_path = os.path.join(ROOT, "profiles", "r04_cpu_baseline_thread_sweep.json")
        sweep_note = None
        if not a.cpu_threads and os.path.exists(sweep_path):
            with open(sweep_path) as f:
                sw = json.load(f)
            if sw.get("best_threads") and sw.get("host_threads") == orc.host_threads():
                cores = int(sw["best_threads"])
                sweep_note = "profiles/r04_cpu_baseline_thread_sweep.json"
        ns = a.n if a.cpu_full else min(a.cpu_sample, a.n)
        if ns <= 0:  # calibrate on 4000 items, then: everything if that fits the budget, else a sample
            nc = min(4000, a.n)
            dsc = orc.Dataset(metric, a.dim, np.arange(nc, dtype=np.uint32), items.codes[:nc],
                              items.headers[:nc], draw_levels(nc, a.M, a.seed))
            t1 = time.perf_counter()
            orc.build(dsc, M=a.M, M0=M0, ef=a.ef, order=orc.ORDER_X86, threads=cores)
            rate = nc / max(time.perf_counter() - t1, 1e-3)
            # the per-insert cost grows with the index: measured full-size rates are 0.3 - 0.5 of the
            # 4000-item rate (C2: 22 s for 1M vectors where the calibration alone says 7 s)
            if not a.cpu_sample_only and a.n / (0.3 * rate) <= a.cpu_full_budget:
                ns = a.n
            else:
                ns = int(min(a.n, max(10000, 0.5 * rate * a.cpu_seconds)))
        lv = draw_levels(ns, a.M, a.seed)
        ds = orc.Dataset(metric, a.dim, np.arange(ns, dtype=np.uint32), items.codes[:ns],
                         items.headers[:ns], lv)
        t1 = time.perf_counter()
        og = orc.build(ds, M=a.M, M0=M0, ef=a.ef, order=orc.ORDER_X86, threads=cores)
        tc = time.perf_counter() - t1
        out["cpu_baseline"] = {
            "value": round(ns / tc, 1), "unit": "vectors/s", "cores": cores, "kind": "port",
            "host": {"logical_cpus": os.cpu_count(), "cpu_quota": orc.host_threads(), "threads_chosen_from": sweep_note},
            "size": "port-full" if ns == a.n else "port-sample",
            "sample": (f"all {a.n} vectors" if ns == a.n else
                       f"first {ns} of the {a.n} vectors (a smaller index is cheaper per insert than the "
                       f"full one, so this flatters the CPU; --cpu-full times all of them)") +
                      f", same params, vectors in RAM, {cores} threads (rayon-like), AVX2+FMA kernels",
            "seconds": round(tc, 2)}
        # the committed full-size comparison of this workload, if there is one (scripts/r4_recall_parity.py: GPU-built vs
        # CPU-built index at the BASELINE size, same levels, same searcher, exact ground truth)
        short = cfg_name.split(" ")[0].lower()
        f5 = os.path.join(ROOT, "profiles", f"r05_{short}_recall_sweep_{a.data}_cpu3.json")
        if short in ("c2", "c3", "c4", "c5") and os.path.exists(f5) and not (a.batch_frac or a.batch_max or a.x86_order):
            with open(f5) as f:
                fj = json.load(f)
            if fj.get("n") == a.n and fj.get("summary") and fj.get("cpu_builds"):
                cb = fj["cpu_builds"]
                gd = [g_ for g_ in fj["gpu_builds"] if g_.get("default")]
                out["cpu_baseline"]["full_size_runs"] = {
                    "cpu_builds": len(cb), "cores": cb[0]["threads"],
                    "vectors_per_s": [c_["vec_per_s"] for c_ in cb], "seconds": [c_["build_s"] for c_ in cb],
                    "queries": fj.get("queries"),
                    "recall_at_10_by_ef_search": {
                        e: {"cpu_median": v["cpu_median"], "cpu_min": v["cpu_min"], "cpu_max": v["cpu_max"],
                            "gpu_default_schedule": next((g_["recall"] for g_ in v["gpu"]
                                                          if gd and g_["batch_max"] == gd[0]["batch_max"]
                                                          and g_["batch_frac"] == gd[0]["batch_frac"]), None)}
                        for e, v in fj["summary"].items()},
                    "source": f"profiles/{os.path.basename(f5)} (scripts/r5_recall_sweep.py --config {short.upper()} "
                              f"--data {a.data} --cpu-builds {len(cb)}; exact f64 ground truth)"}
        fp = os.path.join(ROOT, "profiles", f"r04_{short}_recall_parity" + ("" if a.data == "overlap" else "_" + a.data) + ".json")
        if (short in ("c2", "c3") and os.path.exists(fp) and not os.path.exists(f5)
                and not (a.batch_frac or a.batch_max or a.x86_order)):  # (r04's C4 files: ground truth broken at 10M)
            with open(fp) as f:
                fj = json.load(f)
            if fj.get("n") == a.n and "recall_at_10" in fj:
                out["cpu_baseline"]["full_size_run"] = {
                    "vectors_per_s": fj["cpu_vec_per_s"], "seconds": fj["cpu_build_s"], "cores": fj.get("cpu_threads"),
                    "recall_at_10_cpu_built": fj["recall_at_10"]["100"]["cpu_built"],
                    "recall_at_10_gpu_built": fj["recall_at_10"]["100"]["gpu_built"],
                    "recall_at_10_by_ef_search": fj["recall_at_10"],
                    "source": f"profiles/{os.path.basename(fp)} (scripts/r4_recall_parity.py --config {short.upper()} --data {a.data})"}
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
    if a.gpus == 1 and not a.native and alt != "none" and alt != a.data:
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
        out["roofline_alt"] = roofline_of(ga, dta, a.alt_steps, alt)
        if not a.no_recall and a.queries:
            truth_a = brute_force_topk(torch, a.metric, xa_dev, qa_dev, 10)
            qca, qha = H.encode_vectors(metric, qa_dev.cpu().numpy())
            ida, _, cna = builder.search_knn(qca, qha, k=10, ef_search=a.ef_search)
            out["recall_at_10_alt"] = round(recall_at_k(ida, cna, truth_a), 4)
    if a.native and a.gpus > 1:
        out["n_collectives"] = driver.n_collectives
        # HNY_MGPU_VERIFY=1 makes every replica export and compares them inside hny_multi_builder_run
        out["replicas_identical"] = True if os.environ.get("HNY_MGPU_VERIFY") else None
    if world > 1:  # replicas must be bit-identical: compare a checksum of the exported graph
        import zlib
        cs = zlib.crc32(graph.nbrs.tobytes()) ^ zlib.crc32(graph.offsets.tobytes())
        tc = torch.tensor([cs, -cs], dtype=torch.int64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        out["replicas_identical"] = bool(tc[0].item() == cs and -tc[1].item() == cs)
        out["n_collectives"] = driver.n_collectives
    if rank == 0:
        line = json.dumps(out)
        if a.out:
            with open(a.out, "w") as f:
                f.write(line + "\n")
        print(line, flush=True)
    builder.close()
    if a.native:
        builder._mb.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
