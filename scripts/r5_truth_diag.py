#!/usr/bin/env python3
"""Round 5 diagnosis, part 2: is the f32 brute-force ground truth itself wrong at 10M x 128?
Exact truth = f64 matmul over all items (chunked); compared with bench.brute_force_topk (f32) and with the index."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import hannoy_amd as H
    kind = sys.argv[1] if len(sys.argv) > 1 else "lat8"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    dev = torch.device("cuda", 0)
    dim, M, M0, ef = 128, 16, 32, 100
    x_dev = bench.gen_data(torch, n, dim, kind, 42, dev)
    q_dev = bench.gen_data(torch, 1000, dim, kind, 42, dev, queries=True)
    truth32 = bench.brute_force_topk(torch, "cosine", x_dev, q_dev, 10)
    qd = q_dev.double()
    qd = qd / qd.norm(dim=1, keepdim=True)
    best_v = torch.full((1000, 10), -2.0, dtype=torch.float64, device=dev)
    best_i = torch.zeros((1000, 10), dtype=torch.int64, device=dev)
    for lo in range(0, n, 1 << 20):
        xb = x_dev[lo:lo + (1 << 20)].double()
        xb = xb / xb.norm(dim=1, keepdim=True)
        s = qd @ xb.T
        v, i = torch.topk(s, 10, dim=1)
        cv = torch.cat([best_v, v], 1)
        ci = torch.cat([best_i, i + lo], 1)
        o = torch.topk(cv, 10, dim=1).indices
        best_v, best_i = torch.gather(cv, 1, o), torch.gather(ci, 1, o)
    truth64 = best_i.cpu().numpy()
    out = {"data": kind, "n": n,
           "f32_truth_vs_exact": round(float(np.mean([len(set(truth32[i]) & set(truth64[i])) for i in range(1000)]) / 10), 4)}
    # the same f32 computation, smaller pieces: does the size of the GEMM / top-k matter?
    xn = x_dev / x_dev.norm(dim=1, keepdim=True)
    qn = q_dev / q_dev.norm(dim=1, keepdim=True)
    bv = torch.full((1000, 10), -2.0, device=dev)
    bi = torch.zeros((1000, 10), dtype=torch.int64, device=dev)
    for lo in range(0, n, 1 << 20):
        s = qn @ xn[lo:lo + (1 << 20)].T
        v, i = torch.topk(s, 10, dim=1)
        cv = torch.cat([bv, v], 1)
        ci = torch.cat([bi, i + lo], 1)
        o = torch.topk(cv, 10, dim=1).indices
        bv, bi = torch.gather(cv, 1, o), torch.gather(ci, 1, o)
    t32c = bi.cpu().numpy()
    out["f32_chunked_truth_vs_exact"] = round(float(np.mean([len(set(t32c[i]) & set(truth64[i])) for i in range(1000)]) / 10), 4)
    # one big f32 GEMM, then where do its scores differ from the exact ones?
    s = qn[:256] @ xn.T
    idx = best_i[:256]
    got = torch.gather(s, 1, idx)
    out["max_abs_err_of_big_gemm_on_true_top10"] = float((got.double() - best_v[:256]).abs().max())
    v_big, i_big = torch.topk(s, 10, dim=1)
    out["big_gemm_topk_vs_exact_first256"] = round(float(np.mean([len(set(i_big[i].tolist()) & set(truth64[i])) for i in range(256)]) / 10), 4)
    # topk of the big matrix against a manual check: max of each row
    out["big_gemm_rowmax_matches_topk"] = bool(torch.equal(s.max(dim=1).values, v_big[:, 0]))
    del s, xn
    x = x_dev.cpu().numpy()
    del x_dev
    torch.cuda.empty_cache()
    qc, qh = H.encode_vectors(H.COSINE, q_dev.cpu().numpy())
    items = H.ItemSet.from_f32(H.COSINE, x, levels=H.draw_levels(42, M, n))
    with H.Builder(items, M=M, M0=M0, ef_construction=ef) as b:
        b.run()
        b.finish()
        for e in (100, 400):
            ids, dists, cnt = b.search_knn(qc, qh, k=10, ef_search=e)
            out[f"recall_ef{e}_vs_exact"] = round(bench.recall_at_k(ids, cnt, truth64), 4)
            out[f"recall_ef{e}_vs_f32"] = round(bench.recall_at_k(ids, cnt, truth32), 4)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
