#!/usr/bin/env python3
"""Round 5 diagnosis: why does recall@10 of the 10M x 128 indexes plateau at 0.85 - 0.88 whatever ef_search?
Candidates: (a) ground truth computed in f32 (near-ties at 10M density), (b) items no search can reach
(in-degree 0 on layer 0), (c) a defect of the searcher at large slot ids.  Prints one JSON per n."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import hannoy_amd as H
    kind = sys.argv[1] if len(sys.argv) > 1 else "lat8"
    sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1000000,4000000,10000000").split(",")]
    dev = torch.device("cuda", 0)
    dim, M, M0, ef = 128, 16, 32, 100
    for n in sizes:
        x_dev = bench.gen_data(torch, n, dim, kind, 42, dev)
        q_dev = bench.gen_data(torch, 1000, dim, kind, 42, dev, queries=True)
        qc, qh = H.encode_vectors(H.COSINE, q_dev.cpu().numpy())
        truth32 = bench.brute_force_topk(torch, "cosine", x_dev, q_dev, 10)
        # f64 re-rank of the f32 top-200
        xn = x_dev / x_dev.norm(dim=1, keepdim=True)
        qn = q_dev / q_dev.norm(dim=1, keepdim=True)
        top = torch.cat([torch.topk(qn[i:i + 250] @ xn.T, 200, dim=1).indices for i in range(0, 1000, 250)])
        xd = x_dev[top.reshape(-1)].double().reshape(1000, 200, dim)
        qd = q_dev.double()
        cs = (xd * qd[:, None, :]).sum(2) / (xd.norm(dim=2) * qd.norm(dim=1)[:, None])
        order = torch.argsort(-cs, dim=1)[:, :10]
        truth64 = torch.gather(top, 1, order).cpu().numpy()
        kth_cos = torch.gather(cs, 1, order)[:, 9].cpu().numpy()
        del xn, xd
        same = np.mean([len(set(truth32[i]) & set(truth64[i])) for i in range(1000)]) / 10
        x = x_dev.cpu().numpy()
        del x_dev
        torch.cuda.empty_cache()
        items = H.ItemSet.from_f32(H.COSINE, x, levels=H.draw_levels(42, M, n))
        out = {"data": kind, "n": n, "truth_f32_vs_f64_overlap": round(float(same), 4)}
        with H.Builder(items, M=M, M0=M0, ef_construction=ef) as b:
            b.run()
            g = b.finish()
            l0 = g.rec_layer == 0
            deg_in = np.zeros(n, np.int64)
            # in-degree on layer 0 (ids == slots here: ids are 0..n-1)
            offs = g.offsets
            for r in np.nonzero(l0)[0][:0]:
                pass
            starts, ends = offs[:-1][l0], offs[1:][l0]
            idx = np.concatenate([np.arange(s, e) for s, e in zip(starts[:0], ends[:0])]) if False else None
            mask = np.zeros(len(g.nbrs), bool)
            # records are sorted by (item, layer): mark the layer-0 stretches with a difference array
            d = np.zeros(len(g.nbrs) + 1, np.int32)
            np.add.at(d, starts, 1)
            np.add.at(d, ends, -1)
            mask = np.cumsum(d[:-1]) > 0
            deg_in = np.bincount(g.nbrs[mask], minlength=n)
            out_deg = (ends - starts)
            out["layer0"] = {"mean_out_degree": round(float(out_deg.mean()), 2),
                             "in_degree_0": int((deg_in == 0).sum()),
                             "in_degree_0_frac": round(float((deg_in == 0).mean()), 5),
                             "in_degree_le_1_frac": round(float((deg_in <= 1).mean()), 5)}
            for e in (100, 400, 1600):
                ids, dists, cnt = b.search_knn(qc, qh, k=10, ef_search=e)
                r32 = bench.recall_at_k(ids, cnt, truth32)
                r64 = bench.recall_at_k(ids, cnt, truth64)
                # which true neighbours are missed: by id range and by in-degree
                miss = [t for i in range(1000) for t in truth64[i] if t not in set(ids[i, :cnt[i]].tolist())]
                miss = np.array(miss, np.int64)
                allt = truth64.reshape(-1)
                hi = 1 << 23
                out[f"ef_search_{e}"] = {
                    "recall_vs_f32_truth": round(r32, 4), "recall_vs_f64_truth": round(r64, 4),
                    "missed": int(len(miss)),
                    "missed_with_in_degree_0": int((deg_in[miss] == 0).sum()) if len(miss) else 0,
                    "truth_with_in_degree_0": int((deg_in[allt] == 0).sum()),
                    "missed_id_ge_2^23": int((miss >= hi).sum()) if len(miss) else 0,
                    "truth_id_ge_2^23": int((allt >= hi).sum()),
                    # a hit = an item at least as close (f64) as the true 10th neighbour, up to f32 resolution
                }
            del g
        print(json.dumps(out), flush=True)
        del items, x


if __name__ == "__main__":
    main()
