"""Per-batch wall time of one C2 build (sync after every phase): where the non-sharded time goes."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import hannoy_amd as H
from bench import gen_data

def main():
    n, dim = 1_000_000, 768
    import argparse
    p = argparse.ArgumentParser(); p.add_argument("--frac", type=float, default=0.0); a = p.parse_args()
    dev = torch.device("cuda:0")
    x = gen_data(torch, n, dim, "clustered", 42, dev)
    items = H.ItemSet.from_f32(H.COSINE, x.cpu().numpy())
    with H.Builder(items, M=16, M0=32, ef_construction=100, batch_frac=a.frac) as b:
        for rep in range(2):
            b.reset(); b.sync()
            rows = []
            t_all = time.perf_counter()
            while True:
                bt = b.next_batch()
                if bt.count == 0:
                    break
                t0 = time.perf_counter(); b.search(0, bt.count); b.sync()
                t1 = time.perf_counter(); b.apply(); b.sync()
                t2 = time.perf_counter()
                rows.append((bt.count, bt.level, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
            tot = time.perf_counter() - t_all
        small = [r for r in rows if r[0] < 65536]
        print(json.dumps({"frac": a.frac, "total_s_with_syncs": round(tot, 4), "n_batches": len(rows),
                          "ramp_batches": len(small),
                          "ramp_search_ms": round(sum(r[2] for r in small), 2),
                          "ramp_apply_ms": round(sum(r[3] for r in small), 2),
                          "full_search_ms": round(sum(r[2] for r in rows if r[0] == 65536), 2),
                          "full_apply_ms": round(sum(r[3] for r in rows if r[0] == 65536), 2)}))
        for r in rows[:40]:
            print("count %6d level %d search %.3f ms apply %.3f ms" % r)

main()
