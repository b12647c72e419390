#!/usr/bin/env python3
"""What the first real 1/2/4/8-GPU run should show — a per-batch model written down BEFORE any multi-GPU node has been
available to the build sessions, so that the driver's SCALE run is checkable against a stated expectation.

Inputs: the one-GPU batch timelines profiles/r04_<cfg>_batch_times.json (scripts/r4_batch_times.py: a sync after
the search and after the link phase of every batch) and the phase times of the one-GPU bench lines.
Model (DESIGN.md 6 — what the protocol does per batch on N ranks):
  search  sharded by members: s_b / N, but never below the walk-latency floor L (a rank with fewer members than
          resident waves is bound by ONE walk's latency: L = the search time of the one-GPU batches that are pure
          latency, 100..500 members, at the same stage of the build) and replicated (s_b) below 64 N members;
  exchange one all-gather of the selection records (264 B per member at M0 = 32; 520 B at M0 = 64) on a ring of
          7 x 153 GB/s xGMI links at 60 % efficiency + 30 us launch, a second small one for the re-pruned lists;
  link    emit + radix sort + segments + appends replicated (fraction `rep` of the link phase, from the phase
          times of the bench line), the overflowing targets' re-prunes sharded (1 - rep) / N;
  host    one 4-byte read per batch (~40 us of idle queue);
  export  on rank 0 only (replicated time, the other ranks wait).
Writes profiles/r04_predicted_scaling.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINK_GBS = 153.0 * 0.6  # per direction and link, ring all-gather: every rank forwards (N - 1) / N of the data
CFG = {"c2": dict(rec_bytes=264 + 16, rep=0.45, export_ms=9.0, ranks_floor_members=256),
       "c4": dict(rec_bytes=264 + 16, rep=0.45, export_ms=62.0, ranks_floor_members=256)}


def predict(name, tl, N, rep, export_ms, rec_bytes):
    b = tl["batches"]
    # latency floor: median search time of the batches with 100 .. 500 members (every member has its own wave)
    lat = sorted(x["search_ms"] for x in b if 100 <= x["count"] <= 500)
    L = lat[len(lat) // 2] if lat else 1.0
    total = 0.0
    parts = {"search": 0.0, "exchange": 0.0, "link": 0.0, "host": 0.0}
    for x in b:
        c, s, a = x["count"], x["search_ms"], x["apply_ms"]
        if N == 1 or c < 64 * N:
            sn, ex = s, 0.0
        else:
            sn = max(s / N, min(s, L))
            ex = 2 * 0.03 + c * rec_bytes * (N - 1) / N / (LINK_GBS * 1e9) * 1e3
        an = a if N == 1 else a * rep + a * (1 - rep) / N
        parts["search"] += sn
        parts["exchange"] += ex
        parts["link"] += an
        parts["host"] += 0.04 if N > 1 else 0.0
    total = sum(parts.values()) + export_ms
    return total, parts, L


def main():
    out = {"model": __doc__.split("Model")[1].split("Writes")[0].strip(), "configs": {}}
    for name, p in CFG.items():
        path = os.path.join(ROOT, "profiles", f"r04_{name}_batch_times.json")
        if not os.path.exists(path):
            print("missing", path, file=sys.stderr)
            continue
        tl = json.load(open(path))
        one = None
        rows = {}
        for N in (1, 2, 4, 8):
            t, parts, L = predict(name, tl, N, p["rep"], p["export_ms"], p["rec_bytes"])
            one = one or t
            rows[str(N)] = {"predicted_ms": round(t, 1), "speedup": round(one / t, 2),
                            "vectors_per_s": round(tl["n"] / t * 1e3, 0),
                            **{k: round(v, 1) for k, v in parts.items()}, "export_ms": p["export_ms"]}
        out["configs"][name.upper()] = {"n": tl["n"], "one_gpu_timeline_ms_with_syncs": round(tl["total_ms"], 1),
                                        "walk_latency_floor_ms": round(L, 3), "replicated_fraction_of_link_phase": p["rep"],
                                        "by_n_gpus": rows}
    json.dump(out, open(os.path.join(ROOT, "profiles", "r04_predicted_scaling.json"), "w"), indent=1)
    print(json.dumps(out["configs"], indent=1))


if __name__ == "__main__":
    main()
