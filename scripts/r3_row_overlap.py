#!/usr/bin/env python3
"""How many of the rows a level-0 walk scores are also scored by the walks next to it in processing order?
(DESIGN.md "row sharing": what a grouped walk / distance sharing between neighbouring queries could save.)
CPU only: the oracle builds N clustered vectors with the GPU's schedule and traces the last batch's level-0
walks (ORC_TRACE_EVALS); members are ordered by their layer-0 entry point (a stand-in for the GPU's locality
key) and overlaps are reported for partners at distance 1, 2, 4, ... in that order.
  python scripts/r3_row_overlap.py [n] [dim] [clusters]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from tests.conftest import draw_levels  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
ncl = int(sys.argv[3]) if len(sys.argv) > 3 else max(8, n // 1000)  # ~1 000 points per cluster, like C2
rng = np.random.default_rng(42)
cent = rng.uniform(-1, 1, (ncl, dim)).astype(np.float32)
x = cent[rng.integers(0, ncl, n)] + 0.15 * rng.standard_normal((n, dim)).astype(np.float32)
ds = orc.Dataset.from_f32(orc.COSINE, x, draw_levels(n, 16, 42))
path = os.path.join(tempfile.gettempdir(), "orc_trace.bin")
os.environ["ORC_TRACE_EVALS"] = path
orc.build(ds, M=16, M0=32, ef=100, order=orc.ORDER_WAVE, threads=os.cpu_count(), batch_frac=1.0,
          batch_max=max(4096, n // 12))
t = np.fromfile(path, np.uint32)
starts = np.flatnonzero(t == 0xFFFFFFFF)
recs = []
for a, b in zip(starts, list(starts[1:]) + [len(t)]):
    recs.append((int(t[a + 1]), int(t[a + 2]), t[a + 3:b]))
print(f"{len(recs)} level-0 walks traced, {np.mean([len(r[2]) for r in recs]):.0f} rows scored per walk")
recs.sort(key=lambda r: (r[1], r[0]))  # by entry point into layer 0, then id
sets = [set(r[2].tolist()) for r in recs]
for dist in (1, 2, 4, 8, 16, 64, 256, 1024):
    if dist >= len(sets):
        break
    ov = [len(sets[i] & sets[i + dist]) / max(1, len(sets[i])) for i in range(0, len(sets) - dist, max(1, len(sets) // 2000))]
    print(f"partner at distance {dist:5d} in entry-point order: {100 * np.mean(ov):5.1f} % of a walk's rows are shared")
for G in (2, 4, 8):  # a group of G consecutive members: rows needed once per group / rows needed per member
    tot = uni = 0
    for i in range(0, len(sets) - G, G * max(1, len(sets) // (2000 * G))):
        u = set()
        for j in range(G):
            u |= sets[i + j]
            tot += len(sets[i + j])
        uni += len(u)
    print(f"groups of {G}: distinct rows / scored rows = {uni / tot:.3f}")
