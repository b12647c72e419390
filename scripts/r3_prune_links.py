#!/usr/bin/env python3
"""How many robust_prune rejections could be read off STORED links (a violating selected s lists the candidate,
or the candidate lists s, with the distance the link was created with)?  CPU only: the oracle counts while it
builds (ORC_PRUNE_LINK_STATS).  Result on 60 000 x 256 clustered: 41 % / 42 % / 45 % (either).  The kernel
that used it (wg_prune with a two-chunk list prefetch and an LDS id set of S) skipped those row loads and was
SLOWER — C2 prune 76 vs 67 ms, C3 198 vs 165 — the chunk pipeline is latency-bound, not byte-bound; reverted.
  python scripts/r3_prune_links.py [n] [dim]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ORC_PRUNE_LINK_STATS"] = "1"
from oracle import orc  # noqa: E402
from tests.conftest import draw_levels  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rng = np.random.default_rng(42)
ncl = max(8, n // 1000)
cent = rng.uniform(-1, 1, (ncl, dim)).astype(np.float32)
x = cent[rng.integers(0, ncl, n)] + 0.15 * rng.standard_normal((n, dim)).astype(np.float32)
ds = orc.Dataset.from_f32(orc.COSINE, x, draw_levels(n, 16, 42))
orc.build(ds, M=16, M0=32, ef=100, order=orc.ORDER_WAVE, threads=os.cpu_count(), batch_frac=1.0, batch_max=max(4096, n // 12))
