import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.lib()
    return o


def draw_levels(n, M, seed):
    """Levels with the reference's distribution P(l) = M^-l (1 - 1/M) (hnsw.rs:94-119), from numpy's
    generator: most tests take levels as an INPUT (product and oracle get the same array).  The reference's
    own stream — rand 0.8.5 StdRng = ChaCha12 + WeightedIndex<f32> — is restated in product and oracle
    (hannoy_amd.draw_levels, orc.draw_levels) and pinned by KAT-1/5/9 (tests/test_oracle_kat.py)."""
    rng = np.random.default_rng(seed)
    u = rng.random(n)
    lv = np.floor(-np.log(1.0 - u) / np.log(M)).astype(np.int64)
    return np.minimum(lv, 7).astype(np.uint8)
