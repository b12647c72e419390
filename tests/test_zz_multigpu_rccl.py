"""Multi-GPU over REAL RCCL: these tests need two or more GPUs and skip on the one-GPU boxes the build sessions get, so
they have never run before the driver's round-end tiers.  They live in the file pytest collects LAST, so that with
`-x` a first-contact failure here cannot hide the rest of the suite."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import draw_levels
from test_gpu_parity import _mk, _multi_case, _same_graph, hny  # noqa: F401  (fixtures and helpers)
from test_bench_launch import BENCH, _clean_env, _json_line

pytestmark = pytest.mark.gpu


def _visible_gpus():
    import torch
    return torch.cuda.device_count()  # (counting devices does not initialise the GPU)


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("metric", [0, 3])
def test_native_multi_gpu_over_real_rccl_when_the_box_has_the_gpus(orc, hny, monkeypatch, world, metric):
    """hny_build(n_gpus = N) on N DISTINCT devices, no shim: ncclCommInitAll over the node's GPUs, both all-gathers
    of every batch over xGMI, sharded searches and deferred re-prunes.  Skips on a one-GPU box (every other
    multi-rank test maps its ranks to GPU 0 through HNY_MGPU_SHIM); on a multi-GPU box it runs without anyone
    asking — the exported graph of EVERY replica (HNY_MGPU_VERIFY) must be the oracle's, counters included."""
    have = _visible_gpus()
    if have < world:
        pytest.skip(f"{world} GPUs needed, {have} visible")
    monkeypatch.delenv("HNY_MGPU_SHIM", raising=False)
    monkeypatch.setenv("HNY_MGPU_VERIFY", "1")
    monkeypatch.setenv("HNY_MGPU_MIN_BATCH", "16")
    monkeypatch.setenv("HNY_MGPU_MIN_DEFERRED", "2")
    ds, items, o, kw = _multi_case(orc, hny, metric=metric, n=20000, dim=96 if metric < 3 else 512)
    g = hny.build(items, n_gpus=world, devices=list(range(world)), **kw)
    _same_graph(g, o)
    assert g.n_links_added == o.n_links_added and g.n_evals_walk == o.n_evals_walk
    # the resident form (what bench.py --gpus N --native times): two runs on the same replicas
    with hny.MultiBuilder(items, devices=list(range(world)), **kw) as mb:
        assert mb.world == world
        for _ in range(2):
            _same_graph(mb.run(), o)
        assert mb.n_collectives > 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["native", "self"])
def test_gpus_n_over_real_rccl_when_the_box_has_the_gpus(mode):
    """`bench.py --gpus N [--native]` with N = min(visible GPUs, 8) distinct devices over RCCL — the command the
    driver issues for the scaling bench.  Auto-skips on a one-GPU box."""
    n = min(_visible_gpus(), 8)
    if n < 2:
        pytest.skip("one GPU visible")
    env = dict(_clean_env(), HNY_MGPU_VERIFY="1")
    env.pop("HNY_MGPU_SHIM", None)
    cmd = [sys.executable, BENCH, "--gpus", str(n)] + (["--native"] if mode == "native" else []) + \
          ["--items", "200000", "--dim", "128", "--steps", "1", "--warmup", "1", "--no-cpu", "--no-recall", "--queries", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == n and j["ranks_seen"] == n and j["replicas_identical"] is True
    assert sorted(j["devices"]) == list(range(n)) and j["n_collectives"] > 0 and j["value"] > 0
