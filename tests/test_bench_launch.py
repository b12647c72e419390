"""bench.py's launch path on CPU (SURVEY.md §8d / north_star "1/2/4/8-GPU scaling reported"): `python bench.py
--gpus N` must really start N ranks — round 2's bench parsed --gpus and ran one GPU.  --rendezvous-only
stops after the ranks have met over gloo (no GPU, no build)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HNY_BENCH_SELF_LAUNCHED"):
        env.pop(k, None)
    return env


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_self_launches_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], capture_output=True, text=True,
                       env=_clean_env(), timeout=300)
    assert r.returncode == 0, r.stderr
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["ranks"] == [0, 1] and j["launcher"] == "self"


def test_gpus_3_self_launches_three_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--rendezvous-only"], capture_output=True, text=True,
                       env=_clean_env(), timeout=300)
    assert r.returncode == 0, r.stderr
    j = _json_line(r.stdout)
    assert j["ranks_seen"] == 3 and j["ranks"] == [0, 1, 2]


def test_under_a_launcher_every_process_is_one_rank():
    """the form the driver uses for N > 1: torch.distributed.run starts the ranks, bench.py must not fork again"""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2",
                        "--rendezvous-only"], capture_output=True, text=True, env=_clean_env(), timeout=600)
    assert r.returncode == 0, r.stderr
    j = _json_line(r.stdout)
    assert j["ranks_seen"] == 2 and j["launcher"] == "torchrun"


def test_launcher_and_flag_must_agree():
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode != 0 and "disagree" in r.stderr


def test_a_failing_rank_fails_the_run():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: the ranks would build")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, env=_clean_env(),
                       timeout=300)
    assert r.returncode != 0 and "exited with code" in r.stderr and not r.stdout.strip()


# ---- the same launch modes with real builds, on the one GPU of the test box (ranks share it) ----
import pytest  # noqa: E402

SMALL = ["--items", "30000", "--dim", "64", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-recall", "--queries", "0"]


@pytest.mark.gpu
def test_gpus_2_native_builds_on_two_replicas_through_the_copy_shim():
    """`bench.py --gpus 2 --native`: the resident multi-builder behind the C ABI (hny_multi.cpp) with two replicas on
    GPU 0, the all-gathers replaced by the copy shim, every replica's export compared inside the library"""
    env = dict(_clean_env(), HNY_MGPU_SHIM="1", HNY_MGPU_VERIFY="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--native"] + SMALL, capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["launcher"] == "native" and j["replicas_identical"] is True
    assert j["n_collectives"] > 0 and j["value"] > 0


@pytest.mark.gpu
def test_gpus_2_self_launched_ranks_build_over_gloo_on_one_gpu():
    """`bench.py --gpus 2 --backend gloo`: two self-launched processes (the torchrun harness, multigpu.py), exchange
    staged through the host; rank 0 compares the replicas' checksums"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo"] + SMALL, capture_output=True,
                       text=True, env=_clean_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["ranks_seen"] == 2 and j["launcher"] == "self" and j["replicas_identical"] is True and j["n_collectives"] > 0
