"""The C ABI from plain C: tests/c_client/kat1_client.c is compiled with gcc against
include/hannoy_amd.h and linked to libhannoy_amd.so."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compile(tmp_path):
    import hannoy_amd  # noqa: F401  (makes sure the library exists)
    exe = str(tmp_path / "kat1_client")
    libdir = os.path.join(ROOT, "hannoy_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_client", "kat1_client.c"), "-o", exe,
                           "-L", libdir, "-l:libhannoy_amd.so", f"-Wl,-rpath,{libdir}",
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_c_client_compiles_and_reports_no_device(tmp_path):
    import torch
    exe = _compile(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2, r.stderr  # HNY_ERR_NO_DEVICE: the library has no CPU path


@pytest.mark.gpu
def test_c_client_builds_kat1(tmp_path, kat, orc):
    exe = _compile(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    recs = [tuple(bytes.fromhex(h) for h in line.split()) for line in r.stdout.strip().splitlines()]
    k = kat["kat1"]
    ds = orc.Dataset.from_f32(orc.EUCLIDEAN, np.array(k["vectors"], np.float32), k["levels"])
    og = orc.build(ds, M=3, M0=3, ef=100)
    assert recs == orc.encode_kv(ds, og, 0, True)  # byte-identical Metadata/Version/Links/Item records
    assert len(recs) == 2 + len(k["links"]) + 6
