"""Builds hannoy_amd/libhannoy_amd.so (host driver + gfx950 kernels) in-tree with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhannoy_amd.so")
SOURCES = ["hny_host.cpp", "hny_kernels.hip", "hny_lmdb.cpp"]
HOST_ONLY = {"hny_lmdb.cpp"}  # no device code: compiled as plain C++
HEADERS = [os.path.join(CSRC, "hny_internal.h"),
           os.path.join(os.path.dirname(HERE), "include", "hannoy_amd.h")]
# -ffp-contract=off: FMAs only where the source says fmaf (parity with the oracle's orders);
# correctly rounded f32 divide/sqrt for the cosine / hamming finalisers.
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result"] + os.environ.get("HNY_CFLAGS", "").split()


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    objs = []
    for s in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
        if s in HOST_ONLY:
            cmd = [hipcc(), "-O2", "-std=c++17", "-fPIC", "-Wall", "-x", "c++", "-c",
                   os.path.join(CSRC, s), "-o", obj]
        else:
            cmd = [hipcc()] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc(), "-shared", "--offload-arch=gfx950", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
