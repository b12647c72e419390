"""Builds hannoy_amd/libhannoy_amd.so (host driver + gfx950 kernels) in-tree with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhannoy_amd.so")
SOURCES = ["hny_host.cpp", "hny_multi.cpp", "hny_kernels.hip", "hny_lmdb.cpp"]
HOST_ONLY = {"hny_lmdb.cpp"}  # no device code: compiled as plain C++
HEADERS = [os.path.join(CSRC, "hny_internal.h"),
           os.path.join(CSRC, "hny_rust_sort.h"),
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


# hny_kernels.hip is compiled once per part (see its header): 0 = general kernels + entry points,
# 1..7 = the build kernels specialised for metric part-1
KERNEL_PARTS = range(8)


def build(force=False, verbose=False, out=None, tag=""):
    """out / tag: a second build of the library next to the regular one (objects get the tag in their
    names), for same-box A/B runs through HNY_LIB"""
    if out is None and not force and not needs_build():
        return LIB
    jobs = []
    for s in SOURCES:
        stem = os.path.join(CSRC, os.path.splitext(s)[0] + tag)
        if s in HOST_ONLY:
            jobs.append((stem + ".o", [hipcc(), "-O2", "-std=c++17", "-fPIC", "-Wall", "-x", "c++", "-c",
                                       os.path.join(CSRC, s), "-o", stem + ".o"]))
        elif s == "hny_kernels.hip":
            for part in KERNEL_PARTS:
                obj = f"{stem}_p{part}.o"
                jobs.append((obj, [hipcc()] + FLAGS + [f"-DHNY_PART={part}", "-x", "hip", "-c",
                                                       os.path.join(CSRC, s), "-o", obj]))
        else:
            jobs.append((stem + ".o", [hipcc()] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, s),
                                                           "-o", stem + ".o"]))

    def run(job):
        if verbose:
            print(" ".join(job[1]), flush=True)
        # (one retry: with eight compilers of a 5 000-line translation unit in flight, clang has been seen to
        # die of a signal once in a few dozen builds; the same command succeeds when repeated)
        for attempt in (1, 2):
            rc = subprocess.call(job[1])
            if rc == 0:
                return job[0]
            if attempt == 1:
                print(f"[buildlib] {os.path.basename(job[0])}: compiler exit code {rc}, retrying once", file=sys.stderr, flush=True)
        raise subprocess.CalledProcessError(rc, job[1])
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(len(jobs), int(os.environ.get("HNY_BUILD_JOBS", os.cpu_count() or 1))))
    with ThreadPoolExecutor(workers) as ex:
        objs = list(ex.map(run, jobs))
    lib = out or LIB
    cmd = [hipcc(), "-shared", "--offload-arch=gfx950", "-o", lib] + objs + ["-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    o = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    build(force="--force" in sys.argv, verbose=True, out=o, tag="_ab" if o else "")
    print(LIB)
