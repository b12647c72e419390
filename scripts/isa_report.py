#!/usr/bin/env python3
"""Static report on ONE specialised kernel, without a GPU: compiles hny_kernels.hip for a single row
shape (-DHNY_ONLY_LPR/-DHNY_ONLY_NCH), prints register use / spills, instruction counts of the kernel
and of its loops by nesting depth, and LLVM's uniformity analysis of the same kernel (which loop-carried
values and branches the compiler treats as divergent: each divergent branch costs exec-mask SALU
instead of one s_cbranch_scc).

  python scripts/isa_report.py --part 4 --lpr 8 --nch 1 --kernel 'k_walkILi8ELi1ELb0ELi4ELb0ELi2E'
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "hannoy_amd", "csrc", "hny_kernels.hip")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-x", "hip", "--cuda-device-only"]
LLVM = "/opt/rocm/lib/llvm/bin"


def extract_ir_function(ll, name):
    """the module reduced to one kernel's definition (everything else becomes a declaration-free stub)"""
    out, keep, skipping = [], False, False
    for line in ll.split("\n"):
        if line.startswith("define "):
            keep = name in line
            skipping = not keep
        if not skipping:
            out.append(line)
        if skipping and line == "}":
            skipping = False
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--part", type=int, default=4)
    ap.add_argument("--lpr", type=int, default=8)
    ap.add_argument("--nch", type=int, default=1)
    ap.add_argument("--kernel", default="k_walkILi8ELi1ELb0ELi4ELb0ELi2E")
    ap.add_argument("--out", default="/tmp/isa_report")
    ap.add_argument("--no-uniformity", action="store_true")
    ap.add_argument("--lines", action="store_true", help="list the source lines of divergent branches")
    ap.add_argument("--cflags", default=os.environ.get("HNY_CFLAGS", ""))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    defs = [f"-DHNY_PART={a.part}", f"-DHNY_ONLY_LPR={a.lpr}", f"-DHNY_ONLY_NCH={a.nch}"] + a.cflags.split()
    asm = os.path.join(a.out, "k.s")
    ll = os.path.join(a.out, "k.ll")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + defs + ["-S", SRC, "-o", asm], stderr=subprocess.DEVNULL)
    text = open(asm).read()
    # ---- the kernel's ISA
    m = re.search(r"^(_Z\w*" + re.escape(a.kernel) + r"\w*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M)
    if not m:
        sys.exit("kernel not found: " + a.kernel)
    body = m.group(2).split("\n")
    open(os.path.join(a.out, "kernel.s"), "w").write(m.group(0))
    meta = re.search(r"\.name:\s+" + re.escape(m.group(1)) + r"\n(.*?)\.wavefront_size", text, re.S)
    for key in ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count"):
        mm = re.search(r"\." + key + r":\s+(\d+)", meta.group(1)) if meta else None
        print(f"{key:18s} {mm.group(1) if mm else '?'}")
    depth = 0
    by_depth = {}
    for line in body:
        s = line.strip()
        mm = re.search(r"Depth[= ](\d+)", s)
        if s.startswith(".LBB"):
            depth = int(mm.group(1)) if mm else 0
            continue
        if s.startswith(";") and mm:  # continuation comment lines of a loop header
            depth = max(depth, int(mm.group(1))) if "Loop Header" in s or "Inner Loop" in s else depth
            continue
        if not s or s[0] in ";.":
            continue
        op = s.split()[0]
        cls = ("VALU" if op.startswith("v_") and "readlane" not in op and "writelane" not in op and "readfirstlane" not in op
               else "XLANE" if op.startswith("v_") else "WAIT" if op.startswith(("s_waitcnt", "s_nop")) else
               "BRANCH" if op.startswith(("s_cbranch", "s_branch")) else "SALU" if op.startswith("s_") else
               "LDS" if op.startswith("ds_") else "VMEM")
        by_depth.setdefault(depth, {}).setdefault(cls, 0)
        by_depth[depth][cls] += 1
    print("static instructions by loop depth:")
    for d in sorted(by_depth):
        c = by_depth[d]
        print(f"  depth {d}: " + "  ".join(f"{k}={c[k]}" for k in sorted(c)) + f"  total={sum(c.values())}")
    if a.no_uniformity:
        return
    # ---- uniformity of the same kernel
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + defs + (["-gline-tables-only"] if a.lines else []) +
                          ["-emit-llvm", "-S", SRC, "-o", ll],
                          stderr=subprocess.DEVNULL)
    one = os.path.join(a.out, "one.ll")
    open(one, "w").write(extract_ir_function(open(ll).read(), a.kernel))
    uni = os.path.join(a.out, "uniformity.txt")
    with open(uni, "w") as f:
        # the module is printed by the same run: the analysis names metadata by the numbers of THAT printout
        subprocess.check_call([os.path.join(LLVM, "opt"), "-passes=print<uniformity>", "-S", "-o",
                               os.path.join(a.out, "renum.ll"), one], stderr=f)
    u = open(uni).read()
    if a.lines:  # source lines of the branches the compiler treats as divergent (needs -gline-tables-only)
        irt = open(os.path.join(a.out, "renum.ll")).read()
        loc = {m.group(1): (int(m.group(2)), m.group(3)) for m in
               re.finditer(r"^(!\d+) = (?:distinct )?!DILocation\(line: (\d+)(?:[^\n]*?inlinedAt: (!\d+))?", irt, re.M)}
        def chain(k):
            out = []
            while k in loc:
                out.append(str(loc[k][0]))
                k = loc[k][1]
            return "<-".join(out)
        seen = {}
        for mm in re.finditer(r"DIVERGENT:\s+br i1 [^\n]*?!dbg (!\d+)", u):
            c = chain(mm.group(1)) or "0"
            seen[c] = seen.get(c, 0) + 1
        for c in sorted(seen, key=lambda x: [int(t) for t in x.split("<-")][::-1]):
            print(f"  divergent br at line {c}  x{seen[c]}")
    nbr = len(re.findall(r"DIVERGENT:\s+br ", u))
    nphi = len(re.findall(r"DIVERGENT:\s+%\d+ = phi", u))
    cyc = re.search(r"CYCLES WITH DIVERGENT EXIT:\n((?:  depth.*\n)*)", u)
    print(f"divergent branches {nbr}, divergent phis {nphi}, cycles with a divergent exit "
          f"{len(cyc.group(1).splitlines()) if cyc else 0}   ({uni})")


if __name__ == "__main__":
    main()
