#!/usr/bin/env python3
"""Fold rocprofv3 PC samples (…pc_sampling_*.csv) of the build kernels: samples per kernel, and for k_walk the share per
source line (Instruction_Comment carries file:line when the library was built with -gline-tables-only) and per
instruction class, plus the hottest instructions."""
import collections
import csv
import re
import sys

csv.field_size_limit(1 << 30)


def cls(op):
    if "readlane" in op or "writelane" in op or "readfirstlane" in op:
        return "xlane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    return "vmem"


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    print("columns:", list(rows[0].keys()) if rows else None, " samples:", len(rows))
    if not rows:
        return
    ik = next((k for k in rows[0] if k.lower() == "instruction"), None)
    ck = next((k for k in rows[0] if "comment" in k.lower()), None)
    by_line, by_cls, by_inst = collections.Counter(), collections.Counter(), collections.Counter()
    extra = collections.defaultdict(collections.Counter)
    n = 0
    for r in rows:
        inst = (r.get(ik) or "").strip()
        com = (r.get(ck) or "").strip() if ck else ""
        if not inst:
            continue
        n += 1
        op = inst.split()[0]
        by_cls[cls(op)] += 1
        m = re.search(r"hny_kernels\.hip:(\d+)", com)
        line = int(m.group(1)) if m else 0
        by_line[line] += 1
        by_inst[(line, inst[:70])] += 1
        for k in r:  # stochastic sampling: issue / stall reason columns, whatever this ROCm names them
            kl = k.lower()
            if ("stall" in kl or "reason" in kl or "wave_issued" in kl or "inst_type" in kl) and r[k] not in ("", None):
                extra[k][r[k]] += 1
    print("samples with an instruction:", n)
    print("by class:", {k: round(v / n, 4) for k, v in by_cls.most_common()})
    for k, c in extra.items():
        print(k, {a: round(b / n, 4) for a, b in c.most_common(12)})
    src = None
    try:
        import os
        src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hannoy_amd", "csrc",
                                "hny_kernels.hip")).read().split("\n")
    except OSError:
        pass
    print("by source line (share of samples):")
    for line, c in by_line.most_common(70):
        txt = src[line - 1].strip()[:100] if src and 0 < line <= len(src) else ""
        print(f"  {line:5d} {c / n:7.4f}  {txt}")
    print("hottest instructions:")
    for (line, inst), c in by_inst.most_common(60):
        print(f"  {c / n:7.4f}  {line:5d}  {inst}")


if __name__ == "__main__":
    main()
