#!/usr/bin/env python3
"""Per kernel family: SQ counters of two rocprofv3 --pmc passes, instructions per walk evaluation, and the
instruction-issue roof (one VALU and one scalar instruction per SIMD every quad-cycle).
usage: r4_sq_summary.py TAG pass1.csv pass2.csv bench.json [out.json]
out.json (-> profiles/r04_sq_<n>x<dim>_<metric>_M<M>_ef<ef>_<data>.json) is what bench.py's `roofline_issue` reads."""
import collections
import csv
import json
import sys

FAMS = ("k_walk_heap", "k_walk", "k_prune_n8", "k_apply_n8", "k_prune_wg", "k_apply_wg", "k_apply_append", "k_apply",
        "k_emit", "k_segments")
tag, p1, p2, bj = sys.argv[1:5]
out_json = sys.argv[5] if len(sys.argv) > 5 else None
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for path in (p1, p2):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        fam = next((k for k in FAMS if k in n), "other")
        if path == p2 and r["Counter_Name"] == "SQ_WAVE_CYCLES":
            agg[fam]["SQ_WAVE_CYCLES_p2"] += float(r["Counter_Value"])
        else:
            agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[fam].add(r["Dispatch_Id"])
j = json.loads(open(bj).read())
b = j["build"]
evals = {"k_walk": b["evals_walk"], "k_prune_n8": b.get("evals_prune", 0), "k_prune_wg": b.get("evals_prune", 0)}
secs = {"k_walk": b["t_walk_kernels_s"], "k_prune_n8": b["t_prune_kernels_s"], "k_prune_wg": b["t_prune_kernels_s"]}
print(f"== {tag}: {j['config']['workload'] if 'workload' in j.get('config', {}) else ''} (under rocprofv3 --pmc: "
      f"walk {b['t_walk_kernels_s']} s, {b['evals_walk']} walk evaluations)")
for fam in FAMS:
    if fam not in agg:
        continue
    c = agg[fam]
    print(fam, {k: f"{v:.4g}" for k, v in sorted(c.items())})
    wc = c.get("SQ_WAVE_CYCLES_p2") or c.get("SQ_WAVE_CYCLES") or 1.0
    print("   frac of wave cycles:", {k: round(v / wc, 3) for k, v in c.items()
                                      if k.startswith(("SQ_WAIT", "SQ_ACTIVE"))})
    if evals.get(fam):
        e = evals[fam]
        print("   per evaluation:", {k: round(c[k] / e, 2) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM",
                                                                      "SQ_INSTS_LDS", "SQ_INSTS_SMEM") if k in c})
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_INSTS_VALU"):
        # SQ_BUSY_CYCLES: cycles a shader engine's SQ had waves, summed over the SEs that report (32 on this part);
        # issue roof = the busier of the two issue ports: one instruction per SIMD per quad-cycle each
        valu_q, sca_q = c.get("SQ_ACTIVE_INST_VALU", c["SQ_INSTS_VALU"]), c.get("SQ_ACTIVE_INST_SCA", c.get("SQ_INSTS_SALU", 0))
        if secs.get(fam):
            for clk in (2.1e9, 2.4e9):
                quads = 1024 * secs[fam] * clk / 4
                print(f"   issue-port occupancy at {clk / 1e9:.1f} GHz over the kernel's {secs[fam]} s: "
                      f"VALU {valu_q / quads:.3f}  scalar {sca_q / quads:.3f}")

if out_json:
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "hannoy_amd", "csrc", "hny_kernels.hip"), "rb") as f:
        sha = hashlib.sha1(f.read()).hexdigest()
    res = {"workload": j.get("config", {}).get("workload"), "kernel_source_sha1": sha, "library": tag,
           "passes": "rocprofv3 --pmc, two passes of 8 SQ counters, one build each (scripts/r4_sq.sh); SQ_ACTIVE_* / "
                     "SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles summed over waves",
           "walk_seconds_under_profiler": b["t_walk_kernels_s"]}
    for fam in FAMS:
        if fam in agg:
            res[fam] = dict(agg[fam])
            if evals.get(fam):
                res[fam]["evals"] = evals[fam]
    json.dump(res, open(out_json, "w"), indent=1)
