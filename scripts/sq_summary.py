#!/usr/bin/env python3
"""Sum rocprofv3 --pmc SQ_* counters per kernel family: pmc_summary-style table.
usage: sq_summary.py counter_collection.csv"""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    fam = next((k for k in ("k_walk_heap", "k_walk", "k_prune_wg", "k_apply_wg", "k_apply", "k_emit", "k_segments") if k in n), "other")
    agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
for fam, c in agg.items():
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print(fam, {k: f"{v:.3g}" for k, v in sorted(c.items())})
    print("   frac of wave cycles:", {k: round(v / wc, 3) for k, v in c.items() if k.startswith(("SQ_WAIT", "SQ_ACTIVE"))})
