#!/bin/bash
# Round 5, VERDICT r4 task 4: does putting neighbouring members (locality order) of the long-row prune on ONE XCD — the
# walk's tiling, PruneArgs.xcd_tile — lower k_prune_wg's L2->fabric traffic?  FETCH_SIZE and TCC hit / miss passes of one C2
# build per setting.   gpurun --timeout 900 -- 'bash scripts/r5_prune_tile_fetch.sh'
export TMPDIR=/tmp
out=gpurun_out/r5_prune_tile
rm -rf $out && mkdir -p $out
for tile in 0 64 512; do
  for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=${tile}_$(echo $pass | cut -c1-3)
    HNY_PRUNE_XCD_TILE=$tile timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/$tag -- python3 bench.py --no-cpu --no-recall --queries 0 --steps 1 --warmup 0 --alt-data none > $out/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $out/$tag.log; exit 1; }
  done
  python3 - <<PY
import csv, glob, collections
agg = collections.Counter()
for f in glob.glob("$out/${tile}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_prune_wg" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
rd = 2 * agg["FETCH_SIZE"] * 1024 / 1e9
print("xcd_tile $tile: k_prune_wg read %.1f GB (2 x FETCH_SIZE), L2 hit rate %.3f" % (rd, agg["TCC_HIT_sum"] / max(1.0, agg["TCC_HIT_sum"] + agg["TCC_MISS_sum"])), flush=True)
PY
  rm -rf $out/${tile}_*/
done
