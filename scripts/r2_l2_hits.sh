# L2 hit / miss counters of k_walk on C2 with the single work counter and with the XCD-tiled queue
export TMPDIR=/tmp
out=gpurun_out/r2_l2
rm -rf $out && mkdir -p $out
run() { # name
  local name=$1
  timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $out/$name -- python3 bench.py --no-cpu --no-recall --queries 0 --alt-data none --steps 1 --warmup 0 > $out/$name.log 2>&1
  echo "== $name (HNY_XCD_TILE=${HNY_XCD_TILE:-default})" >> $out/summary.txt
  python3 scripts/sq_summary.py $(find $out/$name -name "*counter_collection.csv") 2>&1 | grep -A0 "^k_walk\|^k_prune_wg" >> $out/summary.txt
  grep -a '"metric"' $out/$name.log | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('   bench: walk', j['build']['t_walk_kernels_s'], 's')" >> $out/summary.txt
  find $out/$name -name "*.csv" -delete
}
HNY_XCD_TILE=0 run single_counter &&
run xcd_tiled
cat $out/summary.txt
