# SQ latency / busy counters of the walk kernels on C5 (5M x 1024-bit Hamming): sub-wave kernel vs one wave per query
export TMPDIR=/tmp
out=gpurun_out/r2_sq
rm -rf $out && mkdir -p $out
CTRS=${CTRS:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"}
ARGS=${ARGS:-"--items 5000000 --dim 1024 --metric hamming --ef 64"}
run() { # name
  local name=$1; shift
  timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $out/$name -- python3 bench.py --no-cpu --no-recall --queries 0 --steps 1 --warmup 0 --alt-data none $ARGS > $out/$name.log 2>&1
  echo "== $name: $ARGS" >> $out/summary.txt
  python3 scripts/sq_summary.py $(find $out/$name -name "*counter_collection.csv") 2>&1 | grep -A1 "^k_walk" >> $out/summary.txt
  grep -a '"metric"' $out/$name.log | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('   bench: walk', j['build']['t_walk_kernels_s'], 's, value', j['value'])" >> $out/summary.txt
  find $out/$name -name "*.csv" -delete
}
run classic
cat $out/summary.txt
