# SQ busy / wait / LDS counters of the prune + apply workgroup kernels on C2 (1M x 768 cosine), two passes
export TMPDIR=/tmp
out=gpurun_out/r2_sqp
rm -rf $out && mkdir -p $out
ARGS=${ARGS:-""}
run() { # name counters...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 bench.py --no-cpu --no-recall --queries 0 --alt-data none --steps 1 --warmup 0 $ARGS > $out/$name.log 2>&1
  echo "== $name" >> $out/summary.txt
  python3 scripts/sq_summary.py $(find $out/$name -name "*counter_collection.csv") 2>&1 | grep -A1 "^k_prune_wg\|^k_apply_wg\|^k_walk" >> $out/summary.txt
  find $out/$name -name "*.csv" -delete
}
run busy SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS &&
run wait SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT &&
run lds SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM
cat $out/summary.txt
