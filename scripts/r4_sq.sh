#!/bin/bash
# SQ instruction / issue / wait counters of the walk kernel, two rocprofv3 --pmc passes (8 SQ slots each), optionally for
# two builds of the library (LIBS="path1 path2", default: the in-tree one):
#   gpurun --timeout 900 -- 'NAME=c5 ARGS="--items 5000000 --dim 1024 --metric hamming --ef 64" LIBS="hannoy_amd/libhannoy_amd_r3.so hannoy_amd/libhannoy_amd.so" bash scripts/r4_sq.sh'
# -> gpurun_out/r4_sq/<NAME>_summary.txt (per kernel family: counters, per-evaluation instruction counts, issue-roof fractions)
#    KEY=5000000x1024_hamming_M16_ef64_clustered also writes gpurun_out/r4_sq/r04_sq_<KEY>.json (-> profiles/: bench.py's roofline_issue)
export TMPDIR=/tmp
out=gpurun_out/r4_sq
mkdir -p $out
NAME=${NAME:-c5}
ARGS=${ARGS:-"--items 5000000 --dim 1024 --metric hamming --ef 64"}
LIBS=${LIBS:-"hannoy_amd/libhannoy_amd.so"}
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES"
: > $out/${NAME}_summary.txt
for lib in $LIBS; do
  tag=$(basename $lib .so)
  export HNY_LIB=$PWD/$lib
  i=0
  for ctrs in "$P1" "$P2"; do
    i=$((i+1))
    d=$out/${NAME}_${tag}_p$i
    rm -rf $d
    timeout -k 10 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -- python3 bench.py --no-cpu --no-recall --queries 0 --steps 1 --warmup 0 --alt-data none $ARGS > $d.log 2>&1 || { echo "pass $i of $tag failed"; tail -5 $d.log; exit 1; }
    cp $(find $d -name "*counter_collection.csv") $out/${NAME}_${tag}_p$i.csv
    grep -a '"metric"' $d.log | tail -1 > $out/${NAME}_${tag}_p$i.json
    rm -rf $d
  done
  python3 scripts/r4_sq_summary.py $tag $out/${NAME}_${tag}_p1.csv $out/${NAME}_${tag}_p2.csv $out/${NAME}_${tag}_p1.json ${KEY:+$out/r04_sq_${KEY}.json} | tee -a $out/${NAME}_summary.txt
  rm -f $out/${NAME}_${tag}_p1.csv $out/${NAME}_${tag}_p2.csv
done
