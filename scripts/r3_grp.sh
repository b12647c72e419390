#!/bin/bash
# grouped walk: parity tests, then same-box A/B against one wave per member
export TMPDIR=/tmp
mkdir -p gpurun_out/r3_grp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "grouped_walk or heap_walk or tie_pool or xcd_tiled_walk" > gpurun_out/r3_grp/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3_grp/pytest.log
HNY_DEBUG_GRP=1 CFGS="${CFGS:-c2 c3}" ORDER="A B A B" STEPS=3 ENV_A="HNY_GRP=1" ENV_B="HNY_GRP=0" bash scripts/r3_ab.sh
grep -h "grouped walk" gpurun_out/r3_ab/c2_A_1.log gpurun_out/r3_ab/c3_A_1.log | tail -4
