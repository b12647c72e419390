#!/bin/bash
# whole GPU suite on the MI355X box: gpurun --timeout 1000 -- 'bash scripts/r3_suite.sh'
export TMPDIR=/tmp
out=gpurun_out/r3_suite
rm -rf $out && mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -15 $out/pytest.log
