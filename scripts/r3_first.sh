#!/bin/bash
# round 3, first GPU call: whole GPU suite, then the N-rank bench entry points on the 1-GPU box
# (gloo test mode: two ranks share the GPU; --native with the copy shim)
export TMPDIR=/tmp
out=gpurun_out/r3_first
rm -rf $out && mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -5 $out/pytest.log
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --items 200000 --steps 2 --warmup 1 --no-cpu --out $out/bench_gpus2_gloo.json > $out/bench_gpus2_gloo.log 2>&1; echo "gloo rc=$?"
tail -2 $out/bench_gpus2_gloo.log
HNY_MGPU_SHIM=1 HNY_MGPU_VERIFY=1 timeout -k 10 300 python bench.py --gpus 2 --native --items 200000 --steps 2 --warmup 1 --no-cpu --out $out/bench_gpus2_native_shim.json > $out/bench_gpus2_native_shim.log 2>&1; echo "native rc=$?"
tail -2 $out/bench_gpus2_native_shim.log
timeout -k 10 300 python bench.py --gpus 1 --native --items 200000 --steps 2 --warmup 1 --no-cpu --out $out/bench_gpus1_native_rccl.json > $out/bench_gpus1_native_rccl.log 2>&1; echo "native1 rc=$?"
tail -2 $out/bench_gpus1_native_rccl.log
