#!/bin/bash
# the PMC passes of every BASELINE workload, after any change to hny_kernels.hip (bench.py checks the source
# hash the summaries carry): two gpurun calls, `bash scripts/r3_pmc_all.sh long` and `... short`; copies the
# summaries into profiles/ under the names bench.py looks up (only gpurun_out/ travels back: copy again locally)
set -e
one() { # NAME  profile-key  ARGS
  NAME=$1 ARGS="$3" bash scripts/r3_pmc.sh > gpurun_out/r3_pmc/$1.txt 2>&1
  cp gpurun_out/r3_pmc/$1_pmc_hbm.json gpurun_out/r3_pmc/r03_pmc_hbm_$2.json
  cp gpurun_out/r3_pmc/$1_kernel_stats.csv gpurun_out/r3_pmc/r03_$1_kernel_stats.csv
  cp gpurun_out/r3_pmc/$1_bench.json gpurun_out/r3_pmc/r03_$1_bench_under_rocprof.json
  echo "== $1 done"; tail -4 gpurun_out/r3_pmc/$1_pmc.txt
}
mkdir -p gpurun_out/r3_pmc
if [ "$1" = "long" ]; then
  one c2 1000000x768_cosine_M16_ef100_clustered ""
  one c2_overlap 1000000x768_cosine_M16_ef100_overlap "--data overlap"
  one c3 1000000x768_euclidean_M32_ef200_clustered "--metric euclidean --M 32 --ef 200"
else
  one c5 5000000x1024_hamming_M16_ef64_clustered "--items 5000000 --dim 1024 --metric hamming --ef 64"
  one c4 10000000x128_cosine_M16_ef100_clustered "--items 10000000 --dim 128"
fi
