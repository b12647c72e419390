#!/bin/bash
# Round-4 profiles of every BASELINE workload, after any change to hny_kernels.hip (bench.py checks the source hash the
# summaries carry): rocprofv3 --kernel-trace --stats + the FETCH_SIZE / WRITE_SIZE PMC passes (scripts/r3_pmc.sh), and
# for the short-row workloads the two SQ passes (scripts/r4_sq.sh).  `bash scripts/r4_pmc_all.sh long|short`;
# everything lands in gpurun_out/r4_pmc/ under the names bench.py looks up — copy to profiles/ locally.
set -e
mkdir -p gpurun_out/r4_pmc gpurun_out/r3_pmc
one() { # NAME  profile-key  ARGS
  NAME=$1 ARGS="$3" bash scripts/r3_pmc.sh > gpurun_out/r4_pmc/$1.txt 2>&1
  cp gpurun_out/r3_pmc/$1_pmc_hbm.json gpurun_out/r4_pmc/r04_pmc_hbm_$2.json
  cp gpurun_out/r3_pmc/$1_kernel_stats.csv gpurun_out/r4_pmc/r04_$1_kernel_stats.csv
  cp gpurun_out/r3_pmc/$1_bench.json gpurun_out/r4_pmc/r04_$1_bench_under_rocprof.json
  echo "== $1 done"; tail -4 gpurun_out/r3_pmc/$1_pmc.txt
}
sq() { # NAME profile-key ARGS
  NAME=$1 KEY=$2 ARGS="$3" bash scripts/r4_sq.sh > gpurun_out/r4_pmc/$1_sq.txt 2>&1
  cp gpurun_out/r4_sq/r04_sq_$2.json gpurun_out/r4_pmc/
  cp gpurun_out/r4_sq/$1_summary.txt gpurun_out/r4_pmc/r04_$1_sq_counters.txt
  echo "== $1 SQ done"; grep -A3 "^k_walk {" gpurun_out/r4_sq/$1_summary.txt | tail -3
}
if [ "$1" = "long" ]; then
  one c2 1000000x768_cosine_M16_ef100_clustered ""
  one c2_overlap 1000000x768_cosine_M16_ef100_overlap "--data overlap"
  one c3 1000000x768_euclidean_M32_ef200_clustered "--metric euclidean --M 32 --ef 200"
else
  one c5 5000000x1024_hamming_M16_ef64_clustered "--items 5000000 --dim 1024 --metric hamming --ef 64"
  sq c5 5000000x1024_hamming_M16_ef64_clustered "--items 5000000 --dim 1024 --metric hamming --ef 64"
  one c4 10000000x128_cosine_M16_ef100_clustered "--items 10000000 --dim 128"
  sq c4 10000000x128_cosine_M16_ef100_clustered "--items 10000000 --dim 128"
fi
