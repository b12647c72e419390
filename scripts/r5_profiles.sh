#!/bin/bash
# Round-5 profiles of the BASELINE workloads on the data their bench lines are taken on (C2 / C3: clustered, C2 also on
# overlap; C4 / C5: overlap — recall >= 0.9 there): rocprofv3 --kernel-trace --stats, the FETCH_SIZE /
# WRITE_SIZE / TCC_HIT+MISS passes (scripts/r5_pmc.sh) and, for rows <= 512 B, the two SQ passes (scripts/r4_sq.sh).
#   gpurun --timeout 1190 -- 'bash scripts/r5_profiles.sh long'      (c2, c2_overlap, c3)
#   gpurun --timeout 1190 -- 'bash scripts/r5_profiles.sh c5'  /  'bash scripts/r5_profiles.sh c4'
# everything lands in gpurun_out/r5_pmc/ under the names bench.py looks up — copy to profiles/.
mkdir -p gpurun_out/r5_pmc
sq() { # NAME KEY ARGS
  NAME=$1 KEY=$2 ARGS="$3" bash scripts/r4_sq.sh > gpurun_out/r5_pmc/$1_sq.txt 2>&1 || { echo "SQ passes of $1 failed"; tail -5 gpurun_out/r5_pmc/$1_sq.txt; return 1; }
  cp gpurun_out/r4_sq/r04_sq_$2.json gpurun_out/r5_pmc/r05_sq_$2.json
  cp gpurun_out/r4_sq/$1_summary.txt gpurun_out/r5_pmc/r05_$1_sq_counters.txt
  echo "== $1 SQ done"; grep -A3 "^k_walk {" gpurun_out/r4_sq/$1_summary.txt | tail -3
}
case "$1" in
  long)
    NAME=c2 KEY=1000000x768_cosine_M16_ef100_clustered ARGS="" bash scripts/r5_pmc.sh &&
    NAME=c2_overlap KEY=1000000x768_cosine_M16_ef100_overlap ARGS="--data overlap" bash scripts/r5_pmc.sh &&
    NAME=c3 KEY=1000000x768_euclidean_M32_ef200_clustered ARGS="--metric euclidean --M 32 --ef 200" bash scripts/r5_pmc.sh ;;
  c5)
    A="--items 5000000 --dim 1024 --metric hamming --ef 64 --data overlap"
    NAME=c5 KEY=5000000x1024_hamming_M16_ef64_overlap ARGS="$A" bash scripts/r5_pmc.sh &&
    sq c5 5000000x1024_hamming_M16_ef64_overlap "$A" ;;
  c4)
    A="--items 10000000 --dim 128 --data overlap"
    NAME=c4 KEY=10000000x128_cosine_M16_ef100_overlap ARGS="$A" bash scripts/r5_pmc.sh &&
    sq c4 10000000x128_cosine_M16_ef100_overlap "$A" ;;
esac
