# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the walk kernels on C5 (5M x 1024-bit Hamming)
export TMPDIR=/tmp
out=gpurun_out/r2_pmc
rm -rf $out && mkdir -p $out
ARGS=${ARGS:-"--no-cpu --no-recall --queries 0 --steps 1 --warmup 0 --alt-data none --items 5000000 --dim 1024 --metric hamming --ef 64"}
run() { # name
  local name=$1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${name}_f -- python3 bench.py $ARGS > $out/${name}_f.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${name}_w -- python3 bench.py $ARGS > $out/${name}_w.log 2>&1
  python3 scripts/pmc_summary.py $(find $out/${name}_f -name "*counter_collection.csv") $(find $out/${name}_w -name "*counter_collection.csv") $out/$name.json > $out/$name.txt 2>&1
  grep -a '"metric"' $out/${name}_f.log | tail -1 > $out/${name}_bench.json
  find $out -name "*.csv" -delete
  echo "== $name"; cat $out/$name.txt
}
run ${NAME:-c5}_classic
