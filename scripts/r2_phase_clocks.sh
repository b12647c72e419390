# wave cycles per walk phase (library built with HNY_CFLAGS=-DHNY_PHASE_CLOCKS): C5, 4M x 128, C2
mkdir -p gpurun_out
A="--no-cpu --no-recall --queries 0 --alt-data none --steps 1 --warmup 0"
for cfg in "c5 --items 5000000 --dim 1024 --metric hamming --ef 64" "c4s --items 4000000 --dim 128" "c2"; do
  set -- $cfg; name=$1; shift
  timeout -k 10 300 python bench.py $A "$@" > gpurun_out/r2_ph_$name.json 2> gpurun_out/r2_ph_$name.err || exit 1
  echo "== $name"; grep "walk wave cycles" gpurun_out/r2_ph_$name.err | tail -1
  python - <<PY
import json
j=json.loads(open("gpurun_out/r2_ph_$name.json").read().strip().splitlines()[-1]); b=j["build"]
print("   walk s", b["t_walk_kernels_s"], "evals_walk", b["evals_walk"])
PY
done
