#!/bin/bash
# A/B of alternative builds of libwittgpu.so on the default bench workload: bash tools/gpu_ab_lib.sh <tag> <lib> [<lib> ...]
# ("default" = the in-tree library). One line per build: value, ms per step, delivery pass.
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = default ]; then unset WG_LIB; else export WG_LIB=$(pwd)/$lib; fi
  timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu --no-second > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - $OUT/bench_$name.json $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-28s %.1f M msgs/s  step %.1f ms  delivery pass %.1f us  frac %.4f" % (sys.argv[2], d["value"] / 1e6, d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
done
