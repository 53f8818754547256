#!/bin/bash
# A/B of alternative builds of libwittgpu.so on BOTH bench workloads (Handel default line, GSFSignature 4096 x 64 copies):
#   bash tools/gpu_ab_both.sh <tag> <lib> [<lib> ...]     ("default" = the in-tree library)
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = default ]; then unset WG_LIB; else export WG_LIB=$(pwd)/$lib; fi
  timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu --no-second > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  timeout 600 python bench.py --workload gsf --nodes 4096 --replicas 64 --init-threads 8 --steps 3 --warmup 1 --no-cpu --no-second > $OUT/gsf_$name.json 2> $OUT/gsf_$name.err
  python - $OUT/bench_$name.json $OUT/gsf_$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-28s handel %.1f M msgs/s  step %.1f ms  delivery pass %.1f us  frac %.4f" % (sys.argv[3], d["value"] / 1e6, d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
print("   ", d["roofline"].get("warmup_phase_device_ms"))
g = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-28s gsf    %.1f M msgs/s  step %.1f ms" % (sys.argv[3], g["value"] / 1e6, g["ms_per_step"]))
print("   ", g["roofline"].get("warmup_phase_device_ms"))
PY
done
