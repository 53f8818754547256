#!/bin/bash
# the default bench workload under several settings of one environment variable: bash tools/gpu_sweep_env.sh <tag> <VAR> <v1> <v2> ...
set -u
TAG=$1; VAR=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
for v in "$@"; do
  env $VAR=$v timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu --no-second > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json "$VAR=$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
ph = d["roofline"].get("warmup_phase_device_ms", {})
print("%-24s %.1f M msgs/s  step %.1f ms  delivery %.1f us  cond_select %.1f ms  cond_rest %.1f ms" % (sys.argv[2], d["value"] / 1e6, d["ms_per_step"], d["roofline"]["avg_launch_us"], ph.get("cond_select", 0), ph.get("cond_rest", 0)))
PY
done
