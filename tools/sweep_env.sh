#!/bin/bash
# (SWEEP_ARGS="--workload gsf --nodes 4096 --replicas 64" for another workload)
# A/B of tuning variables, one bench process each (they are latched at engine creation): bash tools/sweep_env.sh <tag> "VAR=1" "VAR=2 OTHER=3" ...
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
for kv in "$@"; do
  name=$(echo "$kv" | tr ' =/' '___')
  env $kv timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --no-second $SWEEP_ARGS > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - "$kv" $OUT/bench_$name.json <<'PY' | tee -a $OUT/sweep.txt
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    ph = d["roofline"].get("warmup_phase_device_ms") or {}
    print("%-40s %7.1f M msgs/s  step %7.1f ms  deliver %6.1f cond_select %6.1f" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"], ph.get("deliver", 0), ph.get("cond_select", 0)) + "  resolve %6.1f" % ph.get("k_resolve", 0))
except Exception as x:
    print(sys.argv[1], "FAILED", x)
PY
done
