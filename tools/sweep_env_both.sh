#!/bin/bash
# A/B of tuning variables on both bench workloads (Handel default line; GSFSignature 4096 x 64), one process each:
#   bash tools/sweep_env_both.sh <tag> "VAR=1" "VAR=2 OTHER=3" ...     (WL="handel gsf" selects)
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
for kv in "$@"; do
  name=$(echo "$kv" | tr ' =' '__')
  for wl in ${WL:-handel gsf}; do
    if [ $wl = handel ]; then args="--steps 2 --warmup 1"; else args="--workload gsf --nodes 4096 --replicas 64 --init-threads 8 --steps 3 --warmup 1"; fi
    env $kv timeout 600 python bench.py $args --no-cpu --no-second > $OUT/${wl}_$name.json 2> $OUT/${wl}_$name.err
    python - "$kv" $wl $OUT/${wl}_$name.json <<'PY' | tee -a $OUT/sweep.txt
import json, sys
try:
    d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    ph = d["roofline"].get("warmup_phase_device_ms") or {}
    print("%-44s %-6s %7.1f M msgs/s  step %7.1f ms  %s" % (sys.argv[1], sys.argv[2], d["value"] / 1e6, d["ms_per_step"], " ".join("%s=%.1f" % (k.split("(")[0], v) for k, v in ph.items() if v)))
except Exception as x:
    print(sys.argv[1], sys.argv[2], "FAILED", x)
PY
  done
done
