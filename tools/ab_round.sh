#!/bin/bash
# Same-box A/B of environment settings (or builds: WG_LIB=<path>) on several workloads, INTERLEAVED — boxes differ by
# +- 1.5 %, so only runs inside one gpurun call compare:
#   bash tools/ab_round.sh <tag> <rounds> "<set A>" "<set B>" ...        (a set may be "" = defaults)
# AB_WORKLOADS: bench argument strings separated by ';' (default: the headline, GSFSignature 4096 x 256).
TAG=$1; ROUNDS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
IFS=';' read -ra WLS <<< "${AB_WORKLOADS:-;--workload gsf --nodes 4096 --replicas 480}"
[ ${#WLS[@]} -eq 0 ] && WLS=("")
for r in $(seq 1 $ROUNDS); do
  for wl in "${WLS[@]}"; do
    i=0
    for SET in "$@"; do
      i=$((i+1))
      f=$OUT/b_r${r}_s${i}_$(echo "$wl" | tr -c 'a-zA-Z0-9' '_' | cut -c1-24).json
      env $SET timeout 900 python bench.py --steps ${AB_STEPS:-3} --warmup 1 --no-cpu --no-second $wl > $f 2> ${f%.json}.err
      python - "$f" "[$r] ${SET:-defaults} | ${wl:-handel 32768}" <<'PY' | tee -a $OUT/ab.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-70s %7.1f M msgs/s  step %7.1f ms  pass %6.1f us frac %.4f" % (sys.argv[2], d["value"] / 1e6, d["ms_per_step"], r["avg_launch_us"], r["frac"]))
except Exception as x:
    print(sys.argv[2], "FAILED", x)
PY
    done
  done
done
