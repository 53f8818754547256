#!/bin/bash
# A/B of environment settings on the bench workload with per-kernel times: bash tools/ab_env.sh <tag> "<bench args>" "VAR=val ..." "VAR=val ..." ...
# (first setting may be "" = defaults). One rocprofv3 kernel-trace run per setting (1 step), prints the step value and the main kernels.
TAG=$1; BARGS=$2; shift 2
R0=$(pwd); OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for SET in "$@"; do
  i=$((i+1))
  (cd /tmp && env $SET timeout 600 rocprofv3 --kernel-trace --stats -d $R0/$OUT/p$i -o k --output-format csv -- python $R0/bench.py --steps 2 --warmup 1 --no-cpu --no-second $BARGS > $R0/$OUT/bench_$i.json 2> $R0/$OUT/bench_$i.err)
  python tools/prof_summary.py phases $OUT/p$i $OUT/phases_$i.md; rm -rf $OUT/p$i
  echo "== [$i] $SET"; python -c "
import json; d=json.load(open('$OUT/bench_$i.json')); print('value %.1f M ms_per_step %.1f R %d frac %.4f' % (d['value']/1e6, d['ms_per_step'], d['config']['replicas_per_gpu'], d['roofline']['frac']))"
  cut -d"|" -f2,5,23 $OUT/phases_$i.md | sed -n 3,14p
done
