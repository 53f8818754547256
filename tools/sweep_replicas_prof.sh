#!/bin/bash
# kernel stats (rocprofv3) of the bench at several batch sizes: bash rsweep.sh <tag> R1 R2 ...
TAG=$1; shift
R0=$(pwd); OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for R in "$@"; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R0/$OUT/p$R -o k --output-format csv -- python $R0/bench.py --steps 1 --warmup 0 --no-cpu --no-second --replicas $R > $R0/$OUT/bench_R$R.json 2> $R0/$OUT/bench_R$R.err)
  python tools/prof_summary.py stats $OUT/p$R $OUT/kernel_stats_R$R.md; python tools/prof_summary.py phases $OUT/p$R $OUT/phases_R$R.md; rm -rf $OUT/p$R
  echo "== R=$R"; python -c "
import json; d=json.load(open('$OUT/bench_R$R.json')); print('value %.1f M ms_per_step %.1f R %d' % (d['value']/1e6, d['ms_per_step'], d['config']['replicas_per_gpu']))"
  cut -d"|" -f2,5,23 $OUT/phases_R$R.md | head -22
done
