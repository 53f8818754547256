#!/bin/bash
set -u
OUT=gpurun_out/r02s; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/prof_casper -o k --output-format csv -- \
   python $REPO/bench.py --workload casper --steps 1 --warmup 0 --no-cpu > $REPO/$OUT/prof_casper.json 2> $REPO/$OUT/prof_casper.err)
python tools/prof_summary.py stats $OUT/prof_casper $OUT/casper_kernel_stats.md && rm -rf $OUT/prof_casper
head -14 $OUT/casper_kernel_stats.md
