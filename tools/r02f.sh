#!/bin/bash
# sixth GPU call: idle-ms skipping — whole GPU suite, Casper at 262 150 nodes (A/B, +10 % stopped), kernel stats
set -u
OUT=gpurun_out/r02f; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 $OUT/pytest_gpu.log
for sk in 0 1; do
  WG_SKIP_IDLE=$sk timeout 900 python bench.py --workload casper --steps 1 --warmup 0 --no-cpu > $OUT/bench_casper_skip$sk.json 2> $OUT/bench_casper_skip$sk.err
  echo "casper skip=$sk rc=$?"; cat $OUT/bench_casper_skip$sk.json; tail -2 $OUT/bench_casper_skip$sk.err
done
timeout 900 python bench.py --workload casper --casper-stopped 0.1 --steps 2 --warmup 1 > $OUT/bench_casper_stopped10.json 2> $OUT/bench_casper_stopped10.err
echo "casper stopped rc=$?"; cat $OUT/bench_casper_stopped10.json; tail -2 $OUT/bench_casper_stopped10.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/prof_casper -o k --output-format csv -- \
   python $REPO/bench.py --workload casper --steps 1 --warmup 0 --no-cpu > $REPO/$OUT/prof_casper.json 2> $REPO/$OUT/prof_casper.err)
echo "prof rc=$?"
python tools/prof_summary.py stats $OUT/prof_casper $OUT/casper_kernel_stats.md && rm -rf $OUT/prof_casper
head -24 $OUT/casper_kernel_stats.md
cat $OUT/prof_casper.json
