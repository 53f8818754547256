#!/bin/bash
# One gpurun call: GPU parity tests, the bench line, a rocprofv3 kernel trace and the HBM PMC passes.
# Usage (from the repo root on the GPU box): bash tools/gpu_session.sh <tag> [what...]   what = tests bench prof pmc
set -u
TAG=${1:-r01}; shift || true
WHAT=${*:-tests bench prof pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
for w in $WHAT; do
case $w in
tests)
  timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -m gpu ${PYTEST_X--x} -q --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log ;;
bench)
  timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json ;;
prof)
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/prof_stats -o k --output-format csv -- \
     python $REPO/bench.py --steps 1 --warmup 0 --no-cpu --no-second > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof_bench.err)
  echo "prof rc=$?"
  python tools/prof_summary.py stats $OUT/prof_stats $OUT/kernel_stats.md && rm -rf $OUT/prof_stats
  head -30 $OUT/kernel_stats.md ;;
pmc)
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c -d $REPO/$OUT/pmc_$c -o k --output-format csv -- \
       python $REPO/bench.py --steps 1 --warmup 0 --no-cpu --no-second --batches 1 --nodes ${PMC_NODES:-32768} > $REPO/$OUT/pmc_$c.json 2> $REPO/$OUT/pmc_$c.err)
    echo "pmc $c rc=$?"
    python tools/prof_summary.py pmc $OUT/pmc_$c $OUT/pmc_$c.md && rm -rf $OUT/pmc_$c
    head -12 $OUT/pmc_$c.md
  done ;;
esac
done
