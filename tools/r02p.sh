#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes of the Casper config-5 bench (per kernel)
set -u
OUT=gpurun_out/r02p; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $REPO/$OUT/pmc_$c -o k --output-format csv -- \
     python $REPO/bench.py --workload casper --steps 1 --warmup 0 --no-cpu > $REPO/$OUT/pmc_$c.json 2> $REPO/$OUT/pmc_$c.err)
  echo "pmc $c rc=$?"
  python tools/prof_summary.py pmc $OUT/pmc_$c $OUT/casper_pmc_$c.md && rm -rf $OUT/pmc_$c
  head -8 $OUT/casper_pmc_$c.md
done
