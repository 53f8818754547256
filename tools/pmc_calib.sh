#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of tools/micro/pmc_calib (known byte counts per kernel) -> gpurun_out/<tag>/pmc_calib.txt
OUT=gpurun_out/${1:-calib}; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
[ -x tools/micro/pmc_calib ] || hipcc --offload-arch=gfx950 -O2 -o tools/micro/pmc_calib tools/micro/pmc_calib.hip
tools/micro/pmc_calib > $OUT/pmc_calib_plain.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $REPO/$OUT/calib_$c -o k --output-format csv -- $REPO/tools/micro/pmc_calib > $REPO/$OUT/calib_$c.out 2>&1)
  python tools/prof_summary.py pmc $OUT/calib_$c $OUT/calib_$c.md && rm -rf $OUT/calib_$c
done
{ echo "# known bytes (program output, no profiler)"; grep CALIB $OUT/pmc_calib_plain.txt
  echo; echo "# FETCH_SIZE (KB per dispatch)"; cat $OUT/calib_FETCH_SIZE.md
  echo; echo "# WRITE_SIZE (KB per dispatch)"; cat $OUT/calib_WRITE_SIZE.md; } > $OUT/pmc_calib.txt
cat $OUT/pmc_calib.txt
