#!/bin/bash
# fourth GPU call: resident Casper tests incl. stopped attesters; rocprofv3 kernel stats of Casper at 262 150 nodes
set -u
OUT=gpurun_out/r02d; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 600 python -m pytest tests/test_zr_gpu_casper_resident.py -m gpu -q > $OUT/pytest_casper.log 2>&1; echo "casper tests rc=$?"; tail -5 $OUT/pytest_casper.log
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/prof_casper -o k --output-format csv -- \
   python $REPO/bench.py --workload casper --casper-cycle-length 64 --casper-producers 5 --steps 1 --warmup 0 --casper-ms 16000 --no-cpu > $REPO/$OUT/prof_casper.json 2> $REPO/$OUT/prof_casper.err)
echo "prof rc=$?"
python tools/prof_summary.py stats $OUT/prof_casper $OUT/casper_kernel_stats.md && rm -rf $OUT/prof_casper
head -40 $OUT/casper_kernel_stats.md
cat $OUT/prof_casper.json
