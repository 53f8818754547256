#!/bin/bash
# fifth GPU call: k_expand_runs — Casper tests, A/B at 262 150 nodes, rocprofv3 kernel stats of the new path
set -u
OUT=gpurun_out/r02e; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 600 python -m pytest tests/test_zr_gpu_casper_resident.py tests/test_gpu_engine.py tests/test_gpu_casper.py -m gpu -q > $OUT/pytest_casper.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/pytest_casper.log
for rm in 0 64; do
  WG_RUN_MIN=$rm timeout 900 python bench.py --workload casper --casper-cycle-length 64 --casper-producers 5 --steps 1 --warmup 0 --casper-ms 24000 --no-cpu > $OUT/bench_casper_cl64_runmin$rm.json 2> $OUT/bench_casper_cl64_runmin$rm.err
  echo "casper cl=64 runmin=$rm rc=$?"; cat $OUT/bench_casper_cl64_runmin$rm.json; tail -2 $OUT/bench_casper_cl64_runmin$rm.err
done
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/prof_casper -o k --output-format csv -- \
   python $REPO/bench.py --workload casper --casper-cycle-length 64 --casper-producers 5 --steps 1 --warmup 0 --casper-ms 16000 --no-cpu > $REPO/$OUT/prof_casper.json 2> $REPO/$OUT/prof_casper.err)
echo "prof rc=$?"
python tools/prof_summary.py stats $OUT/prof_casper $OUT/casper_kernel_stats.md && rm -rf $OUT/prof_casper
head -24 $OUT/casper_kernel_stats.md
cat $OUT/prof_casper.json
