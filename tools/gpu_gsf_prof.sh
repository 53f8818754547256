#!/bin/bash
# GSFSignature 4096 nodes x 64 copies under rocprofv3: kernel statistics + the GSF parity tests.  bash tools/gpu_gsf_prof.sh <tag>
TAG=${1:-gsfprof}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$(pwd)
timeout 900 python -m pytest tests/test_gpu_gsf.py tests/test_gpu_engine.py tests/test_gpu_send_expand.py -m gpu -x -q > $OUT/pytest_gsf.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gsf.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/prof -o k --output-format csv -- \
   python $REPO/bench.py --workload gsf --nodes 4096 --replicas 64 --init-threads 8 --steps 1 --warmup 0 --no-cpu --no-second > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof_bench.err)
python tools/prof_summary.py stats $OUT/prof $OUT/kernel_stats_gsf.md && rm -rf $OUT/prof; head -30 $OUT/kernel_stats_gsf.md
