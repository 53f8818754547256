#!/bin/bash
# GSFSignature on the GPU box: parity tests, bench line (BASELINE configs[1]), rocprofv3 kernel stats.
TAG=${1:-gsf}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$(pwd)
timeout 900 python -m pytest tests/test_gpu_gsf.py -m gpu -x -q --durations=5 > $OUT/pytest_gsf.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_gsf.log
for R in ${GSF_R:-16 64}; do
  timeout 600 python bench.py --workload gsf --nodes 4096 --replicas $R --init-threads 8 > $OUT/bench_gsf_R$R.json 2> $OUT/bench_gsf_R$R.err; echo "bench R=$R rc=$?"; cat $OUT/bench_gsf_R$R.json
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/prof -o k --output-format csv -- \
   python $REPO/bench.py --workload gsf --nodes 4096 --replicas 16 --init-threads 8 --steps 1 --warmup 0 --no-cpu > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof_bench.err)
python tools/prof_summary.py stats $OUT/prof $OUT/kernel_stats_gsf.md && rm -rf $OUT/prof; head -24 $OUT/kernel_stats_gsf.md
