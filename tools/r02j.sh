#!/bin/bash
# sendAll latency-bin histograms + runMin 8: tests, Casper bench, kernel stats
set -u
OUT=gpurun_out/r02j; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 900 python -m pytest tests/test_zr_gpu_casper_resident.py tests/test_gpu_engine.py tests/test_zs_gpu_send_expand.py tests/test_gpu_batch.py -m gpu -q > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/pytest.log
timeout 900 python bench.py --workload casper --steps 2 --warmup 1 --no-cpu > $OUT/bench_casper.json 2> $OUT/bench_casper.err
echo "casper rc=$?"; cat $OUT/bench_casper.json; tail -2 $OUT/bench_casper.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/prof_casper -o k --output-format csv -- \
   python $REPO/bench.py --workload casper --steps 1 --warmup 0 --no-cpu > $REPO/$OUT/prof_casper.json 2> $REPO/$OUT/prof_casper.err)
python tools/prof_summary.py stats $OUT/prof_casper $OUT/casper_kernel_stats.md && rm -rf $OUT/prof_casper
head -16 $OUT/casper_kernel_stats.md
for ag in 1024 2048 4096; do
  WG_CASPER_ATT_GRID=$ag timeout 600 python bench.py --workload casper --steps 2 --warmup 1 --no-cpu > $OUT/bench_casper_attgrid$ag.json 2> $OUT/bench_casper_attgrid$ag.err
  echo "attgrid=$ag rc=$? $(python -c "import json;j=json.load(open('$OUT/bench_casper_attgrid$ag.json'));print('%.1f M msgs/s, %.0f ms/step'%(j['value']/1e6,j['ms_per_step']))")"
done
for rg in 2048 4096; do
  WG_EXPAND_RUNS_GRID=$rg timeout 600 python bench.py --workload casper --steps 2 --warmup 1 --no-cpu > $OUT/bench_casper_rungrid$rg.json 2> $OUT/bench_casper_rungrid$rg.err
  echo "rungrid=$rg rc=$? $(python -c "import json;j=json.load(open('$OUT/bench_casper_rungrid$rg.json'));print('%.1f M msgs/s, %.0f ms/step'%(j['value']/1e6,j['ms_per_step']))")"
done
