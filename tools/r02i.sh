#!/bin/bash
# WG_RUN_MIN sweep on Casper config 5
set -u
OUT=gpurun_out/r02i; mkdir -p $OUT
for rm in 4 8 16 32 64; do
  WG_RUN_MIN=$rm timeout 600 python bench.py --workload casper --steps 2 --warmup 1 --no-cpu > $OUT/bench_casper_runmin$rm.json 2> $OUT/bench_casper_runmin$rm.err
  echo "runmin=$rm rc=$? $(python -c "import json;j=json.load(open('$OUT/bench_casper_runmin$rm.json'));print('%.1f M msgs/s, %.0f ms/step'%(j['value']/1e6,j['ms_per_step']))")"
done
