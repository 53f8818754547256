#!/bin/bash
set -u
OUT=gpurun_out/r02q; mkdir -p $OUT
timeout 900 python -m pytest tests/test_zr_gpu_casper_resident.py tests/test_gpu_casper.py -m gpu -q > $OUT/pytest_casper.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest_casper.log
timeout 600 python bench.py --workload casper --steps 2 --warmup 1 --no-cpu > $OUT/bench_casper.json 2> $OUT/bench_casper.err
echo "casper rc=$? $(python -c "import json;j=json.load(open('$OUT/bench_casper.json'));print('%.1f M msgs/s, %.0f ms/step'%(j['value']/1e6,j['ms_per_step']))")"
