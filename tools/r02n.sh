#!/bin/bash
set -u
OUT=gpurun_out/r02n; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_zr_gpu_casper_resident.py -m gpu -q --durations=5 > $OUT/pytest_casper.log 2>&1; echo "rc=$?"; tail -25 $OUT/pytest_casper.log
