#!/bin/bash
# Casper config 5 over whole cycles: 1 cycle (64 slots), then SURVEY §8d's horizon of 5 cycles
set -u
OUT=gpurun_out/r02m; mkdir -p $OUT
timeout 900 python bench.py --workload casper --casper-ms 512000 --steps 1 --warmup 0 --no-cpu > $OUT/bench_casper_1cycle.json 2> $OUT/bench_casper_1cycle.err
echo "1 cycle rc=$?"; cat $OUT/bench_casper_1cycle.json; tail -3 $OUT/bench_casper_1cycle.err
timeout 1500 python bench.py --workload casper --casper-ms 2560000 --casper-stopped 0.1 --steps 1 --warmup 0 --no-cpu > $OUT/bench_casper_5cycles_stopped10.json 2> $OUT/bench_casper_5cycles_stopped10.err
echo "5 cycles rc=$?"; cat $OUT/bench_casper_5cycles_stopped10.json; tail -3 $OUT/bench_casper_5cycles_stopped10.err
