#!/bin/bash
# third GPU call: concurrent batches A/B, Casper at BASELINE config 5's node count, default line with roofline.traffic
set -u
OUT=gpurun_out/r02c; mkdir -p $OUT
for b in 2 4; do
  timeout 900 python bench.py --batches $b --no-cpu --init-threads 16 > $OUT/bench_batches$b.json 2> $OUT/bench_batches$b.err
  echo "batches=$b rc=$?"; cat $OUT/bench_batches$b.json; tail -2 $OUT/bench_batches$b.err
done
timeout 1200 python bench.py --workload casper --casper-cycle-length 64 --casper-producers 5 --steps 1 --warmup 0 --casper-ms 24000 > $OUT/bench_casper_cl64.json 2> $OUT/bench_casper_cl64.err
echo "casper cl=64 rc=$?"; cat $OUT/bench_casper_cl64.json; tail -3 $OUT/bench_casper_cl64.err
timeout 1200 python bench.py --init-threads 16 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; cat $OUT/bench.json; tail -2 $OUT/bench.err
