#!/bin/bash
set -u
OUT=gpurun_out/r02k; mkdir -p $OUT
timeout 1200 python tools/shard_invariance.py 32768 4 0 > $OUT/shard_invariance_32768_k4.json 2> $OUT/shard_invariance_32768_k4.err
echo "32768 k=4 rc=$?"; cat $OUT/shard_invariance_32768_k4.json; tail -3 $OUT/shard_invariance_32768_k4.err
timeout 1200 python tools/shard_invariance.py 32768 8 1 > $OUT/shard_invariance_32768_k8.json 2> $OUT/shard_invariance_32768_k8.err
echo "32768 k=8 rc=$?"; cat $OUT/shard_invariance_32768_k8.json; tail -3 $OUT/shard_invariance_32768_k8.err
