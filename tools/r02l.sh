#!/bin/bash
set -u
OUT=gpurun_out/r02l; mkdir -p $OUT
timeout 2400 python tools/shard_invariance.py 65536 4 0 > $OUT/shard_invariance_65536_k4.json 2> $OUT/shard_invariance_65536_k4.err
echo "65536 k=4 rc=$?"; cat $OUT/shard_invariance_65536_k4.json; tail -3 $OUT/shard_invariance_65536_k4.err
