#!/bin/bash
# default bench line (with second_workload) and smoke() as the driver runs them
set -u
OUT=gpurun_out/r02g; mkdir -p $OUT
( time timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; cat $OUT/bench.json; tail -3 $OUT/bench.err; cat $OUT/bench.time
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
