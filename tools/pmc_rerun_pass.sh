#!/bin/bash
# one PMC pass of tools/gpu_final_round.sh again (rocprofv3 itself crashed in it): bash tools/pmc_rerun_pass.sh <tag> <prefix> <FETCH_SIZE|WRITE_SIZE|req> <bench args...>
TAG=$1; pre=$2; c=$3; shift 3
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
ctr=$c; [ $c = req ] && ctr="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum"
for try in 1 2 3; do
  (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --pmc $ctr -d $R/$OUT/p_$pre$c -o k --output-format csv -- python $R/bench.py "$@" --steps 1 --warmup 0 --no-cpu --no-second > $R/$OUT/pmc_$pre$c.json 2> $R/$OUT/pmc_$pre$c.err)
  rc=$?; echo "pmc $pre$c try $try rc=$rc"
  if [ $rc = 0 ]; then python tools/prof_summary.py pmc $OUT/p_$pre$c $OUT/pmc_$pre$c.md && rm -rf $OUT/p_$pre$c; break; fi
  rm -rf $OUT/p_$pre$c
done
