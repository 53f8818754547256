#!/bin/bash
# issue / wait / request-count counters of the per-ms kernels on the bench workload (counters only, one pass per set)
OUT=gpurun_out/${1:-pmc}; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$(pwd)
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ|TCP|TCC|GRBM|TA|TD)_[A-Za-z0-9_]+" | sort -u > $OUT/counters_available.txt
pass() { name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc "$@" -d $REPO/$OUT/p_$name -o k --output-format csv -- \
     python $REPO/bench.py --steps 1 --warmup 0 --no-cpu --no-second ${BENCH_ARGS:-} > $REPO/$OUT/p_$name.json 2> $REPO/$OUT/p_$name.err)
  echo "pass $name rc=$?"
  python tools/prof_summary.py pmc $OUT/p_$name $OUT/pmc_$name.md && rm -rf $OUT/p_$name
  head -30 $OUT/pmc_$name.md
}
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
pass cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
pass req TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_ATOMIC_sum
