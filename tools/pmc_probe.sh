#!/bin/bash
# issue / wait / TLB counters of the per-ms kernels on a reduced workload (N=8192, R=16)
OUT=gpurun_out/${1:-pmc}; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ|TCP|TCC|GRBM|TA|TD)_[A-Za-z0-9_]+" | sort -u > $OUT/counters_available.txt
wc -l $OUT/counters_available.txt
pass() { name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $REPO/$OUT/p_$name -o k --output-format csv -- \
     python $REPO/bench.py --nodes 8192 --replicas 16 --init-threads 16 --warmup 0 --no-cpu > $REPO/$OUT/p_$name.json 2> $REPO/$OUT/p_$name.err)
  echo "pass $name rc=$?"
  python tools/prof_summary.py pmc $OUT/p_$name $OUT/pmc_$name.md && rm -rf $OUT/p_$name
  grep -E "k_deliver|cond_a1" $OUT/pmc_$name.md | head -24
}
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
pass cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
pass tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
pass l2 TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum
