#!/bin/bash
# TLB / L2 counters of the per-ms kernels on the bench workload (one rocprofv3 --pmc pass per group)
OUT=gpurun_out/${1:-pmc_tlb}; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd); shift || true
rocprofv3 -L 2>/dev/null | grep -o -E "\b[A-Za-z0-9_]*(UTCL|TLB|Tlb)[A-Za-z0-9_]*" | sort -u > $OUT/counters_tlb.txt; cat $OUT/counters_tlb.txt | head -60
pass() { name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc "$@" -d $R/$OUT/p_$name -o k --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-second $EXTRA > $R/$OUT/p_$name.json 2> $R/$OUT/p_$name.err)
  echo "pass $name rc=$?"
  python tools/prof_summary.py pmc $OUT/p_$name $OUT/pmc_$name.md && rm -rf $OUT/p_$name
  grep -E "k_handel_(lane|wave|a1|update|copy)" $OUT/pmc_$name.md | head -30
}
EXTRA="$@"
pass tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
