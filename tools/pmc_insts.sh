#!/bin/bash
# memory-instruction counts per kernel of the bench workload (one pass, counters only)
OUT=gpurun_out/${1:-pmc}; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$(pwd)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VALU -d $REPO/$OUT/p -o k --output-format csv -- \
   python $REPO/bench.py --steps 1 --warmup 0 --no-cpu ${BENCH_ARGS:-} > $REPO/$OUT/pmc_bench.json 2> $REPO/$OUT/pmc_bench.err)
echo "rc=$?"
python tools/prof_summary.py pmc $OUT/p $OUT/pmc_insts.md && rm -rf $OUT/p
grep -E "k_deliver|cond_a1|k_resolve|ExpandF" $OUT/pmc_insts.md | head -40
