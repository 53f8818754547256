#!/bin/bash
# L2 hit / miss and memory-side requests of the per-ms kernels at two copy counts (same memory per copy: queue_cap_wide=12):
# bash tools/pmc_copies_cache.sh <tag> 31 32
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$(pwd)
for R in "$@"; do
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $REPO/$OUT/p_$R -o k --output-format csv -- \
     python $REPO/bench.py --steps 1 --warmup 0 --no-cpu --no-second --replicas $R --engine-config queue_cap_wide=12 > $REPO/$OUT/p_$R.json 2> $REPO/$OUT/p_$R.err)
  echo "R=$R rc=$?"
  python tools/prof_summary.py pmc $OUT/p_$R $OUT/pmc_cache_R$R.md && rm -rf $OUT/p_$R
  grep -E "k_handel_lane\(|k_handel_wave|k_handel_a1c|k_handel_lane2" $OUT/pmc_cache_R$R.md | cut -c1-160
done
