R0=$(pwd); OUT=gpurun_out/r23b_R31_R32; mkdir -p $OUT; export TMPDIR=/tmp
for R in 31 32; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R0/$OUT/p$R -o k --output-format csv -- python $R0/bench.py --steps 1 --warmup 0 --no-cpu --no-second --replicas $R --engine-config queue_cap_wide=12 > $R0/$OUT/bench_R$R.json 2> $R0/$OUT/bench_R$R.err)
  python tools/prof_summary.py phases $OUT/p$R $OUT/phases_R$R.md; rm -rf $OUT/p$R
  echo "== R=$R"; cut -d"|" -f2,4,5,23 $OUT/phases_R$R.md | head -30
done
