set -u
OUT=gpurun_out/r08d; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
timeout 900 python bench.py --workload gsf --nodes 4096 --replicas 64 --steps 3 --warmup 1 --no-cpu > $OUT/bench_gsf64.json 2> $OUT/bench_gsf64.err; echo "rc=$?"; tail -3 $OUT/bench_gsf64.err
python - $OUT/bench_gsf64.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value %.1f M msgs/s  ms_per_step %.1f  R %d  delivery pass %.1f us frac %.4f" % (d["value"] / 1e6, d["ms_per_step"], d["config"]["replicas_per_gpu"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
print(d["roofline"].get("warmup_phase_device_ms"))
PY
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/p -o k --output-format csv -- python $R/bench.py --workload gsf --nodes 4096 --replicas 64 --steps 1 --warmup 0 --no-cpu > $R/$OUT/prof_gsf.json 2> $R/$OUT/prof_gsf.err)
python tools/prof_summary.py stats $OUT/p $OUT/gsf_kernel_stats.md; rm -rf $OUT/p
head -24 $OUT/gsf_kernel_stats.md
