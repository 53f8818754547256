set -u
OUT=gpurun_out/r09i; mkdir -p $OUT
for r in 64 128 256; do
  timeout 900 python bench.py --workload gsf --nodes 4096 --replicas $r --steps 3 --warmup 1 --no-cpu > $OUT/bench_gsf_$r.json 2> $OUT/bench_gsf_$r.err
  python - $OUT/bench_gsf_$r.json $r <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("R=%s  %.1f M msgs/s  step %.1f ms  R_run %d  delivery %.1f us frac %.4f" % (sys.argv[2], d["value"] / 1e6, d["ms_per_step"], d["config"]["replicas_per_gpu"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
done
