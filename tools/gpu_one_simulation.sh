set -u
OUT=gpurun_out/${1:-r09k}; mkdir -p $OUT
timeout 600 python bench.py --replicas 1 --steps 3 --warmup 1 --no-cpu --no-second > $OUT/bench_one_copy.json 2> $OUT/one.err; echo "one rc=$?"
timeout 900 python bench.py --mode shard --gpus 1 --steps 3 --warmup 1 --no-cpu --no-second > $OUT/bench_shard1.json 2> $OUT/shard1.err; echo "shard rc=$?"; tail -3 $OUT/shard1.err
python - $OUT <<'PY'
import json, sys, os
for f in ("bench_one_copy.json", "bench_shard1.json"):
    try:
        d = json.loads(open(os.path.join(sys.argv[1], f)).read().strip().splitlines()[-1])
        print(f, "%.1f M msgs/s, %.1f ms per step" % (d["value"] / 1e6, d["ms_per_step"]), d["config"].get("parallelism", "")[:80])
    except Exception as x:
        print(f, "unreadable:", x)
PY
