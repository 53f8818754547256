#!/bin/bash
# One GPU call of a perf iteration: quick Handel parity subset, the bench line (3 timed steps), the same under
# rocprofv3 (kernel stats + per-phase table).   bash tools/gpu_perf_round.sh <tag> [extra bench args]
set -u
TAG=${1:-perf}; shift || true
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
for f in wittgenstein_amd/csrc/*; do  # a failed local build must not be measured as if it were the new code
  if [ "$f" -nt wittgenstein_amd/libwittgpu.so ]; then echo "STALE libwittgpu.so: $f is newer"; exit 1; fi
done
timeout 600 python -m pytest tests/test_gpu_handel.py tests/test_golden.py tests/test_gpu_snapshot.py tests/test_gpu_batch.py -m gpu -x -q > $OUT/pytest_handel.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_handel.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu --no-second "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value %.1f M msgs/s  ms_per_step %.1f  R %d  delivery pass %.1f us frac %.4f" % (d["value"] / 1e6, d["ms_per_step"], d["config"]["replicas_per_gpu"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
print(d["roofline"].get("warmup_phase_device_ms"))
PY
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/p -o k --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-second "$@" > $R/$OUT/prof_bench.json 2> $R/$OUT/prof_bench.err)
python tools/prof_summary.py stats $OUT/p $OUT/kernel_stats.md; python tools/prof_summary.py phases $OUT/p $OUT/phases.md; rm -rf $OUT/p
cut -d"|" -f2,3,4,5,6,12,13,23 $OUT/phases.md | head -24
