#!/bin/bash
# One GPU call: Handel parity subset, bench line, rocprofv3 kernel stats + per-phase table, and the two PMC passes
# (FETCH_SIZE, WRITE_SIZE: separate runs, as MI355X_MICROARCH.md prescribes) -> profiles-ready traffic.json of the
# delivery pass.   bash tools/gpu_traffic_round.sh <tag> [extra bench args]
set -u
TAG=${1:-traffic}; shift || true
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
for f in wittgenstein_amd/csrc/*; do
  if [ "$f" -nt wittgenstein_amd/libwittgpu.so ]; then echo "STALE libwittgpu.so: $f is newer"; exit 1; fi
done
timeout 600 python -m pytest tests/test_gpu_handel.py tests/test_golden.py tests/test_gpu_snapshot.py tests/test_gpu_batch.py -m gpu -x -q > $OUT/pytest_handel.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_handel.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu --no-second "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value %.1f M msgs/s  ms_per_step %.1f  R %d  delivery pass %.1f us frac %.4f" % (d["value"] / 1e6, d["ms_per_step"], d["config"]["replicas_per_gpu"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/p -o k --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-second "$@" > $R/$OUT/prof_bench.json 2> $R/$OUT/prof_bench.err)
python tools/prof_summary.py stats $OUT/p $OUT/kernel_stats.md; python tools/prof_summary.py phases $OUT/p $OUT/phases.md; rm -rf $OUT/p
cut -d"|" -f2,3,4,5,6,12,13,23 $OUT/phases.md | head -24
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c -d $R/$OUT/p_$c -o k --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-second "$@" > $R/$OUT/pmc_$c.json 2> $R/$OUT/pmc_$c.err)
  echo "pmc $c rc=$?"
  python tools/prof_summary.py pmc $OUT/p_$c $OUT/pmc_$c.md && rm -rf $OUT/p_$c
  head -14 $OUT/pmc_$c.md
done
python tools/traffic_from_pmc.py $OUT/pmc_FETCH_SIZE.md $OUT/pmc_WRITE_SIZE.md 32768 24 $OUT/traffic.json "k_handel_lane,k_handel_copy,k_handel_update<,k_handel_wave<" | head -12
