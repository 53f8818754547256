#!/bin/bash
# The long evidence runs of a round, one GPU call: config 4's workload unsharded (Handel 131 072 nodes on one MI355X,
# tools/config4_unsharded.py), Casper IMD config 5 over its whole horizon (2 560 000 simulated ms, 10 % of the attesters
# stopped) with its kernel statistics, GSFSignature config 2 x 64 and x 256 copies, Handel 65 536 nodes.
#   bash tools/gpu_evidence_long.sh <tag>
set -u
TAG=${1:-long}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
echo "== config 4 unsharded"; timeout 1500 python tools/config4_unsharded.py 131072 > $OUT/config4_unsharded_131072.json 2> $OUT/config4.err; echo "rc=$?"; tail -c 900 $OUT/config4_unsharded_131072.json
echo "== casper full horizon"; timeout 900 python bench.py --workload casper --casper-ms 2560000 --casper-stopped 0.1 --steps 1 --warmup 0 --no-cpu > $OUT/bench_casper_full_horizon.json 2> $OUT/casper_full.err; echo "rc=$?"; tail -2 $OUT/casper_full.err
echo "== casper 24 s, kernel stats"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/pc -o k --output-format csv -- python $R/bench.py --workload casper --casper-stopped 0.1 --steps 1 --warmup 0 --no-cpu > $R/$OUT/prof_casper.json 2> $R/$OUT/prof_casper.err)
python tools/prof_summary.py stats $OUT/pc $OUT/casper_kernel_stats.md; rm -rf $OUT/pc; head -12 $OUT/casper_kernel_stats.md
echo "== gsf 64 / 256 copies"; timeout 600 python bench.py --workload gsf --nodes 4096 --replicas 64 --steps 3 --warmup 1 --no-cpu > $OUT/bench_gsf64.json 2> $OUT/gsf64.err; echo "rc=$?"; tail -1 $OUT/gsf64.err
timeout 600 python bench.py --workload gsf --nodes 4096 --replicas 256 --steps 3 --warmup 1 --no-cpu > $OUT/bench_gsf256.json 2> $OUT/gsf256.err; echo "rc=$?"; tail -1 $OUT/gsf256.err
echo "== handel 65536"; timeout 900 python bench.py --nodes 65536 --replicas 6 --steps 2 --warmup 1 --no-cpu --no-second > $OUT/bench_handel65536.json 2> $OUT/handel65536.err; echo "rc=$?"; tail -2 $OUT/handel65536.err
python - $OUT <<'PY'
import json, sys, os
for f in ("bench_casper_full_horizon.json", "bench_gsf64.json", "bench_gsf256.json", "bench_handel65536.json"):
    try:
        d = json.loads(open(os.path.join(sys.argv[1], f)).read().strip().splitlines()[-1])
        print(f, "%.1f M msgs/s, %.1f ms per step, frac %.4f" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["frac"]))
    except Exception as x:
        print(f, "unreadable:", x)
PY
