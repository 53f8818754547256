#!/bin/bash
# per-dispatch kernel durations of one un-warmed bench step (rocprofv3 kernel trace), compacted to name,dur_ns
OUT=gpurun_out/${1:-trace}; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$(pwd)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $REPO/$OUT/p -o k --output-format csv -- \
   python $REPO/bench.py --steps 1 --warmup 0 --no-cpu ${BENCH_ARGS:-} > $REPO/$OUT/trace_bench.json 2> $REPO/$OUT/trace_bench.err)
python - $OUT <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
f = sorted(glob.glob(os.path.join(out, "p", "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(os.path.join(out, "trace_compact.csv"), "w") as o:
    for r in rows:
        n = r["Kernel_Name"].replace("wg::", "")
        n = n.split("(")[0][-60:]
        o.write("%s,%d,%d\n" % (n, int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print("dispatches", len(rows))
PY
rm -rf $OUT/p; ls -la $OUT
