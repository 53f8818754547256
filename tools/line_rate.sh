#!/bin/bash
# The mixed scattered read / write ceiling of the memory side, in the PMC's own unit: tools/micro/line_rate_probe plain (logical
# lines/s) and under rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum (EA requests per second per probe
# kernel) -> gpurun_out/<tag>/line_rate.txt; the highest 8 : 7 figure goes into bench.py's LINE_RATE_CEILING.
#   bash tools/line_rate.sh <tag> [footprint GiB]
TAG=${1:-linerate}; FOOT=${2:-128}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
./tools/micro/line_rate_probe $FOOT > $OUT/line_rate_plain.txt 2>&1; cat $OUT/line_rate_plain.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $R/$OUT/p -o k --output-format csv -- $R/tools/micro/line_rate_probe $FOOT > $R/$OUT/line_rate_pmc_stdout.txt 2>&1)
python - $OUT/p <<'PY' | tee $OUT/line_rate.txt
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
cc = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
kt = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = defaultdict(dict)
for f in cc:
    for r in csv.DictReader(open(f)):
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
print("| probe kernel (RD, WR, write bytes) | ms | EA read req | EA write req | G read req/s | G write req/s | G req/s |")
print("|---|---|---|---|---|---|---|")
for k in sorted(cnt, key=lambda x: int(x)):
    if k not in dur or dur[k][1] < 2_000_000:  # (the short warm-up launches)
        continue
    name, ns = dur[k]
    rd, wr = cnt[k].get("TCC_EA0_RDREQ_sum", 0.0), cnt[k].get("TCC_EA0_WRREQ_sum", 0.0)
    print("| %s | %.3f | %.0f | %.0f | %.2f | %.2f | %.2f |" % (name.split("(")[0].replace("void ", ""), ns / 1e6, rd, wr, rd / ns, wr / ns, (rd + wr) / ns))
PY
rm -rf $OUT/p
