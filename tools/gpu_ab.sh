#!/bin/bash
# A/B of an env-selected engine variant on the bench workload: prints one summary line per variant.
# usage: bash tools/gpu_ab.sh <tag> "<ENV=.. ENV=..>" ["<ENV..>" ...]   (bench args via BENCH_ARGS)
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for v in "$@"; do
  i=$((i+1))
  env $v timeout 900 python bench.py --no-cpu ${BENCH_ARGS:-} > $OUT/ab_$i.json 2> $OUT/ab_$i.err
  python - "$v" $OUT/ab_$i.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); r=d["roofline"]; p=r.get("warmup_phase_device_ms",{})
    print("[%s] value=%.1fM ms_per_step=%.1f deliver_us=%.1f frac=%.4f phases=%s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], r["avg_launch_us"], r["frac"], {k.split("(")[0]:round(v) for k,v in p.items()}))
except Exception as e:
    print("[%s] FAILED %s" % (sys.argv[1], e)); print(open(sys.argv[2].replace(".json",".err")).read()[-800:])
PY
done | tee $OUT/ab_summary.txt
