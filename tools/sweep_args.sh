#!/bin/bash
# A/B of bench.py arguments / environment, one process each: bash tools/sweep_args.sh <tag> "ENV=1 -- --batches 2" ...
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
for spec in "$@"; do
  envs=${spec%%--*}; args=${spec#*-- }
  [ "$envs" = "$spec" ] && envs="" && args="$spec"
  name=$(echo "$spec" | tr ' =/' '___' | tr -s '_-')
  env $envs timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --no-second $args > $OUT/b_$name.json 2> $OUT/b_$name.err
  python - "$spec" $OUT/b_$name.json <<'PY' | tee -a $OUT/sweep.txt
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print("%-44s %7.1f M msgs/s  step %7.1f ms" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"]))
except Exception as x:
    print(sys.argv[1], "FAILED", x)
PY
done
