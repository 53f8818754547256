OUT=gpurun_out/r12e; mkdir -p $OUT
run() { name=$1; shift; env "$@" > /dev/null 2>&1; }
for cfg in "R=1" "R=1 WG_NODE_GRID=256" "R=1 WG_NODE_GRID=512" "R=1 WG_GRAPH=1" "R=1 WG_GRAPH=1 WG_NODE_GRID=256" "R=2" "R=4" "R=8" "R=12"; do
  R=$(echo $cfg | sed 's/R=\([0-9]*\).*/\1/'); envs=$(echo $cfg | sed 's/R=[0-9]* *//')
  env $envs timeout 600 python bench.py --replicas $R --steps 3 --warmup 1 --no-cpu --no-second > $OUT/one.json 2> $OUT/one.err
  python - "$cfg" $OUT/one.json <<'PY' | tee -a $OUT/copies_sweep.txt
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-40s %7.1f M msgs/s  step %7.1f ms" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"]))
except Exception as x:
    print(sys.argv[1], "FAILED", x)
PY
done
