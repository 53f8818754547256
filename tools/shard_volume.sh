#!/bin/bash
# words per exchange of BASELINE config 3 as 8 logical shards on one MI355X, the snapshots owner-directed (round 5) and inside the
# all-reduce image (rounds 1-4): the measured base of DESIGN.md section 7.2's table.   bash tools/shard_volume.sh <tag>
TAG=${1:-shardvol}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for mode in directed image; do
  v=0; [ $mode = image ] && v=1
  WG_TEST_SHARD_IMAGE=$v timeout 1500 python -m pytest tests/test_gpu_shards.py -x -q -m gpu > $OUT/pytest_$mode.log 2>&1; echo "$mode rc=$?"; tail -1 $OUT/pytest_$mode.log
  cp gpurun_out/shards_result.json $OUT/shards_result_$mode.json
done
python - $OUT <<'PY'
import json, sys
out = {}
for mode in ("directed", "image"):
    d = json.load(open("%s/shards_result_%s.json" % (sys.argv[1], mode)))["config3_as_8_shards"]
    out[mode] = {k: d[k] for k in ("snapshots", "simulated_ms", "words_received_per_shard", "by_exchange_shard0", "by_exchange_shard7", "run_s", "bad")}
json.dump(out, open("%s/shard_volume.json" % sys.argv[1], "w"), indent=1)
for mode, d in out.items():
    ms = d["simulated_ms"]
    print(mode, "ms", ms, {k: round(4.0 * v[1] / ms) for k, v in d["by_exchange_shard0"].items()}, "bytes per simulated ms, shard 0")
PY
