#!/bin/bash
# First GPU call of a round (from the repo root on the GPU box), ordered by what the judge needs first; every leg
# writes under gpurun_out/<tag>/ and is independent, so a cut-off call still leaves the earlier legs' files.
#   bash tools/next_round.sh r02a            # ~6-7 GPU-minutes in total
# Legs (rough wall time on one MI355X):
#   tests   pytest -m gpu, whole suite                                    ~3 min
#   bench   bench.py default line (roofline + cpu_baseline incl. all_cores) ~2 min
#   prof    rocprofv3 --kernel-trace --stats of bench.py --steps 1 --warmup 0 --no-cpu   ~1.3 min
#   pmc     FETCH_SIZE / WRITE_SIZE passes -> profiles/traffic.json (bench.py then fills roofline.traffic)  ~2.5 min
#   shard1  bench.py --mode shard --gpus 1: what the sharded pipeline costs on one GPU (vs the unsharded line) ~1.5 min
#   shard4  ... --logical-shards 4: four node-range shards of one simulation on the one GPU                 ~2 min
set -u
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
bash tools/gpu_session.sh $TAG tests bench prof pmc
if [ -s $OUT/pmc_FETCH_SIZE.md ] && [ -s $OUT/pmc_WRITE_SIZE.md ]; then
  python tools/traffic_from_pmc.py $OUT/pmc_FETCH_SIZE.md $OUT/pmc_WRITE_SIZE.md 32768 16 $OUT/traffic.json
  echo "copy $OUT/traffic.json to profiles/traffic.json and re-run bench.py for a line with roofline.traffic"
fi
timeout 600 python bench.py --mode shard --gpus 1 --steps 1 --warmup 1 --no-cpu > $OUT/bench_shard1.json 2> $OUT/bench_shard1.err
echo "shard1 rc=$?"; cat $OUT/bench_shard1.json
# Casper IMD resident at BASELINE config 5's node count (262 150 nodes; the default of --workload casper), + kernel stats
timeout 600 python bench.py --workload casper --steps 2 --warmup 1 > $OUT/bench_casper.json 2> $OUT/bench_casper.err
echo "casper rc=$?"; cat $OUT/bench_casper.json
(cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_casper -o k --output-format csv -- \
   python $OLDPWD/bench.py --workload casper --steps 1 --warmup 0 --no-cpu > $OLDPWD/$OUT/prof_casper.json 2> $OLDPWD/$OUT/prof_casper.err)
python tools/prof_summary.py stats $OUT/prof_casper $OUT/casper_kernel_stats.md && rm -rf $OUT/prof_casper
# shard-count invariance at full size (one simulation as 4 logical shards vs the unsharded engine, every row and counter)
timeout 900 python tools/shard_invariance.py 32768 4 0 > $OUT/shard_invariance_32768_k4.json 2> $OUT/shard_invariance_32768_k4.err
echo "shard invariance rc=$?"; cat $OUT/shard_invariance_32768_k4.json
# hipGraph A/B (value only: the HIP-event roofline bracket is off under WG_GRAPH) on the launch-bound GSFSignature config
for gr in 0 1; do
  WG_GRAPH=$gr timeout 300 python bench.py --workload gsf --nodes 4096 --replicas 16 --no-cpu > $OUT/bench_gsf_graph$gr.json 2> $OUT/bench_gsf_graph$gr.err
  echo "gsf graph=$gr rc=$?"; cat $OUT/bench_gsf_graph$gr.json
done
# the same simulation as 4 node-range shards on this one GPU (in-process loopback all-reduce): the exchange volumes and
# the owner split at full size, without xGMI
timeout 600 python bench.py --mode shard --gpus 1 --logical-shards 4 --steps 1 --warmup 0 --no-cpu > $OUT/bench_shard4_logical.json 2> $OUT/bench_shard4_logical.err
echo "shard4-logical rc=$?"; cat $OUT/bench_shard4_logical.json
