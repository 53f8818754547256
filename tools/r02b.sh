#!/bin/bash
# second GPU call of the session: shard fix, PMC calibration, shard bench, Handel 65 536, Casper at larger sizes
set -u
OUT=gpurun_out/r02b; mkdir -p $OUT
free -g | tee $OUT/host.txt; nproc | tee -a $OUT/host.txt
timeout 600 python -m pytest tests/test_zz_gpu_shards.py -m gpu -q > $OUT/pytest_shards.log 2>&1; echo "shards rc=$?"; tail -5 $OUT/pytest_shards.log
bash tools/pmc_calib.sh r02b > /dev/null 2>&1; cat $OUT/pmc_calib.txt
timeout 600 python bench.py --mode shard --gpus 1 --steps 1 --warmup 1 --no-cpu > $OUT/bench_shard1.json 2> $OUT/bench_shard1.err
echo "shard1 rc=$?"; cat $OUT/bench_shard1.json; tail -3 $OUT/bench_shard1.err
for cl in 8 16; do
  timeout 900 python bench.py --workload casper --casper-cycle-length $cl --casper-producers 5 --steps 1 --warmup 0 --casper-ms 24000 > $OUT/bench_casper_cl$cl.json 2> $OUT/bench_casper_cl$cl.err
  echo "casper cl=$cl rc=$?"; cat $OUT/bench_casper_cl$cl.json; tail -3 $OUT/bench_casper_cl$cl.err
done
AVAIL=$(free -g | awk '/^Mem:/{print $7}')
TH=$(( AVAIL / 45 )); [ $TH -gt 4 ] && TH=4
if [ $TH -ge 1 ]; then
  timeout 1500 python bench.py --nodes 65536 --replicas 4 --init-threads $TH --steps 1 --warmup 0 --no-cpu > $OUT/bench_handel65536.json 2> $OUT/bench_handel65536.err
  echo "handel65536 rc=$?"; cat $OUT/bench_handel65536.json; tail -5 $OUT/bench_handel65536.err
else
  echo "handel65536 skipped: only $AVAIL GB of host memory available"
fi
