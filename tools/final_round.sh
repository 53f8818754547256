#!/bin/bash
# final evidence of a round on the final code: whole GPU suite, default bench line (Handel + second_workload), rocprofv3
# kernel stats of the Handel bench, FETCH_SIZE / WRITE_SIZE passes -> traffic.json
set -u
TAG=${1:-r02z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
PYTEST_X= PYTEST_TIMEOUT=1500 bash tools/gpu_session.sh $TAG tests bench prof pmc
if [ -s $OUT/pmc_FETCH_SIZE.md ] && [ -s $OUT/pmc_WRITE_SIZE.md ]; then
  python tools/traffic_from_pmc.py $OUT/pmc_FETCH_SIZE.md $OUT/pmc_WRITE_SIZE.md 32768 16 $OUT/traffic.json > /dev/null
fi
