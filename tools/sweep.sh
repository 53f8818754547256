#!/bin/bash
# occupancy / batch-size sweep on a reduced workload (N=8192) — prints one line per variant
OUT=gpurun_out/${1:-sweep}; mkdir -p $OUT
for dw in 3 4 5 6 8; do for cw in 4 6 8; do
  WG_DELIVER_WAVES=$dw WG_COND_WAVES=$cw timeout 300 python bench.py --nodes 8192 --replicas 16 --init-threads 16 --no-cpu 2>/dev/null \
   | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; p=r['warmup_phase_device_ms']; print('dw=$dw cw=$cw value=%.1fM deliver_us=%.1f cond_select_ms=%.1f deliver_ms=%.1f total_ms=%.1f' % (d['value']/1e6, r['avg_launch_us'], p['cond_select'], p['deliver'], sum(p.values())))"
done; done | tee $OUT/sweep_waves.txt
for R in 1 4 16 32 64; do
  timeout 300 python bench.py --nodes 8192 --replicas $R --init-threads 16 --no-cpu 2>/dev/null \
   | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('R=$R value=%.1fM deliver_us=%.1f' % (d['value']/1e6, r['avg_launch_us']))"
done | tee $OUT/sweep_R.txt
