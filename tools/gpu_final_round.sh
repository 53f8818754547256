#!/bin/bash
# The evidence set of a round's final code, ONE GPU session: the two PMC passes of the bench command -> profiles/traffic.json
# (stamped with the commit), then the whole GPU suite, smoke(), the driver's literal bench line (which reads that traffic.json),
# the same command under rocprofv3 --kernel-trace --stats (kernel statistics + per-phase table) and the Casper delivery
# pass's PMC passes -> profiles/traffic_casper.json.      WG_COMMIT=<hash> bash tools/gpu_final_round.sh <tag>
# (On a gpurun box only gpurun_out/ comes back: afterwards copy gpurun_out/<tag>/traffic*.json over profiles/traffic*.json in
# the repo — bench.py reads those — and the other files to profiles/<tag>_*. SKIP_SUITE=1 leaves the GPU suite to its own call.)
set -u
TAG=${1:-final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
for f in wittgenstein_amd/csrc/*; do
  if [ "$f" -nt wittgenstein_amd/libwittgpu.so ]; then echo "STALE libwittgpu.so: $f is newer"; exit 1; fi
done
# three counter passes of one bench command (bytes read, bytes written, request counts), each in its own run
passes() { pre=$1; shift
  for c in FETCH_SIZE WRITE_SIZE req; do
    ctr=$c; [ $c = req ] && ctr="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum"
    for try in 1 2 3; do  # (rocprofv3 itself segfaults now and then, more often on the 512-copy workload: a pass is tried three times)
      (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --pmc $ctr -d $R/$OUT/p_$pre$c -o k --output-format csv -- python $R/bench.py "$@" --steps 1 --warmup 0 --no-cpu --no-second > $R/$OUT/pmc_$pre$c.json 2> $R/$OUT/pmc_$pre$c.err)
      rc=$?; echo "pmc $pre$c try $try rc=$rc"
      [ $rc = 0 ] && break
      rm -rf $OUT/p_$pre$c
    done
    python tools/prof_summary.py pmc $OUT/p_$pre$c $OUT/pmc_$pre$c.md && rm -rf $OUT/p_$pre$c
  done
}
HANDEL_PASS="k_handel_lane,k_handel_update<,k_handel_lane2,k_handel_copy,k_handel_dissem<,k_handel_wave<"
passes ""
python tools/traffic_from_pmc.py $OUT/pmc_FETCH_SIZE.md $OUT/pmc_WRITE_SIZE.md 32768 $OUT/pmc_FETCH_SIZE.json $OUT/traffic.json "$HANDEL_PASS" $OUT/pmc_req.md > /dev/null
cp $OUT/traffic.json profiles/traffic.json; export WG_TRAFFIC_SESSION=1
# the side workloads of the driver's line: Handel at the north star's target size (8 copies) and GSFSignature (256 copies)
passes h65536_ --nodes 65536 --replicas 8
python tools/traffic_from_pmc.py $OUT/pmc_h65536_FETCH_SIZE.md $OUT/pmc_h65536_WRITE_SIZE.md 65536 $OUT/pmc_h65536_FETCH_SIZE.json $OUT/traffic_handel65536.json "$HANDEL_PASS" $OUT/pmc_h65536_req.md > /dev/null
cp $OUT/traffic_handel65536.json profiles/traffic_handel65536.json
passes gsf_ --workload gsf --nodes 4096 --replicas 480
python tools/traffic_from_pmc.py $OUT/pmc_gsf_FETCH_SIZE.md $OUT/pmc_gsf_WRITE_SIZE.md 4096 $OUT/pmc_gsf_FETCH_SIZE.json $OUT/traffic_gsf.json "k_gsf_docycle16,k_gsf_docycle<,k_gsf_lane,k_deliver_inbox<GsfProto" $OUT/pmc_gsf_req.md > /dev/null
cp $OUT/traffic_gsf.json profiles/traffic_gsf.json
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c -d $R/$OUT/pc_$c -o k --output-format csv -- python $R/bench.py --workload casper --casper-stopped 0.1 --steps 1 --warmup 0 --no-cpu > $R/$OUT/pmc_casper_$c.json 2> $R/$OUT/pmc_casper_$c.err)
  echo "pmc casper $c rc=$?"
  python tools/prof_summary.py pmc $OUT/pc_$c $OUT/pmc_casper_$c.md && rm -rf $OUT/pc_$c
done
python tools/traffic_from_pmc.py $OUT/pmc_casper_FETCH_SIZE.md $OUT/pmc_casper_WRITE_SIZE.md 262150 1 $OUT/traffic_casper.json "k_casper_classify,k_casper_attestations,k_deliver<CasperProto" > /dev/null
python - $OUT/traffic_casper.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); d["stopped_fraction"] = 0.1
json.dump(d, open(sys.argv[1], "w"), indent=1)
PY
cp $OUT/traffic_casper.json profiles/traffic_casper.json
[ "${SKIP_SUITE:-0}" = 1 ] || { timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log; }
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_argv.json 2> $OUT/bench_driver_argv.err; echo "bench rc=$?"; tail -3 $OUT/bench_driver_argv.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/p -o k --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-second > $R/$OUT/prof_bench.json 2> $R/$OUT/prof_bench.err)
python tools/prof_summary.py stats $OUT/p $OUT/kernel_stats.md; python tools/prof_summary.py phases $OUT/p $OUT/phases.md; rm -rf $OUT/p
# ... and of the two side workloads of the driver's line (their avg_launch_us is then checkable from rocprof too)
side() { pre=$1; shift
  for try in 1 2 3; do
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/p_$pre -o k --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-second "$@" > $R/$OUT/prof_bench_$pre.json 2> $R/$OUT/prof_bench_$pre.err)
    rc=$?; echo "prof $pre try $try rc=$rc"; [ $rc = 0 ] && break; rm -rf $OUT/p_$pre
  done
  python tools/prof_summary.py stats $OUT/p_$pre $OUT/kernel_stats_$pre.md; python tools/prof_summary.py phases $OUT/p_$pre $OUT/phases_$pre.md; rm -rf $OUT/p_$pre
}
side h65536 --nodes 65536 --replicas 8
side gsf --workload gsf --nodes 4096 --replicas 480
python - $OUT/bench_driver_argv.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f M msgs/s  ms_per_step %.1f  delivery pass %.1f us  frac %.4f  traffic %.1f MB (algorithmic %.1f MB)" % (
    d["value"] / 1e6, d["ms_per_step"], r["avg_launch_us"], r["frac"], (r["traffic"] or 0) / 1e6, r["algorithmic_bytes_per_launch"] / 1e6))
print(r.get("traffic_source")); print("cpu_baseline", d.get("cpu_baseline", {}).get("value"), "second", (d.get("second_workload") or {}).get("value"))
PY
head -12 $OUT/kernel_stats.md
