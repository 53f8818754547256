#!/bin/bash
# Investigation build: in-kernel cycle counters (KPROF_MARK) of the Handel node-visit kernels.
# Builds wittgenstein_amd/libwittgpu_kprof.so (-DWG_KPROF) and runs one un-timed RunMultipleTimes pass with it; the
# counters are those of the batch's FIRST member (s_memtime units, summed over its wavefronts).
# KPROF_WORKLOAD=gsf KPROF_R=64 KPROF_N=4096: GSFSignature; the slots then mean (the printed names are Handel's): 00 node visits of
# k_deliver_inbox, 01 cycles node_begin, 02 cycles of the visit, 03 nodes passed over (delivered by a lean kernel), 04 / 05 / 06 / 07
# cycles of updateVerifiedSignatures' rows / scalars / accelerated calls / doneAt, 09 onNewSig calls of the wavefront path, 11 updates,
# 12 improving updates, 14 / 30 / 15 cycles of an event's prologue / action() / epilogue.
OUT=gpurun_out/${1:-kprof}; mkdir -p $OUT
bash wittgenstein_amd/csrc/build.sh -DWG_KPROF -o $(pwd)/wittgenstein_amd/libwittgpu_kprof.so 2>&1 | grep -E "error"
WG_LIB=$(pwd)/wittgenstein_amd/libwittgpu_kprof.so python - <<'PY' 2>&1 | tee $OUT/kprof.txt
import sys, os
sys.path.insert(0, os.getcwd())
import bench, wittgenstein_amd as w
R = int(os.environ.get("KPROF_R", "16")); n = int(os.environ.get("KPROF_N", "32768"))
sims, batch = bench.make_batch(w, n, range(R), 0, R, os.environ.get("KPROF_WORKLOAD", "handel"))
sims[0].network().profile(1)
batch.run_multiple_times(chunk=10, maxTime=20000)
pr = sims[0].network().profile_read()
for k, v in pr.items():
    print("%-44s %16.0f" % (k, v["total_ns"]))
PY
