#!/usr/bin/env python
"""BASELINE.json config 4's WORKLOAD — Handel, 131 072 nodes, 10 % dead — run once at full size on ONE
MI355X, unsharded (the rows, receptionRanks and emission lists of 131 072 nodes are ~ 240 GB: they fit the 288 GB of one
GPU), through the size-independent checks of tests/test_gpu_handel.py::test_full_size_properties_32768: every live node
done, message accounting closes, verifiedInd and toVerifyInd disjoint, stopped nodes silent and absent from every
totalIncoming row, every node holding its own signature. The oracle cannot hold this size (parity by invariants).
usage: config4_unsharded.py [nodes=131072] [seed=0]      (prints one JSON line; exit 1 on a failed check)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def run(n=131072, seed=0):
    import wittgenstein_amd as w
    NB, NL = "RANDOM_SPEED=CONSTANT_TOR=0.00", "NetworkLatencyByDistanceWJitter"
    down = int(n * 0.10)
    params = (n, int(n * (1 - 0.10) * 0.99), 4, 50, 10, 20, 10, down)  # bench.py's handel_params(n)
    t0 = time.perf_counter()
    g = w.Handel(w.HandelParameters(*params, NB, NL, 0), seed=seed)
    g.init()
    init_s = time.perf_counter() - t0
    net = g.network()
    dev_bytes = net.device_bytes() if hasattr(net, "device_bytes") else None
    t1 = time.perf_counter()
    delivered, steps = 0, 0
    while True:  # C/RunMultipleTimes.java:50-64
        did = net.runMs(10)
        delivered += net.last_stats["delivered"]
        steps += 1
        if did and not g.cont_if():
            break
        assert net.time < 20000, "no convergence"
    run_s = time.perf_counter() - t1
    live = net.read("down") == 0
    done = net.read("doneAt")
    checks = {}
    checks["live_nodes"] = int(live.sum()) == n - down
    checks["every_live_node_done"] = bool((done[live] > 0).all())
    checks["no_stopped_node_done"] = bool((done[~live] == 0).all())
    recv, sent = net.read("msgReceived"), net.read("msgSent")
    checks["accounting_closes"] = int(recv.sum()) == int(net.delivered_by_level().sum()) == delivered
    checks["stopped_nodes_silent"] = bool((recv[~live] == 0).all() and (sent[~live] == 0).all())
    vi = net.read_bits("verifiedIndSignatures")
    tv = net.read_bits("toVerifyInd")
    checks["verifiedInd_and_toVerifyInd_disjoint"] = bool(((vi & tv) == 0).all())
    del vi, tv
    ti = net.read_bits("totalIncoming")
    # (|totalIncoming| >= threshold at the END is not an invariant: updateVerifiedSignatures clears lastAggVerified when the
    # new aggregate intersects it, P/Handel.java:716-724, so totalIncoming can shrink after doneAt was set)
    own = (ti[np.arange(n), np.arange(n) >> 6] >> (np.arange(n) & 63).astype(np.uint64)) & np.uint64(1)
    checks["own_signature_held"] = bool((own[live] == 1).all())  # HLevel() of level 0, :413-421
    checks["no_signature_of_a_stopped_node"] = True
    dead_ids = np.nonzero(~live)[0]
    for d in dead_ids[:64]:  # a stopped node never sends: nobody holds its signature
        if ((ti[live, d >> 6] >> np.uint64(d & 63)) & np.uint64(1)).any():
            checks["no_signature_of_a_stopped_node"] = False
    out = {"nodes": n, "seed": seed, "params": list(params), "init_s": round(init_s, 1), "run_s": round(run_s, 2),
           "runMs10_calls": steps, "time": net.time, "delivered": int(delivered),
           "delivered_msgs_per_s": delivered / run_s, "device_bytes": dev_bytes,
           "doneAt_max": int(done.max()), "checks": checks, "ok": all(checks.values())}
    out["init_on_device"] = bool(g.init_on_device)
    return out


def main():
    out = run(int(sys.argv[1]) if len(sys.argv) > 1 else 131072, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(json.dumps(out), flush=True)
    sys.exit(0 if out["ok"] else 1)


if __name__ == "__main__":
    main()
