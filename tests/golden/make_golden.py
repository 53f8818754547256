#!/usr/bin/env python
"""Generates tests/golden/*.json from the CPU oracle (oracle/liboracle.so).

NOT reference output: no JVM exists in the build image, so these are traces of the C++ restatement
("parity unpinned" for seed-dependent outputs). They pin the oracle against accidental
change and give the GPU tests a fixture that does not need the oracle at run time.
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # (wittgenstein_amd.protocols.choose_attesters, pure Python)
import oracle_lib as o  # noqa: E402

NB, NL = "RANDOM_SPEED=CONSTANT_TOR=0.00", "NetworkLatencyByDistanceWJitter"


def digest(a):
    return hashlib.sha256(a.tobytes()).hexdigest()[:16]


def pingpong(n, seed):
    p = o.PingPong(n, NB, NL, seed=seed)
    steps = []
    for _ in range(12):  # P/PingPong.java:94-101 main(): runMs(50) steps
        p.run_ms(50)
        steps.append(int(p.read("pong")[0]))
    return {"protocol": "PingPong", "nodes": n, "seed": seed, "pong0_every_50ms": steps, "rng": p.info()["rng"],
            "msgReceived": digest(p.read("msgReceived")), "bytesSent": digest(p.read("bytesSent"))}


def handel(params, seed, chunk=10):
    n, thr, pair, lw, ec, per, fp, down, desync = params
    h = o.Handel(n, thr, pair, lw, ec, per, fp, down, NB, NL, desync, seed=seed)
    while True:  # C/RunMultipleTimes.java:50-64
        did = h.run_ms(chunk)
        if did and not h.cont_if():
            break
    i = h.info(False)
    return {"protocol": "Handel", "params": list(params), "seed": seed, "chunk": chunk, "time": i["time"],
            "delivered": i["delivered"], "rng": i["rng"],
            "doneAt": [int(v) for v in h.read("doneAt")] if n <= 64 else digest(h.read("doneAt")),
            "sigsChecked": digest(h.read("sigsChecked")), "msgReceived": digest(h.read("msgReceived")),
            "totalIncoming": digest(h.read_bits("totalIncoming")),
            "deliveredByLevel": [int(v) for v in h.stats()["deliveredByLevel"]]}


SCALARS = ["doneAt", "msgReceived", "msgSent", "bytesSent", "bytesReceived", "sigsChecked", "sigQueueSize",
           "msgFiltered", "currWindowSize", "addedCycle"]
LEVELS = ["posInLevel", "outgoingFinished", "queueLen"]
BITS = ["totalIncoming", "lastAggVerified", "verifiedIndSignatures", "toVerifyInd", "finishedPeers"]


def handel_full(params, seed, chunk=10, marks=()):
    """BASELINE config 3 at full size: the run loop of C/RunMultipleTimes.java:50-64 to the stop
    predicate (P/Handel.java:1044-1053), with a digest of EVERY observable tests/parity.py compares in lock-step —
    per-node scalars, per-(node, level) scalars, the five bitset rows (P/Handel.java:349-352, 373-394) — at the end
    and, for `marks` (simulated times), on the way. ~7.5 min and ~17 GB on one core at 32 768 nodes."""
    n, thr, pair, lw, ec, per, fp, down, desync = params
    h = o.Handel(n, thr, pair, lw, ec, per, fp, down, NB, NL, desync, seed=seed)

    def snap():
        i = h.info(False)
        d = {"time": i["time"], "delivered": i["delivered"], "rng": i["rng"]}
        for f in SCALARS:
            d[f] = digest(h.read(f))
        for f in LEVELS:
            d[f] = digest(h.read_level(f))
        for f in BITS:
            d[f] = digest(h.read_bits(f))
        return d
    at = {}
    while True:
        did = h.run_ms(chunk)
        t = h.info(False)["time"]
        if t in marks:
            at[str(t)] = snap()
        if did and not h.cont_if():
            break
    out = {"protocol": "Handel", "params": list(params), "seed": seed, "chunk": chunk, "init_s": h.init_seconds(),
           "final": snap(), "marks": at, "deliveredByLevel": [int(v) for v in h.stats()["deliveredByLevel"]],
           "done_nodes": int((h.read("doneAt") > 0).sum())}
    return out


CASPER_FIELDS = ["msgReceived", "msgSent", "bytesSent", "bytesReceived", "headHeight", "headProposalTime", "headId",
                 "attestationsByHeadSize", "blocksReceived", "attestationsHeld"]


def casper_config5_shape(per, stopped_frac=0.10, seed=0, chunk=8000, chunks=3):
    """BASELINE config 5's shape (cycleLength 64, 5 block producers, `per` attesters voting per slot, every vote and block a
    sendAll to all nodes; stopped_frac of the attesters stop()ped after init(): SURVEY.md §8d's "+10 %") for chunks x chunk
    simulated ms: digests of every observable tests/test_zr_gpu_casper_resident.py::diff compares, after every chunk.
    per = 1024 (65 542 nodes, 134 M deliveries): ~8 min on one core. The oracle keeps every attestation in every node's
    HashSet, so config 5's own 4096 per slot (16 x the events) is out of a fixture's reach."""
    from wittgenstein_amd import protocols as P  # (choose_attesters: pure Python, java.util.Random restated)
    params = (64, False, 5, per, 1000, 1)
    c = o.CasperIMD(params, None, None, seed=seed)
    first = 1 + params[2]
    ids = P.choose_attesters(range(first, first + params[0] * per), int(stopped_frac * params[0] * per), seed=seed + 1)
    if ids:
        c.stop(ids)
    marks = []
    for _ in range(chunks):
        c.run_ms(chunk)
        i = c.info()
        d = {"time": i["time"], "rng": i["rng"], "queue": i["queue"], "delivered": i["delivered"]}
        for f in CASPER_FIELDS:
            d[f] = digest(c.read(f))
        marks.append(d)
    return {"protocol": "CasperIMD", "params": list(params), "seed": seed, "chunk": chunk, "stopped": len(ids),
            "stop_seed": seed + 1, "nodes": 1 + params[2] + params[0] * per, "marks": marks,
            "observer_head_height": int(c.read("headHeight")[0])}


if __name__ == "__main__":
    o.build()
    if len(sys.argv) > 1 and sys.argv[1] == "casper":
        per = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
        res = casper_config5_shape(per)
        with open(os.path.join(HERE, "casper_config5_shape_%d.json" % res["nodes"]), "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
        print("wrote Casper trace at", res["nodes"], "nodes:", res["marks"][-1]["delivered"], "deliveries")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "config3":
        # separate file: tests/golden/handel_config3_32768.json (regenerating the small traces takes seconds, this 8 min)
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
        down = int(n * 0.10)
        res = handel_full((n, int(n * 0.9 * 0.99), 4, 50, 10, 20, 10, down, 0), 0, marks=(200, 600, 1000))
        with open(os.path.join(HERE, "handel_config3_%d.json" % n), "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
        print("wrote config 3 trace at", n, "nodes: time", res["final"]["time"], "delivered", res["final"]["delivered"])
        sys.exit(0)
    out = {"pingpong_1000_seed0": pingpong(1000, 0), "pingpong_1000_seed3": pingpong(1000, 3),
           "handel_64_handeltest": handel((64, 60, 6, 10, 5, 5, 10, 2, 100), 0),
           "handel_256_seed0": handel((256, 228, 4, 50, 10, 20, 10, 25, 0), 0),
           "handel_1024_seed1": handel((1024, 912, 4, 50, 10, 20, 10, 102, 0), 1)}
    with open(os.path.join(HERE, "oracle_traces.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", len(out), "traces")
