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


if __name__ == "__main__":
    o.build()
    out = {"pingpong_1000_seed0": pingpong(1000, 0), "pingpong_1000_seed3": pingpong(1000, 3),
           "handel_64_handeltest": handel((64, 60, 6, 10, 5, 5, 10, 2, 100), 0),
           "handel_256_seed0": handel((256, 228, 4, 50, 10, 20, 10, 25, 0), 0),
           "handel_1024_seed1": handel((1024, 912, 4, 50, 10, 20, 10, 102, 0), 1)}
    with open(os.path.join(HERE, "oracle_traces.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", len(out), "traces")
