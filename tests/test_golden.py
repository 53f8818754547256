"""tests/golden/oracle_traces.json (oracle-generated, see make_golden.py): the oracle must still produce
them (CPU), and the MI355X engine must produce them without the oracle in the loop (GPU)."""
import hashlib
import json
import os

import numpy as np
import pytest

import parity

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_traces.json")))


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def test_oracle_reproduces_golden(oracle):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    assert mg.pingpong(1000, 0) == GOLD["pingpong_1000_seed0"]
    assert mg.handel((64, 60, 6, 10, 5, 5, 10, 2, 100), 0) == GOLD["handel_64_handeltest"]
    assert mg.handel((256, 228, 4, 50, 10, 20, 10, 25, 0), 0) == GOLD["handel_256_seed0"]


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["pingpong_1000_seed0", "pingpong_1000_seed3"])
def test_gpu_pingpong_golden(key):
    import wittgenstein_amd as w
    g0 = GOLD[key]
    g = w.PingPong(w.PingPongParameters(g0["nodes"], parity.NB, parity.NL), seed=g0["seed"])
    g.init()
    steps = []
    for _ in range(12):
        g.network().runMs(50)
        steps.append(int(g.network().read("pong")[0]))
    assert steps == g0["pong0_every_50ms"]
    assert g.network().rng_state() == g0["rng"]
    assert digest(g.network().read("msgReceived")) == g0["msgReceived"]
    assert digest(g.network().read("bytesSent")) == g0["bytesSent"]


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["handel_64_handeltest", "handel_256_seed0", "handel_1024_seed1"])
def test_gpu_handel_golden(key):
    import wittgenstein_amd as w
    g0 = GOLD[key]
    n, thr, pair, lw, ec, per, fp, down, desync = g0["params"]
    # (the HandelTest parameters queue up to 39 entries per level with runMs(10) chunks: raise the default cap of 32)
    g = w.Handel(w.HandelParameters(n, thr, pair, lw, ec, per, fp, down, parity.NB, parity.NL, desync), seed=g0["seed"],
                 config={"queue_cap": 64})
    g.init()
    net = g.network()
    delivered = 0
    while True:  # C/RunMultipleTimes.java:50-64
        did = net.runMs(g0["chunk"])
        delivered += net.last_stats["delivered"]
        if did and not g.cont_if():
            break
    assert net.time == g0["time"] and delivered == g0["delivered"] and net.rng_state() == g0["rng"]
    done = net.read("doneAt")
    assert ([int(v) for v in done] if n <= 64 else digest(done)) == g0["doneAt"]
    assert digest(net.read("sigsChecked")) == g0["sigsChecked"]
    assert digest(net.read("msgReceived")) == g0["msgReceived"]
    assert digest(net.read_bits("totalIncoming")) == g0["totalIncoming"]
    assert [int(v) for v in net.delivered_by_level()[:len(g0["deliveredByLevel"])]] == g0["deliveredByLevel"]
