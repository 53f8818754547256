"""The city-based latency models (AwsRegionNetworkLatency, NetworkLatencyByCity, NetworkLatencyByCityWJitter —
C/NetworkLatency.java:86-233) and the city node builders (C/NodeBuilder.java:98-147, C/RegistryNodeBuilders.java:44-58)
on the engine: wg_set_latency_city probed through wg_latency_probe against the oracle for every mode, then whole
protocols built on AWS / CITIES nodes in lock-step with the oracle. Bodies also run on the CPU wave emulator
(tests/test_emu_kernels.py). Data: tests/golden/city_data.json."""
import os

import numpy as np
import pytest

import oracle_lib as o
import parity
import wittgenstein_amd as w
from wittgenstein_amd import geo
from wittgenstein_amd.core import Network

pytestmark = pytest.mark.gpu
DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "city_data.json")


def reg():
    o.load_city_data(DATA)
    return geo.register(DATA)


@pytest.mark.parametrize("model", ["AwsRegionNetworkLatency", "NetworkLatencyByCity", "NetworkLatencyByCityWJitter"])
def test_city_latency_probe_matches_the_oracle(model):
    r = reg()
    b = r["AWS"] if model.startswith("Aws") else r["CITIES"]
    tab, ping, jit = r["aws_tables"] if model.startswith("Aws") else r["city_tables"]
    mode = {"AwsRegionNetworkLatency": 0, "NetworkLatencyByCity": 1, "NetworkLatencyByCityWJitter": 2}[model]
    rng = np.random.RandomState(3)
    n = 600
    city = rng.randint(0, len(b.names), n)
    city[:len(b.names)] = np.arange(len(b.names))[:n]            # every city is some node's
    ex = rng.choice([0, 0, 0, 500], n)
    net = Network.create()
    net.add_nodes(b.merc_x[city], b.merc_y[city], extraLatency=ex)
    net.setCityLatency(mode, city, tab=tab if mode != 2 else None, ping=ping if mode == 2 else None,
                       jitter=None if mode == 1 else jit)
    k = 6000
    frm = rng.randint(0, n, k).astype(np.int32)
    to = rng.randint(0, n, k).astype(np.int32)
    frm[:30], to[:30] = np.arange(30), np.arange(30)             # from == to -> 1 (C/NetworkLatency.java:28-30)
    delta = rng.randint(0, 100, k).astype(np.int32)
    delta[:200] = np.arange(200) % 100
    got = net.latency_probe(frm, to, delta)
    nl = o.LatencyModel(model)
    for i in range(k):
        f, t = int(frm[i]), int(to[i])
        want = nl.city(b.names[city[f]], b.names[city[t]], int(delta[i]), same=(f == t), e1=int(ex[f]), e2=int(ex[t]))
        assert got[i] == want, (i, b.names[city[f]], b.names[city[t]], int(delta[i]), got[i], want)


@pytest.mark.parametrize("nb,nl", [("CITIES_SPEED=CONSTANT_TOR=0.00", "NetworkLatencyByCityWJitter"),
                                   ("CITIES_SPEED=GAUSSIAN_TOR=0.10", "NetworkLatencyByCity"),
                                   ("AWS_SPEED=CONSTANT_TOR=0.00", "AwsRegionNetworkLatency")])
def test_pingpong_on_city_nodes(nb, nl):
    reg()
    p = w.PingPong(w.PingPongParameters(400, nb, nl), seed=4)
    p.init()
    c = o.PingPong(400, nb, nl, seed=4)
    assert (p.network().read("x") == c.read("x")).all() and (p.network().read("y") == c.read("y")).all()
    for _ in range(8):
        p.network().runMs(50)
        c.run_ms(50)
        d = parity.diff_pingpong(p, c)
        assert not d, d


def test_handel_on_city_nodes():
    """HandelScenarios.defaultParams' topology (P/HandelScenarios.java:82-88: CITIES nodes, NetworkLatencyByCityWJitter)"""
    reg()
    g, c = parity.handel_pair((128, 114, 4, 50, 10, 20, 10, 12, 0), nb="CITIES_SPEED=CONSTANT_TOR=0.00",
                              nl="NetworkLatencyByCityWJitter", seed=1, config={"queue_cap": 64})
    k = 0
    while c.cont_if() and k < 400:
        g.network().runMs(10)
        c.run_ms(10)
        k += 1
        if k % 10 == 0:
            d = parity.diff_handel(g, c)
            assert not d, (k, d)
    d = parity.diff_handel(g, c)
    assert not d, d
    assert not g.cont_if() and not c.cont_if()


def test_city_latency_needs_city_nodes():
    reg()
    p = w.PingPong(w.PingPongParameters(10, "RANDOM_SPEED=CONSTANT_TOR=0.00", "NetworkLatencyByCity"))
    with pytest.raises(w.IllegalStateException):   # C/NetworkLatency.java:175-178 "default city location"
        p.init()
