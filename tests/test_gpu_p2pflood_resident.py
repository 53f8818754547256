"""P2PFlood (P/P2PFlood.java over C/P2PNetwork.java, C/messages/FloodMessage.java) RESIDENT on the device
(wittgenstein_amd/csrc/proto_p2pflood.hip.h): every first receipt forwards to the peers, shuffled with the shared rd, as
one MultipleDestWithDelayEnvelope created by `resolve` (deferred shuffle + explicit arrivals). Against the CPU oracle
(oracle/p2pflood.hpp, pinned against PT/P2PFloodTest.testSimpleRun) after every chunk: received set size, doneAt, peer
count, the four Node counters; network.time, msgs.size(), rd state."""
import numpy as np
import pytest

import oracle_lib as o
from wittgenstein_amd import protocols as P

FIELDS = {"msgReceived": "msgReceived", "msgSent": "msgSent", "bytesSent": "bytesSent", "bytesReceived": "bytesReceived",
          "doneAt": "doneAt", "down": "down", "floodReceived": "received", "peerCount": "peerCount", "x": "x", "y": "y"}


def diff(g, c):
    net, out = g.network(), []
    for gf, cf in FIELDS.items():
        a, b = net.read(gf), c.read(cf)
        bad = np.nonzero(a != b)[0]
        if len(bad):
            out.append("%s: %d nodes differ, first node %d: device %d oracle %d" % (gf, len(bad), bad[0], a[bad[0]], b[bad[0]]))
    i = c.info()
    mine = (net.time, net.rng_state(), net.msgs.size())
    if mine != (i["time"], i["rng"], i["queue"]):
        out.append("time / rd / msgs.size(): device %r oracle %r" % (mine, i))
    return out


def lockstep(params, nl, seed, chunk, chunks):
    """params = P2PFloodParameters ctor order: (nodeCount, deadNodeCount, delayBeforeResent, msgCount, msgToReceive,
    peersCount, delayBetweenSends)"""
    g = P.P2PFlood(P.P2PFloodParameters(*params, None, nl), seed=seed)
    g.init()
    c = o.P2PFlood(params, None, nl, seed=seed)
    assert not diff(g, c), "after init()"
    for k in range(chunks):
        g.network().runMs(chunk)
        c.run_ms(chunk)
        d = diff(g, c)
        assert not d, "t=%d: %s" % (g.network().time, d)
    return g, c


@pytest.mark.gpu
def test_simple_run():  # PT/P2PFloodTest.java:12-31 (constant-speed builder): every live node holds the message once
    g, c = lockstep((100, 10, 50, 1, 1, 10, 30), "NetworkNoLatency", seed=0, chunk=1000, chunks=20)
    rec, down = g.network().read("floodReceived"), g.network().read("down")
    assert ((rec == 1) == (down == 0)).all()


@pytest.mark.gpu
def test_three_messages_by_distance():
    g, c = lockstep((300, 20, 20, 3, 1, 6, 10), None, seed=4, chunk=100, chunks=40)
    assert c.info()["delivered"] > 4000


@pytest.mark.gpu
def test_dense_graph_one_ms_delays_every_ms():
    lockstep((64, 0, 5, 2, 1, 12, 1), "NetworkFixedLatency(7)", seed=9, chunk=1, chunks=600)


@pytest.mark.gpu
def test_2000_nodes():  # PT/P2PFloodTest.testCopy's size
    g, c = lockstep((2000, 10, 50, 1, 1, 10, 30), None, seed=0, chunk=250, chunks=8)
    assert c.info()["delivered"] > 10000
