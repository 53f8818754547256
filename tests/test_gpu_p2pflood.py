"""The reference's P2P layer (C/P2PNetwork.java, C/P2PNode.java, C/messages/FloodMessage.java) and P2PFlood
(P/P2PFlood.java) on the engine in host-callback mode vs the CPU oracle (oracle/p2pflood.hpp, pinned against
PT/P2PFloodTest.testSimpleRun): the peer graph is built with the shared rd (setPeers), every flood hop is a
MultipleDestWithDelayEnvelope. Compared before the run and after every chunk: per node peers (count and an
order-sensitive digest), received set size, doneAt, down, the four Node counters; network.time, msgs.size(), rd."""
import numpy as np
import pytest

import oracle_lib as o
from examples.hostmode import p2p

GET = {"msgReceived": lambda n: n.msgReceived, "msgSent": lambda n: n.msgSent, "bytesSent": lambda n: n.bytesSent,
       "bytesReceived": lambda n: n.bytesReceived, "doneAt": lambda n: n.doneAt, "down": lambda n: int(n.down),
       "received": lambda n: len(n.getMsgReceived(-1)), "peerCount": lambda n: len(n.peers),
       "peerDigest": lambda n: sum((k + 1) * q.nodeId for k, q in enumerate(n.peers)), "x": lambda n: n.x,
       "y": lambda n: n.y}


def lockstep(params, nl, seed, chunk, chunks):
    """params = P2PFloodParameters ctor order: (nodeCount, deadNodeCount, delayBeforeResent, msgCount, msgToReceive,
    peersCount, delayBetweenSends)"""
    g = p2p.P2PFlood(p2p.P2PFloodParameters(*params, None, nl))
    g.network.rd.setSeed(seed)
    g.init()
    c = o.P2PFlood(params, None, nl, seed=seed)
    for k in range(chunks + 1):
        for f, fn in GET.items():
            a, b = np.array([fn(n) for n in g.network.allNodes], np.int64), c.read(f)
            bad = np.nonzero(a != b)[0]
            assert not len(bad), "t=%d %s: %d nodes differ, first node %d: engine %d oracle %d" % (
                g.network.time, f, len(bad), bad[0], a[bad[0]], b[bad[0]])
        i = c.info()
        assert (g.network.time, g.network._eng.rng_state(), g.network.msgs.size()) == (i["time"], i["rng"], i["queue"])
        if k < chunks:
            g.network.runMs(chunk)
            c.run_ms(chunk)
    return g, c


@pytest.mark.gpu
def test_p2pflood_simple_run():  # PT/P2PFloodTest.java:12-31 through the engine (constant-speed builder)
    g, c = lockstep((100, 10, 50, 1, 1, 10, 30), "NetworkNoLatency", seed=0, chunk=1000, chunks=20)
    assert len(g.network.allNodes) == 100
    for n in g.network.allNodes:
        assert len(n.getMsgReceived(-1)) == (0 if n.isDown() else 1)


@pytest.mark.gpu
def test_p2pflood_three_messages_by_distance():
    g, c = lockstep((300, 20, 20, 3, 1, 6, 10), None, seed=4, chunk=100, chunks=40)
    assert c.info()["delivered"] > 4000
    assert all(len(n.getMsgReceived(-1)) == 3 for n in g.network.allNodes if not n.isDown())


@pytest.mark.gpu
def test_empty_destination_list_costs_a_draw():  # C/Network.java:430: the seed is drawn before the list is looked at
    g = p2p.P2PFlood(p2p.P2PFloodParameters(8, 0, 1, 1, 1, 2, 1, None, "NetworkNoLatency"))
    g.init()
    net = g.network
    before = net._eng.rng_state()
    net.send(p2p.FloodMessage(1, 0, 1), net.allNodes[0], [], net.time + 1, 1, _force_multi=True)
    assert net._eng.rng_state() == (before * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
    before = net._eng.rng_state()
    net.send(p2p.FloodMessage(1, 0, 1), net.allNodes[0], [])   # the 3-argument overload returns first (:353-356)
    assert net._eng.rng_state() == before
