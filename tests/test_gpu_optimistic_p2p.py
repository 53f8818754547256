"""protocols.OptimisticP2PSignature (P/OptimisticP2PSignature.java) on the engine in host-callback mode — the second protocol
over the reference's P2PNetwork beside P2PFlood — vs the CPU oracle (oracle/optimistic_p2p.hpp, pinned
against PT/OptimisticP2PSignatureTest in tests/test_oracle_protocols.py). Compared before the run and after every chunk: per
node the peer list (count and an order-sensitive digest), done, doneAt, verifiedSignatures.cardinality(), the four Node
counters, the position; network.time, msgs.size(), the rd state."""
import numpy as np
import pytest

import oracle_lib as o
from examples.hostmode import optimistic_p2p as op

GET = {"msgReceived": lambda n: n.msgReceived, "msgSent": lambda n: n.msgSent, "bytesSent": lambda n: n.bytesSent,
       "bytesReceived": lambda n: n.bytesReceived, "doneAt": lambda n: n.doneAt, "done": lambda n: int(n.done),
       "sigs": lambda n: len(n.verifiedSignatures), "peerCount": lambda n: len(n.peers),
       "peerDigest": lambda n: sum((k + 1) * q.nodeId for k, q in enumerate(n.peers)), "x": lambda n: n.x, "y": lambda n: n.y}


def lockstep(params, nl, seed, chunk, chunks):
    """params = OptimisticP2PSignatureParameters ctor order: (nodeCount, threshold, connectionCount, pairingTime)"""
    g = op.OptimisticP2PSignature(op.OptimisticP2PSignatureParameters(*params, None, nl))
    g.network.rd.setSeed(seed)
    g.init()
    c = o.OptimisticP2PSignature(params, None, nl, seed=seed)
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
def test_optimistic_p2p_simple():  # PT/OptimisticP2PSignatureTest.java:14-32 through the engine (constant-speed builder)
    n = 100
    g, c = lockstep((n, n // 2 + 1, 13, 3), None, seed=0, chunk=20, chunks=15)
    assert len(g.network.allNodes) == n and c.info()["delivered"] > 50000
    for node in g.network.allNodes:
        assert not node.isDown() and node.doneAt > 0 and node.done and len(node.verifiedSignatures) > n // 2


@pytest.mark.gpu
def test_optimistic_p2p_copy_and_high_threshold():  # :34-50's parameters: two copies agree with each other and the oracle
    g1, c = lockstep((200, 160, 10, 2), None, seed=7, chunk=25, chunks=8)
    g2 = g1.copy()
    g2.network.rd.setSeed(7)
    g2.init()
    g2.network.runMs(200)
    for n1, n2 in zip(g1.network.allNodes, g2.network.allNodes):
        assert (n1.done, n1.doneAt) == (n2.done, n2.doneAt)
    assert c.info()["delivered"] > 100000


@pytest.mark.gpu
def test_optimistic_p2p_without_latency_batched_steps(monkeypatch):
    """the same run through the batched-step calls (wg_step_begin / wg_step_end: one round trip per simulated ms instead
    of one per delivery)"""
    monkeypatch.setenv("WG_HOST_BATCH", "1")
    g, c = lockstep((120, 100, 8, 1), "NetworkNoLatency", seed=3, chunk=5, chunks=6)
    assert (c.read("done") == 1).all()


@pytest.mark.gpu
def test_envelope_ring_overflow_is_loud():
    """an envelope ring (wg_config.chain_slots) that comes round onto an envelope with destinations still to reach is an
    error, not a corrupted queue"""
    from wittgenstein_amd.core import EngineCapacityError
    g = op.OptimisticP2PSignature(op.OptimisticP2PSignatureParameters(100, 51, 13, 3, None, None), config={"chain_slots": 256})
    g.init()
    with pytest.raises(EngineCapacityError):
        g.network.runMs(300)
