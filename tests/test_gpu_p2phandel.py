"""protocols.P2PHandel (P/P2PHandel.java) on the engine in host-callback mode — the third protocol over the reference's P2PNetwork
 — vs the CPU oracle (oracle/p2phandel.hpp, pinned against PT/P2PHandelTest in tests/test_oracle_protocols.py).
Compared before the run and after every chunk: per node the peer list, the four Node counters, doneAt, the position, the verified
signatures (count and an order-sensitive digest of the bits), what the node believes of its peers, and `toVerify` — its size, the
LENGTH OF ITS HASH TABLE and a digest of its elements in java.util.HashSet's ITERATION order (checkSigs2 ORs the others into the
first one, an object shared with other nodes); network.time, msgs.size(), the rd state."""
import numpy as np
import pytest

import oracle_lib as o
from examples.hostmode import p2phandel as ph


def _digest(b):  # oracle/capi.cpp bits_digest
    v, k, i, x = 0, 1, 0, b.v
    while x:
        if x & 1:
            v += k * (i + 1)
            k += 1
        x >>= 1
        i += 1
    return v


GET = {"msgReceived": lambda n: n.msgReceived, "msgSent": lambda n: n.msgSent, "bytesSent": lambda n: n.bytesSent,
       "bytesReceived": lambda n: n.bytesReceived, "doneAt": lambda n: n.doneAt, "x": lambda n: n.x, "y": lambda n: n.y,
       "sigs": lambda n: n.verifiedSignatures.cardinality(), "sigsDigest": lambda n: _digest(n.verifiedSignatures),
       "toVerify": lambda n: n.toVerify.size, "toVerifyCapacity": lambda n: n.toVerify.capacity(),
       "toVerifyOrder": lambda n: sum((k + 1) * (_digest(b) % 1000003) for k, b in enumerate(n.toVerify.items())),
       "peerCount": lambda n: len(n.peers), "peerDigest": lambda n: sum((k + 1) * q.nodeId for k, q in enumerate(n.peers)),
       "justRelay": lambda n: int(n.justRelay), "peersState": lambda n: sum(b.cardinality() for b in n.peersState.values())}


def lockstep(params, nl, seed, chunk, chunks):
    """params = P2PHandelParameters ctor order (signingNodeCount, relayingNodeCount, threshold, connectionCount, pairingTime,
    sigsSendPeriod, doubleAggregateStrategy, sendSigsStrategy, sendState)"""
    g = ph.P2PHandel(ph.P2PHandelParameters(*params, None, nl))
    g.network.rd.setSeed(seed)
    g.init()
    c = o.P2PHandel(params, None, nl, seed=seed)
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


def test_bitset_hash_and_hashset_order_known_answers():
    """java.util.BitSet.hashCode() and the HashSet model on values that can be worked out by hand from the JDK's source:
    hashCode of {} is 1234; of {0} is 1234 ^ 1 = 1235; of {64} (word 1 = 1, times 2) is 1234 ^ 2 = 1232; bit 63 reaches the
    high half: words[0] = 2^63 -> h = 2^63 ^ 1234 -> (int) ((h >> 32) ^ h) = 0x80000000 ^ 1234 (as a signed shift it also fills
    the upper bits, which the int cast drops)."""
    B = ph.BitSet
    assert B().hashCode() == 1234 and B(1).hashCode() == 1235 and B(1 << 64).hashCode() == 1232
    assert B(1 << 63).hashCode() == (0x80000000 ^ 1234)
    s = ph.JavaHashSet()
    a, b, c = B(1), B(2), B(1 | 1 << 16)            # 1235, 1232, 1234 ^ 65537 -> spread: buckets 3, 0, (h ^ h >>> 16) & 15
    for x in (a, b, c):
        assert s.add(x)
    assert not s.add(B(2)) and s.size == 3            # equal content: not added
    order = [x.v for x in s.items()]
    assert order[0] == 2 and set(order) == {1, 2, 1 | 1 << 16} and s.capacity() == 16
    b.or_(B(8))                                       # mutated after insertion: it stays in its bucket, found by identity only
    assert [x.v for x in s.items()][0] == 10 and s.add(B(2)) and s.size == 4   # the old content is addable again
    s.clear()
    assert s.size == 0 and s.capacity() == 16         # clear() keeps the table


@pytest.mark.gpu
def test_p2phandel_default_shape():  # PT/P2PHandelTest's fixture: P2PHandelScenarios.defaultParams(32, 0.0, 4, ..) on the RANDOM builder
    g, c = lockstep((32, 0, 31, 4, 4, 20, True, "dif", False), "NetworkLatencyByDistanceWJitter", seed=0, chunk=40, chunks=40)
    assert all(n.doneAt > 0 for n in g.network.allNodes)


@pytest.mark.gpu
@pytest.mark.parametrize("params", [(64, 0, 60, 3, 2, 5, True, "all", False),        # testSimpleRunWithoutState :44-55
                                    (20, 0, 20, 3, 2, 50, True, "cmp_diff", True),   # testSimpleRunWithState :57-68
                                    (40, 8, 36, 6, 3, 10, True, "cmp_all", False)])  # relaying nodes, the third strategy
def test_p2phandel_runs_to_done_in_lockstep(params):
    g, c = lockstep(params, "NetworkLatencyByDistanceWJitter", seed=1, chunk=100, chunks=30)
    assert all(n.doneAt > 0 for n in g.network.allNodes if not n.justRelay)


@pytest.mark.gpu
def test_p2phandel_single_best_strategy_batched_steps(monkeypatch):
    """checkSigs1 (the sets persist between checks and grow: iteration order, Iterator.remove and remove(best) all matter),
    through the batched-step calls"""
    monkeypatch.setenv("WG_HOST_BATCH", "1")
    g, c = lockstep((40, 0, 25, 8, 2, 5, False, "dif", True), "NetworkLatencyByDistanceWJitter", seed=0, chunk=25, chunks=14)
    assert max(n.toVerify.capacity() for n in g.network.allNodes) >= 32   # the tables did resize


@pytest.mark.gpu
def test_p2phandel_a_treeified_bucket_is_refused_like_the_oracle():
    from wittgenstein_amd.core import IllegalStateException
    g = ph.P2PHandel(ph.P2PHandelParameters(100, 0, 25, 10, 2, 5, False, "dif", True, None, "NetworkLatencyByDistanceWJitter"))
    g.network.rd.setSeed(1)
    g.init()
    c = o.P2PHandel((100, 0, 25, 10, 2, 5, False, "dif", True), None, "NetworkLatencyByDistanceWJitter", seed=1)
    with pytest.raises(o.OracleError):
        c.run_ms(1000)
    with pytest.raises(IllegalStateException):
        g.network.runMs(1000)
    assert g.network.time == c.info()["time"] == 192   # both stop in the same simulated ms
