"""San Fermin (P/SanFerminSignature.java + P/SanFerminHelper.java) on the engine in host-callback mode vs the CPU
oracle (oracle/sanfermin.hpp, pinned against PT/SanFerminTest): swap requests / replies (single and multi-destination
sends), timeout and pairing tasks, pickNextNodes' Collections.shuffle on the shared rd. Compared after every chunk:
per node aggValue, currentPrefixLength, doneAt, thresholdAt, sent / received requests, done, isSwapping, the four Node
counters; network.time, msgs.size(), rd state; finishedNodes."""
import numpy as np
import pytest

import oracle_lib as o
from examples.hostmode import sanfermin as sf

GET = {"msgReceived": lambda n: n.msgReceived, "msgSent": lambda n: n.msgSent, "bytesSent": lambda n: n.bytesSent,
       "bytesReceived": lambda n: n.bytesReceived, "aggValue": lambda n: n.aggValue,
       "currentPrefixLength": lambda n: n.currentPrefixLength, "doneAt": lambda n: n.doneAt,
       "thresholdAt": lambda n: n.thresholdAt, "sentRequests": lambda n: n.sentRequests,
       "receivedRequests": lambda n: n.receivedRequests, "done": lambda n: int(n.done),
       "isSwapping": lambda n: int(n.isSwapping), "x": lambda n: n.x, "y": lambda n: n.y}


def lockstep(params, seed, chunk, chunks, nl=None):
    """params = SanFerminSignatureParameters ctor order: (nodeCount, threshold, pairingTime, signatureSize,
    replyTimeout, candidateCount)"""
    g = sf.SanFerminSignature(sf.SanFerminSignatureParameters(*params, False, None, nl))
    g.network.rd.setSeed(seed)  # RunMultipleTimes: copy, rd.setSeed(i), init()
    g.init()
    c = o.SanFerminSignature(params, None, nl, seed=seed)
    for k in range(chunks):
        g.network.runMs(chunk)
        c.run_ms(chunk)
        for f, fn in GET.items():
            a, b = np.array([fn(n) for n in g.allNodes], np.int64), c.read(f)
            bad = np.nonzero(a != b)[0]
            assert not len(bad), "t=%d %s: %d nodes differ, first node %d: engine %d oracle %d" % (
                g.network.time, f, len(bad), bad[0], a[bad[0]], b[bad[0]])
        i = c.info()
        assert (g.network.time, g.network._eng.rng_state(), g.network.msgs.size(), len(g.finishedNodes)) == \
               (i["time"], i["rng"], i["queue"], i["finished"])
    return g, c


@pytest.mark.gpu
def test_sanfermin_64_matches_oracle():
    g, c = lockstep((64, 64, 2, 48, 300, 1), seed=3, chunk=50, chunks=40)
    assert len(g.finishedNodes) >= 56 and all(n.aggValue == 64 for n in g.finishedNodes)


@pytest.mark.gpu
def test_sanfermin_default_size_multi_candidates():  # the no-arg parameters' size (:69-81), 3 candidates per try
    g, c = lockstep((1024, 1024, 2, 48, 300, 3), seed=0, chunk=100, chunks=40)
    assert c.info()["delivered"] > 20000 and len(g.finishedNodes) > 900


@pytest.mark.gpu
def test_sanfermin_fixed_latency_short_timeout():
    lockstep((128, 100, 3, 48, 40, 2), seed=8, chunk=10, chunks=100, nl="NetworkFixedLatency(25)")
