"""San Fermin, Cappos' variant (P/SanFerminCappos.java) on the engine in host-callback mode
(examples/hostmode/sanfermin_cappos.py) vs the CPU oracle (oracle/sanfermin_cappos.hpp): Swap messages with and without
a reply wanted (single and multi-destination sends), timeout and pairing tasks, pickNextNodes' shuffle on the shared rd.
Compared after every chunk: per node totalNumberOfSigs(-1), currentPrefixLength, doneAt, thresholdAt, the size of the
signature cache, done, isSwapping, the four Node counters, positions; network.time, msgs.size(), rd state, finishedNodes."""
import numpy as np
import pytest

import oracle_lib as o
from examples.hostmode import sanfermin_cappos as sc

GET = {"msgReceived": lambda n: n.msgReceived, "msgSent": lambda n: n.msgSent, "bytesSent": lambda n: n.bytesSent,
       "bytesReceived": lambda n: n.bytesReceived, "totalNumberOfSigs": lambda n: n.totalNumberOfSigs(-1),
       "currentPrefixLength": lambda n: n.currentPrefixLength, "doneAt": lambda n: n.doneAt,
       "thresholdAt": lambda n: n.thresholdAt, "cachedLevels": lambda n: len(n.signatureCache),
       "cachedValues": lambda n: sum(len(v) for v in n.signatureCache.values()), "done": lambda n: int(n.done),
       "isSwapping": lambda n: int(n.isSwapping), "x": lambda n: n.x, "y": lambda n: n.y}


def lockstep(params, seed, chunk, chunks, nl=None, batched=None):
    """params = SanFerminParameters ctor order: (nodeCount, threshold, pairingTime, signatureSize, timeout, candidateCount)"""
    g = sc.SanFerminCappos(sc.SanFerminParameters(*params, None, nl), batched=batched)
    g.network.rd.setSeed(seed)  # RunMultipleTimes: copy, rd.setSeed(i), init() — which builds the nodes
    g.init()
    c = o.SanFerminCappos(params, None, nl, seed=seed)
    for k in range(chunks):
        g.network.runMs(chunk)
        c.run_ms(chunk)
        for f, fn in GET.items():
            a, b = np.array([fn(n) for n in g.allNodes], np.int64), c.read(f)
            bad = np.nonzero(a != b)[0]
            assert not len(bad), "t=%d %s: %d nodes differ, first node %d: engine %d oracle %d" % (
                g.network.time, f, len(bad), bad[0], a[bad[0]], b[bad[0]])
        i = c.info()
        assert (g.network.time, g.network._eng.rng_state(), g.network.msgs.size(), len(g.params.finishedNodes)) == \
               (i["time"], i["rng"], i["queue"], i["finished"])
    return g, c


@pytest.mark.gpu
def test_cappos_64_matches_oracle():
    g, c = lockstep((64, 32, 2, 48, 150, 4), seed=3, chunk=50, chunks=40)
    assert len(g.params.finishedNodes) >= 56 and all(n.totalNumberOfSigs(-1) == 64 for n in g.params.finishedNodes)


@pytest.mark.gpu
def test_cappos_many_candidates_short_timeout_on_batched_steps():
    lockstep((128, 128, 3, 48, 40, 8), seed=9, chunk=25, chunks=40, nl="NetworkFixedLatency(30)", batched=True)
