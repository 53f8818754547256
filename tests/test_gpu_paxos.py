"""protocols.Paxos (P/Paxos.java) on the engine in host-callback mode vs the CPU oracle (oracle/paxos.hpp, pinned against
PT/PaxosTest and the final check of Paxos.play() in tests/test_oracle_protocols.py). Compared before the run and after every
chunk: per node the four Node counters, doneAt and the position; per acceptor maxAgreed, acceptedSeq / acceptedVal, agreedTo; per
proposer the proposed and the accepted value, the sequence numbers and every counter; network.time, msgs.size(), the rd state
(every send to the acceptors shuffles them with it). init() sends between node constructions: HostNetwork.deferred_init."""
import numpy as np
import pytest

import oracle_lib as o
from examples.hostmode import paxos as px

NONE = -1


def _opt(v):
    return NONE if v is None else v


ALL = {"msgReceived": lambda n: n.msgReceived, "msgSent": lambda n: n.msgSent, "bytesSent": lambda n: n.bytesSent,
       "bytesReceived": lambda n: n.bytesReceived, "doneAt": lambda n: n.doneAt, "x": lambda n: n.x, "y": lambda n: n.y}
ACC = {"maxAgreed": lambda n: n.maxAgreed, "acceptedSeq": lambda n: _opt(n.acceptedSeq), "acceptedVal": lambda n: _opt(n.acceptedVal),
       "agreedTo": lambda n: NONE if n.agreedTo is None else n.agreedTo.nodeId}
PROP = {"valueProposed": lambda n: n.valueProposed, "valueAccepted": lambda n: _opt(n.valueAccepted), "seqIP": lambda n: n.seqIP,
        "seqAccepted": lambda n: n.seqAccepted, "agreeCount": lambda n: n.agreeCount, "reject1Count": lambda n: n.reject1Count,
        "reject2Count": lambda n: n.reject2Count, "timeoutCount": lambda n: n.timeoutCount, "proposalIP": lambda n: int(n.proposalIP),
        "agreeCountIP": lambda n: n.agreeCountIP, "acceptCountIP": lambda n: n.acceptCountIP}


def lockstep(params, nl, seed, chunk, chunks):
    """params = PaxosParameters ctor order: (acceptorCount, proposerCount, timeout)"""
    g = px.Paxos(px.PaxosParameters(*params, None, nl))
    g.network.rd.setSeed(seed)
    g.init()
    c = o.Paxos(params, None, nl, seed=seed)
    for k in range(chunks + 1):
        nodes = g.network.allNodes
        for fields, kind in ((ALL, px.PaxosNode), (ACC, px.AcceptorNode), (PROP, px.ProposerNode)):
            for f, fn in fields.items():
                a = np.array([fn(n) if isinstance(n, kind) else -2 for n in nodes], np.int64)
                b = c.read(f)
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
def test_paxos_simple():  # PT/PaxosTest.java:9-19 through the engine
    g, c = lockstep((3, 1, 1000), None, seed=0, chunk=500, chunks=20)
    assert len(g.network.allNodes) == 4 and g.majority == 2
    for n in g.proposers:
        assert n.seqIP > 0 and n.valueAccepted == n.valueProposed and n.doneAt > 0


@pytest.mark.gpu
@pytest.mark.parametrize("params,seed", [((3, 2, 1000), 0), ((3, 3, 1000), 1), ((7, 5, 600), 2)])
def test_paxos_contended_and_copy(params, seed):  # testCopy's shape (:21-32), play()'s, and a larger one with both kinds of rejection
    g1, c = lockstep(params, "NetworkLatencyByDistanceWJitter", seed=seed, chunk=100, chunks=30)
    assert g1.finalCheck() and all(p.doneAt > 0 for p in g1.proposers)        # play()'s final check :473-486
    g2 = g1.copy()
    g2.network.rd.setSeed(seed)
    g2.init()
    g2.network.runMs(3000)
    assert [n.msgReceived for n in g1.network.allNodes] == [n.msgReceived for n in g2.network.allNodes]


@pytest.mark.gpu
def test_paxos_timeouts_batched_steps(monkeypatch):
    """a timeout shorter than a round trip (every proposal times out and is retried: the timeout tasks and the proposals' re-sends
    interleave), through the batched-step calls (wg_step_begin / wg_step_end)"""
    monkeypatch.setenv("WG_HOST_BATCH", "1")
    g, c = lockstep((5, 3, 120), "NetworkLatencyByDistanceWJitter", seed=4, chunk=40, chunks=25)
    assert sum(p.timeoutCount for p in g.proposers) > 5
