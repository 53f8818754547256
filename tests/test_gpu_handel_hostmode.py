"""Handel (P/Handel.java) on the engine in host-callback mode (examples/hostmode/handel.py) vs the CPU oracle
(oracle/handel.hpp) — the honest parameters and the two attack scenarios, byzantineSuicide (:538-559, 577-584, 688-694)
and hiddenByzantine (:813-817, 840-917), as an unmodified protocol class would run them on the engine (the resident device
form runs both too: tests/test_gpu_handel.py). Compared after every chunk: the
node counters and scalars, every level's posInLevel / outgoingFinished / queue length / suicideBizAfter, the five bitsets
of every level and the blacklist, network.time, msgs.size() and the rd state."""
import numpy as np
import pytest

import oracle_lib as o
import parity
from examples.hostmode import handel as hh

SCALARS = {"doneAt": lambda n: n.doneAt, "msgReceived": lambda n: n.msgReceived, "msgSent": lambda n: n.msgSent,
           "bytesSent": lambda n: n.bytesSent, "bytesReceived": lambda n: n.bytesReceived,
           "sigsChecked": lambda n: n.sigsChecked, "sigQueueSize": lambda n: n.sigQueueSize,
           "msgFiltered": lambda n: n.msgFiltered, "currWindowSize": lambda n: n.currWindowSize,
           "addedCycle": lambda n: n.addedCycle, "down": lambda n: int(n.isDown()), "x": lambda n: n.x, "y": lambda n: n.y}
LEVELS = {"posInLevel": lambda l: l.posInLevel, "outgoingFinished": lambda l: int(l.outgoingFinished),
          "queueLen": lambda l: len(l.toVerifyAgg), "suicideBizAfter": lambda l: l.suicideBizAfter}
BITS = {"totalIncoming": lambda l: l.totalIncoming, "lastAggVerified": lambda l: l.lastAggVerified,
        "verifiedIndSignatures": lambda l: l.verifiedIndSignatures, "toVerifyInd": lambda l: l.toVerifyInd,
        "finishedPeers": lambda l: l.finishedPeers}


def rows(values, n):  # python ints -> [N][W] uint64
    w = (n + 63) // 64
    out = np.zeros((len(values), w), np.uint64)
    for i, v in enumerate(values):
        for k in range(w):
            out[i, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def diff(g, c, n):
    bad = []
    nodes = g.network.allNodes
    for f, fn in SCALARS.items():
        a, b = np.array([fn(x) for x in nodes], np.int64), c.read(f)
        if (a != b).any():
            i = int(np.nonzero(a != b)[0][0])
            bad.append("%s: node %d engine %d oracle %d" % (f, i, a[i], b[i]))
    for f, fn in LEVELS.items():
        a, b = np.array([[fn(l) for l in x.levels] for x in nodes], np.int32), c.read_level(f)
        if (a != b).any():
            bad.append("%s differs at (node, level) %s" % (f, tuple(np.argwhere(a != b)[0])))
    for f, fn in BITS.items():
        vals = []
        for x in nodes:
            v = 0
            for l in x.levels:
                v |= fn(l)
            vals.append(v)
        if (rows(vals, n) != c.read_bits(f)).any():
            bad.append("%s differs" % f)
    if (rows([x.blacklist for x in nodes], n) != c.read_bits("blacklist")).any():
        bad.append("blacklist differs")
    i = c.info()
    mine = (g.network.time, g.network._eng.rng_state(), g.network.msgs.size())
    if mine != (i["time"], i["rng"], i["queue"]):
        bad.append("time / rd / queue: engine %s oracle %s" % (mine, (i["time"], i["rng"], i["queue"])))
    return bad


def lockstep(params, seed=0, chunk=10, max_chunks=400, batched=None, **mode):
    """params = HandelParameters ctor order up to desynchronizedStart: (nodeCount, threshold, pairingTime, levelWaitTime,
    extraCycle, disseminationPeriodMs, fastPath, nodesDown, desynchronizedStart)"""
    n = params[0]
    g = hh.Handel(hh.HandelParameters(*params[:8], parity.NB, parity.NL, params[8], mode.get("byzantineSuicide", False),
                                      mode.get("hiddenByzantine", False)), batched=batched)
    g.network.rd.setSeed(seed)
    g.init()
    c = o.Handel(*params[:8], parity.NB, parity.NL, params[8], seed=seed,
                 byzantine_suicide=mode.get("byzantineSuicide", False), hidden_byzantine=mode.get("hiddenByzantine", False))
    k = 0
    while c.cont_if() and k < max_chunks:
        g.network.runMs(chunk)
        c.run_ms(chunk)
        d = diff(g, c, n)
        assert not d, "t=%d: %s" % (g.network.time, d)
        k += 1
    assert g.contIf() == c.cont_if()
    return g, c


P64 = (64, 50, 4, 50, 5, 20, 10, 6, 0)


@pytest.mark.gpu
def test_honest_handel_through_host_callbacks():
    g, c = lockstep(P64, seed=1)
    assert not g.contIf() and all(n.blacklist == 0 for n in g.network.allNodes)


@pytest.mark.gpu
def test_byzantine_suicide():
    """down nodes' invalid signatures are injected in front of the window, checked, and their senders blacklisted"""
    g, c = lockstep(P64, seed=1, byzantineSuicide=True)
    nodes = g.network.allNodes
    down = sum(1 << n.nodeId for n in nodes if n.isDown())
    assert not g.contIf()
    assert any(n.blacklist for n in nodes) and all(n.blacklist & ~down == 0 for n in nodes)  # only byzantine nodes
    _, honest = lockstep(P64, seed=1)  # the same seed without the attack: fewer verifications were wasted
    assert int(c.read("sigsChecked").sum()) > int(honest.read("sigsChecked").sum())


@pytest.mark.gpu
def test_hidden_byzantine():
    """valid but nearly useless signatures of down nodes pushed into the last level's queue: nobody is blacklisted"""
    g, c = lockstep(P64, seed=2, hiddenByzantine=True)
    assert not g.contIf() and all(n.blacklist == 0 for n in g.network.allNodes)
    assert any(n.hiddenByzantine is not None for n in g.network.allNodes)


@pytest.mark.gpu
def test_byzantine_suicide_desynchronized_on_batched_steps():
    lockstep((32, 24, 3, 30, 4, 10, 5, 4, 40), seed=5, batched=True, byzantineSuicide=True)
