"""Casper IMD (P/CasperIMD.java) on the engine in host-callback mode vs the CPU oracle (oracle/casper.hpp, pinned
against PT/CasperIMDTest / PT/CasperByzantineTest): the block tree and attestation sets stay host objects as in the
reference; every `sendAll` (an N-destination envelope per block / attestation), task and latency draw goes through the
engine. Compared after every chunk: per node msgReceived / msgSent / bytesSent / bytesReceived, head (height,
proposalTime, id), the number of heads attested, blocks received — the observables of PT/CasperIMDTest.java:263-274 —
plus network.time and the state of the shared rd."""
import numpy as np
import pytest

import oracle_lib as o
from examples.hostmode import casper

FIELDS = {
    "msgReceived": lambda n: n.msgReceived, "msgSent": lambda n: n.msgSent, "bytesSent": lambda n: n.bytesSent,
    "bytesReceived": lambda n: n.bytesReceived, "headHeight": lambda n: n.head.height,
    "headProposalTime": lambda n: n.head.proposalTime, "headId": lambda n: n.head.id,
    "attestationsByHeadSize": lambda n: len(n.attestationsByHead), "x": lambda n: n.x, "y": lambda n: n.y,
    "blocksReceived": lambda n: len(n.blocksReceivedByBlockId),
    "attestationsHeld": lambda n: sum(len(s) for s in n.attestationsByHead.values()),
}


BYZ = {"WF": casper.ByzBlockProducerWF, "plain": casper.ByzBlockProducerPlain, "SF": casper.ByzBlockProducerSF,
       "NS": casper.ByzBlockProducerNS}


def pair(params, seed, nl=None, byz_delay=0, byz="WF"):
    g = casper.CasperIMD(casper.CasperParemeters(*params, None, nl))
    g.network.rd.setSeed(seed)                       # RunMultipleTimes: copy, rd.setSeed(i), init()
    g.init(BYZ[byz](g, byz_delay))
    return g, o.CasperIMD(params, None, nl, seed=seed, byz_delay=byz_delay, byz=byz)


def diff(g, c):
    out = []
    for f, fn in FIELDS.items():
        a, b = np.array([fn(n) for n in g.network.allNodes], np.int64), c.read(f)
        bad = np.nonzero(a != b)[0]
        if len(bad):
            out.append("%s: %d nodes differ, first node %d: engine %d oracle %d" % (f, len(bad), bad[0], a[bad[0]], b[bad[0]]))
    info = c.info()
    if (g.network.time, g.network._eng.rng_state()) != (info["time"], info["rng"]):
        out.append("time / rd state")
    return out


def lockstep(params, seed, chunk, chunks, **kw):
    g, c = pair(params, seed, **kw)
    assert not diff(g, c), "after init()"
    for k in range(chunks):
        g.network.runMs(chunk)
        c.run_ms(chunk)
        d = diff(g, c)
        assert not d, "t=%d: %s" % (g.network.time, d)
    return g, c


@pytest.mark.gpu
def test_casper_small_matches_oracle():
    g, c = lockstep((2, False, 2, 10, 1000, 1), seed=5, chunk=2000, chunks=15)
    assert g.observer.head.height >= 2 and c.info()["delivered"] > 500


@pytest.mark.gpu
def test_casper_reference_test_parameters():  # PT/CasperIMDTest.java:10-11: 5 producers, 5 x 80 attesters, 40 s
    g, c = lockstep((5, False, 5, 80, 1000, 1), seed=0, chunk=4000, chunks=10)
    assert len(g.network.allNodes) == 406 and g.observer.head.height == 4
    assert c.info()["delivered"] > 100000


@pytest.mark.gpu
def test_casper_byzantine_delay_and_random_ties():  # ByzBlockProducerWF(-2000) (PT/CasperByzantineTest.java:41), rd in best()
    lockstep((3, True, 3, 8, 1000, 1), seed=9, chunk=1000, chunks=50, byz_delay=-2000)


@pytest.mark.gpu
@pytest.mark.parametrize("byz,delay", [("plain", 0), ("plain", 1500), ("SF", 0), ("SF", -1000), ("NS", 0), ("NS", 2500)])
def test_casper_other_byzantine_producers(byz, delay, chunks=40):
    """init(badNode) with the reference's other byzantine block producers (P/CasperIMD.java:511-633: the plain delayed one,
    "skip father", "no skip") instead of the ByzBlockProducerWF init() installs: the Java classes mirrored on the engine's
    host-callback mode, in lock-step with the oracle's restatement of the same lines — incl. the producer's own counters."""
    g, c = lockstep((3, False, 3, 8, 1000, 1), seed=6, chunk=2000, chunks=chunks, byz=byz, byz_delay=delay)
    b, oc = g.bps[0], c.byz_counters()
    assert (b.onDirectFather, b.onOlderAncestor, b.incNotTheBestFather, getattr(b, "skipped", 0), b.toSend) == (
        oc["onDirectFather"], oc["onOlderAncestor"], oc["incNotTheBestFather"], oc["skipped"], oc["toSend"])
    assert b.toSend >= 1 + 3 * (chunks // 14)  # it produced its blocks: one per three slots
    assert g.observer.head.height >= chunks // 5


@pytest.mark.gpu
def test_byzantine_wf_timeline():  # PT/CasperByzantineTest.java:12-35 through the engine
    g = casper.CasperIMD(casper.CasperParemeters(1, False, 2, 2, 1000, 1, None, "NetworkNoLatency"))
    byz = casper.ByzBlockProducerWF(g, 0)
    g.init(byz)
    net, obs = g.network, g.observer
    net.run(9)
    assert obs.head is g.genesis
    net.run(1)
    assert obs.head.height == 1 and obs.head.producer is byz
    net.run(8)
    assert obs.head.height == 2 and obs.head.producer is not byz
    net.run(8)
    assert obs.head.height == 3 and obs.head.producer is byz
