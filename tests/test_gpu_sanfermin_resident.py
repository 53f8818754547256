"""San Fermin (P/SanFerminSignature.java + P/SanFerminHelper.java) RESIDENT on the device
(wittgenstein_amd/csrc/proto_sanfermin.hip.h; pickNextNodes' Collections.shuffle deferred to `resolve`) vs the CPU oracle
(oracle/sanfermin.hpp, pinned against PT/SanFerminTest): after every chunk per node aggValue, currentPrefixLength,
doneAt, thresholdAt, sent / received requests, done, isSwapping, the four Node counters; network.time, msgs.size(), rd."""
import numpy as np
import pytest

import oracle_lib as o
from wittgenstein_amd import protocols as P

FIELDS = ["msgReceived", "msgSent", "bytesSent", "bytesReceived", "aggValue", "currentPrefixLength", "doneAt",
          "thresholdAt", "sentRequests", "receivedRequests", "x", "y"]


def diff(g, c):
    net, out = g.network(), []
    for f in FIELDS:
        a, b = net.read(f), c.read(f)
        bad = np.nonzero(a != b)[0]
        if len(bad):
            out.append("%s: %d nodes differ, first node %d: device %d oracle %d" % (f, len(bad), bad[0], a[bad[0]], b[bad[0]]))
    fl = net.read("sfFlags")
    if not np.array_equal(fl & 1, c.read("done")) or not np.array_equal((fl >> 2) & 1, c.read("isSwapping")):
        out.append("done / isSwapping")
    i = c.info()
    mine = (net.time, net.rng_state(), net.msgs.size())
    if mine != (i["time"], i["rng"], i["queue"]):
        out.append("time / rd / msgs.size(): device %r oracle %r" % (mine, i))
    return out


def lockstep(params, seed, chunk, chunks, nl=None, config=None):
    """params = SanFerminSignatureParameters ctor order: (nodeCount, threshold, pairingTime, signatureSize,
    replyTimeout, candidateCount)"""
    g = P.SanFerminSignature(P.SanFerminSignatureParameters(*params, False, None, nl), seed=seed, config=config)
    g.init()
    c = o.SanFerminSignature(params, None, nl, seed=seed)
    assert not diff(g, c), "after init()"
    for k in range(chunks):
        g.network().runMs(chunk)
        c.run_ms(chunk)
        d = diff(g, c)
        assert not d, "t=%d: %s" % (g.network().time, d)
    assert g.cont_if() == (int((c.read("done") == 0).sum()) > 0)
    return g, c


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 4, 8, 32])
def test_tiny_networks(n):  # bitsets inside one 64-bit word
    lockstep((n, n, 2, 48, 300, 1), seed=n, chunk=20, chunks=40)


@pytest.mark.gpu
def test_sanfermin_64_every_ms():
    g, c = lockstep((64, 64, 2, 48, 300, 1), seed=3, chunk=1, chunks=900)
    assert c.info()["finished"] >= 56


@pytest.mark.gpu
@pytest.mark.parametrize("cand", [1, 2, 3, 7])
def test_candidate_counts_shuffle_draws(cand):  # shuffles of 2..8 candidates: n - 1 deferred rd draws before the seed
    lockstep((256, 256, 2, 48, 300, cand), seed=cand, chunk=25, chunks=60)


@pytest.mark.gpu
def test_sanfermin_default_size():  # the no-arg parameters (:69-81): 1024 nodes
    g, c = lockstep((1024, 1024, 2, 48, 300, 1), seed=0, chunk=100, chunks=40)
    assert c.info()["finished"] > 900 and c.info()["delivered"] > 15000


@pytest.mark.gpu
def test_fixed_latency_short_timeout_and_threshold():
    g, c = lockstep((128, 100, 3, 48, 40, 2), seed=8, chunk=10, chunks=120, nl="NetworkFixedLatency(25)")
    assert (c.read("thresholdAt") > 0).sum() > 100


@pytest.mark.gpu
def test_sanfermin_batch_run_multiple_times():
    """RunMultipleTimes over San Fermin copies (wg_batch_run_multiple_times: the loop on the device): nodes that run out of
    candidates never finish, so the loop ends at maxTime (C/RunMultipleTimes.java:50-64)"""
    import wittgenstein_amd as w
    params, seeds = (64, 64, 2, 48, 300, 1), [1, 2, 3]
    gs = []
    for sd in seeds:
        g = P.SanFerminSignature(P.SanFerminSignatureParameters(*params), seed=sd)
        g.init()
        gs.append(g)
    delivered, ms = w.Batch([g.network() for g in gs]).run_multiple_times(chunk=10, maxTime=3000)
    for g, sd, d in zip(gs, seeds, delivered):
        c = o.SanFerminSignature(params, seed=sd)
        while True:
            did = c.run_ms(10)
            if not (c.info()["time"] < 3000 and (not did or int((c.read("done") == 0).sum()) > 0)):
                break
        assert not diff(g, c), diff(g, c)
        assert d == c.info()["delivered"]
