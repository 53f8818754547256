"""Shared helpers: run the device engine and the CPU oracle in lock-step and diff their state.
(The analogue of the reference's testCopy pattern, PT/HandelTest.java:14-34.)"""
import numpy as np

import oracle_lib as o
import wittgenstein_amd as w

NB = "RANDOM_SPEED=CONSTANT_TOR=0.00"
NL = "NetworkLatencyByDistanceWJitter"

SCALARS = ["doneAt", "msgReceived", "msgSent", "bytesSent", "bytesReceived", "sigsChecked", "sigQueueSize",
           "msgFiltered", "currWindowSize", "addedCycle"]
LEVELS = ["posInLevel", "outgoingFinished", "queueLen", "suicideBizAfter"]
BITS = ["totalIncoming", "lastAggVerified", "verifiedIndSignatures", "toVerifyInd", "finishedPeers", "blacklist"]


def handel_pair(params, nb=NB, nl=NL, seed=0, config=None, byzantine_suicide=False, hidden_byzantine=False, bad_nodes=None):
    """params = (nodeCount, threshold, pairing, levelWait, extraCycle, period, fastPath, nodesDown, desync)"""
    n, thr, pair, lw, ec, per, fp, down, desync = params
    g = w.Handel(w.HandelParameters(n, thr, pair, lw, ec, per, fp, down, nb, nl, desync, byzantineSuicide=byzantine_suicide,
                                    hiddenByzantine=hidden_byzantine, badNodes=bad_nodes), seed=seed, config=config)
    g.init()
    c = o.Handel(n, thr, pair, lw, ec, per, fp, down, nb, nl, desync, seed=seed, byzantine_suicide=byzantine_suicide,
                 hidden_byzantine=hidden_byzantine, bad_nodes=bad_nodes)
    return g, c


def diff_handel(g, c, bits=True, levels=True):
    """returns a list of human-readable mismatches (empty = identical observable state)"""
    net = g.network()
    out = []
    gi, ci = net.time, c.info(False)
    if gi != ci["time"]:
        out.append("time %d != %d" % (gi, ci["time"]))
    if net.rng_state() != ci["rng"]:
        out.append("rng state %x != %x" % (net.rng_state(), ci["rng"]))
    for f in SCALARS:
        a, b = net.read(f), c.read(f)
        bad = np.nonzero(a != b)[0]
        if len(bad):
            out.append("%s: %d nodes differ, first node %d: gpu %d oracle %d" % (f, len(bad), bad[0], a[bad[0]], b[bad[0]]))
    if levels:
        for f in LEVELS:
            a, b = net.read_level(f), c.read_level(f)
            bad = np.argwhere(a != b)
            if len(bad):
                i, l = bad[0]
                out.append("%s: %d (node,level) differ, first (%d,%d): gpu %d oracle %d" % (f, len(bad), i, l, a[i, l], b[i, l]))
    if bits:
        for f in BITS:
            a, b = net.read_bits(f), c.read_bits(f)
            if a.shape != b.shape:
                b = b[:, :a.shape[1]]
            bad = np.argwhere(a != b)
            if len(bad):
                i, wd = bad[0]
                out.append("%s: %d words differ, first node %d word %d: gpu %x oracle %x" % (f, len(bad), i, wd, a[i, wd], b[i, wd]))
    return out


def diff_pingpong(g, c):
    net = g.network()
    out = []
    ci = c.info()
    if net.time != ci["time"]:
        out.append("time")
    if net.rng_state() != ci["rng"]:
        out.append("rng state %x != %x" % (net.rng_state(), ci["rng"]))
    if net.msgs.size() != ci["queue"]:
        out.append("queue %d != %d" % (net.msgs.size(), ci["queue"]))
    for f in ["pong", "msgReceived", "msgSent", "bytesSent", "bytesReceived"]:
        a, b = net.read(f), c.read(f)
        bad = np.nonzero(a != b)[0]
        if len(bad):
            out.append("%s: %d nodes differ, first node %d: gpu %d oracle %d" % (f, len(bad), bad[0], a[bad[0]], b[bad[0]]))
    return out
