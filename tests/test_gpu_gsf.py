"""GSFSignature (P/GSFSignature.java) resident on the MI355X vs the CPU oracle, lock-step: every per-node
counter, the three bitset rows, per-level cursors, queue size, rd state. BASELINE config 2 is
test_gsf_4096_config2. Bodies are also run on the CPU wave emulator by tests/test_emu_kernels.py."""
import numpy as np
import pytest

import oracle_lib as o
import wittgenstein_amd as w

NB = "RANDOM_SPEED=CONSTANT_TOR=0.00"
NBG = "RANDOM_SPEED=GAUSSIAN_TOR=0.00"  # RegistryNodeBuilders.name(RANDOM, true, 0) — PT/GSFSignatureTest.java:11
NL = "NetworkLatencyByDistanceWJitter"

SCALARS = [("doneAt", "doneAt"), ("msgReceived", "msgReceived"), ("msgSent", "msgSent"), ("bytesSent", "bytesSent"),
           ("bytesReceived", "bytesReceived"), ("sigChecked", "sigChecked"), ("gsfSigQueueSize", "sigQueueSize"),
           ("toVerifySize", "toVerifySize"), ("verifiedCardinality", "verifiedCardinality")]
BITS = ["verifiedSignatures", "individualSignatures", "indivVerifiedSig"]


def pair(params, nb=NB, seed=0, config=None):
    """params = GSFSignatureParameters ctor order: (nodeCount, threshold, pairingTime, timeoutPerLevelMs,
    periodDurationMs, acceleratedCallsCount, nodesDown)"""
    n, thr, pt, to, per, acc, down = params
    g = w.GSFSignature(w.GSFSignatureParameters(n, thr, pt, to, per, acc, down, nb, NL), seed=seed, config=config)
    g.init()
    return g, o.GSFSignature(n, thr, pt, to, per, acc, down, nb, NL, seed=seed)


def diff(g, c, queue=True):
    net, out = g.network(), []
    ci = c.info(queue)
    if net.time != ci["time"]:
        out.append("time %d != %d" % (net.time, ci["time"]))
    if net.rng_state() != ci["rng"]:
        out.append("rng state %x != %x" % (net.rng_state(), ci["rng"]))
    if queue and net.msgs.size() != ci["queue"]:
        out.append("msgs.size() %d != %d" % (net.msgs.size(), ci["queue"]))
    for gf, cf in SCALARS:
        a, b = net.read(gf), c.read(cf)
        bad = np.nonzero(a != b)[0]
        if len(bad):
            out.append("%s: %d nodes differ, first node %d: gpu %d oracle %d" % (gf, len(bad), bad[0], a[bad[0]], b[bad[0]]))
    for f in ("posInLevel", "remainingCalls"):
        a, b = net.read_level(f), c.read_level(f)
        bad = np.argwhere(a != b)
        if len(bad):
            i, l = bad[0]
            out.append("%s: %d (node,level) differ, first (%d,%d): gpu %d oracle %d" % (f, len(bad), i, l, a[i, l], b[i, l]))
    live = c.read("down") == 0  # a stopped node never gets levels (P/GSFSignature.java:627-634)
    for f in BITS:
        a, b = net.read_bits(f)[live], c.read_bits(f)[live]
        bad = np.argwhere(a != b)
        if len(bad):
            i, wd = bad[0]
            out.append("%s: %d words differ, first live node #%d word %d: gpu %x oracle %x" % (f, len(bad), i, wd, a[i, wd], b[i, wd]))
    return out


def lockstep(params, nb=NB, seed=0, config=None, step=1, total=400, to_convergence=False):
    g, c = pair(params, nb, seed, config)
    assert not diff(g, c), diff(g, c)
    t = 0
    while t < total or (to_convergence and c.cont_if()):
        assert t < 20000
        did = g.network().runMs(step)
        cdid = c.run_ms(step)
        t += step
        d = diff(g, c)
        assert not d, "t=%d: %s" % (t, d)
        assert did == cdid
        assert g.cont_if() == c.cont_if()
    return g, c


@pytest.mark.gpu
def test_reference_test_parameters_every_ms():  # PT/GSFSignatureTest.java:13-14 (32 nodes) and testSend :59-64
    g, c = pair((32, 1, 3, 20, 10, 10, 0), NBG)
    g.network().runMs(1)
    c.run_ms(1)
    assert g.network().msgs.size() == 64
    g, c = lockstep((32, 1, 3, 20, 10, 10, 0), NBG, total=300)
    assert (g.network().read("verifiedCardinality") == 32).all()  # testSimpleRun :95-105


@pytest.mark.gpu
def test_simple_threshold_with_dead_nodes():  # testSimpleThreshold :107-124
    g, c = lockstep((64, 32, 3, 20, 10, 10, 12), NBG, total=400)
    card, down = g.network().read("verifiedCardinality"), g.network().read("down") != 0
    assert down.sum() == 12 and (card[down] == 1).all() and (card[~down] >= 32).all()


@pytest.mark.gpu
def test_event_order_visit_of_every_active_node(monkeypatch):
    """WG_GSF_REST_LIST=0: k_deliver_inbox looks at every active node's inbox count (the form before k_gsf_lane listed the
    nodes it leaves) — the same lock-step runs as the default's, dead nodes and multi-word levels included"""
    monkeypatch.setenv("WG_GSF_REST_LIST", "0")
    lockstep((64, 32, 3, 20, 10, 10, 12), NBG, total=400)
    lockstep((512, 500, 3, 50, 10, 10, 0), seed=5, step=10, total=150)


@pytest.mark.gpu
def test_copy_parameters_long_lists():  # testCopy's parameters :126-131; toVerify grows past one wavefront (170 entries)
    lockstep((128, 96, 6, 10, 5, 10, 25), NBG, total=600, config={"queue_cap": 256})


@pytest.mark.gpu
def test_256_chunks_of_10_to_convergence():
    g, c = lockstep((256, 250, 3, 50, 10, 10, 0), seed=3, step=10, total=0, to_convergence=True)
    assert not g.cont_if()


@pytest.mark.gpu
def test_tor_latency_few_accelerated_calls():
    lockstep((256, 200, 2, 30, 7, 3, 20), "RANDOM_SPEED=CONSTANT_TOR=0.33", seed=11, step=7, total=700,
             config={"queue_cap": 512, "horizon_ms": 2048})


@pytest.mark.gpu
def test_512_multiword_levels():
    lockstep((512, 500, 3, 50, 10, 10, 0), seed=5, step=10, total=300)


@pytest.mark.gpu
def test_queue_capacity_overflow_is_loud():
    g, c = pair((128, 96, 6, 10, 5, 10, 25), NBG, config={"queue_cap": 64})
    with pytest.raises(w.EngineCapacityError):
        for _ in range(100):
            g.network().runMs(10)


@pytest.mark.gpu
def test_unsupported_shapes_are_loud():
    with pytest.raises(w.UnsupportedError):
        w.GSFSignature(w.GSFSignatureParameters(31, 1, 3, 20, 10, 10, 0, NB, NL)).init()
    with pytest.raises(w.IllegalArgumentException):  # ctor check P/GSFSignature.java:69-74
        w.GSFSignatureParameters(32, 30, 3, 20, 10, 10, 5, NB, NL)


@pytest.mark.gpu
def test_gsf_4096_config2():
    """BASELINE.json configs[1]: GSFSignature 4096 nodes, threshold (int)(0.99 * 4096), pairing 3, timeout 50,
    period 10, accelerated 10, RunMultipleTimes' runMs(10) loop to newConfIf — bit-exact at every chunk."""
    g, c = lockstep((4096, 4055, 3, 50, 10, 10, 0), step=10, total=0, to_convergence=True)
    assert not g.cont_if()
    live = c.read("down") == 0
    assert (g.network().read("verifiedCardinality")[live] >= 4055).all()
    assert (g.network().delivered_by_level()[:c.levels] == c.stats()["deliveredByLevel"].astype(np.int64)).all()


@pytest.mark.gpu
def test_gsf_batch_matches_single_runs():
    """wg_batch of GSF copies with distinct seeds (RunMultipleTimes): each member equals the oracle run alone."""
    seeds = [0, 1, 2]
    gs = []
    for sd in seeds:
        g = w.GSFSignature(w.GSFSignatureParameters(256, 250, 3, 50, 10, 10, 0, NB, NL), seed=sd)
        g.init()
        gs.append(g)
    batch = w.Batch([g.network() for g in gs])
    batch.run_multiple_times(chunk=10, maxTime=20000)
    for sd, g in zip(seeds, gs):
        c = o.GSFSignature(256, 250, 3, 50, 10, 10, 0, NB, NL, seed=sd)
        while True:
            did = c.run_ms(10)
            if not ((c.info()["time"] < 20000) and (not did or c.cont_if())):
                break
        d = diff(g, c)
        assert not d, d
