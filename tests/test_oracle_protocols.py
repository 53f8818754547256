"""Oracle-level protocol properties the reference's tests pin (PT/PingPongTest, PT/HandelTest) plus
invariants used as size-independent parity properties on the GPU."""
import numpy as np
import pytest

import oracle_lib as o


def test_pingpong_all_pongs(oracle):
    p = o.PingPong(1000)  # null names -> RANDOM builder, ByDistanceWJitter (registry defaults)
    p.run_ms(10000)
    pong = p.read("pong")
    assert pong[0] == 1000 and (pong[1:] == 0).all()  # PT/PingPongTest.java:8-19
    assert p.read("msgReceived").sum() == 2000 and p.info()["queue"] == 0


def test_handel_invariants(oracle):
    n, down = 256, 25
    h = o.Handel(n, int(n * 0.9 * 0.99), 4, 50, 10, 20, 10, down)
    while h.cont_if() and h.info(False)["time"] < 20000:
        h.run_ms(10)
    assert not h.cont_if()
    live = h.read("down") == 0
    assert live.sum() == n - down
    assert (h.read("doneAt")[live] > 0).all()
    ti, waited = h.read_bits("totalIncoming"), h.read_bits("waitedSigs")
    own = np.zeros_like(ti)
    for i in range(n):
        own[i, i // 64] = np.uint64(1) << np.uint64(i % 64)
    assert ((ti & ~(waited | own)) == 0).all()  # totalIncoming subset of waitedSigs (+ own sig at level 0)
    # (the final cardinality may be below the threshold again: lastAggVerified is replaced, not merged,
    #  when a new aggregate intersects it — P/Handel.java:713-716 — so only doneAt is monotone)
    st = h.stats()
    assert st["deliveredByLevel"].sum() == h.info(False)["delivered"]


# ---- GSFSignature: PT/GSFSignatureTest.java restated against the oracle ----------------------------------
GSF_NB = "RANDOM_SPEED=GAUSSIAN_TOR=0.00"  # RegistryNodeBuilders.name(RANDOM, true, 0)  (PT/GSFSignatureTest.java:11)
GSF_NL = "NetworkLatencyByDistanceWJitter"


def _run_handel(**mode):
    n, down = 256, 25
    h = o.Handel(n, int(n * 0.9 * 0.99), 4, 50, 10, 20, 10, down, seed=3, **mode)
    while h.cont_if() and h.info(False)["time"] < 20000:
        h.run_ms(10)
    return h


def test_handel_attack_scenarios(oracle):
    """P/Handel.java's byzantineSuicide (:538-559, 577-584, 688-694) and hiddenByzantine (:813-817, 840-917). The
    reference has no test of either (PT/HandelTest is honest); what the code itself guarantees is checked: only down
    nodes are ever blacklisted, suicide signatures cost verifications, the hidden attack blacklists nobody, and both runs
    still reach the threshold on every live node."""
    honest, suicide, hidden = _run_handel(), _run_handel(byzantine_suicide=True), _run_handel(hidden_byzantine=True)
    for h in (honest, suicide, hidden):
        assert not h.cont_if() and (h.read("doneAt")[h.read("down") == 0] > 0).all()
    down = suicide.read("down") != 0
    bl = suicide.read_bits("blacklist")
    ids = [i for i in range(256) if (bl[:, i // 64] >> np.uint64(i % 64) & np.uint64(1)).any()]
    assert ids and all(down[i] for i in ids)
    assert not bl[down].any()  # a down node runs nothing
    assert suicide.read("sigsChecked").sum() > honest.read("sigsChecked").sum()
    assert (suicide.read_level("suicideBizAfter")[~down] >= -1).all() and (honest.read_level("suicideBizAfter") == -1).all()
    assert not hidden.read_bits("blacklist").any() and not honest.read_bits("blacklist").any()
    # the same seed draws the same topology and bad nodes in all three (init() does not depend on the scenario)
    assert (honest.read("x") == suicide.read("x")).all() and (honest.read("down") == hidden.read("down")).all()


def test_sanfermin_cappos_aggregates_everything(oracle):
    """P/SanFerminCappos.java has no test in the reference; what the protocol promises: a node that finishes holds the
    whole set (its own signature + the best cached value of every level), thresholdAt precedes doneAt, and with a
    generous timeout and every candidate tried nearly everybody finishes."""
    c = o.SanFerminCappos((256, 128, 2, 48, 150, 8), seed=1)
    for _ in range(80):
        c.run_ms(50)
    done = c.read("done") != 0
    assert done.sum() >= 240 and c.info()["finished"] == done.sum()
    assert (c.read("totalNumberOfSigs")[done] == 256).all() and (c.read("currentPrefixLength")[done] == 0).all()
    th, da = c.read("thresholdAt"), c.read("doneAt")
    assert (th[done] > 0).all() and (th[done] <= da[done]).all()
    assert (c.read("msgSent") > 0).all() and c.read("msgReceived").sum() == c.info()["delivered"]


def test_gsf_init_and_max_sig_in_level(oracle):  # testInit :22-47, testMaxSigInLevel :49-57
    g = o.GSFSignature(32, 1, 3, 20, 10, 10, 0, GSF_NB, GSF_NL)
    assert g.levels == 6
    assert [len(g.read_peers(0, l)) for l in range(6)] == [0, 1, 2, 4, 8, 16]
    assert g.read_peers(0, 1)[0] == 1
    lv = g.read_bits("levelVerified")
    assert lv[0, 0] == 1  # level 0 holds the node's own signature, the other levels nothing
    waited = g.read_bits("waitedSigs")
    assert bin(int(waited[0, 0])).count("1") == 1 + 1 + 2 + 4 + 8 + 16  # expectedSigs 1,1,2,4,8,16


def test_gsf_send(oracle):  # testSend :59-64: after runMs(1) every node has sent its signature to its peer
    g = o.GSFSignature(32, 1, 3, 20, 10, 10, 0, GSF_NB, GSF_NL)
    g.run_ms(1)
    assert g.info()["queue"] == 64


def test_gsf_dead_nodes(oracle):  # testDeadNodes :73-80
    g = o.GSFSignature(32, int(0.8 * 32), 3, 20, 10, 10, int(0.1 * 32), GSF_NB, GSF_NL)
    assert g.read("down").sum() == 3


def test_gsf_simple_run_and_threshold(oracle):  # testSimpleRun :95-105, testSimpleThreshold :107-124
    g = o.GSFSignature(32, 1, 3, 20, 10, 10, 0, GSF_NB, GSF_NL)
    g.run_ms(10000)
    assert (g.read("verifiedCardinality") == 32).all()
    g = o.GSFSignature(64, int(.50 * 64), 3, 20, 10, 10, int(.2 * 64), GSF_NB, GSF_NL)
    g.run_ms(10000)
    card, down = g.read("verifiedCardinality"), g.read("down") != 0
    assert (card[down] == 1).all()
    assert ((card[~down] >= 32) & (card[~down] <= 64)).all()


def _gsf_state(g):
    return ([g.read(f) for f in ("doneAt", "msgReceived", "msgSent", "bytesSent", "sigChecked", "sigQueueSize",
                                 "toVerifySize")] +
            [g.read_bits(b) for b in ("verifiedSignatures", "levelVerified", "individualSignatures", "indivVerifiedSig")] +
            [g.read_level("posInLevel"), g.read_level("remainingCalls"), g.info()["rng"], g.info()["queue"]])


def test_gsf_copy_is_deterministic(oracle):  # testCopy :126-147 (lock-step, every ms)
    a = o.GSFSignature(128, int(.75 * 128), 6, 10, 5, 10, int(.2 * 128), GSF_NB, GSF_NL)
    b = o.GSFSignature(128, int(.75 * 128), 6, 10, 5, 10, int(.2 * 128), GSF_NB, GSF_NL)
    for _ in range(400):
        a.run_ms(1)
        b.run_ms(1)
        assert a.info()["queue"] == b.info()["queue"]
        assert (a.read("doneAt") == b.read("doneAt")).all()
        assert (a.read_bits("verifiedSignatures") == b.read_bits("verifiedSignatures")).all()
        assert (a.read("toVerifySize") == b.read("toVerifySize")).all()


def test_gsf_aliasing_is_unobservable(oracle):
    """updateVerifiedSignatures ORs into the BitSet of the message object (P/GSFSignature.java:390,419), which a
    multi-destination send shares between its receivers. The device protocol keeps a private copy per receiver;
    that is exact because the shared objects are always whole level blocks (or larger) — shown here by running
    the oracle with and without sharing, and by the payload-shape counter — and because GSFNode.verifiedSignatures
    is always the union of the levels' verifiedSignatures (the device keeps one row for both)."""
    for (n, thr, pair, to, per, acc, down, nb, seed) in [
            (128, 96, 6, 10, 5, 10, 25, GSF_NB, 0), (512, 500, 3, 50, 10, 10, 0, "RANDOM_SPEED=CONSTANT_TOR=0.00", 3),
            (256, 200, 2, 30, 7, 3, 20, "RANDOM_SPEED=CONSTANT_TOR=0.33", 11)]:
        a = o.GSFSignature(n, thr, pair, to, per, acc, down, nb, GSF_NL, seed)
        b = o.GSFSignature(n, thr, pair, to, per, acc, down, nb, GSF_NL, seed)
        b.set_copy_on_delivery()
        for _ in range(100):
            a.run_ms(7)
            b.run_ms(7)
            for x, y in zip(_gsf_state(a), _gsf_state(b)):
                assert np.array_equal(x, y)
        assert a.stats()["shapeViolations"] == 0
        live = a.read("down") == 0
        assert np.array_equal(a.read_bits("levelVerified")[live], a.read_bits("verifiedSignatures")[live])


# ---- OptimisticP2PSignature: PT/OptimisticP2PSignatureTest.java restated against the oracle --------------
def test_optimistic_p2p_simple(oracle):  # testSimple :14-32
    n = 100
    p = o.OptimisticP2PSignature((n, n // 2 + 1, 13, 3), GSF_NB, GSF_NL)
    p.run_ms(10 * 1000)
    assert len(p.read("done")) == n
    assert (p.read("done") == 1).all() and (p.read("doneAt") > 0).all() and (p.read("sigs") > n // 2).all()
    assert (p.read("peerCount") >= 3).all()  # P2PNetwork.setPeers :48-55 (minimum == false)


def test_optimistic_p2p_copy_is_deterministic(oracle):  # testCopy :34-50
    a = o.OptimisticP2PSignature((200, 160, 10, 2), GSF_NB, GSF_NL)
    b = o.OptimisticP2PSignature((200, 160, 10, 2), GSF_NB, GSF_NL)
    a.run_ms(200)
    b.run_ms(200)
    assert (a.read("done") == b.read("done")).all() and (a.read("doneAt") == b.read("doneAt")).all()
    assert a.info() == b.info() and a.info()["delivered"] > 10000


# ---- Slush / Snowflake: PT/SlushTest.java, PT/SnowflakeTest.java restated against the oracle (oracle/slush.hpp) ----------
@pytest.mark.parametrize("snow,params", [(False, (100, 7, 7, 4.0 / 7.0)), (True, (100, 5, 7, 4.0 / 7.0, 3))])
def test_slush_snowflake_simple(oracle, snow, params):  # testSimple (SlushTest :14-25, SnowflakeTest :14-25)
    p = o.Slush(params, GSF_NB, GSF_NL, snowflake=snow)
    p.run_ms(10 * 1000)
    col = p.read("myColor")
    assert len(col) == 100 and col[0] in (1, 2) and (col == col[0]).all()  # every node ends on node 0's colour
    assert (p.read("answersInProgress") == 0).all() and p.info()["queue"] == 0  # every query was answered K times
    if not snow:
        assert (p.read("round") == 7).all()  # M rounds each (P/Slush.java:165-168)
    else:
        assert (p.read("cnt") == 3 + 1).all()  # stops past beta confirmations (P/Snowflake.java:186-191)


@pytest.mark.parametrize("snow,params", [(False, (60, 5, 7, 4.0 / 7.0)), (True, (60, 5, 7, 4.0 / 7.0, 3))])
def test_slush_snowflake_copy_is_deterministic(oracle, snow, params):  # testCopy (SlushTest :27-45, SnowflakeTest :27-47)
    a = o.Slush(params, GSF_NB, GSF_NL, snowflake=snow)
    b = o.Slush(params, GSF_NB, GSF_NL, snowflake=snow)
    a.run_ms(200)
    b.run_ms(200)
    for f in ("myColor", "myQueryNonce", "round", "cnt"):
        assert (a.read(f) == b.read(f)).all()
    assert a.info() == b.info() and a.info()["delivered"] > 100


# ---- Paxos: PT/PaxosTest.java restated against the oracle (oracle/paxos.hpp) ------------------------------------------------
def test_paxos_simple(oracle):  # testSimple :9-19
    p = o.Paxos((3, 1, 1000))
    p.run_ms(10 * 1000)
    assert len(p.read("seqIP")) == 4                      # 3 acceptors + 1 proposer; majority = 3 / 2 + 1 = 2
    prop = p.read("seqIP") != -2
    assert prop.sum() == 1 and (p.read("seqIP")[prop] > 0).all()
    # one proposer alone: its first proposal is agreed to and accepted by the three acceptors
    assert (p.read("valueAccepted")[prop] == p.read("valueProposed")[prop]).all() and (p.read("doneAt")[prop] > 0).all()
    assert (p.read("acceptedVal")[~prop] == p.read("valueProposed")[prop][0]).all() and (p.read("timeoutCount")[prop] == 0).all()


def test_paxos_copy_is_deterministic(oracle):  # testCopy :21-32
    a, b = o.Paxos((3, 2, 1000)), o.Paxos((3, 2, 1000))
    a.run_ms(2000)
    b.run_ms(2000)
    assert (a.read("msgReceived") == b.read("msgReceived")).all() and a.info() == b.info()


@pytest.mark.parametrize("seed", range(6))
def test_paxos_agreement(oracle, seed):
    """the final check of Paxos.play() (P/Paxos.java:473-486) on its own parameters (3 acceptors, 3 proposers, timeout 1000) and
    on a larger contended shape (rejections at both stages): every proposer that is done accepted the same value, and it is one that was proposed"""
    for params in ((3, 3, 1000), (7, 5, 600)):
        p = o.Paxos(params, None, "NetworkLatencyByDistanceWJitter", seed=seed)
        for _ in range(100):  # RunMultipleTimes' loop: runMs(10) while some proposer is not done (:488-497), at most 5 s here
            p.run_ms(50)
            prop = p.read("seqIP") != -2
            if (p.read("doneAt")[prop] > 0).all():
                break
        acc = p.read("valueAccepted")[prop]
        done = acc[acc >= 0]
        assert len(done) > 0 and (done == done[0]).all() and done[0] in set(p.read("valueProposed")[prop].tolist())


# ---- Dfinity: PT/DfinityTest.java restated against the oracle (oracle/dfinity.hpp) -------------------------------------------
def test_dfinity_run(oracle):  # testRun :20-24 with the fixture of :10-18 (NetworkNoLatency)
    p = o.Dfinity((10, 10, 10, 1, 1, 0), GSF_NB, "NetworkNoLatency")
    assert p.n == 1 + 10 + 10 + 10                      # the observer, the attesters, the producers, the beacon committee
    p.run_ms(11 * 1000)
    assert p.read("headHeight")[0] == 3                 # Assert.assertEquals(3, dfinity.network.observer.head.height)
    assert (p.read("headHeight") == 3).all() and len(set(p.read("headId").tolist())) == 1  # one chain, everywhere


def test_dfinity_copy_is_deterministic_and_the_chain_grows(oracle):  # the parameters of the (disabled) testCopy :28-47
    a = o.Dfinity((10, 50, 25, 100, 1, 5), GSF_NB, "NetworkLatencyByDistanceWJitter", seed=3)
    b = o.Dfinity((10, 50, 25, 100, 1, 5), GSF_NB, "NetworkLatencyByDistanceWJitter", seed=3)
    for _ in range(20):
        a.run_ms(1000)
        b.run_ms(1000)
        assert a.info() == b.info()
        for f in ("headTime", "majorityBlocks", "majorityHeightSum", "lastRandomBeacon"):
            assert (a.read(f) == b.read(f)).all()
    assert a.read("headHeight")[0] >= 2 and (a.read("headHeight") >= a.read("headHeight")[0] - 1).all()


# ---- P2PHandel: PT/P2PHandelTest.java restated against the oracle (oracle/p2phandel.hpp) -------------------------------------
def _p2phandel_default(nodes=32, cc=4):  # P2PHandelScenarios.defaultParams(32, 0.0, 4, null, null) (:261-277) on the RANDOM builder
    return (nodes, 0, int(nodes * 0.99), cc, 4, 20, True, "dif", False)


def test_p2phandel_setup_checksigs_sigupdate(oracle):  # testSetup :17-25, testCheckSigs :74-83, testSigUpdate :85-91
    p = o.P2PHandel(_p2phandel_default(), GSF_NB, GSF_NL)
    assert (p.read("sigs") == 1).all() and (p.read("sigsDigest") == np.arange(32) + 1).all()   # own signature only
    assert (p.read("peerCount") >= 3).all()
    empty, queued, card = p.probe()
    assert empty == 1 and queued == 1 and card == 2


def test_p2phandel_compressed_size(oracle):  # testCompressedSize :104-143: the reference's fourteen values
    p = o.P2PHandel(_p2phandel_default(), GSF_NB, GSF_NL)
    for want, binary in [(1, "1111"), (1, "1111 1111"), (1, "1111 1111 1111 1111"),
                         (3, "0000 0000 0000 0000  0000 0000 0000 0000 1111 1111 1111 1111  1111 1111 1111 0000"),
                         (1, "0000 0000 0000 0000  0000 0000 0000 0000 1111 1111 1111 1111  1111 1111 1111 1111 0000"),
                         (2, "0000 0000 0000 0000  1111 1111 1111 1111 1111 1111 1111 1111  1111 1111 1111 1111 0000"),
                         (3, "1111 1111 1111 1111  1111 1111 1111 0000"), (1, "1111 1111 0000"), (3, "0001 1111 1111 0000"),
                         (3, "0001 1111 1111 1111"), (2, "0000 1111 1111 1111  0000"), (4, "1101 0111"), (3, "1111 1110")]:
        assert p.compressed_size(binary) == want, binary


@pytest.mark.parametrize("params", [(64, 0, 60, 3, 2, 5, True, "all", False), (20, 0, 20, 3, 2, 50, True, "cmp_diff", True)])
def test_p2phandel_runs_to_done(oracle, params):  # testSimpleRunWithoutState :44-55, testSimpleRunWithState :57-68
    p = o.P2PHandel(params, GSF_NB, GSF_NL)
    while (p.read("doneAt") == 0).any() and p.info()["time"] < 20000:   # RunMultipleTimes.contUntilDone()
        p.run_ms(1000)
    assert (p.read("doneAt") > 0).all() and (p.read("sigs") >= params[2]).all()


@pytest.mark.parametrize("params", [(100, 0, 25, 10, 2, 5, False, "dif", False), (40, 0, 25, 8, 2, 5, False, "dif", True)])
def test_p2phandel_repeatability(oracle, params):
    """testRepeatability :27-42 (checkSigs1: the single-best strategy) on its own shape without the state messages, and with them
    on 40 nodes. The reference's exact parameters (100 nodes WITH state messages) grow a bucket of toVerify's table into a
    red-black tree (java.util.HashMap.treeifyBin: BitSet hashes of small sets cluster), whose iteration order follows the
    tree's shape and, for equal hashes, System.identityHashCode: not restatable — the oracle refuses it loudly (next test)."""
    a, b = o.P2PHandel(params, GSF_NB, GSF_NL), o.P2PHandel(params, GSF_NB, GSF_NL)
    a.run_ms(10 * 1000)
    b.run_ms(10 * 1000)
    assert (a.read("doneAt") == b.read("doneAt")).all() and (a.read("doneAt") > 0).all() and a.info() == b.info()


def test_p2phandel_a_treeified_bucket_is_refused(oracle):
    p = o.P2PHandel((100, 0, 25, 10, 2, 5, False, "dif", True), GSF_NB, GSF_NL)
    with pytest.raises(o.OracleError) as ei:
        p.run_ms(10 * 1000)
    assert "became a tree" in str(ei.value)
