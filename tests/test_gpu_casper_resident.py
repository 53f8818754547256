"""Casper IMD (P/CasperIMD.java) RESIDENT on the device (wittgenstein_amd/csrc/proto_casper.hip.h: attestation sets as
bitsets over an attestation index, sendAll resolved by k_sendall_*, 8-second periodic tasks through the far buffer) vs
the CPU oracle (oracle/casper.hpp, pinned against PT/CasperIMDTest / PT/CasperByzantineTest). After every chunk, per node:
msgReceived / msgSent / bytesSent / bytesReceived, head (height, proposalTime, id), attestationsByHead.size(), blocks
received, attestations held — the observables of PT/CasperIMDTest.java:263-274 — plus network.time, msgs.size() (far
envelopes included) and the rd state."""
import numpy as np
import pytest

import oracle_lib as o
from wittgenstein_amd import protocols as P
from wittgenstein_amd.core import UnsupportedError

FIELDS = ["msgReceived", "msgSent", "bytesSent", "bytesReceived", "headHeight", "headProposalTime", "headId",
          "attestationsByHeadSize", "blocksReceived", "attestationsHeld", "x", "y"]


def diff(g, c):
    net, out = g.network(), []
    for f in FIELDS:
        a, b = net.read(f), c.read(f)
        bad = np.nonzero(a != b)[0]
        if len(bad):
            out.append("%s: %d nodes differ, first node %d: device %d oracle %d" % (f, len(bad), bad[0], a[bad[0]], b[bad[0]]))
    i = c.info()
    mine = (net.time, net.rng_state(), net.msgs.size())
    if mine != (i["time"], i["rng"], i["queue"]):
        out.append("time / rd / msgs.size(): device %r oracle %r" % (mine, i))
    return out


def lockstep(params, seed, chunk, chunks, byz_delay=0, nl=None, max_slots=16, stopped=0):
    """params = CasperParemeters ctor order: (cycleLength, randomOnTies, blockProducersCount, attestersPerRound,
    blockConstructionTime, attestationConstructionTime); stopped = attesters stop()ped after init() (BASELINE config
    5's "+10 % Byzantine" as SURVEY.md §8d defines it), the same nodes on both sides"""
    g = P.CasperIMD(P.CasperParemeters(*params, None, nl), seed=seed, byz_delay=byz_delay, max_slots=max_slots)
    g.init()
    c = o.CasperIMD(params, None, nl, seed=seed, byz_delay=byz_delay)
    if stopped:
        ids = g.stop_attesters(stopped, seed=seed + 1)
        assert len(set(ids)) == stopped and all(i in g.attester_ids() for i in ids)
        c.stop(ids)
        g.stopped_ids = ids
    assert not diff(g, c), "after init()"
    for k in range(chunks):
        g.network().runMs(chunk)
        c.run_ms(chunk)
        d = diff(g, c)
        assert not d, "t=%d: %s" % (g.network().time, d)
    return g, c


@pytest.mark.gpu
def test_casper_small():
    g, c = lockstep((2, False, 2, 6, 1000, 1), seed=5, chunk=2000, chunks=25)
    assert c.read("headHeight")[0] >= 5 and c.info()["delivered"] > 400


@pytest.mark.gpu
def test_casper_byzantine_delay():  # ByzBlockProducerWF(-2000), PT/CasperByzantineTest.java:41
    lockstep((3, False, 3, 8, 1000, 1), seed=9, chunk=1000, chunks=60, byz_delay=-2000)


@pytest.mark.gpu
def test_casper_reference_test_parameters():  # PT/CasperIMDTest.java:10-11: 5 producers, 5 x 80 attesters
    g, c = lockstep((5, False, 5, 80, 1000, 1), seed=0, chunk=4000, chunks=12)
    assert g.network().node_count == 406 and c.info()["delivered"] > 150000
    assert g.network().read("headHeight")[0] == 5


@pytest.mark.gpu
def test_casper_ten_percent_attesters_stopped():  # BASELINE config 5's "+10 %": 40 of 400 attesters stop()ped
    g, c = lockstep((5, False, 5, 80, 1000, 1), seed=3, chunk=4000, chunks=12, stopped=40)
    held = g.network().read("attestationsHeld")
    ids = g.stopped_ids
    assert ids == P.choose_attesters(g.attester_ids(), 40, seed=4)
    assert int(g.network().read("msgReceived")[ids].sum()) == 0   # a stopped node receives nothing ...
    assert int(g.network().read("msgSent")[ids].sum()) == 0       # ... and its attester task never runs
    live = np.setdiff1d(np.arange(1, g.network().node_count), ids)
    assert int(held[live].min()) > 0 and int(held[ids].max()) == 0


@pytest.mark.gpu
def test_casper_stopped_attesters_byzantine_delay():
    lockstep((3, False, 3, 8, 1000, 1), seed=9, chunk=1000, chunks=40, byz_delay=-2000, stopped=3)


@pytest.mark.gpu
def test_casper_long_chain_runs_one_wavefront_each(monkeypatch):
    """k_expand_runs: chain runs of >= runMin hops (64 for this protocol: a sendAll to N nodes has runs of ~N/300) are
    unrolled one wavefront per run instead of in place by the scanning lane; WG_RUN_MIN=2 sends every run of 2+ hops of
    these small networks through it — same events, same order, same everything as the oracle"""
    monkeypatch.setenv("WG_RUN_MIN", "2")
    lockstep((5, False, 5, 80, 1000, 1), seed=1, chunk=4000, chunks=8, stopped=17)
    lockstep((3, False, 3, 8, 1000, 1), seed=2, chunk=500, chunks=40, byz_delay=-2000)


@pytest.mark.gpu
def test_casper_4096_attesters_a_slot_runs():  # one round of BASELINE config 5's attesters per slot, 2 cycles of 2 slots
    g = P.CasperIMD(P.CasperParemeters(2, False, 2, 4096, 1000, 1), seed=1, max_slots=8)
    g.init()
    net = g.network()
    net.runMs(22000)
    n = net.node_count
    assert n == 1 + 2 + 8192
    assert net.read("headHeight")[0] == 2                      # the observer follows the chain
    assert int(net.read("attestationsHeld").min()) >= 4096     # every node holds the first round's attestations
    assert net.last_stats["delivered"] > 2 * 4096 * n * 0.9    # two rounds of sendAll, minus what is still in flight


@pytest.mark.gpu
def test_config5_size_fast_paths_equal_the_plain_ones(monkeypatch):
    """BASELINE config 5's node count (262 150 nodes, 10 % of the attesters stopped) is beyond the oracle; the
    size-independent property instead: the paths added for this size — chain runs unrolled one wavefront per run
    (k_expand_runs), idle stretches skipped (k_skip_idle) — give exactly what the plain paths give (in-place runs, every
    ms enqueued), which are the ones the small lock-step cases pin against the oracle. Two and a half slots: two rounds
    of 4096 votes and two blocks, every one a sendAll to all nodes."""
    def run(run_min, skip):
        monkeypatch.setenv("WG_RUN_MIN", run_min)
        monkeypatch.setenv("WG_SKIP_IDLE", skip)
        g = P.CasperIMD(P.CasperParemeters(64, False, 5, 4096, 1000, 1), seed=2, max_slots=5)
        g.init()
        ids = g.stop_attesters(26214, seed=3)
        net = g.network()
        net.runMs(20500)
        out = {f: net.read(f) for f in FIELDS}
        out["state"] = (net.time, net.rng_state(), net.msgs.size(), net.last_stats["delivered"])
        return out, ids
    plain, ids = run("0", "0")
    fast, ids2 = run("8", "1")
    assert ids == ids2 and plain["state"] == fast["state"], (plain["state"], fast["state"])
    for f in FIELDS:
        assert np.array_equal(plain[f], fast[f]), f
    n = 1 + 5 + 64 * 4096
    live = np.ones(n, bool)
    live[ids] = False
    assert plain["state"][3] > 2 * (4096 * 0.85) * (n - 26214) * 0.95  # two rounds of votes from the live voters of those slots
    assert int(fast["headHeight"][0]) == 2 and int(fast["headHeight"][live].min()) == 2   # everybody follows the chain
    assert int(fast["msgReceived"][ids].sum()) == 0 and int(fast["attestationsHeld"][live].min()) > 4096


@pytest.mark.gpu
def test_casper_16390_nodes_ten_percent_stopped_against_the_oracle():
    """BASELINE config 5's shape at the largest size the oracle runs inside a test (it keeps every attestation in every
    node's HashSet): cycleLength 64, 5 producers, 256 attesters per slot = 16 390 nodes, 10 % of the attesters stop()ped,
    24 simulated seconds = 8.4 M deliveries, three slots of votes (each a sendAll to all 16 390) — every node's counters,
    head, attestation and block sets, the queue size and the rd state after every 8-second chunk"""
    g, c = lockstep((64, False, 5, 256, 1000, 1), seed=0, chunk=8000, chunks=3, max_slots=5, stopped=1638)
    assert g.network().node_count == 16390 and c.info()["delivered"] > 6000000
    assert int(g.network().read("headHeight")[0]) == 2


@pytest.mark.gpu
def test_casper_65542_nodes_against_the_oracle_trace():
    """BASELINE config 5's shape at a quarter of its size — cycleLength 64, 5 producers, 1024 attesters voting per slot =
    65 542 nodes, 10 % of the attesters stop()ped, 24 simulated seconds = 108.9 M deliveries — against the ORACLE's run of the
    same configuration (tests/golden/casper_config5_shape_65542.json, tests/golden/make_golden.py casper: 8 minutes on one
    core, so a committed fixture rather than a lock-step partner): a digest of every observable `diff` compares, the
    queue size, the rd state and the delivered count after every 8-second chunk."""
    import hashlib
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "casper_config5_shape_65542.json")))
    params = tuple(gold["params"])
    g = P.CasperIMD(P.CasperParemeters(*params, None, None), seed=gold["seed"], max_slots=5)
    g.init()
    ids = g.stop_attesters(gold["stopped"], seed=gold["stop_seed"])
    assert len(ids) == gold["stopped"] and g.network().node_count == gold["nodes"] == 65542
    net = g.network()
    delivered = 0
    for want in gold["marks"]:
        net.runMs(gold["chunk"])
        delivered += net.last_stats["delivered"]
        got = {"time": net.time, "rng": net.rng_state(), "queue": net.msgs.size(), "delivered": delivered}
        for f in FIELDS[:10]:
            got[f] = hashlib.sha256(np.ascontiguousarray(net.read(f), dtype=np.int64).tobytes()).hexdigest()[:16]
        bad = {k: (got.get(k), v) for k, v in want.items() if got.get(k) != v}
        assert not bad, "t=%d: %s" % (net.time, bad)
    assert int(net.read("headHeight")[0]) == gold["observer_head_height"] == 2


def random_on_ties_cases(long=True):
    """randomOnTies (P/CasperIMD.java:250-253, the CasperParemeters() default) resident: a tie's rd.nextBoolean() takes its
    place in the rd sequence from the draws of every earlier event of the ms, so the events that can call best() are
    delivered by one wavefront in global event order (k_casper_seq). ByzBlockProducerWF(+7000) forks the chain with equal
    votes on both branches — ties do draw (checked: the rd state differs from the randomOnTies == false run of the oracle) —
    and (-2000) / no delay are the reference's own byzantine / plain configurations, where no tie arises."""
    cases = [((3, True, 3, 8, 1000, 1), 1, 1000, 80, 7000, 0, True), ((2, True, 2, 6, 1000, 1), 3, 500, 160, 7000, 2, True)]
    if long:
        cases += [((3, True, 3, 8, 1000, 1), 9, 1000, 60, -2000, 0, False), ((5, True, 5, 80, 1000, 1), 1, 4000, 10, -2000, 17, False)]
    for params, seed, chunk, chunks, byz, stopped, ties in cases:
        g, c = lockstep(params, seed=seed, chunk=chunk, chunks=chunks, byz_delay=byz, stopped=stopped)
        plain = o.CasperIMD((params[0], False) + params[2:], None, None, seed=seed, byz_delay=byz)
        if stopped:
            plain.stop(g.stopped_ids)
        plain.run_ms(chunk * chunks)
        assert (c.info()["rng"] != plain.info()["rng"]) == ties, (params, byz)
        assert c.read("headHeight")[0] >= 4


@pytest.mark.gpu
def test_random_on_ties_resident():
    random_on_ties_cases()


def two_blocks_in_one_ms_cases():
    """a byzantine producer delayed by exactly one slot builds in the ms of the next producer's periodic task (ByzBlockProducerWF
    :635-692 with delay = SLOT_DURATION): two blocks in one simulated ms, ids in the order receiveUntil reaches their tasks.
    The resident engine refused this until round 3 (ERR_SAME_MS_BLOCKS); such ms now go through k_casper_seq."""
    for params, seed in (((3, False, 3, 8, 1000, 1), 9), ((2, True, 2, 6, 1000, 1), 4)):
        g, c = lockstep(params, seed=seed, chunk=1000, chunks=60, byz_delay=8000, stopped=0)
        ids = c.read("headId")
        assert c.read("headHeight")[0] >= 4 and len(set(ids.tolist())) >= 1


@pytest.mark.gpu
def test_two_blocks_in_one_ms_resident():
    two_blocks_in_one_ms_cases()
