"""k shards of one simulation in ONE process (wittgenstein_amd.shards.LoopbackGroup: k engines, k host threads, the
all-reduce sums the k buffers in place) — on the CPU wave emulator here, on one MI355X in tests/test_gpu_shards.py.
Shard-count invariance against the oracle, as tests/test_shards_gloo.py checks it over processes."""
import os
import subprocess

import numpy as np
import pytest

import wittgenstein_amd._lib as L
import parity

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")


@pytest.fixture(scope="module", autouse=True)
def emulated_kernels(oracle):
    subprocess.run(["make", "-s", "-C", EMU_DIR], check=True)
    saved = (L._lib, L.LIB_PATH)
    L._lib, L.LIB_PATH = None, os.path.join(EMU_DIR, "libwittgpu_emu.so")
    try:
        L.lib()
        yield
    finally:
        L._lib, L.LIB_PATH = saved


def handel_loopback(k, params, seed, device_memory, chunks_max=400, check_every=5, alltoall=True):
    """runs k shards in lock-step with the oracle; returns the list of mismatches (empty = identical)"""
    import wittgenstein_amd as w
    from wittgenstein_amd import shards
    import oracle_lib as o
    n, thr, pair, lw, ec, per, fp, down, desync = params
    grp = shards.LoopbackGroup(k, device_memory=device_memory)
    sims = []
    for s in range(k):
        g = w.Handel(w.HandelParameters(n, thr, pair, lw, ec, per, fp, down, parity.NB, parity.NL, desync), seed=seed,
                     config=grp.config(s, alltoall=alltoall, queue_cap=64))
        g.init()
        sims.append(g)
    c = o.Handel(n, thr, pair, lw, ec, per, fp, down, parity.NB, parity.NL, desync, seed=seed)
    nets = [g.network() for g in sims]

    class Whole:  # what parity.diff_handel reads, assembled from the shards' own rows
        time = property(lambda self: nets[0].time)

        def rng_state(self):
            assert len({net.rng_state() for net in nets}) == 1
            return nets[0].rng_state()

        def read(self, f):
            return grp.gather([net.read(f) for net in nets], nets)

        def read_level(self, f):
            return grp.gather([net.read_level(f) for net in nets], nets)

        def read_bits(self, f):
            return grp.gather([net.read_bits(f) for net in nets], nets)

    class G:
        def network(self):
            return Whole()

    bad, steps = [], 0
    while c.cont_if() and steps < chunks_max and not bad:
        grp.run(lambda s: nets[s].runMs(10))
        c.run_ms(10)
        steps += 1
        if steps % check_every == 0:
            bad += parity.diff_handel(G(), c)
        if any(g.cont_if() for g in sims) != c.cont_if():
            bad.append("cont_if")
    bad += parity.diff_handel(G(), c)
    dl = c.stats()["deliveredByLevel"]
    if not np.array_equal(nets[0].delivered_by_level()[:len(dl)].astype(np.uint64), dl):
        bad.append("delivered_by_level")
    if c.cont_if():
        bad.append("did not converge in %d chunks" % chunks_max)
    handel_loopback.by_exchange = [shards.traffic_by_exchange(net) for net in nets]  # (wg_shard_traffic, of the last call)
    return bad, [shards.traffic(net) for net in nets]


@pytest.mark.parametrize("k,params", [(2, (64, 57, 4, 50, 10, 20, 10, 6, 0)), (4, (32, 28, 4, 20, 5, 10, 10, 2, 0))])
def test_handel_logical_shards_match_the_oracle(k, params):
    bad, traffic = handel_loopback(k, params, seed=1, device_memory=False)
    assert bad == []
    assert len({t[0] for t in traffic}) == 1 and traffic[0][0] > 0     # every shard issued the same collectives (the words a shard RECEIVES differ: the snapshots go to their readers)


@pytest.mark.parametrize("k", [2, 3])
def test_dissemination_snapshots_go_to_the_shard_that_reads_them(k):
    """Round 5: with an all-to-all transport a dissemination's snapshot (SendSigs.sigs, P/Handel.java:254) travels as the
    sub-rows the messages to OTHER shards' nodes point at, to those shards only (HandelState::xout, chunks of 16 words) —
    not as the whole copied row inside an all-reduce image to every shard. Same run as the oracle's either way (256 nodes:
    rows of several words; 3 shards: ranges that do not line up with the level blocks)."""
    params = (256, 228, 4, 50, 10, 20, 10, 25, 0)
    bad, directed = handel_loopback(k, params, seed=2, device_memory=False, check_every=10)
    assert bad == []
    by_d = handel_loopback.by_exchange
    bad, image = handel_loopback(k, params, seed=2, device_memory=False, check_every=10, alltoall=False)
    assert bad == []
    by_i = handel_loopback.by_exchange
    for s in range(k):  # wg_shard_traffic: the totals of wg_shard_info split by exchange; only the snapshots' rows differ between the two forms
        assert sum(c for c, _ in by_d[s].values()) == directed[s][0] and sum(x for _, x in by_d[s].values()) == directed[s][1]
        assert sum(c for c, _ in by_i[s].values()) == image[s][0] and sum(x for _, x in by_i[s].values()) == image[s][1]
        for kind in ("events", "outbox", "envelopes", "candidates"):
            assert by_d[s][kind] == by_i[s][kind] and by_d[s][kind][0] > 0
        assert by_i[s]["directed_counts"] == (0, 0) and by_d[s]["directed_counts"][1] == k * k * by_d[s]["directed_counts"][0]
        assert by_i[s]["snapshots"][1] > 0 and by_d[s]["snapshots"][1] > 0
    assert len({t[0] for t in directed}) == 1 and len(set(image)) == 1
    # (at 256 nodes a sub-row is two words in a 17-word chunk: the volumes are compared where rows are wide — 8 192 nodes on
    # the MI355X, tests/test_gpu_shards.py)


def handel_shards_vs_unsharded(k, params, seed, device_memory, chunk=10, queue_cap=64, traffic=None, alltoall=True):
    """k logical shards against the UNSHARDED engine (no oracle: usable at sizes the oracle cannot reach in a test).
    Both run RunMultipleTimes' loop; compared at the end: every per-node scalar, the per-level scalars, all five bitset
    rows, time, rd state, delivered count. Returns the list of mismatches."""
    import wittgenstein_amd as w
    from wittgenstein_amd import shards
    n, thr, pair, lw, ec, per, fp, down, desync = params
    hp = w.HandelParameters(n, thr, pair, lw, ec, per, fp, down, parity.NB, parity.NL, desync)
    ref = w.Handel(hp, seed=seed, config={"queue_cap": queue_cap})
    ref.init()
    rd_, rms = w.Batch([ref.network()]).run_multiple_times(chunk=chunk, maxTime=20000)
    grp = shards.LoopbackGroup(k, device_memory=device_memory)
    sims = []
    for s in range(k):
        sims.append(w.Handel(hp, seed=seed, config=grp.config(s, alltoall=alltoall, queue_cap=queue_cap)))
        sims[-1].init()
    nets = [g.network() for g in sims]
    delivered = ms = 0
    while True:
        did = grp.run(lambda s: nets[s].runMs(chunk))[0]
        delivered += nets[0].last_stats["delivered"]
        ms += chunk
        if not (nets[0].time < 20000 and (not did or any(g.cont_if() for g in sims))):
            break
    if traffic is not None:
        traffic["words"] = [shards.traffic(net)[1] for net in nets]
    bad = []
    if (delivered, ms) != (rd_[0], rms[0]):
        bad.append("delivered / simulated ms: shards %r unsharded %r" % ((delivered, ms), (rd_[0], rms[0])))
    rnet = ref.network()
    if (nets[0].time, nets[0].rng_state()) != (rnet.time, rnet.rng_state()):
        bad.append("time / rd state")
    for f in parity.SCALARS:
        if not np.array_equal(grp.gather([net.read(f) for net in nets], nets), rnet.read(f)):
            bad.append(f)
    for f in parity.LEVELS:
        if not np.array_equal(grp.gather([net.read_level(f) for net in nets], nets), rnet.read_level(f)):
            bad.append(f)
    for f in parity.BITS:
        if not np.array_equal(grp.gather([net.read_bits(f) for net in nets], nets), rnet.read_bits(f)):
            bad.append(f)
    return bad, int((rnet.read("doneAt") > 0).sum()), delivered


def test_logical_shards_equal_the_unsharded_engine():
    bad, done, delivered = handel_shards_vs_unsharded(2, (64, 57, 4, 50, 10, 20, 10, 6, 0), seed=3, device_memory=False)
    assert bad == [] and done == 58 and delivered > 0


def p2pflood_loopback(k, params, nl, seed, chunk, chunks, device_memory=False):
    """P2PFlood (P/P2PFlood.java) resident on k logical shards in lock-step with the oracle: every first receipt's shuffled
    MultipleDestWithDelayEnvelope — destinations AND explicit arrivals — goes through the replicated envelope creation
    (k_shard_multi_fill / k_shard_multi_create); every observable of tests/test_gpu_p2pflood_resident.py::diff"""
    import oracle_lib as o
    from wittgenstein_amd import protocols as P, shards
    import test_gpu_p2pflood_resident as tf
    grp = shards.LoopbackGroup(k, device_memory=device_memory)
    sims = []
    for s in range(k):
        sims.append(P.P2PFlood(P.P2PFloodParameters(*params, None, nl), seed=seed, config=grp.config(s)))
        sims[-1].init()
    c = o.P2PFlood(params, None, nl, seed=seed)
    nets = [g.network() for g in sims]

    class Whole:
        time = property(lambda self: nets[0].time)
        msgs = property(lambda self: nets[0].msgs)

        def rng_state(self):
            assert len({net.rng_state() for net in nets}) == 1
            return nets[0].rng_state()

        def read(self, f):
            return grp.gather([net.read(f) for net in nets], nets)

    class G:
        def network(self):
            return Whole()

    assert not tf.diff(G(), c), "after init()"
    for _ in range(chunks):
        grp.run(lambda s: nets[s].runMs(chunk))
        c.run_ms(chunk)
        d = tf.diff(G(), c)
        assert not d, "t=%d: %s" % (nets[0].time, d)
    return c, [shards.traffic(net) for net in nets]


def test_p2pflood_logical_shards_match_the_oracle():
    c, traffic = p2pflood_loopback(2, (100, 10, 50, 1, 1, 10, 30), "NetworkNoLatency", seed=0, chunk=1000, chunks=20)
    assert c.info()["delivered"] > 800 and len(set(traffic)) == 1 and traffic[0][0] > 0
    c, _ = p2pflood_loopback(3, (300, 20, 20, 3, 1, 6, 10), None, seed=4, chunk=100, chunks=40)
    assert c.info()["delivered"] > 4000
