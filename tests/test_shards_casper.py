"""Casper IMD (P/CasperIMD.java) resident on node-range shards of ONE simulation (wg_shard_configure): per-node rows with the
owner of the node, block / attestation tables replicated and filled by exchange (CasperState::xtab, k_casper_shard_apply),
sendAll through the replicated envelope creation (k_shard_multi_*, k_sendall_* on every shard), the 8-second periodic
tasks through the far buffer every shard keeps alike. k shards in one process (shards.LoopbackGroup) on the CPU wave
emulator here — on one MI355X in tests/test_gpu_shards.py — in lock-step with the oracle (oracle/casper.hpp, pinned
against PT/CasperIMDTest / PT/CasperByzantineTest): after every chunk every observable of
tests/test_gpu_casper_resident.py::diff, assembled from the shards' own rows."""
import os
import subprocess

import pytest

import wittgenstein_amd._lib as L

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


def casper_loopback(k, params, seed, chunk, chunks, byz_delay=0, max_slots=16, stopped=0, device_memory=False):
    """k logical shards in lock-step with the oracle; returns (oracle, traffic of the shards)"""
    import oracle_lib as o
    from wittgenstein_amd import protocols as P, shards
    import test_gpu_casper_resident as tcr
    grp = shards.LoopbackGroup(k, device_memory=device_memory)
    sims = []
    for s in range(k):
        g = P.CasperIMD(P.CasperParemeters(*params, None, None), seed=seed, byz_delay=byz_delay, max_slots=max_slots,
                        config=grp.config(s))
        g.init()
        sims.append(g)
    c = o.CasperIMD(params, None, None, seed=seed, byz_delay=byz_delay)
    if stopped:
        ids = [g.stop_attesters(stopped, seed=seed + 1) for g in sims][0]
        c.stop(ids)
    nets = [g.network() for g in sims]

    class Whole:  # what tcr.diff reads, assembled from the shards' own rows (a shard reports zeros for the others' nodes)
        time = property(lambda self: nets[0].time)
        msgs = property(lambda self: nets[0].msgs)

        def rng_state(self):
            assert len({net.rng_state() for net in nets}) == 1
            return nets[0].rng_state()

        def read(self, f):
            if f in ("x", "y"):
                return nets[0].read(f)
            return grp.gather([net.read(f) for net in nets], nets)

    class G:
        def network(self):
            return Whole()

    assert not tcr.diff(G(), c), "after init()"
    for _ in range(chunks):
        grp.run(lambda s: nets[s].runMs(chunk))
        c.run_ms(chunk)
        d = tcr.diff(G(), c)
        assert not d, "t=%d: %s" % (nets[0].time, d)
        assert len({net.msgs.size() for net in nets}) == 1  # the scheduler is replicated
    return c, [shards.traffic(net) for net in nets]


def test_casper_two_shards_match_the_oracle():
    c, traffic = casper_loopback(2, (2, False, 2, 6, 1000, 1), seed=5, chunk=2000, chunks=14)
    assert c.read("headHeight")[0] >= 2 and c.info()["delivered"] > 200
    assert len(set(traffic)) == 1 and traffic[0][0] > 0  # every shard issued the same collectives


def test_casper_three_shards_byzantine_delay_and_stopped_attesters():  # ByzBlockProducerWF(-2000) + 3 attesters stop()ped
    c, _ = casper_loopback(3, (3, False, 3, 8, 1000, 1), seed=9, chunk=1000, chunks=30, byz_delay=-2000, stopped=3)
    assert c.read("headHeight")[0] >= 2


def test_casper_reference_test_parameters_four_shards_ten_percent_stopped():
    """PT/CasperIMDTest.java:10-11's network (5 producers, 5 x 80 attesters: 406 nodes) with 40 attesters stop()ped — BASELINE
    config 5's "+10 %" as SURVEY.md §8d defines it — on 4 shards: three slots, 80 votes (each a sendAll to all 406) per slot"""
    c, _ = casper_loopback(4, (5, False, 5, 80, 1000, 1), seed=3, chunk=4000, chunks=7, stopped=40)
    assert c.info()["delivered"] > 50000


def test_casper_shards_with_chain_runs_one_wavefront_each(monkeypatch):
    """k_expand_runs on a sharded engine (a sendAll's runs of consecutive same-ms hops unrolled one wavefront per run: the
    attestation events are not threaded, every shard releases the envelope's slot): WG_RUN_MIN=2 sends every run of 2+
    hops of these small networks through it"""
    monkeypatch.setenv("WG_RUN_MIN", "2")
    casper_loopback(2, (5, False, 5, 80, 1000, 1), seed=1, chunk=4000, chunks=5, stopped=17)
    casper_loopback(3, (3, False, 3, 8, 1000, 1), seed=2, chunk=500, chunks=40, byz_delay=-2000)


def test_casper_random_on_ties_on_shards():
    """randomOnTies (P/CasperIMD.java:250-253, the CasperParemeters() default) on a sharded engine: a tie's rd.nextBoolean() takes
    its index in the rd sequence from the draws of every earlier event of the ms — other shards' events too — so the ordered
    visit of the ms's blocks and tasks goes round the shards (k_casper_seq_shard, one two-word collective per change of
    owner). ByzBlockProducerWF(+7000) forks the chain with equal votes on both branches: ties do draw (the rd state differs
    from the randomOnTies == false run of the oracle)."""
    import oracle_lib as o
    for k, params, seed, chunk, chunks, stopped in ((2, (2, True, 2, 6, 1000, 1), 3, 500, 160, 2), (3, (3, True, 3, 8, 1000, 1), 1, 1000, 80, 0)):
        c, traffic = casper_loopback(k, params, seed=seed, chunk=chunk, chunks=chunks, byz_delay=7000, stopped=stopped)
        assert len(set(traffic)) == 1 and c.read("headHeight")[0] >= 4
        if not stopped:
            plain = o.CasperIMD((params[0], False) + params[2:], None, None, seed=seed, byz_delay=7000)
            plain.run_ms(chunk * chunks)
            assert c.info()["rng"] != plain.info()["rng"]
