"""wg_batch_*: independent simulations advanced in lock-step by one launch sequence (the device form of
C/RunMultipleTimes.java:44-64) must leave every member in exactly the state the CPU oracle reaches when it
runs that seed alone — members stop at different times, stopped members are not advanced."""
import numpy as np
import pytest

import oracle_lib as o
import parity
import wittgenstein_amd as w

pytestmark = pytest.mark.gpu


def ratios(n, dead=0.10):
    down = int(n * dead)
    return (n, int(n * (1 - dead) * 0.99), 4, 50, 10, 20, 10, down, 0)


# (12 copies: every kernel of the ordering / append chain then runs on a grid scaled by the batch size — engine.h
# grid_per_engine(): scans from 5 copies, the multisplit from 3, k_resolve from 9 — instead of its per-engine constant)
@pytest.mark.parametrize("n,seeds", [(256, [0, 1, 2, 3, 4]), (1024, [7, 8, 9]), (512, list(range(20, 32)))])
def test_handel_batch_matches_oracle_per_seed(n, seeds):
    pairs = [parity.handel_pair(ratios(n), seed=s) for s in seeds]
    batch = w.Batch([g.network() for g, _ in pairs])
    delivered, sim_ms = batch.run_multiple_times(chunk=10)
    stop = []
    for (g, c), d, ms in zip(pairs, delivered, sim_ms):
        while c.cont_if():
            c.run_ms(10)
        diff = parity.diff_handel(g, c)
        assert not diff, diff
        assert c.info(False)["time"] == ms and c.info(False)["delivered"] == d
        assert not g.cont_if()
        stop.append(ms)
    assert len(set(stop)) > 1 or len(seeds) == 1  # members really stopped at different times


@pytest.mark.parametrize("n,seeds", [(256, list(range(40, 58)))])
def test_gsf_batch_matches_oracle_per_seed(n, seeds):
    """GSFSignature copies batched (what bench.py's third_workload runs by the hundred): from 8 members on a launch deals its
    blocks to the XCDs by engine (engine_kernels.hip.h wg_place), from 16 on the per-engine grids follow the engine's own work
    (engine.h grid_per_engine), and the accelerated calls' multi-destination envelopes keep their latencies beside the ids
    (CHAIN_LAT) — every member still ends exactly where its own oracle run ends (P/GSFSignature.java:670-682 continuation)."""
    import test_gpu_gsf as tg
    pairs = [tg.pair((n, int(0.99 * n), 3, 50, 10, 10, 0), seed=s) for s in seeds]
    batch = w.Batch([g.network() for g, _ in pairs])
    delivered, sim_ms = batch.run_multiple_times(chunk=10, maxTime=20000)
    for (g, c), d, ms in zip(pairs, delivered, sim_ms):
        while c.cont_if():
            c.run_ms(10)
        diff = tg.diff(g, c)
        assert not diff, diff
        assert c.info(False)["time"] == ms and c.info(False)["delivered"] == d
        assert not g.cont_if()


def test_batch_every_ms_lockstep_with_oracle():
    pairs = [parity.handel_pair((128, 100, 1, 10, 4, 7, 10, 12, 0), seed=s) for s in (0, 5)]
    batch = w.Batch([g.network() for g, _ in pairs])
    for t in range(400):
        batch.runMs(1)
        for g, c in pairs:
            c.run_ms(1)
        if t % 20 == 0:
            for g, c in pairs:
                d = parity.diff_handel(g, c)
                assert not d, "t=%d %s" % (t, d)
    for g, c in pairs:
        assert not parity.diff_handel(g, c)


def test_pingpong_batch_and_active_mask():
    seeds = [0, 1, 2]
    gs = []
    for s in seeds:
        g = w.PingPong(w.PingPongParameters(1000, parity.NB, parity.NL), seed=s)
        g.init()
        gs.append(g)
    batch = w.Batch([g.network() for g in gs])
    batch.runMs(100)
    batch.runMs(150, active=[True, False, True])  # member 1 is not advanced
    assert [g.network().time for g in gs] == [250, 100, 250]
    for g, s, until in zip(gs, seeds, (250, 100, 250)):
        c = o.PingPong(1000, parity.NB, parity.NL, seed=s)
        c.run_ms(100)
        if until > 100:
            c.run_ms(150)
        assert not parity.diff_pingpong(g, c)
    # a member can still be driven on its own afterwards
    gs[1].network().runMs(150)
    c = o.PingPong(1000, parity.NB, parity.NL, seed=1)
    c.run_ms(100)
    c.run_ms(150)
    assert not parity.diff_pingpong(gs[1], c)


def test_batch_rejects_mismatched_members():
    a = w.PingPong(w.PingPongParameters(100, parity.NB, parity.NL))
    b = w.PingPong(w.PingPongParameters(200, parity.NB, parity.NL))
    a.init()
    b.init()
    with pytest.raises(w.IllegalArgumentException):
        w.Batch([a.network(), b.network()])
    with pytest.raises(w.IllegalArgumentException):
        w.Batch([a.network(), a.network()])


@pytest.mark.parametrize("n", [512])
def test_run_multiple_times_device_loop_equals_host_loop(n):
    """wg_batch_run_multiple_times (loop condition evaluated on the device, chunks enqueued back to back) and the
    same loop driven from the host with one wg_batch_run_ms per chunk reach the same state, per member."""
    seeds = [3, 4, 5] if n > 64 else [3, 4]
    res = []
    for on_device in (True, False):
        gs = [parity.handel_pair(ratios(n), seed=s)[0] for s in seeds]
        batch = w.Batch([g.network() for g in gs])
        out = batch.run_multiple_times(chunk=10, maxTime=20000, on_device=on_device)
        res.append((out, [(g.network().time, g.network().rng_state(), g.network().read("doneAt").tolist(),
                           g.network().read("msgReceived").tolist()) for g in gs]))
    assert res[0] == res[1]
