"""wg_snapshot / wg_restore (include/wittgpu.h "init() image"): an engine restored to its init() image and run again
is indistinguishable from a freshly initialised one — the cheap form of RunMultipleTimes' `p.copy(); rd.setSeed(i);
init()` (C/RunMultipleTimes.java:44-48) that bench.py uses between timed steps. Checked against the oracle (which is
re-created and re-initialised from scratch for every pass, as the reference does)."""
import pytest

import oracle_lib as o
import parity
import wittgenstein_amd as w

pytestmark = pytest.mark.gpu


def _run_handel_pair(g, c, chunk=10, limit=400):
    steps = 0
    while c.cont_if() and steps < limit:
        g.network().runMs(chunk)
        c.run_ms(chunk)
        steps += 1
    d = parity.diff_handel(g, c)
    assert not d, d
    assert not g.cont_if()


def test_handel_restore_equals_fresh_init():
    params = (256, 228, 4, 50, 10, 20, 10, 25, 0)
    g, c = parity.handel_pair(params, seed=3)
    size = g.network().snapshot()
    assert size > 0
    _run_handel_pair(g, c)
    t_end, rng_end = g.network().time, g.network().rng_state()
    assert g.network().read("sigsChecked").sum() > 0  # receptionRanks were bumped (P/Handel.java:825)
    for _ in range(2):  # restore twice: the image itself must survive a run
        g.network().restore()
        assert g.network().time == 0
        c = o.Handel(*params[:8], parity.NB, parity.NL, params[8], seed=3)  # Protocol.copy() + init(), from scratch
        d = parity.diff_handel(g, c)
        assert not d, d
        _run_handel_pair(g, c, chunk=7)  # (another chunk size: the restored engine is not tied to the first run's)
    g.network().restore()
    c = o.Handel(*params[:8], parity.NB, parity.NL, params[8], seed=3)
    _run_handel_pair(g, c)
    assert (g.network().time, g.network().rng_state()) == (t_end, rng_end)


def test_snapshot_only_before_the_first_event():
    g, _ = parity.handel_pair((64, 57, 4, 50, 10, 20, 10, 6, 0))
    g.network().runMs(30)
    with pytest.raises(w.IllegalStateException):
        g.network().snapshot()
    g2, _ = parity.handel_pair((64, 57, 4, 50, 10, 20, 10, 6, 0))
    with pytest.raises(w.IllegalStateException):
        g2.network().restore()  # no image


def test_batch_restore_and_run_multiple_times_again():
    """bench.py's step: R copies, snapshot once, then per step restore + wg_batch_run_multiple_times."""
    params = (128, 114, 4, 50, 10, 20, 10, 12, 0)
    sims = []
    for seed in range(3):
        g = w.Handel(w.HandelParameters(*params[:8], parity.NB, parity.NL, params[8]), seed=seed)
        g.init()
        g.network().snapshot()
        sims.append(g)
    batch = w.Batch([g.network() for g in sims])
    first = batch.run_multiple_times(chunk=10, maxTime=20000)
    state = [(g.network().time, g.network().rng_state(), g.network().read("doneAt").tolist()) for g in sims]
    for _ in range(2):
        for g in sims:
            g.network().restore()
        again = batch.run_multiple_times(chunk=10, maxTime=20000)
        assert again == first
        assert state == [(g.network().time, g.network().rng_state(), g.network().read("doneAt").tolist()) for g in sims]
    for seed, g in enumerate(sims):  # and it is the oracle's run for that seed
        c = o.Handel(*params[:8], parity.NB, parity.NL, params[8], seed=seed)
        while True:
            did = c.run_ms(10)
            if did and not c.cont_if():
                break
        d = parity.diff_handel(g, c)
        assert not d, (seed, d)


def test_pingpong_and_gsf_restore():
    gp = w.PingPong(w.PingPongParameters(1000, parity.NB, parity.NL))
    gp.init()
    gp.network().snapshot()
    for _ in range(2):
        cp = o.PingPong(1000, parity.NB, parity.NL)
        for _ in range(4):
            gp.network().runMs(50)
            cp.run_ms(50)
        d = parity.diff_pingpong(gp, cp)
        assert not d, d
        gp.network().restore()
    import test_gpu_gsf as tg
    params = (64, 57, 3, 20, 10, 10, 6)
    gg, _ = tg.pair(params, tg.NBG, seed=2)
    gg.network().snapshot()
    for _ in range(2):  # (GSFSignature keeps every allocation in the image: nothing protocol-specific to recompute)
        cg = o.GSFSignature(*params, tg.NBG, tg.NL, seed=2)
        for _ in range(30):
            gg.network().runMs(10)
            cg.run_ms(10)
        d = tg.diff(gg, cg)
        assert not d, d
        gg.network().restore()
