"""The ENGINE-OWNED RCCL communicator (wg_rccl_unique_id / wg_shard_configure_rccl, wg_config.rccl_id): the sharded
pipeline with its per-ms sums as ncclAllReduce calls on the engine's own stream — no caller callback, no
torch.distributed in the process at all (a Java host has neither). One GPU on the box: a one-rank communicator; every
kernel of the sharded path and every collective runs, results against the oracle bit for bit. Shard-count invariance
over 2/3/4 ranks is tests/test_shards_gloo.py (caller-supplied collective, gloo); 8 logical shards at BASELINE config
3's size against the oracle's golden trace is tests/test_gpu_shards.py."""
import numpy as np
import pytest

import oracle_lib as o
import parity
import wittgenstein_amd as w
from wittgenstein_amd import shards

pytestmark = pytest.mark.gpu


def test_unique_id_is_128_opaque_bytes_and_fresh_each_time():
    a, b = shards.rccl_unique_id(), shards.rccl_unique_id()
    assert len(a) == len(b) == 128 and a != b


def test_pingpong_through_the_engines_own_communicator():
    p = w.PingPong(w.PingPongParameters(1000), seed=0, config=shards.config_rccl())
    p.init()
    c = o.PingPong(1000, seed=0)
    for _ in range(10):  # P/PingPong.java:94-101
        p.network().runMs(50)
        c.run_ms(50)
        d = parity.diff_pingpong(p, c)
        assert not d, d
    assert shards.shard_range(p.network()) == (0, 1000)
    calls, words = shards.traffic(p.network())
    assert calls > 0 and words >= 2000 + 5 * 1000  # one packed word per event + one record image per Pong


@pytest.mark.parametrize("params", [(64, 57, 4, 50, 10, 20, 10, 6, 0), (256, 230, 4, 50, 10, 20, 10, 25, 100)])
def test_handel_through_the_engines_own_communicator(params):
    g, c = parity.handel_pair(params, seed=2, config=shards.config_rccl(queue_cap=64))
    k = 0
    while c.cont_if() and k < 400:
        g.network().runMs(10)
        c.run_ms(10)
        k += 1
        if k % 5 == 0:
            d = parity.diff_handel(g, c)
            assert not d, (k, d)
    d = parity.diff_handel(g, c)
    assert not d, d
    assert not c.cont_if() and not g.cont_if()
    dl = c.stats()["deliveredByLevel"]
    assert (g.network().delivered_by_level()[:len(dl)].astype(np.uint64) == dl).all()
    assert shards.traffic(g.network())[0] > 0


def test_gsf_through_the_engines_own_communicator():
    import test_gpu_gsf as tg
    g, c = tg.pair((256, 250, 3, 50, 10, 10, 5), seed=3, config=shards.config_rccl())
    k = 0
    while c.cont_if() and k < 600:
        g.network().runMs(5)
        c.run_ms(5)
        k += 1
    d = tg.diff(g, c)
    assert not d, d
    assert not g.cont_if()


def test_casper_through_the_engines_own_communicator():  # sendAll + far envelopes + the table exchange through ncclAllReduce
    import test_gpu_casper_resident as tcr
    from wittgenstein_amd import protocols as P
    params = (5, False, 5, 80, 1000, 1)
    g = P.CasperIMD(P.CasperParemeters(*params, None, None), seed=3, max_slots=16, config=shards.config_rccl())
    g.init()
    c = o.CasperIMD(params, None, None, seed=3)
    ids = g.stop_attesters(40, seed=4)
    c.stop(ids)
    for _ in range(12):
        g.network().runMs(4000)
        c.run_ms(4000)
        d = tcr.diff(g, c)
        assert not d, (g.network().time, d)
    assert c.info()["delivered"] > 100000 and shards.traffic(g.network())[0] > 0


def test_p2pflood_through_the_engines_own_communicator():
    import test_gpu_p2pflood_resident as tf
    from wittgenstein_amd import protocols as P
    params = (300, 20, 20, 3, 1, 6, 10)
    g = P.P2PFlood(P.P2PFloodParameters(*params, None, None), seed=4, config=shards.config_rccl())
    g.init()
    c = o.P2PFlood(params, None, None, seed=4)
    for _ in range(40):
        g.network().runMs(100)
        c.run_ms(100)
        d = tf.diff(g, c)
        assert not d, (g.network().time, d)
    assert c.info()["delivered"] > 4000 and shards.traffic(g.network())[0] > 0


def test_configure_after_allocation_is_refused():
    net = w.Network.create({})
    net.add_nodes([1, 2], [1, 2])
    net.load_protocol(1)
    import ctypes as C
    from wittgenstein_amd import _lib as L
    buf = (C.c_uint8 * 128).from_buffer_copy(shards.rccl_unique_id())
    assert L.lib().wg_shard_configure_rccl(net._h, 0, 1, buf) == L.WG_ESTATE
