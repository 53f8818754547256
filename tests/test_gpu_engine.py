"""Engine semantics on the MI355X through the C ABI: the reference's CT/NetworkTest.java cases restated
against libwittgpu.so with the resident PingPong protocol as the probe (a delivered Pong increments
`pong` at the destination = the reference tests' AtomicInteger; a delivered Ping sends a Pong back).
CT/ = core/src/test/java/net/consensys/wittgenstein/core/."""
import numpy as np
import pytest

import oracle_lib as o
import parity
import wittgenstein_amd as w
from wittgenstein_amd import Network

pytestmark = pytest.mark.gpu

PING, PONG = 0, 1
PINGPONG = 1


def four_nodes(latency="NetworkNoLatency", xs=(1, 1, 1, 1), ys=(1, 1, 1, 1), **kw):
    """fixture of CT/NetworkTest.java:12-33: four nodes at (1,1), NetworkNoLatency"""
    n = Network.create(kw.get("config"))
    n.setNetworkLatency(latency)
    n.add_nodes(list(xs), list(ys), extraLatency=kw.get("extra"))
    n.load_protocol(PINGPONG)
    return n


def test_simple_message_and_time():  # CT/NetworkTest.java:34-55
    n = four_nodes()
    n.send(PONG, 1, 1, 2)
    assert n.msgs.size() == 1
    assert n.read("pong").sum() == 0
    assert n.run(5)
    assert list(n.read("pong")) == [0, 0, 1, 0]
    assert n.time == 5000


def test_register_task():  # :57-68, :436-450
    n = four_nodes()
    n.registerTask(7, 100, 0)
    n.runMs(99)
    assert n.last_stats["tasks"] == 0
    n.runMs(1)
    assert n.last_stats["tasks"] == 1
    assert n.msgs.size() == 0
    n.runMs(5000)
    assert n.last_stats["tasks"] == 0


def test_all_flavors_of_send():  # :70-97 — 4 envelopes, 6 deliveries
    n = four_nodes()
    n.send(PONG, 1, 1, 2)
    n.send(PONG, 1, 1, 2)
    n.send(PONG, 1, 1, [2, 3])
    n.send(PONG, 1, 1, [2, 3])
    assert n.msgs.size() == 4
    n.run(1)
    assert n.msgs.size() == 0
    assert list(n.read("pong")) == [0, 0, 4, 2]
    assert n.read("msgSent")[1] == 6


def test_multiple_message_with_delays():  # :119-144 and testMsgArrival :187-207 (arrivals 2, 13, 24)
    n = four_nodes()
    n.send(PONG, 1, 0, [1, 2, 3], delayBetween=10)
    assert [n.msgs.sizeAt(t) for t in (2, 13, 24)] == [1, 0, 0]  # one chained envelope, re-pushed per hop
    n.runMs(2)
    assert n.read("pong").sum() == 1
    n.runMs(11)
    assert n.read("pong").sum() == 2
    n.runMs(11)
    assert n.read("pong").sum() == 3
    assert n.msgs.size() == 0


def test_delays_across_horizon_pages():  # :146-185 (slot edges of the reference's 60 s MsgsSlot)
    n = four_nodes(config={"horizon_ms": 4096})
    n.send(PONG, 1000, 0, [1, 2, 3], delayBetween=1500)
    assert n.msgs.size() == 1
    n.runMs(1001)
    assert n.msgs.size() == 1 and n.read("pong").sum() == 1
    n.runMs(5000)
    assert n.msgs.size() == 0 and n.read("pong").sum() == 3


def test_stats():  # :273-298
    n = four_nodes()
    n.send(PONG, 1, 0, [1, 2, 3])
    n.send(PONG, 1, 0, 1)
    n.runMs(2)
    assert list(n.read("msgReceived")) == [0, 2, 1, 1]
    assert list(n.read("bytesReceived")) == [0, 2, 1, 1]
    assert list(n.read("msgSent")) == [4, 0, 0, 0]
    assert list(n.read("bytesSent")) == [4, 0, 0, 0]


def test_partitions():  # :348-422
    n = Network.create()
    n.setNetworkLatency("NetworkNoLatency")
    n.add_nodes([200, 400, 600, 800], [1, 1, 1, 1])
    n.load_protocol(PINGPONG)
    n.partition(0.25)
    n.send(PONG, 1, 0, 1)
    assert n.msgs.size() == 1
    n.send(PONG, 1, 1, 2)  # crosses the cut at x = 500: dropped at send time
    assert n.msgs.size() == 1
    n.send(PONG, 1, 2, 3)
    assert n.msgs.size() == 2
    n.partition(0.35)  # second cut at 700: node 3 alone
    n.send(PONG, 1, 2, 3)
    n.send(PONG, 1, 3, 0)
    assert n.msgs.size() == 2
    with pytest.raises(w.IllegalArgumentException):
        n.partition(0.35)
    with pytest.raises(w.IllegalArgumentException):
        n.partition(1.0)
    n.runMs(5)
    # 2 -> 3 was sent inside one partition but the second cut separates them before delivery (:606)
    assert list(n.read("pong")) == [0, 1, 0, 0]
    assert list(n.read("msgSent")) == [1, 1, 2, 1]  # msgSent counts dropped messages too (:476-477)


def test_task_on_stopped_node_and_periodic():  # :452-479
    n = four_nodes()
    n.registerTask(7, 1000, 0)
    n.set_node_down(0)
    n.runMs(5000)
    assert n.last_stats["tasks"] == 0
    n = four_nodes()
    n.registerPeriodicTask(7, 1000, 100, 0)
    counts = []
    for ms in (500, 500, 100, 50):
        n.runMs(ms)
        counts.append(n.last_stats["tasks"])
    assert counts == [0, 1, 1, 0]
    n.set_node_down(0)
    n.runMs(1000)
    assert n.last_stats["tasks"] == 0  # a periodic task skipped once is gone (C/messages/PeriodicTask.java:39-47)
    n.set_node_down(0, False)
    n.runMs(1000)
    assert n.last_stats["tasks"] == 0


def test_delivery_to_down_node_and_down_at_send():  # C/Network.java:478,606
    n = four_nodes()
    n.send(PONG, 5, 0, [1, 2, 3])
    n.set_node_down(2)  # goes down after the send: skipped at delivery, chain continues
    n.runMs(10)
    assert list(n.read("pong")) == [0, 1, 0, 1]
    n.send(PONG, n.time + 1, 0, [1, 2, 3])  # down at send: dropped at send, still counted in msgSent
    n.runMs(10)
    assert list(n.read("pong")) == [0, 2, 0, 2]
    assert n.read("msgSent")[0] == 6


def test_argument_errors():  # C/Network.java:319-321,371,374,471
    n = four_nodes()
    with pytest.raises(w.IllegalArgumentException):
        n.runMs(0)
    with pytest.raises(w.IllegalArgumentException):
        n.send(PONG, 1, 9, 1)
    with pytest.raises(w.IllegalArgumentException):
        n.send(PONG, 1, 0, 9)
    with pytest.raises(w.IllegalStateException):
        n.send(PONG, 0, 0, 1)  # sendTime <= time
    n.runMs(10)
    with pytest.raises(w.IllegalStateException):
        n.registerTask(7, 5, 0)  # arriving in the past (:249-252)


def test_msg_discard_time():  # C/Network.java:40,481
    n = Network.create()
    n.setNetworkLatency("NetworkFixedLatency(100)")
    n.setMsgDiscardTime(100)
    n.add_nodes([1, 500], [1, 1])
    n.load_protocol(PINGPONG)
    n.send(PONG, 1, 0, 1)
    assert n.msgs.size() == 0 and n.read("msgSent")[0] == 1


@pytest.mark.parametrize("name", ["NetworkLatencyByDistanceWJitter", "IC3NetworkLatency", "NetworkFixedLatency(100)",
                                  "NetworkUniformLatency(200)", "NetworkNoLatency", "EthScanNetworkLatency"])
def test_latency_models_match_oracle(name):
    """the device LUT path of NetworkLatency.getLatency vs the oracle's double-precision evaluation"""
    rng = np.random.RandomState(7)
    k = 4000
    xs = np.concatenate([[1, 1, 1000, 2000, 1, 2000], rng.randint(1, 2001, k)])
    ys = np.concatenate([[1, 1, 556, 1112, 1112, 1], rng.randint(1, 1113, k)])
    ex = np.concatenate([[0, 0, 0, 500, 0, 500], rng.choice([0, 0, 0, 500], k)])
    n = Network.create()
    n.setNetworkLatency(name)
    n.add_nodes(xs, ys, extraLatency=ex)
    frm = rng.randint(0, len(xs), 3 * k).astype(np.int32)
    to = rng.randint(0, len(xs), 3 * k).astype(np.int32)
    frm[:20], to[:20] = np.arange(20) % 6, np.arange(20) % 6  # from == to -> 1 (C/NetworkLatency.java:28-30)
    delta = rng.randint(0, 100, 3 * k).astype(np.int32)
    delta[:200] = np.arange(200) % 100
    got = n.latency_probe(frm, to, delta)
    for i in range(len(frm)):
        f, t = int(frm[i]), int(to[i])
        exp = o.latency(name, int(xs[f]), int(ys[f]), int(ex[f]), int(xs[t]), int(ys[t]), int(ex[t]), int(delta[i]),
                        same=(f == t))
        assert got[i] == exp, (i, f, t, int(delta[i]), got[i], exp)


def test_full_bydistance_lut_matches_oracle():
    """all 1145 x 100 (dist, delta) cells of NetworkLatencyByDistanceWJitter"""
    table = o.latency_table()
    # realise every distance 0..1144 as a pair of nodes: (1,1) and a point at that rounded-down distance
    pts = {}
    for x in range(1, 1002):
        for y in (1, 100, 300, 557):
            d = int(np.floor(np.sqrt((x - 1) ** 2 + (y - 1) ** 2)))
            pts.setdefault(d, (x, y))
    ds = sorted(d for d in pts if d <= 1144)
    assert len(ds) >= 1100
    n = Network.create()
    n.setNetworkLatency("NetworkLatencyByDistanceWJitter")
    n.add_nodes([1] + [pts[d][0] for d in ds], [1] + [pts[d][1] for d in ds])
    frm = np.zeros(len(ds) * 100, np.int32)
    to = np.repeat(np.arange(1, len(ds) + 1), 100).astype(np.int32)
    delta = np.tile(np.arange(100), len(ds)).astype(np.int32)
    got = n.latency_probe(frm, to, delta).reshape(len(ds), 100)
    exp = np.array([np.maximum(1, table[d]) for d in ds])
    assert (got == exp).all()


def test_pingpong_reference_run():  # PT/PingPongTest.java:8-19 + P/PingPong.java:94-101 main()
    g = w.PingPong(w.PingPongParameters())
    g.init()
    c = o.PingPong(1000)
    for _ in range(10):
        g.network().runMs(50)
        c.run_ms(50)
        assert not parity.diff_pingpong(g, c)
    g.network().runMs(9500)
    c.run_ms(9500)
    assert not parity.diff_pingpong(g, c)
    pong = g.network().read("pong")
    assert pong[0] == 1000 and (pong[1:] == 0).all()
    assert g.network().msgs.size() == 0


@pytest.mark.parametrize("chunk", [1, 7, 1000])
def test_pingpong_chunking_and_seeds(chunk):
    for seed in (1, 12345):
        g = w.PingPong(w.PingPongParameters(500, parity.NB, parity.NL), seed=seed)
        g.init()
        c = o.PingPong(500, parity.NB, parity.NL, seed=seed)
        t = 0
        while t < 1000:
            g.network().runMs(chunk)
            c.run_ms(chunk)
            t += chunk
        assert not parity.diff_pingpong(g, c)

