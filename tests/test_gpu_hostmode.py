"""Host-callback mode (WG_PROTO_HOST, wg_next_delivery): queue, ordering, latency and rd in the engine, action()
on the host. The reference's own task tests (CT/NetworkTest.java) restated against wittgenstein_amd.hostnet, and
PingPong (P/PingPong.java) written against that API compared with the CPU oracle run of the same protocol."""
import pytest

import oracle_lib as o
from wittgenstein_amd import hostnet as hn
from wittgenstein_amd.core import IllegalArgumentException

NL = "NetworkLatencyByDistanceWJitter"


class Counter:
    def __init__(self):
        self.v = 0

    def inc(self):
        self.v += 1


def net4(latency="NetworkNoLatency"):  # CT/NetworkTest.java:25-32: four nodes n0..n3
    net = hn.HostNetwork(latency)
    nodes = [hn.Node(net) for _ in range(4)]
    for n in nodes:
        net.addNode(n)
    return net, nodes


@pytest.mark.gpu
def test_register_task():  # CT/NetworkTest.java:57-68
    net, n = net4()
    c = Counter()
    net.registerTask(c.inc, 100, n[0])
    net.runMs(99)
    assert c.v == 0
    net.runMs(1)
    assert c.v == 1


@pytest.mark.gpu
def test_task_and_stopped_node():  # :436-460
    net, n = net4()
    c = Counter()
    net.registerTask(c.inc, 1000, n[0])
    net.runMs(500)
    assert c.v == 0
    net.runMs(500)
    assert c.v == 1
    net.runMs(100)
    net.runMs(5000)
    assert c.v == 1
    net, n = net4()
    c = Counter()
    net.registerTask(c.inc, 1000, n[0])
    net.set_down(n[0])
    net.runMs(5000)
    assert c.v == 0


@pytest.mark.gpu
def test_periodic_task():  # :462-479
    net, n = net4()
    c = Counter()
    net.registerPeriodicTask(c.inc, 1000, 100, n[0])
    net.runMs(500)
    assert c.v == 0
    net.runMs(500)
    assert c.v == 1
    net.runMs(100)
    assert c.v == 2
    net.runMs(50)
    assert c.v == 2
    net.set_down(n[0])
    net.runMs(1000)
    assert c.v == 2


@pytest.mark.gpu
def test_conditional_task():  # :481-506
    net, n = net4()
    c = Counter()
    flag = [False]
    net.registerConditionalTask(c.inc, 1000, 100, n[0], lambda: flag[0], lambda: True)
    net.runMs(500)
    assert c.v == 0
    net.runMs(500)
    assert c.v == 0
    flag[0] = True
    net.runMs(1)
    assert c.v == 1
    net.runMs(99)
    assert c.v == 1
    net.runMs(1)
    assert c.v == 2
    net.set_down(n[0])
    net.runMs(1000)
    assert c.v == 2


class _Rec(hn.Message):
    def __init__(self, log):
        self.log = log

    def action(self, network, frm, to):
        self.log.append((network.time, frm.nodeId, to.nodeId))


@pytest.mark.gpu
def test_multiple_destinations_with_delay_and_lifo():
    """CT/NetworkTest.java:187-207 (delay 10, NoLatency, sendTime 1 -> arrivals 2, 13, 24) and
    CT/EnvelopeStorageTest.java:34-46 (two envelopes of one ms: the one pushed last is delivered first)."""
    net, n = net4()
    log = []
    net.send(_Rec(log), n[0], [n[1], n[2], n[3]], sendTime=1, delayBetween=10)
    net.runMs(30)
    assert [(t, to) for t, _, to in log] == [(2, 1), (13, 2), (24, 3)]
    net, n = net4()
    log = []
    net.send(_Rec(log), n[1], n[0])  # arrives at 2
    net.send(_Rec(log), n[2], n[0])  # arrives at 2, pushed last -> first out
    net.runMs(5)
    assert [f for _, f, _ in log] == [2, 1]
    with pytest.raises(IllegalArgumentException):
        net.sendArriveAt(_Rec(log), net.time, n[0], n[1])  # :385-388


class _PPNode(hn.Node):
    def __init__(self, net):
        super().__init__(net)
        self.pong = 0


class _Ping(hn.Message):  # P/PingPong.java:20-25
    def action(self, network, frm, to):
        network.send(_Pong(), to, frm)


class _Pong(hn.Message):  # :27-32
    def action(self, network, frm, to):
        to.pong += 1


@pytest.mark.gpu
@pytest.mark.parametrize("n", [200])
def test_pingpong_through_host_callbacks_matches_oracle(n):
    """P/PingPong.java:81-87 written against the host-side Network: node construction, the 1 -> n multi-destination
    envelope (a chain whose hops are handed out in the reference's order) and every Pong's send draw from the one
    shared rd — pong counts, all four node counters and the rd state equal the oracle's after every 50 ms."""
    net = hn.HostNetwork(NL)
    nodes = [_PPNode(net) for _ in range(n)]
    for nd in nodes:
        net.addNode(nd)
    net.sendAll(_Ping(), nodes[0])
    c = o.PingPong(n, None, NL)
    for _ in range(6):
        net.runMs(50)
        c.run_ms(50)
        assert [nd.pong for nd in nodes] == c.read("pong").tolist()
        for f in ("msgReceived", "msgSent", "bytesSent", "bytesReceived"):
            assert [getattr(nd, f) for nd in nodes] == c.read(f).tolist(), f
        assert net._eng.rng_state() == c.info()["rng"]
        assert net.time == c.info()["time"]
    assert 0 < nodes[0].pong <= n
    # the binding forgets a Message with its last envelope (wg_host_released): after the run only what is still in flight is held —
    # the sendAll's Ping while it has destinations to reach, Pongs on their way — not the ~ n Pong objects the run made
    assert len(net._handles) == len(net._refs) == len(net._handle_of) <= 1 + net.msgs.size()
    net.runMs(2000)
    assert net.msgs.size() == 0 and not net._handles and net._next_handle <= n + 2  # freed handles are used again


@pytest.mark.gpu
def test_handles_are_released_for_dropped_and_repeated_sends(monkeypatch):
    """wg_host_released covers the ends the deliveries do not show: a send that reaches no destination (every destination down),
    a delivery consumed because the receiver stopped meanwhile (C/Network.java:606), the same Message object under several
    envelopes (one handle, released once per envelope) — on both forms of the boundary"""
    for batched in ("0", "1"):
        monkeypatch.setenv("WG_HOST_BATCH", batched)
        net, n = net4()
        log = []
        m = _Rec(log)
        net.set_down(n[3])
        net.send(m, n[0], n[3])                      # dropped at send time (:478)
        net.send(m, n[0], [n[1], n[2]])              # one envelope, two hops
        net.send(m, n[0], n[1])                      # the same object again: the same handle
        assert len(net._handles) == 1 and list(net._refs.values()) == [3]
        net.runMs(1)
        assert list(net._refs.values()) == [2]       # the dropped one is gone
        net.set_down(n[2])                           # its hop will be consumed, not delivered
        net.runMs(10)
        assert [to for _, _, to in log] == [1, 1] and not net._handles and not net._refs


@pytest.mark.gpu
def test_a_refused_send_gives_its_handle_back():
    """a native call the engine refuses creates no envelope, so wg_host_released never reports one: the binding gives the
    reference back itself — the Message is not pinned and its handle is reused (the Java binding's
    try / catch around its native calls is the same rule)"""
    net, n = net4()
    log = []
    m = _Rec(log)
    net.runMs(5)
    with pytest.raises(hn.IllegalArgumentException):
        net.sendArriveAt(m, net.time, n[0], n[1])       # "wrong arrival time" (C/Network.java:386)
    assert not net._handles and not net._refs and not net._handle_of
    with pytest.raises(Exception):
        net.send(m, n[0], [n[1]], sendTime=net.time)    # sendTime <= time is refused (:371)
    assert not net._handles and not net._refs
    net.send(m, n[0], n[1])                             # ... and the object can still be sent
    assert list(net._refs.values()) == [1]
    net.runMs(2000)
    assert [to for _, _, to in log] == [1] and not net._handles


@pytest.mark.gpu
def test_deferred_init_leaves_rd_where_init_left_it():
    """deferred_init replays the kept sends with rd put back to each send's own state — and must END with rd where init()
    itself ended: an init() that draws AFTER its last send (or only registers tasks) otherwise hands the run a stale stream
    (the Java GpuNetwork.deferredInit did). Compared with the same init() done without the block."""
    def build(deferred):
        net = hn.HostNetwork(None)
        nodes = []
        log = []
        import contextlib
        with (net.deferred_init() if deferred else contextlib.nullcontext()):
            for _ in range(4):
                nodes.append(hn.Node(net))
                net.addNode(nodes[-1])
            if deferred:  # (sends between constructions are what the block is for; without it they come after the nodes)
                pass
            net.send(_Rec(log), nodes[0], nodes[1])
            net.send(_Rec(log), nodes[2], [nodes[1], nodes[3]])
            tail = [net.rd.nextInt() for _ in range(3)]   # init() goes on drawing after its last send
            net.registerTask(lambda: log.append(("task", net.time, -1)), 7, nodes[0])
        return net, tail, log
    a, tail_a, log_a = build(False)
    b, tail_b, log_b = build(True)
    assert tail_a == tail_b and a._eng.rng_state() == b._eng.rng_state()
    for net in (a, b):
        net.runMs(300)
    assert log_a == log_b and a._eng.rng_state() == b._eng.rng_state() and a.rd.nextInt() == b.rd.nextInt()


# ---- the batched form of the same boundary: wg_step_begin / wg_step_end (a ms of deliveries per call, their pushes back in
# one call, include/wittgpu.h) must hand out and file everything exactly as wg_next_delivery does
BATCHED = ["test_register_task", "test_task_and_stopped_node", "test_periodic_task", "test_conditional_task",
           "test_multiple_destinations_with_delay_and_lifo"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", BATCHED)
def test_batched_steps_reference_vectors(name, monkeypatch):
    monkeypatch.setenv("WG_HOST_BATCH", "1")
    globals()[name]()


@pytest.mark.gpu
def test_batched_steps_pingpong_matches_oracle(monkeypatch):
    monkeypatch.setenv("WG_HOST_BATCH", "1")
    test_pingpong_through_host_callbacks_matches_oracle(300)


@pytest.mark.gpu
def test_batched_steps_protocols_match_oracle(monkeypatch):
    """San Fermin (rd draws inside action(): the shuffle), P2PFlood (multi-destination envelopes with delays, an empty
    list's draw), Casper IMD (sendAll, far tasks) and the scheduler fuzz (partitions, stops, discard) on batched steps"""
    monkeypatch.setenv("WG_HOST_BATCH", "1")
    import test_gpu_casper as tc
    import test_gpu_p2pflood as tpf
    import test_gpu_sanfermin as tsf
    import test_gpu_fuzz as tf
    tsf.test_sanfermin_64_matches_oracle()
    tsf.test_sanfermin_fixed_latency_short_timeout()
    tpf.test_p2pflood_three_messages_by_distance()
    tpf.test_empty_destination_list_costs_a_draw()
    tc.lockstep((2, False, 2, 6, 1000, 1), seed=5, chunk=3000, chunks=3)
    tf.test_fuzz_latency_models(None)
    tf.test_fuzz_partitions_stops_and_discard(2)


@pytest.mark.gpu
def test_batched_step_errors_are_loud():
    import ctypes as C
    from wittgenstein_amd import _lib as L
    net = hn.HostNetwork(None, batched=True)
    nodes = [hn.Node(net) for _ in range(4)]
    for n in nodes:
        net.addNode(n)
    net._start()
    lib, h = L.lib(), net._eng._h
    assert lib.wg_step_end(h, None, 0, None) == L.WG_ESTATE  # no step open
    net.registerTask(lambda: None, 5, nodes[0])
    arr = (L.wg_delivery * 8)()
    n = C.c_int32()
    assert lib.wg_step_begin(h, 10, hn.INT_MAX, arr, 8, C.byref(n)) == 0 and n.value == 1 and arr[0].kind == 1
    assert lib.wg_step_begin(h, 10, hn.INT_MAX, arr, 8, C.byref(n)) == L.WG_ESTATE  # the step is still open
    d = L.wg_delivery()
    got = C.c_int32()
    assert lib.wg_next_delivery(h, 10, hn.INT_MAX, C.byref(d), C.byref(got)) == L.WG_ESTATE
    op = (L.wg_step_op * 1)()
    op[0].after, op[0].kind, op[0].msg, op[0].time, op[0].from_ = 3, 2, 99, 7, 1  # a delivery the step did not hand out
    assert lib.wg_step_end(h, op, 1, None) == L.WG_EINVAL
    # a refused step applies NOTHING and stays open (the ms has been handed out: the caller resubmits)
    ops = (L.wg_step_op * 2)()
    ops[0].after, ops[0].kind, ops[0].msg, ops[0].time, ops[0].from_, ops[0].to, ops[0].n = 0, 1, 7, 9, 1, 2, 1  # sendArriveAt: fine
    ops[1].after, ops[1].kind, ops[1].msg, ops[1].time, ops[1].from_, ops[1].to, ops[1].n = 0, 0, 7, 5, 1, 99, 1  # send to node 99
    q0 = C.c_int64()
    assert lib.wg_queue_size(h, C.byref(q0)) == 0
    assert lib.wg_step_end(h, ops, 2, None) == L.WG_EINVAL
    q1 = C.c_int64()
    assert lib.wg_step_begin(h, 10, hn.INT_MAX, arr, 8, C.byref(n)) == L.WG_ESTATE  # still open
    ops[1].to = 3
    ops[1].time = 5  # sendTime == time: IllegalStateException site :471
    assert lib.wg_step_end(h, ops, 2, None) == L.WG_ESTATE
    ops[1].time = 6
    assert lib.wg_step_end(h, ops, 2, None) == 0  # accepted as a whole
    assert lib.wg_queue_size(h, C.byref(q1)) == 0 and q1.value == q0.value + 2


@pytest.mark.gpu
def test_batched_step_is_closed_when_an_action_raises():
    """an exception out of a protocol's action() must not leave the engine in 'step open': the step is closed with the pushes
    made so far (their multi-destination envelopes' re-pushes included) and the exception reaches the caller"""
    net = hn.HostNetwork(None, batched=True)
    nodes = [hn.Node(net) for _ in range(3)]
    for nd in nodes:
        net.addNode(nd)
    net._start()

    class Boom(Exception):
        pass

    ran = []

    def bad():
        ran.append(net.time)
        raise Boom()

    net.registerTask(bad, 5, nodes[0])
    net.registerTask(lambda: ran.append(-net.time), 8, nodes[1])
    with pytest.raises(Boom):
        net.runMs(10)
    assert ran == [5]
    net.runMs(10)  # the engine is usable: the later task still runs
    assert ran == [5, -8]
