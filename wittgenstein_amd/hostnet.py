"""core.Network for protocols whose Message.action() stays on the host (host-callback mode, wg_next_delivery).

The message queue with its LIFO / multi-destination-chain ordering, NetworkLatency sampling and the shared
`rd` live in libwittgpu.so on the MI355X; Node / Message / Task objects and action() stay here — the Python
stand-in for what the reference's Java classes do (C/Network.java, C/Node.java, C/messages/*.java), so that
any protocol written against that API runs unchanged in structure. Method names follow the Java API."""
import ctypes as C
import os

from . import _lib as L
from .core import IllegalArgumentException, IllegalStateException, Network as _EngineNetwork

INT_MAX = 2**31 - 1
_MASK48 = (1 << 48) - 1


def _s32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


class EngineRandom:
    """network.rd (C/Network.java:32): java.util.Random whose 48-bit state is the engine's, so draws made by
    protocol code and by send() interleave exactly as in the reference."""

    def __init__(self, net):
        self._net = net

    def _next(self, bits):
        net = self._net
        if net._rd_held is not None:  # a batched step is open: rd is the caller's until wg_step_end
            s = net._rd_held = (net._rd_held * 0x5DEECE66D + 0xB) & _MASK48
            return _s32(s >> (48 - bits))
        s = (net._eng.rng_state() * 0x5DEECE66D + 0xB) & _MASK48
        net._eng._ck(L.lib().wg_rng_set_state(net._eng._h, C.c_uint64(s)))
        return _s32(s >> (48 - bits))

    def setSeed(self, seed):
        self._net._eng.set_seed(seed)

    def nextInt(self, bound=None):
        if bound is None:
            return self._next(32)
        if bound <= 0:
            raise IllegalArgumentException("bound must be positive")
        r = self._next(31)
        m = bound - 1
        if bound & m == 0:
            return _s32((bound * r) >> 31)
        u = r
        while True:
            r = u % bound
            if _s32(u - r + m) >= 0:
                return r
            u = self._next(31)

    def nextBoolean(self):
        return self._next(1) != 0

    def nextDouble(self):
        return ((self._next(26) << 27) + self._next(27)) * (1.0 / (1 << 53))


class Node:
    """C/Node.java: identity, position (NodeBuilderWithRandomPosition, C/NodeBuilder.java:77-96) and counters."""

    def __init__(self, net):
        r = net.rd.nextInt()
        rx = _s32(r >> 16)
        ry = _s32((r << 16) & 0xFFFFFFFF)
        self.x = abs(rx) % 2000 + 1
        self.y = abs(ry) % 1112 + 1
        self.nodeId = net._allocate_id()
        self.down = False
        self.extraLatency = 0
        self.msgReceived = self.msgSent = self.bytesSent = self.bytesReceived = 0
        self.doneAt = 0

    def isDown(self):
        return self.down

    def start(self):
        self.down = False

    def stop(self):
        self.down = True


class Message:
    """C/messages/Message.java:15-29"""

    def size(self):
        return 1

    def action(self, network, frm, to):
        raise NotImplementedError


class Task(Message):
    """C/messages/Task.java:8-31"""

    def __init__(self, r):
        self.r = r

    def size(self):
        return 0

    def action(self, network, frm, to):
        self.r()


class PeriodicTask(Task):
    """C/messages/PeriodicTask.java:10-47"""

    def __init__(self, r, sender, period, cond=lambda: True):
        super().__init__(r)
        self.sender, self.period, self.cond = sender, period, cond

    def action(self, network, frm, to):
        self.r()
        if self.cond():
            network.sendArriveAt(self, network.time + self.period, self.sender, self.sender)


class ConditionalTask:
    """C/messages/ConditionalTask.java:6-36"""

    def __init__(self, startIf, repeatIf, r, minStartTime, frm, duration):
        self.startIf, self.repeatIf, self.r = startIf, repeatIf, r
        self.minStartTime, self.frm, self.duration = minStartTime, frm, duration


class HostNetwork:
    """C/Network.java over the engine's host-callback mode."""

    def __init__(self, networkLatencyName=None, config=None, batched=None, batch_cap=4096):
        if batched is None:
            batched = os.environ.get("WG_HOST_BATCH", "0") not in ("", "0")
        # batched: a simulated ms travels as ONE wg_step_begin / wg_step_end pair instead of one wg_next_delivery per
        # delivered message (include/wittgpu.h); same results, two FFI crossings per ms
        self._batched, self._batch_cap = bool(batched), int(batch_cap)
        self._rd_held = None    # rd's state while a batched step is open
        self._ops = None        # the step's pushes: (after, kind, handle, payload, time, from, dests, delay, seed)
        self._cur = 0
        self._eng = _EngineNetwork.create(config)
        self._eng.setNetworkLatency(networkLatencyName)
        self.rd = EngineRandom(self)
        self.allNodes = []
        self.conditionalTasks = []
        # handle -> Message / Task object, for as long as an envelope of it is in flight: the engine reports the end of every
        # envelope (wg_host_released) and the object is let go with its last one, as the reference lets go of its Envelope
        self._handles = {}
        self._handle_of = {}   # id(obj) -> handle (a re-armed PeriodicTask, a Message sent twice: one handle)
        self._refs = {}        # handle -> envelopes in flight
        self._free = []
        self._rel = (C.c_uint32 * 1024)()
        self._next_handle = 1
        self._next_id = 0
        self._ready = False
        self._deferred = None  # init()'s sends and tasks while nodes are still being added (deferred_init)
        self.time = 0

    # ---- nodes
    def _allocate_id(self):  # NodeBuilder.allocateNodeId (C/NodeBuilder.java:61-63)
        i = self._next_id
        self._next_id += 1
        return i

    def addNode(self, node):  # :651-659
        if self._ready:
            raise IllegalStateException("nodes must be added before the first send / run")
        if node.nodeId != len(self.allNodes):
            raise IllegalArgumentException("bad node id")
        self.allNodes.append(node)

    def getNodeById(self, i):
        return self.allNodes[i]

    def _start(self):
        if self._ready:
            return
        self._eng.add_nodes([n.x for n in self.allNodes], [n.y for n in self.allNodes],
                            [n.extraLatency for n in self.allNodes], [1 if n.down else 0 for n in self.allNodes])
        self._eng.load_protocol(0)  # WG_PROTO_HOST
        self._ready = True

    def deferred_init(self):
        """`with net.deferred_init():` around an init() that sends BETWEEN node constructions (P/Paxos.java:283-296: every
        ProposerNode starts its first proposal before the next one is built). The engine takes its node table whole, so inside
        the block rd is drawn on the host (the node constructors' draws, the protocol's, the sends' seed draws, in the
        reference's order), the sends and tasks are kept, and at the end — the nodes added — they are issued in their order,
        each send with rd put back to where its seed draw (C/Network.java:377, 430) found it: the same envelopes, the same
        push order, the same rd afterwards as the interleaved original."""
        import contextlib

        @contextlib.contextmanager
        def block():
            if self._ready:
                raise IllegalStateException("deferred_init() after the first send / run")
            self._rd_held = self._eng.rng_state()
            self._deferred = []
            try:
                yield self
                ops, end = self._deferred, self._rd_held
            finally:
                self._deferred, self._rd_held = None, None
            self._start()
            lib, h = L.lib(), self._eng._h
            for op in ops:
                if op[0] == "send":
                    _, before, m, sendTime, frm, ids, delay = op
                    self._eng._ck(lib.wg_rng_set_state(h, C.c_uint64(before)))
                    self._native(m, lambda hm: self._eng.send(hm, sendTime, frm, ids, delay))
                elif op[0] == "arrive":
                    self._native(op[1], lambda hm: self._eng._ck(lib.wg_send_arrive_at(h, hm, 0, op[2], op[3], op[4])))
                else:
                    self._native(op[1], lambda hm: self._eng.registerTask(hm, op[2], op[3]))
            self._eng._ck(lib.wg_rng_set_state(h, C.c_uint64(end)))
        return block()

    def _handle(self, obj):
        """the handle an envelope of `obj` travels under (one more envelope of it in flight)"""
        h = self._handle_of.get(id(obj))
        if h is None:
            if self._free:
                h = self._free.pop()
            else:
                h = self._next_handle
                self._next_handle += 1
            self._handles[h] = obj
            self._handle_of[id(obj)] = h
            self._refs[h] = 0
        self._refs[h] += 1
        return h

    def _native(self, obj, call):
        """call(handle) with a reference taken on `obj`'s handle for the envelope the call creates — given back if the engine
        refuses the call (WG_ENOMEM for chain_slots / chain_dests, "Arriving in the past", ...): no envelope exists then, so
        wg_host_released would never report it and the object would stay pinned for the rest of the run"""
        h = self._handle(obj)
        try:
            return call(h)
        except BaseException:
            self._unref(h)
            raise

    def _unref(self, h):
        r = self._refs[h] - 1
        if r:
            self._refs[h] = r
        else:
            del self._handle_of[id(self._handles.pop(h))]
            del self._refs[h]
            self._free.append(h)

    def _drain_released(self):
        """envelopes that ended since the last call (wg_host_released): their objects are forgotten with their last envelope"""
        lib, h, n = L.lib(), self._eng._h, C.c_int32()
        while True:
            self._eng._ck(lib.wg_host_released(h, self._rel, len(self._rel), C.byref(n)))
            for i in range(n.value):
                self._unref(self._rel[i])
            if n.value < len(self._rel):
                return

    def set_down(self, node, down=True):
        node.down = down
        if self._ready:
            self._eng.set_node_down(node.nodeId, down)

    # ---- sends (C/Network.java:341-390, 418-447)
    def sendAll(self, m, fromNode, sendTime=None):
        self.send(m, fromNode, self.allNodes, self.time + 1 if sendTime is None else sendTime, _force_multi=True)

    def send(self, m, fromNode, dests, sendTime=None, delayBetween=0, _force_multi=False):
        if self._deferred is None:
            self._start()
        if isinstance(dests, Node):
            dests = [dests]
        elif not _force_multi and sendTime is None:
            if not dests:
                return  # the 3-argument overload returns without drawing (:354-356)
        if sendTime is None:
            sendTime = self.time + 1
        ids = [d.nodeId for d in dests]
        # createMessageArrival counts the sender's statistics for every destination, dropped or not (:476-477)
        fromNode.msgSent += len(ids)
        fromNode.bytesSent += len(ids) * m.size()
        if self._deferred is not None:  # (deferred_init: the seed draw is made now, the send itself at the end of the block)
            before = self._rd_held
            self.rd.nextInt()
            self._deferred.append(("send", before, m, sendTime, fromNode.nodeId, ids, delayBetween))
            return
        if self._ops is not None:  # inside a batched step: the seed draw (:377 / :430) is made here, in action() order
            seed = self.rd.nextInt()
            if ids:
                self._ops.append((self._cur, 0, self._handle(m), 0, sendTime, fromNode.nodeId, ids, delayBetween, seed))
            return
        self._native(m, lambda h: self._eng.send(h, sendTime, fromNode.nodeId, ids, delayBetween))

    def sendArriveAt(self, m, arriveAt, fromNode, toNode):
        if self._deferred is not None:
            self._deferred.append(("arrive", m, int(arriveAt), fromNode.nodeId, toNode.nodeId))
            return
        self._start()
        if self._ops is not None:
            if arriveAt <= self.time:
                raise IllegalArgumentException("wrong arrival time: arriveAt=%d, time=%d" % (arriveAt, self.time))
            self._ops.append((self._cur, 1, self._handle(m), 0, int(arriveAt), fromNode.nodeId, [toNode.nodeId], 0, 0))
            return
        self._native(m, lambda h: self._eng._ck(L.lib().wg_send_arrive_at(self._eng._h, h, 0, int(arriveAt), fromNode.nodeId,
                                                                          toNode.nodeId)))

    def _register(self, obj, startAt, fromNode):
        if self._deferred is not None:
            self._deferred.append(("task", obj, int(startAt), fromNode.nodeId))
            return
        self._start()
        if self._ops is not None:
            self._ops.append((self._cur, 2, self._handle(obj), 0, int(startAt), fromNode.nodeId, [], 0, 0))
            return
        self._native(obj, lambda h: self._eng.registerTask(h, startAt, fromNode.nodeId))

    def registerTask(self, task, startAt, fromNode):  # :505-508
        self._register(Task(task), startAt, fromNode)

    def registerPeriodicTask(self, task, startAt, period, fromNode, cond=lambda: True):  # :510-519
        self._register(PeriodicTask(task, fromNode, period, cond), startAt, fromNode)

    def registerConditionalTask(self, task, startAt, duration, fromNode, startIf, repeatIf):  # :521-531
        self.conditionalTasks.append(ConditionalTask(startIf, repeatIf, task, startAt, fromNode, duration))

    # ---- partitions, discard time, queue (C/Network.java:693-707, :103-107, :201-220): straight to the engine
    def partition(self, part):
        self._start()
        self._eng.partition(part)

    def endPartition(self):
        self._eng.endPartition()

    def setMsgDiscardTime(self, ms):
        self._eng.setMsgDiscardTime(ms)

    @property
    def msgs(self):
        self._start()
        return self._eng.msgs

    # ---- the loop
    def run(self, seconds):
        return self.runMs(seconds * 1000)

    def runMs(self, ms):  # :318-338
        if ms <= 0:
            raise IllegalArgumentException("Should be greater than 0. ms=%d" % ms)
        self._start()
        if self.time == 0:
            for n in self.allNodes:
                if not n.isDown():
                    n.start()
        endAt = self.time + ms
        self._drain_released()  # (sends of init() that reached no destination)
        did = self._receiveUntil(endAt)
        self.time = endAt
        self._eng._ck(L.lib().wg_set_time(self._eng._h, endAt))
        return did

    def _cond_time(self, cts, until):
        src = self.conditionalTasks if cts is None else cts
        t = [ct.minStartTime for ct in src if ct.minStartTime <= until and not ct.frm.isDown()]
        return min(t) if t else INT_MAX

    def _receiveUntil(self, until):  # :587-637 with nextMessage :533-570
        if self._batched:
            return self._receiveUntil_batched(until)
        lib, h = L.lib(), self._eng._h
        d = L.wg_delivery()
        got = C.c_int32()
        did = False
        cts = None  # nextMessage()'s private copy of the conditional tasks, made at the first edge of a call
        while True:
            self._eng._ck(lib.wg_next_delivery(h, until, self._cond_time(cts, until), C.byref(d), C.byref(got)))
            if not got.value:
                self._drain_released()  # (envelopes whose last hop was consumed, not delivered: no delivery showed their end)
                return did
            self.time = d.time
            if d.kind == 2:  # time++ edge: the conditional-task scan of :543-566
                cts = self._edge(cts, until)
                continue
            did = True
            cts = None  # a delivery ends the nextMessage() call
            self._deliver(d)
            self._drain_released()

    def _edge(self, cts, until):
        if cts is None:
            cts = list(self.conditionalTasks)
        for ct in list(cts):
            if ct.minStartTime > until or ct.frm.isDown():
                cts.remove(ct)
                continue
            if ct.minStartTime <= self.time:
                cts.remove(ct)
                if ct.startIf():
                    ct.r()
                    ct.minStartTime = self.time + ct.duration
                    if not ct.repeatIf():
                        self.conditionalTasks.remove(ct)
        return cts

    def _deliver(self, d):
        m = self._handles[d.msg]
        frm, to = self.allNodes[d.from_], self.allNodes[d.to]
        if not isinstance(m, Task):  # :607-613 (`!(mc instanceof Task)`)
            if m.size() == 0:
                raise IllegalStateException("Message size should be greater than zero: %r" % m)
            to.msgReceived += 1
            to.bytesReceived += m.size()
        m.action(self, frm, to)

    def _receiveUntil_batched(self, until):
        """the same loop over wg_step_begin / wg_step_end: a ms of deliveries per call, their pushes back in one call"""
        lib, eng = L.lib(), self._eng
        h = eng._h
        cap = self._batch_cap
        arr = (L.wg_delivery * cap)()
        n = C.c_int32()
        did = False
        cts = None
        while True:
            eng._ck(lib.wg_step_begin(h, until, self._cond_time(cts, until), arr, cap, C.byref(n)))
            if not n.value:
                self._drain_released()
                return did
            self._rd_held = eng.rng_state()
            self._ops = []
            failed = None
            try:
                for i in range(n.value):
                    d = arr[i]
                    self._cur = i
                    self.time = d.time
                    if d.kind == 2:
                        cts = self._edge(cts, until)
                        continue
                    did = True
                    cts = None
                    if self.allNodes[d.to].isDown():  # stopped by an earlier action() of this very step (:606)
                        continue
                    self._deliver(d)
            except BaseException as x:  # an action() raised: the step is still closed below, with the pushes made so far
                failed = x
            finally:
                ops, self._ops = self._ops, None
                eng._ck(lib.wg_rng_set_state(h, C.c_uint64(self._rd_held)))
                self._rd_held = None
            oa = (L.wg_step_op * max(1, len(ops)))()
            dests = []
            for k, (after, kind, handle, payload, t, frm, ids, delay, seed) in enumerate(ops):
                o = oa[k]
                o.after, o.kind, o.msg, o.payload, o.time, o.from_ = after, kind, handle, payload, t, frm
                o.n, o.delay, o.seed = len(ids), delay, seed
                if len(ids) == 1:
                    o.to = ids[0]
                elif ids:
                    o.to = len(dests)
                    dests.extend(ids)
            da = (C.c_int32 * max(1, len(dests)))(*dests)
            rc = lib.wg_step_end(h, oa, len(ops), da)
            if rc != L.WG_OK:
                # wg_step_end refuses a step as a whole and leaves it open (nothing applied): close it without the
                # actions' pushes — the multi-destination envelopes' owed re-pushes are kept — and report the refusal
                msg = lib.wg_last_error(h).decode()
                lib.wg_step_end(h, oa, 0, da)
                for op in ops:  # (their envelopes were never made)
                    self._unref(op[2])
                self._drain_released()
                if failed is None:
                    from .core import _raise
                    _raise(rc, msg)
            self._drain_released()
            if failed is not None:
                raise failed
