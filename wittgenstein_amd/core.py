"""Host-side mirror of the reference's `core` package surface for the scheduler path:
Network (C/Network.java), the exceptions it throws, node read-back.

Method names follow the Java API (runMs, run, time, msgs.size()) so tests read like the
reference's own; everything executes in libwittgpu.so on the MI355X.
"""
import ctypes as C

import numpy as np

from . import _lib as L


class IllegalArgumentException(ValueError):
    """java.lang.IllegalArgumentException sites of C/Network.java (:320,371,374,386,427,695)."""


class IllegalStateException(RuntimeError):
    """java.lang.IllegalStateException sites of C/Network.java (:137,250,333,472,599,609,656,671)."""


class EngineCapacityError(MemoryError):
    """A device ring / pool configured through wg_config overflowed (WG_ENOMEM)."""


class HipError(RuntimeError):
    """HIP runtime failure or no device (WG_EHIP)."""


class UnsupportedError(NotImplementedError):
    """A reference feature the resident device protocol does not cover (WG_EUNSUPPORTED)."""


_EXC = {L.WG_EINVAL: IllegalArgumentException, L.WG_ESTATE: IllegalStateException, L.WG_ENOMEM: EngineCapacityError,
        L.WG_EHIP: HipError, L.WG_EUNSUPPORTED: UnsupportedError, L.WG_EHOSTINIT: UnsupportedError}


def _raise(rc, msg):
    raise _EXC.get(rc, RuntimeError)(msg)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


FIELDS = {"doneAt": 0, "msgReceived": 1, "msgSent": 2, "bytesSent": 3, "bytesReceived": 4, "down": 5, "x": 6, "y": 7,
          "extraLatency": 8, "pong": 16, "sigsChecked": 32, "sigQueueSize": 33, "msgFiltered": 34,
          "currWindowSize": 35, "addedCycle": 36, "startAt": 37, "nodePairingTime": 38,
          "sigChecked": 48, "gsfSigQueueSize": 49, "toVerifySize": 50, "verifiedCardinality": 51,
          "aggValue": 64, "currentPrefixLength": 65, "sfFlags": 66, "sentRequests": 67, "receivedRequests": 68,
          "thresholdAt": 69, "headHeight": 80, "headProposalTime": 81, "headId": 82, "attestationsByHeadSize": 83,
          "blocksReceived": 84, "attestationsHeld": 85, "floodReceived": 96, "peerCount": 97}
LEVEL_FIELDS = {"posInLevel": 0, "outgoingFinished": 1, "queueLen": 2, "remainingCalls": 3, "suicideBizAfter": 5}
BITS = {"totalIncoming": 0, "lastAggVerified": 1, "verifiedIndSignatures": 2, "toVerifyInd": 3, "finishedPeers": 4,
        "blacklist": 5, "verifiedSignatures": 8, "individualSignatures": 9, "indivVerifiedSig": 10}


class MessageStorage:
    """Network.msgs (C/Network.java:201-299): size() and sizeAt(t) only — the queue lives in HBM."""

    def __init__(self, net):
        self._net = net

    def size(self):
        v = C.c_int64()
        self._net._ck(L.lib().wg_queue_size(self._net._h, C.byref(v)))
        return v.value

    def sizeAt(self, t):
        v = C.c_int64()
        self._net._ck(L.lib().wg_queue_size_at(self._net._h, int(t), C.byref(v)))
        return v.value


class Network:
    """core.Network over a wg_engine handle."""

    def __init__(self, handle):
        self._h = handle
        self._owner = L.lib()  # the library that created the handle destroys it
        self.msgs = MessageStorage(self)
        self.last_stats = None

    # ---- raw construction: what a Java init() does through the JNI shim ------------------------
    @classmethod
    def create(cls, config=None):
        """new Network<>() (C/Network.java:14-49)."""
        cfg = L.make_config(config)
        h = C.c_void_p()
        rc = L.lib().wg_create(C.byref(cfg), C.byref(h))
        if rc != L.WG_OK:
            _raise(rc, L.lib().wg_last_error(None).decode())
        return cls(h)

    def add_nodes(self, x, y, extraLatency=None, down=None, speedRatio=None):
        """Network.addNode for len(x) nodes with dense ids (C/Network.java:651-659)."""
        x = np.ascontiguousarray(x, np.int32)
        y = np.ascontiguousarray(y, np.int32)
        ex = None if extraLatency is None else np.ascontiguousarray(extraLatency, np.int32)
        dn = None if down is None else np.ascontiguousarray(down, np.uint8)
        sp = None if speedRatio is None else np.ascontiguousarray(speedRatio, np.float64)
        self._ck(L.lib().wg_add_nodes(self._h, len(x), _p(x, C.c_int32), _p(y, C.c_int32),
                                      None if ex is None else _p(ex, C.c_int32),
                                      None if dn is None else _p(dn, C.c_uint8), None,
                                      None if sp is None else _p(sp, C.c_double)))

    def setNetworkLatency(self, name):
        """Network.setNetworkLatency(RegistryNetworkLatencies.getByName(name))."""
        self._ck(L.lib().wg_set_latency_by_name(self._h, name.encode() if name else None))

    def setMeasuredLatency(self, longDistrib):
        arr = np.ascontiguousarray(longDistrib, np.int32)
        self._ck(L.lib().wg_set_latency(self._h, 4, _p(arr, C.c_int32), len(arr)))

    def setCityLatency(self, mode, city_of_node, tab=None, ping=None, jitter=None):
        """wg_set_latency_city: AwsRegionNetworkLatency (mode 0), NetworkLatencyByCity (1), NetworkLatencyByCityWJitter (2)
        over the caller's city list — the tables of wittgenstein_amd.geo (C/NetworkLatency.java:86-233)."""
        c = np.ascontiguousarray(city_of_node, np.int32)
        nc = (np.asarray(tab if tab is not None else ping)).shape[0]
        t = None if tab is None else np.ascontiguousarray(tab, np.int32)
        pg = None if ping is None else np.ascontiguousarray(ping, np.float32)
        j = None if jitter is None else np.ascontiguousarray(jitter, np.float64)
        self._ck(L.lib().wg_set_latency_city(self._h, int(mode), int(nc), _p(c, C.c_int32),
                                             None if t is None else _p(t, C.c_int32),
                                             None if pg is None else _p(pg, C.c_float),
                                             None if j is None else _p(j, C.c_double)))

    def setMsgDiscardTime(self, ms):
        self._ck(L.lib().wg_set_discard_time(self._h, int(ms)))

    def load_protocol(self, proto_id):
        self._ck(L.lib().wg_protocol_load(self._h, proto_id, None, None))

    def set_seed(self, seed):
        self._ck(L.lib().wg_rng_set_seed(self._h, C.c_int64(seed)))

    def send(self, msg, sendTime, frm, dests, delayBetween=0, payload=0):
        """Network.send(m, sendTime, from, dests[, delaysBetweenMessage]) (C/Network.java:369-382,418-447)."""
        d = np.ascontiguousarray(np.atleast_1d(dests), np.int32)
        self._ck(L.lib().wg_send(self._h, int(msg), int(payload), int(sendTime), int(frm), _p(d, C.c_int32), len(d),
                                 int(delayBetween)))

    def registerTask(self, task, startAt, node, arg=0):
        self._ck(L.lib().wg_register_task(self._h, int(task), int(arg), int(startAt), int(node)))

    def registerPeriodicTask(self, task, startAt, period, node):
        self._ck(L.lib().wg_register_periodic_task(self._h, int(task), int(startAt), int(period), int(node)))

    def set_node_down(self, node, down=True):
        """Node.stop() / Node.start() (C/Node.java:120-131)."""
        self._ck(L.lib().wg_set_node_down(self._h, int(node), int(bool(down))))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._owner.wg_destroy(h)

    def _ck(self, rc):
        if rc != L.WG_OK:
            _raise(rc, L.lib().wg_last_error(self._h).decode())

    @property
    def time(self):
        t = C.c_int32()
        self._ck(L.lib().wg_time(self._h, C.byref(t)))
        return t.value

    @property
    def node_count(self):
        return L.lib().wg_node_count(self._h)

    def runMs(self, ms):
        """Network.runMs (C/Network.java:318-338); returns didSomething."""
        did = C.c_uint8()
        st = L.wg_run_stats()
        self._ck(L.lib().wg_run_ms(self._h, int(ms), C.byref(did), C.byref(st)))
        self.last_stats = {n: getattr(st, n) for n, _ in L.wg_run_stats._fields_}
        return bool(did.value)

    def run(self, seconds):
        return self.runMs(seconds * 1000)

    def partition(self, part):
        """Network.partition (C/Network.java:693-703): cut at (int)(MAX_X * part)."""
        if part <= 0 or part >= 1:
            raise IllegalArgumentException("part needs to be a percentage between 0 & 100 excluded")
        cuts = getattr(self, "_cuts", [])
        x = int(np.float32(2000) * np.float32(part))
        if x in cuts:
            raise IllegalArgumentException("this partition exists already")
        cuts = sorted(cuts + [x])
        arr = (C.c_int32 * len(cuts))(*cuts)
        self._ck(L.lib().wg_set_partitions(self._h, arr, len(cuts)))
        self._cuts = cuts

    def endPartition(self):
        self._ck(L.lib().wg_set_partitions(self._h, None, 0))
        self._cuts = []

    def rng_state(self):
        s = C.c_uint64()
        self._ck(L.lib().wg_rng_get_state(self._h, C.byref(s)))
        return s.value

    def read(self, field):
        n = self.node_count
        out = np.zeros(n, np.int64)
        self._ck(L.lib().wg_read_i64(self._h, FIELDS[field], _p(out, C.c_int64), n))
        return out

    def levels(self):
        v = C.c_int32()
        self._ck(L.lib().wg_levels(self._h, C.byref(v)))
        return v.value

    def read_level(self, field):
        n, l = self.node_count, self.levels()
        out = np.zeros((n, l), np.int32)
        self._ck(L.lib().wg_read_level_i32(self._h, LEVEL_FIELDS[field], _p(out, C.c_int32), n, l))
        return out

    def read_ranks(self):
        """HNode.receptionRanks of every node (P/Handel.java:285), [node][sender]"""
        n = self.node_count
        out = np.zeros((n, n), np.int32)
        self._ck(L.lib().wg_read_level_i32(self._h, 4, _p(out, C.c_int32), n, n))
        return out

    def read_bits(self, field):
        n = self.node_count
        w = max(1, n // 64)
        out = np.zeros((n, w), np.uint64)
        self._ck(L.lib().wg_read_bits(self._h, BITS[field], _p(out, C.c_uint64), n, w))
        return out

    def device_bytes(self):
        """device memory held by the engine and its resident protocol (a sharded engine: this shard's)"""
        v = C.c_int64()
        self._ck(L.lib().wg_device_bytes(self._h, C.byref(v)))
        return v.value

    def delivered_by_level(self):
        out = np.zeros(32, np.int64)
        self._ck(L.lib().wg_delivered_by_level(self._h, _p(out, C.c_int64)))
        return out

    def snapshot(self):
        """wg_snapshot: keep the engine as init() left it (device-resident image); valid before the first event.
        restore() then replaces RunMultipleTimes' `p.copy(); rd.setSeed(i); init()` for the same seed
        (C/RunMultipleTimes.java:44-48). Returns the image size in bytes."""
        self._ck(L.lib().wg_snapshot(self._h))
        self._snap_cuts = list(getattr(self, "_cuts", []))
        v = C.c_int64()
        self._ck(L.lib().wg_snapshot_bytes(self._h, C.byref(v)))
        return v.value

    def restore(self):
        """wg_restore: back to the image of snapshot()."""
        self._ck(L.lib().wg_restore(self._h))
        self.last_stats = None
        self._cuts = getattr(self, "_snap_cuts", getattr(self, "_cuts", []))

    def profile(self, mode=1):
        """HIP events on the engine's stream around the kernels of the per-ms pipeline: 0 off, 1 every phase,
        2 the delivery kernel only. Resets the accumulated spans."""
        self._ck(L.lib().wg_profile_enable(self._h, int(mode)))

    def profile_read(self):
        arr = (L.wg_profile_entry * 64)()
        n = C.c_int32()
        self._ck(L.lib().wg_profile_read(self._h, arr, 64, C.byref(n)))
        return {arr[i].name.decode(): {"spans": arr[i].spans, "total_ns": arr[i].total_ns} for i in range(n.value)}

    def profile_reference(self, ref=None):
        """wg_profile_set_reference: keep the spans' start / end relative to `ref`'s (default: this engine's) reference event"""
        self._ck(L.lib().wg_profile_set_reference(self._h, None if ref is None else ref._h))

    def profile_spans(self, cls=2):
        """(start_ns, end_ns) arrays of the spans of phase `cls` (2 = deliver) since the last read, on the reference's time axis"""
        n = C.c_int32()
        self._ck(L.lib().wg_profile_read_spans(self._h, int(cls), None, None, 0, C.byref(n)))
        a, b = np.zeros(n.value, np.float64), np.zeros(n.value, np.float64)
        if n.value:
            self._ck(L.lib().wg_profile_read_spans(self._h, int(cls), _p(a, C.c_double), _p(b, C.c_double), n.value, C.byref(n)))
        return a, b

    def latency_probe(self, frm, to, delta):
        frm = np.ascontiguousarray(frm, np.int32)
        to = np.ascontiguousarray(to, np.int32)
        delta = np.ascontiguousarray(delta, np.int32)
        out = np.zeros(len(frm), np.int32)
        self._ck(L.lib().wg_latency_probe(self._h, len(frm), _p(frm, C.c_int32), _p(to, C.c_int32),
                                          _p(delta, C.c_int32), _p(out, C.c_int32)))
        return out


class Batch:
    """A batch of independent simulations advanced in lock-step on one MI355X (wg_batch_*): the device
    form of RunMultipleTimes' loop over copies (C/RunMultipleTimes.java:44-64)."""

    def __init__(self, networks):
        self.networks = list(networks)
        n = len(self.networks)
        arr = (C.c_void_p * n)(*[net._h for net in self.networks])
        h = C.c_void_p()
        rc = L.lib().wg_batch_create(arr, n, C.byref(h))
        if rc != L.WG_OK:
            _raise(rc, L.lib().wg_batch_last_error(None).decode())
        self._h = h
        self._owner = L.lib()
        self.last_stats = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._owner.wg_batch_destroy(h)

    def _ck(self, rc):
        if rc != L.WG_OK:
            _raise(rc, L.lib().wg_batch_last_error(self._h).decode())

    def runMs(self, ms, active=None):
        """Network.runMs(ms) on every member with active[i] (None = all); returns didSomething per member."""
        n = len(self.networks)
        did = (C.c_uint8 * n)()
        st = (L.wg_run_stats * n)()
        act = None if active is None else (C.c_uint8 * n)(*[1 if a else 0 for a in active])
        self._ck(L.lib().wg_batch_run_ms(self._h, int(ms), act, did, st))
        self.last_stats = [{f: getattr(st[i], f) for f, _ in L.wg_run_stats._fields_} for i in range(n)]
        for net, s in zip(self.networks, self.last_stats):
            net.last_stats = s
        return [bool(d) for d in did]

    def cont_if(self):
        n = len(self.networks)
        out = (C.c_int32 * n)()
        self._ck(L.lib().wg_batch_cont_if(self._h, out))
        return [bool(v) for v in out]

    def run_multiple_times(self, chunk=10, maxTime=0, on_device=True):
        """RunMultipleTimes.run's inner loop (C/RunMultipleTimes.java:50-64) for all members at once:
            do { didSomething = runMs(10); }
            while ((maxTime == 0 || time < maxTime) && (!didSomething || contIf.test(c)));
        A member whose loop ended is no longer advanced. Returns per-member (delivered, simulated_ms).
        on_device: the loop condition is evaluated by the engine (wg_batch_run_multiple_times), chunks are
        enqueued back to back; False drives it from here with one wg_batch_run_ms per chunk."""
        n = len(self.networks)
        if on_device:
            dl = (C.c_int64 * n)()
            ms = (C.c_int64 * n)()
            self._ck(L.lib().wg_batch_run_multiple_times(self._h, int(chunk), int(maxTime), dl, ms))
            return [int(v) for v in dl], [int(v) for v in ms]
        delivered, sim_ms = [0] * n, [0] * n
        active = [True] * n
        while any(active):
            did = self.runMs(chunk, active)
            cont = self.cont_if()
            for i in range(n):
                if active[i]:
                    delivered[i] += self.last_stats[i]["delivered"]
                    sim_ms[i] += chunk
                    active[i] = (maxTime == 0 or self.networks[i].time < maxTime) and (not did[i] or cont[i])
        return delivered, sim_ms
