"""Node-range sharding of ONE simulation over the GPUs of a box: one process per GPU, `torch.distributed` over
RCCL ("nccl" on ROCm) carrying the two per-ms sums the engine asks for (include/wittgpu.h, "node-range sharding";
DESIGN.md §7). The reference has no counterpart (it is single-threaded, C/Network.java:7-11); what is kept is its
result: every shard-count gives the same pong counts / counters / rd state as the unsharded engine.

    cfg = shards.config(dist)                   # rank / world size of the default process group
    p = PingPong(PingPongParameters(1000), config=cfg); p.init()
    p.network().runMs(50)
    pong = shards.gather(dist, p.network().read("pong"))   # whole-network view (SUM of the per-shard views)
"""
import ctypes as C

import numpy as np

from . import _lib as L


class _DeviceWords:
    """int32 words at a raw device address, for torch.as_tensor (zero copy through __cuda_array_interface__)"""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<i4", "data": (int(ptr), False),
                                         "version": 3, "strides": None}


def make_allreduce(dist, group=None, device_memory=True):
    """wg_allreduce_fn over a torch.distributed process group. device_memory=False is for tests/emu only (the
    kernel sources on the CPU wave emulator keep "device" buffers in host memory; backend gloo)."""
    import torch

    def fn(_ctx, buf, count):
        try:
            if device_memory:
                t = torch.as_tensor(_DeviceWords(buf, count), device="cuda")
                dist.all_reduce(t, group=group)
                torch.cuda.synchronize()
            else:
                a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_int32)), shape=(int(count),))
                t = torch.from_numpy(a)
                dist.all_reduce(t, group=group)
            return 0
        except Exception:  # an exception must not unwind through the C frames of the engine
            import traceback
            traceback.print_exc()
            return 1

    return L.ALLREDUCE_FN(fn)


def make_alltoallv(dist, group=None, device_memory=True):
    """wg_alltoallv_fn over a torch.distributed process group (all_to_all_single with split sizes): what carries the rows only
    ONE shard needs — Handel's dissemination snapshots to the shard that owns the receiver (include/wittgpu.h)."""
    import torch

    def fn(_ctx, sendbuf, sc, so, recvbuf, rc, ro):
        try:
            k = dist.get_world_size(group)
            sc_, so_, rc_, ro_ = ([int(a[i]) for i in range(k)] for a in (sc, so, rc, ro))

            def words(ptr, off, cnt):
                if device_memory:
                    return torch.as_tensor(_DeviceWords(ptr + 4 * off, cnt), device="cuda") if cnt else torch.empty(0, dtype=torch.int32, device="cuda")
                if not cnt:
                    return torch.empty(0, dtype=torch.int32)
                return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr + 4 * off, C.POINTER(C.c_int32)), shape=(cnt,)))
            # (the regions of the send buffer are not back to back: gather them, exchange, scatter what arrived)
            send = torch.cat([words(sendbuf, so_[d], sc_[d]) for d in range(k)])
            recv = torch.empty(sum(rc_), dtype=torch.int32, device=send.device)
            dist.all_to_all_single(recv, send, output_split_sizes=rc_, input_split_sizes=sc_, group=group)
            at = 0
            for r in range(k):
                if rc_[r]:
                    words(recvbuf, ro_[r], rc_[r]).copy_(recv[at:at + rc_[r]])
                at += rc_[r]
            if device_memory:
                torch.cuda.synchronize()
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    return L.ALLTOALLV_FN(fn)


_KEEP = []  # the ctypes thunks must outlive every engine that holds their address


def config(dist, group=None, device_memory=True, **capacities):
    """wg_config fields (a dict for the `config=` argument of the protocol mirrors / Network.create) that make
    the engine shard `rank` of `world_size`."""
    fn = make_allreduce(dist, group, device_memory)
    a2a = make_alltoallv(dist, group, device_memory)
    _KEEP.extend([fn, a2a])
    cfg = dict(capacities)
    cfg.update(shard=dist.get_rank(group), nshards=dist.get_world_size(group),
               allreduce=C.cast(fn, C.c_void_p).value, alltoallv=C.cast(a2a, C.c_void_p).value)
    return cfg


def rccl_unique_id():
    """wg_rccl_unique_id: 128 opaque bytes from which every shard's engine builds the shared RCCL communicator; taken by
    shard 0's process and handed to the others (here: one torch.distributed object broadcast)."""
    buf = (C.c_uint8 * 128)()
    rc = L.lib().wg_rccl_unique_id(buf)
    if rc != L.WG_OK:
        raise RuntimeError("wg_rccl_unique_id: %s" % L.lib().wg_last_error(None).decode())
    return bytes(buf)


def config_rccl(dist=None, rank=0, world=1, group=None, **capacities):
    """wg_config fields for a shard whose per-ms sums travel over the ENGINE'S OWN RCCL communicator
    (wg_shard_configure_rccl): no callback, no host synchronisation per collective. With a torch.distributed process
    group the unique id is broadcast from rank 0; without one (`dist=None`) this is a single-process group of `world`
    = 1 (tests, `bench.py --mode shard --gpus 1`)."""
    if dist is not None:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        uid = box[0]
    else:
        uid = rccl_unique_id()
    cfg = dict(capacities)
    cfg.update(shard=rank, nshards=world, rccl_id=uid)
    return cfg


def shard_range(net):
    lo, hi = C.c_int32(), C.c_int32()
    net._ck(L.lib().wg_shard_info(net._h, C.byref(lo), C.byref(hi), None, None))
    return lo.value, hi.value


def traffic(net):
    """(all-reduce calls, int32 words summed) so far"""
    c, w = C.c_int64(), C.c_int64()
    net._ck(L.lib().wg_shard_info(net._h, None, None, C.byref(c), C.byref(w)))
    return c.value, w.value


TRAFFIC_KINDS = ("events", "outbox", "envelopes", "snapshots", "candidates", "directed_counts", "other")


def traffic_by_exchange(net):
    """{exchange: (calls, int32 words)} so far (wg_shard_traffic): the per-ms exchanges of DESIGN.md §7.2, one by one"""
    c, w = (C.c_int64 * 8)(), (C.c_int64 * 8)()
    net._ck(L.lib().wg_shard_traffic(net._h, c, w))
    return {k: (c[i], w[i]) for i, k in enumerate(TRAFFIC_KINDS)}


def _host_device(dist, group=None):
    """where a host-side value has to live to be reduced over this group: RCCL ("nccl") only moves device tensors,
    gloo only host tensors"""
    import torch
    try:
        nccl = "nccl" in str(dist.get_backend(group)).lower()
    except Exception:  # LoopbackGroup and other in-process stand-ins: host tensors
        nccl = False
    return torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")


def gather(dist, per_shard, group=None, device=None):
    """whole-network view of a per-node read-back: a shard reports its own nodes and zeros for the others"""
    import torch
    t = torch.as_tensor(np.ascontiguousarray(per_shard)).to(device or _host_device(dist, group))
    dist.all_reduce(t, group=group)
    return t.cpu().numpy()


def cont_if(dist, protocol, group=None):
    """the RunMultipleTimes continuation predicate of a sharded protocol: a shard evaluates it over its own nodes
    (Handel.newContIf, P/Handel.java:1044-1053, is an OR over live nodes)"""
    import torch
    t = torch.tensor([1 if protocol.cont_if() else 0], dtype=torch.int32, device=_host_device(dist, group))
    dist.all_reduce(t, group=group)
    return bool(t.item())


def run_multiple_times(dist, protocol, chunk=10, maxTime=0, group=None):
    """RunMultipleTimes.run's inner loop (C/RunMultipleTimes.java:50-64) for ONE sharded copy:
        do { didSomething = runMs(chunk); } while ((maxTime == 0 || time < maxTime) && (!didSomething || contIf(p)));
    Returns (delivered messages, simulated ms) of the whole network — both are replicated on every shard."""
    net = protocol.network()
    delivered = sim_ms = 0
    while True:
        did = net.runMs(chunk)
        delivered += net.last_stats["delivered"]
        sim_ms += chunk
        if not ((maxTime == 0 or net.time < maxTime) and (not did or cont_if(dist, protocol, group))):
            return delivered, sim_ms


class WholeNetwork:
    """Whole-network read-back of a sharded Network: every per-node view is this shard's own rows plus, through a
    SUM across shards, everybody else's. Replicated quantities (time, rd state, queue sizes) pass through."""

    def __init__(self, dist, net, group=None):
        self._dist, self._net, self._group = dist, net, group
        self.lo, self.hi = shard_range(net)
        self.msgs = net.msgs

    def _own(self, a):
        a = np.array(a, copy=True)
        a[:self.lo] = 0
        a[self.hi:] = 0
        if a.dtype == np.uint64:  # (torch has no unsigned 64-bit all-reduce)
            return gather(self._dist, a.view(np.int64), self._group).view(np.uint64)
        return gather(self._dist, a, self._group)

    time = property(lambda self: self._net.time)
    node_count = property(lambda self: self._net.node_count)
    last_stats = property(lambda self: self._net.last_stats)

    def rng_state(self):
        return self._net.rng_state()

    def runMs(self, ms):
        return self._net.runMs(ms)

    def levels(self):
        return self._net.levels()

    def read(self, field):
        return self._own(self._net.read(field))

    def read_level(self, field):
        return self._own(self._net.read_level(field))

    def read_bits(self, field):
        return self._own(self._net.read_bits(field))

    def delivered_by_level(self):
        return self._net.delivered_by_level()


class LoopbackGroup:
    """k shards of one simulation on ONE device in one process: every shard is its own wg_engine (own stream, own
    state for its node range), the shards' runMs calls run on k host threads and meet in the all-reduce, which sums the
    k buffers in place on the device. What a one-GPU box can measure of the sharded pipeline (exchange volumes, the
    owner split, shard-count invariance on the hardware); xGMI and RCCL are not involved.

        grp = LoopbackGroup(4)
        sims = [Handel(params, seed=0, config=grp.config(s)) for s in range(4)]; [g.init() for g in sims]
        grp.run(lambda s: sims[s].network().runMs(10))
    """

    def __init__(self, k, device_memory=True):
        import threading
        self.k, self.device_memory = k, device_memory
        self._barrier = threading.Barrier(k)
        self._bufs = [None] * k
        self._thunks = []

    def _allreduce(self, shard, buf, count):
        import threading
        try:
            self._bufs[shard] = (buf, int(count))
            leader = self._barrier.wait() == 0
            if leader:
                if len({c for _, c in self._bufs}) != 1:
                    raise RuntimeError("shards disagree on the size of a collective: %r" % [c for _, c in self._bufs])
                if self.device_memory:
                    import torch
                    ts = [torch.as_tensor(_DeviceWords(b, c), device="cuda") for b, c in self._bufs]
                    total = torch.stack(ts).sum(0, dtype=torch.int32)
                    for t in ts:
                        t.copy_(total)
                    torch.cuda.synchronize()
                else:
                    arrs = [np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_int32)), shape=(c,)) for b, c in self._bufs]
                    total = np.sum(arrs, axis=0, dtype=np.int32)
                    for a in arrs:
                        a[:] = total
            self._barrier.wait()
            return 0
        except threading.BrokenBarrierError:
            return 2
        except Exception:
            import traceback
            traceback.print_exc()
            self._barrier.abort()
            return 1

    def _alltoallv(self, shard, sendbuf, sc, so, recvbuf, rc, ro):
        """every shard's regions for shard d copied into d's receive buffer (device-to-device on the one GPU)"""
        import threading
        try:
            k = self.k
            self._bufs[shard] = (sendbuf, [int(sc[i]) for i in range(k)], [int(so[i]) for i in range(k)], recvbuf,
                                 [int(rc[i]) for i in range(k)], [int(ro[i]) for i in range(k)])
            leader = self._barrier.wait() == 0
            if leader:
                for src in range(k):
                    sb, scs, sos = self._bufs[src][0], self._bufs[src][1], self._bufs[src][2]
                    for dst in range(k):
                        n = scs[dst]
                        rb, rcs, ros = self._bufs[dst][3], self._bufs[dst][4], self._bufs[dst][5]
                        if n != rcs[src]:
                            raise RuntimeError("shards %d -> %d disagree on an all-to-all count: %d sent, %d expected" % (src, dst, n, rcs[src]))
                        if not n:
                            continue
                        if self.device_memory:
                            import torch
                            torch.as_tensor(_DeviceWords(rb + 4 * ros[src], n), device="cuda").copy_(
                                torch.as_tensor(_DeviceWords(sb + 4 * sos[dst], n), device="cuda"))
                        else:
                            C.memmove(rb + 4 * ros[src], sb + 4 * sos[dst], 4 * n)
                if self.device_memory:
                    import torch
                    torch.cuda.synchronize()
            self._barrier.wait()
            return 0
        except threading.BrokenBarrierError:
            return 2
        except Exception:
            import traceback
            traceback.print_exc()
            self._barrier.abort()
            return 1

    def config(self, shard, alltoall=True, **capacities):
        """wg_config fields that make an engine shard `shard` of this group (alltoall=False: the all-reduce alone, the form of
        rounds 1-4 — every snapshot row to every shard)"""
        fn = L.ALLREDUCE_FN(lambda _ctx, buf, count, s=shard: self._allreduce(s, buf, count))
        self._thunks.append(fn)
        _KEEP.append(fn)
        cfg = dict(capacities)
        cfg.update(shard=shard, nshards=self.k, allreduce=C.cast(fn, C.c_void_p).value)
        if alltoall:
            a2a = L.ALLTOALLV_FN(lambda _ctx, sb, sc, so, rb, rc, ro, s=shard: self._alltoallv(s, sb, sc, so, rb, rc, ro))
            self._thunks.append(a2a)
            _KEEP.append(a2a)
            cfg["alltoallv"] = C.cast(a2a, C.c_void_p).value
        return cfg

    def run(self, fn):
        """fn(shard) on k threads at once (the shards meet in their collectives); returns the k results"""
        import threading
        out, err = [None] * self.k, [None] * self.k

        def work(s):
            try:
                out[s] = fn(s)
            except BaseException as e:  # noqa: B036 — re-raised below, after every thread has been released
                err[s] = e
                self._barrier.abort()
        ts = [threading.Thread(target=work, args=(s,)) for s in range(self.k)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        self._barrier.reset()
        for e in err:
            if e is not None:
                raise e
        return out

    def gather(self, per_shard_views, nets):
        """whole-network view: every shard's own rows (a shard reports zeros / stale rows for the others)"""
        total = None
        for a, net in zip(per_shard_views, nets):
            lo, hi = shard_range(net)
            a = np.array(a, copy=True)
            a[:lo] = 0
            a[hi:] = 0
            total = a if total is None else total + a
        return total
