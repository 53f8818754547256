"""Differential stress of the scheduler path itself — core.Network's send family, MultipleDestEnvelope /
MultipleDestWithDelayEnvelope chains, LIFO buckets, tasks, periodic and conditional tasks, rd use inside action(),
partitions, Node.stop()/start(), setMsgDiscardTime — through the engine's host-callback mode against the oracle.
The protocol is test infrastructure on both sides (oracle/fuzz.hpp, tests/fuzz_protocol.py): every action() derives
what it does from a per-node hash, so one envelope delivered out of order, one latency off by a millisecond or one rd
draw out of place diverges the hashes for the rest of the run. Compared after every chunk: per node hash, delivery
count, msgReceived / msgSent / bytesSent / bytesReceived; network.time, msgs.size(), the rd state."""
import numpy as np
import pytest

import fuzz_protocol as fz
import oracle_lib as o

GET = {"h": lambda n: n.h, "c": lambda n: n.c, "msgReceived": lambda n: n.msgReceived, "msgSent": lambda n: n.msgSent,
       "bytesSent": lambda n: n.bytesSent, "bytesReceived": lambda n: n.bytesReceived}


def diff(g, c):
    out = []
    for f, fn in GET.items():
        a, b = np.array([fn(n) for n in g.nodes], np.int64), c.read(f)
        bad = np.nonzero(a != b)[0]
        if len(bad):
            out.append("%s: %d nodes differ, first node %d: engine %d oracle %d" % (f, len(bad), bad[0], a[bad[0]], b[bad[0]]))
    i = c.info()
    mine = (g.network.time, g.network._eng.rng_state(), g.network.msgs.size())
    if mine != (i["time"], i["rng"], i["queue"]):
        out.append("time / rd / msgs.size(): engine %r oracle %r" % (mine, i))
    return out


def run(n, ttl, nl, seed, chunk, chunks, ops=(), config=None):
    """ops: {chunk index: [(name, arg), ...]} applied to both sides before that chunk"""
    g = fz.Fuzz(n, ttl, nl, seed=seed, config=config)
    g.init()
    c = o.Fuzz(n, ttl, nl, seed=seed)
    assert not diff(g, c), "after init()"
    ops = dict(ops)
    for k in range(chunks):
        for name, arg in ops.get(k, ()):
            c.op(name, arg)
            if name == "partition":
                g.network.partition(arg / 1000.0)
            elif name == "endPartition":
                g.network.endPartition()
            elif name in ("stop", "start"):
                g.network.set_down(g.node(arg), name == "stop")
            else:
                g.network.setMsgDiscardTime(arg)
        g.network.runMs(chunk)
        c.run_ms(chunk)
        d = diff(g, c)
        assert not d, "chunk %d (t=%d): %s" % (k, g.network.time, d)
    return g, c


@pytest.mark.gpu
@pytest.mark.parametrize("nl", [None, "NetworkNoLatency", "NetworkFixedLatency(3)", "IC3NetworkLatency"])
def test_fuzz_latency_models(nl):
    g, c = run(48, 12, nl, seed=11, chunk=25, chunks=60)
    assert c.info()["delivered"] > 5000 and c.info()["tasks"] > 500


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_partitions_stops_and_discard(seed):
    ops = {10: [("partition", 300)], 14: [("stop", 3), ("stop", 7)], 18: [("endPartition", 0), ("setMsgDiscardTime", 150)],
           24: [("start", 3)], 27: [("setMsgDiscardTime", 1 << 30)], 30: [("partition", 500), ("partition", 200)],
           36: [("endPartition", 0), ("start", 7)]}
    g, c = run(64, 14, None, seed=seed, chunk=30, chunks=50, ops=ops)
    assert c.info()["delivered"] > 500


@pytest.mark.gpu
def test_fuzz_long_chunks_and_single_ms_chunks_agree():
    a, _ = run(32, 10, None, seed=5, chunk=1, chunks=600)
    b, _ = run(32, 10, None, seed=5, chunk=200, chunks=3)
    assert [n.h for n in a.nodes] == [n.h for n in b.nodes]
