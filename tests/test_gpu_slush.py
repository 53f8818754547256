"""protocols.Slush and protocols.Snowflake (P/Slush.java, P/Snowflake.java) on the engine in host-callback mode vs the CPU oracle
(oracle/slush.hpp, pinned against PT/SlushTest and PT/SnowflakeTest in tests/test_oracle_protocols.py). Compared before the run
and after every chunk: per node the colour, the query nonce, round / cnt, the queries still waiting for answers, the four Node
counters, the position; network.time, msgs.size(), the rd state (the remotes of every query are drawn from it)."""
import numpy as np
import pytest

import oracle_lib as o
from examples.hostmode import slush as sl

GET = {"msgReceived": lambda n: n.msgReceived, "msgSent": lambda n: n.msgSent, "bytesSent": lambda n: n.bytesSent,
       "bytesReceived": lambda n: n.bytesReceived, "myColor": lambda n: n.myColor, "myQueryNonce": lambda n: n.myQueryNonce,
       "round": lambda n: n.round, "cnt": lambda n: getattr(n, "cnt", 0), "answersInProgress": lambda n: len(n.answerIP),
       "x": lambda n: n.x, "y": lambda n: n.y}


def lockstep(snow, params, nl, seed, chunk, chunks):
    """params in the reference's ctor order: Slush (NODES_AV, M, K, A), Snowflake (nodeAv, M, K, A, B)"""
    g = (sl.Snowflake(sl.SnowflakeParameters(*params, None, nl)) if snow else sl.Slush(sl.SlushParameters(*params, None, nl)))
    g.network.rd.setSeed(seed)
    g.init()
    c = o.Slush(params, None, nl, seed=seed, snowflake=snow)
    for k in range(chunks + 1):
        for f, fn in GET.items():
            a, b = np.array([fn(n) for n in g.network.allNodes], np.int64), c.read(f)
            bad = np.nonzero(a != b)[0]
            assert not len(bad), "t=%d %s: %d nodes differ, first node %d: engine %d oracle %d" % (
                g.network.time, f, len(bad), bad[0], a[bad[0]], b[bad[0]])
        i = c.info()
        assert (g.network.time, g.network._eng.rng_state(), g.network.msgs.size()) == (i["time"], i["rng"], i["queue"])
        if k < chunks:
            g.network.runMs(chunk)
            c.run_ms(chunk)
    return g, c


@pytest.mark.gpu
def test_slush_simple():  # PT/SlushTest.java:14-25 through the engine: every node ends on node 0's colour after M rounds
    g, c = lockstep(False, (100, 7, 7, 4.0 / 7.0), "NetworkLatencyByDistanceWJitter", seed=0, chunk=250, chunks=12)
    nodes = g.network.allNodes
    assert len(nodes) == 100 and all(n.myColor == nodes[0].myColor and n.round == 7 for n in nodes)
    assert g.network.msgs.size() == 0 and c.info()["delivered"] > 10000


@pytest.mark.gpu
def test_snowflake_simple():  # PT/SnowflakeTest.java:14-25
    g, c = lockstep(True, (100, 5, 7, 4.0 / 7.0, 3), "NetworkLatencyByDistanceWJitter", seed=0, chunk=250, chunks=16)
    nodes = g.network.allNodes
    assert all(n.myColor == nodes[0].myColor for n in nodes) and all(n.cnt == 4 for n in nodes)
    assert g.getDominantColor()[nodes[0].myColor] == 100


@pytest.mark.gpu
@pytest.mark.parametrize("snow,params", [(False, (60, 5, 7, 4.0 / 7.0)), (True, (60, 5, 7, 4.0 / 7.0, 3))])
def test_copies_agree_and_match_the_oracle(snow, params):  # testCopy (SlushTest :27-45, SnowflakeTest :27-47), other seeds
    g1, _ = lockstep(snow, params, "NetworkLatencyByDistanceWJitter", seed=5, chunk=20, chunks=10)
    g2 = g1.copy()
    g2.network.rd.setSeed(5)
    g2.init()
    g2.network.runMs(200)
    for n1, n2 in zip(g1.network.allNodes, g2.network.allNodes):
        assert (n1.myColor, n1.myQueryNonce, n1.round) == (n2.myColor, n2.myQueryNonce, n2.round)


@pytest.mark.gpu
def test_slush_batched_steps_without_latency(monkeypatch):
    """the same through the batched-step calls (wg_step_begin / wg_step_end: one round trip per simulated ms); NetworkNoLatency puts
    every answer of a query into the same ms"""
    monkeypatch.setenv("WG_HOST_BATCH", "1")
    g, c = lockstep(False, (80, 6, 5, 0.6), "NetworkNoLatency", seed=2, chunk=3, chunks=12)
    assert (c.read("round") == 6).all()
