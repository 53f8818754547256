"""protocols.Dfinity (P/Dfinity.java) on the engine in host-callback mode vs the CPU oracle (oracle/dfinity.hpp, pinned against
PT/DfinityTest.testRun in tests/test_oracle_protocols.py). Compared before the run and after every chunk: per node the four Node
counters, the position, the head (height, id, proposal time), the last random beacon, the blocks received and the committee-majority
sets; per attester voteForHeight, the kept proposals, the vote sets; per producer waitForBlockHeight; per beacon node its height,
lastRDSent, rd and exchange sets; network.time, msgs.size(), the rd state (every committee is shuffled with it before a send)."""
import numpy as np
import pytest

import oracle_lib as o
from examples.hostmode import dfinity as df

ALL = {"msgReceived": lambda n: n.msgReceived, "msgSent": lambda n: n.msgSent, "bytesSent": lambda n: n.bytesSent,
       "bytesReceived": lambda n: n.bytesReceived, "x": lambda n: n.x, "y": lambda n: n.y, "headHeight": lambda n: n.head.height,
       "headId": lambda n: n.head.id, "headTime": lambda n: n.head.proposalTime, "lastRandomBeacon": lambda n: n.lastRandomBeacon,
       "blocksReceived": lambda n: len(n.blocksReceivedByBlockId), "majorityBlocks": lambda n: len(n.committeeMajorityBlocks),
       "majorityHeightSum": lambda n: sum(n.committeeMajorityHeight)}
ATT = {"voteForHeight": lambda n: n.voteForHeight, "proposals": lambda n: len(n.proposals), "votes": lambda n: len(n.votes)}
BP = {"waitForBlockHeight": lambda n: n.waitForBlockHeight}
RB = {"rbHeight": lambda n: n.height, "lastRDSent": lambda n: n.lastRDSent, "rbRd": lambda n: n.rd, "exchanged": lambda n: len(n.exchanged)}


def lockstep(params, nl, seed, chunk, chunks):
    """params = DfinityParameters ctor order: (blockProducersCount, attestersCount, attestersPerRound, blockConstructionTime,
    attestationConstructionTime, percentageDeadAttester)"""
    g = df.Dfinity(df.DfinityParameters(*params, None, nl))  # (the observer is drawn here, from the unseeded rd, as in the reference)
    g.network.rd.setSeed(seed)
    g.init()
    c = o.Dfinity(params, None, nl, seed=seed)
    for k in range(chunks + 1):
        nodes = g.network.allNodes
        assert len(nodes) == c.n
        for fields, kind in ((ALL, df.DfinityNode), (ATT, df.AttesterNode), (BP, df.BlockProducerNode), (RB, df.RandomBeaconNode)):
            for f, fn in fields.items():
                a = np.array([fn(n) if isinstance(n, kind) else -2 for n in nodes], np.int64)
                b = c.read(f)
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
def test_dfinity_run():  # PT/DfinityTest.java:10-24 through the engine: the observer's head is at height 3 after run(11)
    g, c = lockstep((10, 10, 10, 1, 1, 0), "NetworkNoLatency", seed=0, chunk=1000, chunks=11)
    assert g.observer.head.height == 3 and all(n.head is g.observer.head for n in g.network.allNodes)


@pytest.mark.gpu
def test_dfinity_rounds_of_committees_with_latency():
    """several attester and producer rounds (50 attesters in committees of 25, 10 producers in rounds of 5), a block construction
    time of 100 ms, real latencies: proposals that arrive before their height's beacon are kept, votes cross, the beacon
    committee's exchanges wait for 2 x roundTime"""
    g, c = lockstep((10, 50, 25, 100, 1, 5), "NetworkLatencyByDistanceWJitter", seed=3, chunk=500, chunks=40)
    assert g.observer.head.height >= 2 and c.info()["delivered"] > 5000


@pytest.mark.gpu
def test_dfinity_batched_steps(monkeypatch):
    """the same through the batched-step calls (wg_step_begin / wg_step_end: one round trip per simulated ms)"""
    monkeypatch.setenv("WG_HOST_BATCH", "1")
    g, c = lockstep((10, 20, 10, 50, 2, 0), "NetworkLatencyByDistanceWJitter", seed=1, chunk=700, chunks=20)
    assert g.observer.head.height >= 2
