"""Oracle-level protocol properties the reference's tests pin (PT/PingPongTest, PT/HandelTest) plus
invariants used as size-independent parity properties on the GPU."""
import numpy as np

import oracle_lib as o


def test_pingpong_all_pongs(oracle):
    p = o.PingPong(1000)  # null names -> RANDOM builder, ByDistanceWJitter (registry defaults)
    p.run_ms(10000)
    pong = p.read("pong")
    assert pong[0] == 1000 and (pong[1:] == 0).all()  # PT/PingPongTest.java:8-19
    assert p.read("msgReceived").sum() == 2000 and p.info()["queue"] == 0


def test_handel_invariants(oracle):
    n, down = 256, 25
    h = o.Handel(n, int(n * 0.9 * 0.99), 4, 50, 10, 20, 10, down)
    while h.cont_if() and h.info(False)["time"] < 20000:
        h.run_ms(10)
    assert not h.cont_if()
    live = h.read("down") == 0
    assert live.sum() == n - down
    assert (h.read("doneAt")[live] > 0).all()
    ti, waited = h.read_bits("totalIncoming"), h.read_bits("waitedSigs")
    own = np.zeros_like(ti)
    for i in range(n):
        own[i, i // 64] = np.uint64(1) << np.uint64(i % 64)
    assert ((ti & ~(waited | own)) == 0).all()  # totalIncoming subset of waitedSigs (+ own sig at level 0)
    # (the final cardinality may be below the threshold again: lastAggVerified is replaced, not merged,
    #  when a new aggregate intersects it — P/Handel.java:713-716 — so only doneAt is monotone)
    st = h.stats()
    assert st["deliveredByLevel"].sum() == h.info(False)["delivered"]
