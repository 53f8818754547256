"""Developer tool (not a test): lock-step GPU vs oracle with verbose first-mismatch output."""
import sys
import time

import conftest  # noqa: F401  (sys.path)
import oracle_lib as o
import parity
import wittgenstein_amd as w


def pingpong(n=1000, step=10, until=600):
    g = w.PingPong(w.PingPongParameters(n, parity.NB, parity.NL))
    g.init()
    c = o.PingPong(n, parity.NB, parity.NL)
    print("pingpong init diff:", parity.diff_pingpong(g, c))
    t = 0
    while t < until:
        g.network().runMs(step)
        c.run_ms(step)
        t += step
        d = parity.diff_pingpong(g, c)
        if d:
            print("PINGPONG MISMATCH at t=%d" % t, d)
            return False
    print("pingpong ok until", t, "pong0", g.network().read("pong")[0], "stats", g.network().last_stats)
    return True


def handel(params, step=1, until=3000, seed=0, bits=True):
    t0 = time.time()
    g, c = parity.handel_pair(params, seed=seed)
    print("handel", params, "init gpu %.2fs total %.2fs" % (g.init_seconds, time.time() - t0))
    d = parity.diff_handel(g, c, bits=bits)
    if d:
        print("INIT MISMATCH", d)
        return False
    t = 0
    while t < until and c.cont_if():
        g.network().runMs(step)
        c.run_ms(step)
        t += step
        d = parity.diff_handel(g, c, bits=bits)
        if d:
            print("HANDEL MISMATCH at t=%d" % t)
            for x in d:
                print("   ", x)
            return False
    print("handel ok until", t, "gpu cont_if", g.cont_if(), "doneAt max", g.network().read("doneAt").max())
    return True


if __name__ == "__main__":
    ok = pingpong()
    ok = handel((64, 60, 6, 10, 5, 5, 10, 2, 100)) and ok
    ok = handel((64, 57, 4, 50, 10, 20, 10, 6, 0)) and ok
    ok = handel((256, 228, 4, 50, 10, 20, 10, 25, 0), step=1) and ok
    ok = handel((1024, 912, 4, 50, 10, 20, 10, 102, 0), step=10) and ok
    sys.exit(0 if ok else 1)
