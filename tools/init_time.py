#!/usr/bin/env python
"""Wall time of Handel's init() (P/Handel.java:957-1014) for one copy and for a batch of copies on host threads, with the
engine's own breakdown (WG_INIT_VERBOSE=1).   python tools/init_time.py [nodes] [copies]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

os.environ["WG_INIT_VERBOSE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import wittgenstein_amd as w

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 24


def one(seed):
    t = time.perf_counter()
    g = bench.make_sim(w, n, seed, 0)
    t1 = time.perf_counter()
    g.network().snapshot()
    return g, t1 - t, time.perf_counter() - t1


g, a, b = one(0)
print("first copy: init() %.3f s (on the device: %s), wg_snapshot %.3f s" % (a, g.init_on_device, b), flush=True)
os.environ["WG_INIT_VERBOSE"] = "0"
t = time.perf_counter()
with ThreadPoolExecutor(max_workers=copies - 1) as ex:
    rest = list(ex.map(one, range(1, copies)))
print("%d more copies on %d threads: %.3f s wall; per copy init() %.3f .. %.3f s, wg_snapshot %.3f .. %.3f s"
      % (copies - 1, copies - 1, time.perf_counter() - t, min(r[1] for r in rest), max(r[1] for r in rest),
         min(r[2] for r in rest), max(r[2] for r in rest)))
