#!/usr/bin/env python
"""Shard-count invariance at sizes the oracle cannot reach: ONE Handel simulation as k
logical shards on this GPU (shards.LoopbackGroup) against the UNSHARDED engine on the same seed — delivered count,
simulated ms, time, rd state, every per-node scalar, the per-level scalars and all five bitset rows must be equal.
usage: shard_invariance.py <nodes> <shards> [seed]      (prints one JSON line; exit 1 on any mismatch)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    n, k = int(sys.argv[1]), int(sys.argv[2])
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    import torch
    assert torch.cuda.is_available(), "needs an MI355X"
    torch.cuda.set_device(0)
    import test_shards_loopback as tl
    down = int(n * 0.10)
    params = (n, int(n * (1 - 0.10) * 0.99), 4, 50, 10, 20, 10, down, 0)  # bench.py's handel_params(n)
    t0 = time.perf_counter()
    bad, done, delivered = tl.handel_shards_vs_unsharded(k, params, seed=seed, device_memory=True)
    out = {"nodes": n, "logical_shards": k, "seed": seed, "params": params, "mismatches": bad, "live_nodes_done": done,
           "live_nodes": n - down, "delivered": delivered, "wall_s_incl_host_init": time.perf_counter() - t0}
    print(json.dumps(out), flush=True)
    sys.exit(1 if bad or done != n - down else 0)


if __name__ == "__main__":
    main()
