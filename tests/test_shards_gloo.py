"""Node-range sharding of ONE simulation (include/wittgpu.h "node-range sharding", wittgenstein_amd/shards.py)
on CPU: S processes over gloo, each running the engine's kernel sources on the CPU wave emulator (test
infrastructure) for the nodes it owns. What is checked is shard-count invariance against the oracle: PingPong's
pong counts, every Node counter, the queue size, the delivered count and the state of the shared rd after every
runMs — i.e. that the global LIFO order and the rd index of every send survive the split."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch, torch.distributed as dist
import wittgenstein_amd._lib as L
L.LIB_PATH = os.path.join(%(root)r, "tests", "emu", "libwittgpu_emu.so")   # test infrastructure: no GPU here
import wittgenstein_amd as w
from wittgenstein_amd import shards
import oracle_lib as o
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
N, SEED, NL = %(n)d, %(seed)d, %(nl)r
cfg = shards.config(dist, device_memory=False)
p = w.PingPong(w.PingPongParameters(N, None, NL), seed=SEED, config=cfg); p.init()
net = p.network()
ref = o.PingPong(N, None, NL, seed=SEED)
lo, hi = shards.shard_range(net)
bad = []
delivered = 0
for step in range(%(steps)d):
    net.runMs(%(chunk)d); ref.run_ms(%(chunk)d)
    delivered += net.last_stats["delivered"]
    for f in ("pong", "msgReceived", "msgSent", "bytesSent", "bytesReceived"):
        mine = net.read(f)
        if mine[:lo].any() or mine[hi:].any():
            bad.append((step, f, "writes outside the shard"))
        if not np.array_equal(shards.gather(dist, mine), ref.read(f)):
            bad.append((step, f, "differs from the oracle"))
    info = ref.info()
    if (net.time, net.msgs.size(), net.rng_state(), delivered) != (info["time"], info["queue"], info["rng"], info["delivered"]):
        bad.append((step, "time/queue/rng/delivered", (net.time, net.msgs.size(), net.rng_state(), delivered), info))
calls, words = shards.traffic(net)
res = [None] * world
dist.all_gather_object(res, {"rank": rank, "range": [lo, hi], "bad": bad[:5], "calls": calls, "words": words,
                             "pong0": int(shards.gather(dist, net.read("pong"))[0]), "delivered": delivered})
if rank == 0:
    print("RESULT " + json.dumps(res))
dist.destroy_process_group()
'''


def _run(tmp_path, world, port, **kw):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")], check=True)
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(kw, root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0]
    return json.loads(line[len("RESULT "):])


@pytest.mark.parametrize("world,n,nl", [(2, 300, None), (3, 257, "NetworkLatencyByDistanceWJitter")])
def test_sharded_pingpong_matches_the_oracle(oracle, tmp_path, world, n, nl):
    res = _run(tmp_path, world, 29541 + world, n=n, seed=3, nl=nl, steps=8, chunk=50)
    assert len(res) == world
    covered = sorted(tuple(r["range"]) for r in res)
    assert covered[0][0] == 0 and covered[-1][1] == n and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    for r in res:
        assert r["bad"] == [], r
        assert r["calls"] > 0 and r["calls"] == res[0]["calls"] and r["words"] == res[0]["words"]
    assert res[0]["pong0"] == n                      # PT/PingPongTest.java:8-19: node 0 collects every pong
    assert res[0]["delivered"] == 2 * n
