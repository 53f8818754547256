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


def _free_port():
    """a port nobody listens on right now (the CPU suite runs on several xdist workers: fixed ports would collide)"""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _run(tmp_path, world, port, **kw):
    port = _free_port()
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


@pytest.mark.parametrize("world,n,nl", [(2, 300, None), (3, 257, "NetworkLatencyByDistanceWJitter"),
                                        (8, 200, "NetworkLatencyByDistanceWJitter")])
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


HANDEL_WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch, torch.distributed as dist
import wittgenstein_amd._lib as L
L.LIB_PATH = os.path.join(%(root)r, "tests", "emu", "libwittgpu_emu.so")   # test infrastructure: no GPU here
import wittgenstein_amd as w
from wittgenstein_amd import shards
import oracle_lib as o
import parity
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
params = %(params)r
g, c = parity.handel_pair(params, seed=%(seed)d, config=shards.config(dist, device_memory=False, queue_cap=64))
whole = shards.WholeNetwork(dist, g.network())
class G:
    def network(self): return whole
bad, steps, delivered = [], 0, 0
while True:
    g.network().runMs(%(chunk)d); c.run_ms(%(chunk)d); steps += 1
    delivered += g.network().last_stats["delivered"]
    if steps %% %(check_every)d == 0:
        bad += [(steps, m) for m in parity.diff_handel(G(), c)]
    go, want = shards.cont_if(dist, g), c.cont_if()
    if go != want: bad.append((steps, "cont_if %%s != %%s" %% (go, want)))
    if bad or not want or steps >= %(max_steps)d: break
bad += [(steps, m) for m in parity.diff_handel(G(), c)]
dl = c.stats()["deliveredByLevel"]
if not np.array_equal(g.network().delivered_by_level()[:len(dl)].astype(np.uint64), dl): bad.append((steps, "delivered_by_level"))
calls, words = shards.traffic(g.network())
res = [None] * world
dist.all_gather_object(res, {"rank": rank, "bad": [str(b) for b in bad[:6]], "steps": steps, "calls": calls, "words": words,
                             "delivered": delivered, "expect": c.info(False)["delivered"], "done": not c.cont_if(),
                             "doneAt": int((whole.read("doneAt") > 0).sum())})
if rank == 0:
    print("RESULT " + json.dumps(res))
dist.destroy_process_group()
'''


def _run_handel(tmp_path, world, port, **kw):
    global WORKER
    keep, WORKER = WORKER, HANDEL_WORKER
    try:
        return _run(tmp_path, world, port, **kw)
    finally:
        WORKER = keep


# (nodeCount, threshold, pairing, levelWait, extraCycle, period, fastPath, nodesDown, desync)
@pytest.mark.parametrize("world,params", [
    (2, (64, 57, 4, 50, 10, 20, 10, 6, 0)),      # PT/HandelTest's size, fast path on, dead nodes
    (4, (64, 60, 6, 10, 5, 5, 10, 2, 100)),      # PT/HandelTest.java:36-49 parameters, 4 shards
    (2, (64, 57, 4, 50, 10, 20, 10, 6, 600)),    # starts beyond the bucket horizon: host-held envelopes injected mid-run
])
def test_sharded_handel_matches_the_oracle(oracle, tmp_path, world, params):
    res = _run_handel(tmp_path, world, 29551 + world + params[0] % 7, params=params, seed=1, chunk=10, check_every=5,
                      max_steps=400)
    assert len(res) == world
    for r in res:
        assert r["bad"] == [], r
        assert r["done"] and r["delivered"] == r["expect"] > 0
        # (the same collectives on every shard; the words a shard RECEIVES differ since round 5: a dissemination's snapshot goes to
        # the shards whose nodes read it — an all-to-all over gloo here —, not to every shard inside an all-reduce image)
        assert r["calls"] == res[0]["calls"]
    live = params[0] - params[7]
    assert res[0]["doneAt"] == live          # every live node reached the threshold (PT/HandelTest.java:36-49)


GSF_WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch, torch.distributed as dist
import wittgenstein_amd._lib as L
L.LIB_PATH = os.path.join(%(root)r, "tests", "emu", "libwittgpu_emu.so")   # test infrastructure: no GPU here
import wittgenstein_amd as w
from wittgenstein_amd import shards
import oracle_lib as o
import test_gpu_gsf as tg
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
g, c = tg.pair(%(params)r, nb=%(nb)r, seed=%(seed)d, config=shards.config(dist, device_memory=False))
whole = shards.WholeNetwork(dist, g.network())
class G:
    def network(self): return whole
bad, steps, delivered = [], 0, 0
while True:
    g.network().runMs(%(chunk)d); c.run_ms(%(chunk)d); steps += 1
    delivered += g.network().last_stats["delivered"]
    if steps %% %(check_every)d == 0:
        bad += [(steps, m) for m in tg.diff(G(), c)]
    go, want = shards.cont_if(dist, g), c.cont_if()
    if go != want: bad.append((steps, "cont_if %%s != %%s" %% (go, want)))
    if bad or not want or steps >= %(max_steps)d: break
bad += [(steps, m) for m in tg.diff(G(), c)]
calls, words = shards.traffic(g.network())
res = [None] * world
dist.all_gather_object(res, {"rank": rank, "bad": [str(b) for b in bad[:6]], "steps": steps, "calls": calls, "words": words,
                             "delivered": delivered, "expect": c.info(False)["delivered"], "done": not c.cont_if()})
if rank == 0:
    print("RESULT " + json.dumps(res))
dist.destroy_process_group()
'''


# GSFSignatureParameters ctor order: (nodeCount, threshold, pairingTime, timeoutPerLevelMs, periodDurationMs,
# acceleratedCallsCount, nodesDown)
@pytest.mark.parametrize("world,params,nb", [
    (2, (64, 63, 3, 50, 10, 10, 0), "RANDOM_SPEED=CONSTANT_TOR=0.00"),
    (4, (128, 100, 3, 20, 10, 10, 12), "RANDOM_SPEED=GAUSSIAN_TOR=0.00"),   # dead nodes, speed ratios, 4 shards
])
def test_sharded_gsf_matches_the_oracle(oracle, tmp_path, world, params, nb):
    global WORKER
    keep, WORKER = WORKER, GSF_WORKER
    try:
        res = _run(tmp_path, world, 29571 + world, params=params, nb=nb, seed=4, chunk=5, check_every=4, max_steps=600)
    finally:
        WORKER = keep
    assert len(res) == world
    for r in res:
        assert r["bad"] == [], r
        assert r["done"] and r["delivered"] == r["expect"] > 0
        assert r["calls"] == res[0]["calls"] and r["words"] == res[0]["words"]


RMT_WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, torch.distributed as dist
import wittgenstein_amd._lib as L
L.LIB_PATH = os.path.join(%(root)r, "tests", "emu", "libwittgpu_emu.so")   # test infrastructure: no GPU here
from wittgenstein_amd import shards
import parity
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
g, c = parity.handel_pair(%(params)r, seed=%(seed)d, config=shards.config(dist, device_memory=False, queue_cap=64))
d, ms = shards.run_multiple_times(dist, g, chunk=10, maxTime=20000)      # what bench.py --mode shard times
cms = 0
while True:                                                              # C/RunMultipleTimes.java:50-64 on the oracle
    did = c.run_ms(10); cms += 10
    if not (c.info(False)["time"] < 20000 and (not did or c.cont_if())): break
res = [None] * world
dist.all_gather_object(res, {"delivered": d, "ms": ms, "expect": c.info(False)["delivered"], "expect_ms": cms})
if rank == 0:
    print("RESULT " + json.dumps(res))
dist.destroy_process_group()
'''


def test_sharded_run_multiple_times_loop(oracle, tmp_path):
    global WORKER
    keep, WORKER = WORKER, RMT_WORKER
    try:
        res = _run(tmp_path, 2, 29591, params=(64, 57, 4, 50, 10, 20, 10, 6, 0), seed=7)
    finally:
        WORKER = keep
    for r in res:
        assert (r["delivered"], r["ms"]) == (r["expect"], r["expect_ms"]) and r["delivered"] > 0


SF_WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch, torch.distributed as dist
import wittgenstein_amd._lib as L
L.LIB_PATH = os.path.join(%(root)r, "tests", "emu", "libwittgpu_emu.so")   # test infrastructure: no GPU here
from wittgenstein_amd import shards, protocols as P
import oracle_lib as o
import test_gpu_sanfermin_resident as ts
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
params = %(params)r
g = P.SanFerminSignature(P.SanFerminSignatureParameters(*params), seed=%(seed)d, config=shards.config(dist, device_memory=False))
g.init()
c = o.SanFerminSignature(params, seed=%(seed)d)
whole = shards.WholeNetwork(dist, g.network())
class G:
    def network(self): return whole
bad = []
for k in range(%(chunks)d):
    g.network().runMs(%(chunk)d); c.run_ms(%(chunk)d)
    bad += [(k, m) for m in ts.diff(G(), c)]
    if bad: break
if shards.cont_if(dist, g) != (int((c.read("done") == 0).sum()) > 0): bad.append("cont_if")
calls, words = shards.traffic(g.network())
res = [None] * world
dist.all_gather_object(res, {"rank": rank, "bad": [str(b) for b in bad[:6]], "calls": calls, "words": words,
                             "finished": c.info()["finished"]})
if rank == 0:
    print("RESULT " + json.dumps(res))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,params", [(2, (64, 64, 2, 48, 300, 1)), (4, (128, 128, 2, 48, 300, 3))])
def test_sharded_sanfermin_matches_the_oracle(oracle, tmp_path, world, params):  # shuffled multi-destination requests
    global WORKER
    keep, WORKER = WORKER, SF_WORKER
    try:
        res = _run(tmp_path, world, 29601 + world, params=params, seed=3, chunk=50, chunks=30)
    finally:
        WORKER = keep
    for r in res:
        assert r["bad"] == [], r
        assert r["calls"] == res[0]["calls"] and r["words"] == res[0]["words"] and r["finished"] > params[0] * 0.8


CASPER_WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch, torch.distributed as dist
import wittgenstein_amd._lib as L
L.LIB_PATH = os.path.join(%(root)r, "tests", "emu", "libwittgpu_emu.so")   # test infrastructure: no GPU here
from wittgenstein_amd import shards, protocols as P
import oracle_lib as o
import test_gpu_casper_resident as tcr
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
params = %(params)r
g = P.CasperIMD(P.CasperParemeters(*params, None, None), seed=%(seed)d, byz_delay=%(byz)d, max_slots=16,
                config=shards.config(dist, device_memory=False))
g.init()
c = o.CasperIMD(params, None, None, seed=%(seed)d, byz_delay=%(byz)d)
if %(stopped)d:
    ids = g.stop_attesters(%(stopped)d, seed=%(seed)d + 1)   # (every shard stops the same nodes: `down` is replicated)
    c.stop(ids)
whole = shards.WholeNetwork(dist, g.network())
class G:
    def network(self): return whole
bad = [("init", m) for m in tcr.diff(G(), c)]
for k in range(%(chunks)d):
    if bad: break
    g.network().runMs(%(chunk)d); c.run_ms(%(chunk)d)
    bad += [(k, m) for m in tcr.diff(G(), c)]
calls, words = shards.traffic(g.network())
res = [None] * world
dist.all_gather_object(res, {"rank": rank, "bad": [str(b) for b in bad[:6]], "calls": calls, "words": words,
                             "delivered": c.info()["delivered"], "height": int(c.read("headHeight")[0])})
if rank == 0:
    print("RESULT " + json.dumps(res))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,params,byz,stopped,chunk,chunks", [
    (2, (5, False, 5, 80, 1000, 1), 0, 40, 4000, 7),       # PT/CasperIMDTest.java:10-11's network, 10 % of the attesters stop()ped
    (4, (3, False, 3, 8, 1000, 1), -2000, 3, 1000, 30),    # ByzBlockProducerWF(-2000), PT/CasperByzantineTest.java:41
    (2, (2, True, 2, 6, 1000, 1), 7000, 2, 500, 160)])     # randomOnTies with a fork: the ordered visit goes round the ranks
def test_sharded_casper_matches_the_oracle(oracle, tmp_path, world, params, byz, stopped, chunk, chunks):
    """Casper IMD resident on node-range shards (per-node rows by owner, block / attestation tables replicated and filled by
    exchange, sendAll resolved on every shard, periodic tasks through every shard's far buffer): every observable of
    tests/test_gpu_casper_resident.py::diff after every chunk, over gloo ranks"""
    global WORKER
    keep, WORKER = WORKER, CASPER_WORKER
    try:
        res = _run(tmp_path, world, 29701 + world, params=params, seed=3, byz=byz, stopped=stopped, chunk=chunk, chunks=chunks)
    finally:
        WORKER = keep
    for r in res:
        assert r["bad"] == [], r
        assert r["calls"] == res[0]["calls"] and r["words"] == res[0]["words"] and r["height"] >= 2
    assert res[0]["delivered"] > (50000 if world == 2 and not params[1] else 300)


FLOOD_WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch, torch.distributed as dist
import wittgenstein_amd._lib as L
L.LIB_PATH = os.path.join(%(root)r, "tests", "emu", "libwittgpu_emu.so")   # test infrastructure: no GPU here
from wittgenstein_amd import shards, protocols as P
import oracle_lib as o
import test_gpu_p2pflood_resident as tf
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
params = %(params)r
g = P.P2PFlood(P.P2PFloodParameters(*params, None, %(nl)r), seed=%(seed)d, config=shards.config(dist, device_memory=False))
g.init()
c = o.P2PFlood(params, None, %(nl)r, seed=%(seed)d)
whole = shards.WholeNetwork(dist, g.network())
class G:
    def network(self): return whole
bad = [("init", m) for m in tf.diff(G(), c)]
for k in range(%(chunks)d):
    if bad: break
    g.network().runMs(%(chunk)d); c.run_ms(%(chunk)d)
    bad += [(k, m) for m in tf.diff(G(), c)]
calls, words = shards.traffic(g.network())
res = [None] * world
dist.all_gather_object(res, {"rank": rank, "bad": [str(b) for b in bad[:6]], "calls": calls, "words": words,
                             "delivered": c.info()["delivered"]})
if rank == 0:
    print("RESULT " + json.dumps(res))
dist.destroy_process_group()
'''


def test_sharded_p2pflood_matches_the_oracle(oracle, tmp_path):  # shuffled envelopes with explicit arrivals, over 2 gloo ranks
    global WORKER
    keep, WORKER = WORKER, FLOOD_WORKER
    try:
        res = _run(tmp_path, 2, 29801, params=(300, 20, 20, 3, 1, 6, 10), nl=None, seed=4, chunk=100, chunks=40)
    finally:
        WORKER = keep
    for r in res:
        assert r["bad"] == [], r
        assert r["calls"] == res[0]["calls"] and r["words"] == res[0]["words"] and r["delivered"] > 4000
