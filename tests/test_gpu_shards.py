"""Node-range sharding on the MI355X (include/wittgpu.h "node-range sharding"): the sharded pipeline — owner-only
delivery and send resolution, the exchange images, the replicated creation of multi-destination envelopes, the
snapshot exchanges and global draw / registration orders of Handel and GSFSignature — run through RCCL (torch.distributed backend "nccl") on DEVICE memory.
The GPU box has one GPU, so the process group has one rank: every collective is the identity, every kernel of the
sharded path runs, and the result must be the oracle's bit for bit. Shard-count invariance proper (2, 3, 4 shards)
is covered over gloo by tests/test_shards_gloo.py and, on this one GPU, by k engines in one process whose
all-reduce sums their buffers in place (wittgenstein_amd.shards.LoopbackGroup; leg 5 below).

The rank runs in its own process, as it does in production (one process per GPU): torch brings its own HIP runtime,
which has to be the first one initialised in a process — bench.py and this worker import torch before the engine
library, the rest of the GPU suite never loads torch."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json, ctypes as C
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import wittgenstein_amd as w
from wittgenstein_amd import shards
import oracle_lib as o, parity
import test_shards_loopback as tl
o.build()
out = {}
def leg1():
    # 1. the all-reduce thunk on device words (one rank: the sum is the value itself)
    fn = shards.make_allreduce(dist, device_memory=True)
    t = torch.arange(1000, dtype=torch.int32, device="cuda")
    out["thunk_rc"] = fn(None, C.c_void_p(t.data_ptr()), 1000)
    out["thunk_ok"] = bool((t.cpu().numpy() == np.arange(1000)).all())
def leg2():
    # 2. PingPong 1000 nodes (BASELINE config 1), runMs(50) x 10 in lock-step with the oracle
    p = w.PingPong(w.PingPongParameters(1000), seed=0, config=shards.config(dist)); p.init()
    c = o.PingPong(1000, seed=0)
    bad = []
    for _ in range(10):
        p.network().runMs(50); c.run_ms(50)
        bad += parity.diff_pingpong(p, c)
    out["pingpong_bad"] = bad[:5]
    out["pingpong_range"] = list(shards.shard_range(p.network()))
    out["pong0"] = int(p.network().read("pong")[0])
    out["pingpong_traffic"] = list(shards.traffic(p.network()))
def leg3():
    # 3. Handel in lock-step with the oracle to convergence
    out["handel"] = []
    for params in [(64, 57, 4, 50, 10, 20, 10, 6, 0), (256, 230, 4, 50, 10, 20, 10, 25, 100)]:
        g, c = parity.handel_pair(params, seed=2, config=shards.config(dist, queue_cap=64))
        bad, k = [], 0
        while c.cont_if() and k < 400 and not bad:
            g.network().runMs(10); c.run_ms(10); k += 1
            if k %% 5 == 0: bad += parity.diff_handel(g, c)
        bad += parity.diff_handel(g, c)
        dl = c.stats()["deliveredByLevel"]
        out["handel"].append({"bad": bad[:5], "done": (not c.cont_if()) and (not g.cont_if()), "chunks": k,
                              "by_level": bool((g.network().delivered_by_level()[:len(dl)].astype(np.uint64) == dl).all()),
                              "traffic": list(shards.traffic(g.network()))})
def leg4():
    # 4. GSFSignature in lock-step with the oracle to convergence
    import test_gpu_gsf as tg
    g, c = tg.pair((256, 250, 3, 50, 10, 10, 5), seed=3, config=shards.config(dist))
    whole = shards.WholeNetwork(dist, g.network())
    class G:
        def network(self): return whole
    bad, k = [], 0
    while c.cont_if() and k < 600 and not bad:
        g.network().runMs(5); c.run_ms(5); k += 1
        if k %% 4 == 0: bad += tg.diff(G(), c)
    bad += tg.diff(G(), c)
    out["gsf"] = {"bad": bad[:5], "done": (not c.cont_if()) and (not g.cont_if()), "chunks": k,
                  "traffic": list(shards.traffic(g.network()))}
def leg5():
    # 5. real shard-count invariance on the one GPU: k engines in this process, the all-reduce sums their buffers in place
    out["loopback"] = []
    for k, params in [(2, (64, 57, 4, 50, 10, 20, 10, 6, 0)), (4, (256, 230, 4, 50, 10, 20, 10, 25, 100))]:
        bad, traffic = tl.handel_loopback(k, params, seed=1, device_memory=True)
        # (the same collectives on every shard; the words a shard RECEIVES differ since round 5: snapshots go to their readers)
        out["loopback"].append({"k": k, "bad": [str(b) for b in bad[:5]], "same_collectives": len({t[0] for t in traffic}) == 1,
                                "calls": traffic[0][0]})
def leg6():
    # 6. at a size the oracle cannot reach inside a test: 4 logical shards == the unsharded engine, bit for bit
    tr = {}
    bad, done, delivered = tl.handel_shards_vs_unsharded(4, (8192, 7299, 4, 50, 10, 20, 10, 819, 0), seed=0, device_memory=True, traffic=tr)
    out["vs_unsharded_8192"] = {"bad": bad[:6], "done": done, "delivered": delivered, "words": tr["words"]}
    # ... and the same four shards with the all-reduce alone (the form of rounds 1-4: every snapshot row to every shard)
    tr2 = {}
    bad2, done2, delivered2 = tl.handel_shards_vs_unsharded(4, (8192, 7299, 4, 50, 10, 20, 10, 819, 0), seed=0, device_memory=True,
                                                            traffic=tr2, alltoall=False)
    out["vs_unsharded_8192_image"] = {"bad": bad2[:6], "done": done2, "delivered": delivered2, "words": tr2["words"]}
def leg7():
    # 7. BASELINE config 3's size (Handel 32 768 nodes, 10 percent dead, seed 0) as 8 logical shards — what each GPU of config
    # 4's box runs, one eighth of the rows each — against the ORACLE's golden trace of that run
    # (tests/golden/handel_config3_32768.json): every per-node scalar, per-level scalar and bitset row, rd, clock.
    import hashlib, time
    from concurrent.futures import ThreadPoolExecutor
    gold = json.load(open(os.path.join(%(root)r, "tests", "golden", "handel_config3_32768.json")))
    n, thr, pair, lw, ec, per, fp, down, desync = gold["params"]
    hp = w.HandelParameters(n, thr, pair, lw, ec, per, fp, down, parity.NB, parity.NL, desync)
    K = 8
    # (WG_TEST_SHARD_IMAGE=1: the same run with the snapshots inside the all-reduce image, the form of rounds 1-4 — the
    # volumes of DESIGN.md section 7.2's table; tools/shard_volume.sh)
    directed = os.environ.get("WG_TEST_SHARD_IMAGE", "0") != "1"
    grp = shards.LoopbackGroup(K, device_memory=True)
    sims = [w.Handel(hp, seed=gold["seed"], config=grp.config(s, alltoall=directed)) for s in range(K)]
    t0 = time.time()
    with ThreadPoolExecutor(max_workers=K) as ex:  # (init() is host work: ctypes releases the GIL)
        list(ex.map(lambda g: g.init(), sims))
    t_init = time.time() - t0
    nets = [g.network() for g in sims]
    delivered = 0
    while True:  # C/RunMultipleTimes.java:50-64
        did = grp.run(lambda s: nets[s].runMs(gold["chunk"]))[0]
        delivered += nets[0].last_stats["delivered"]
        if did and not any(g.cont_if() for g in sims):
            break
        if nets[0].time > 5000: break
    dig = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
    got = {"time": nets[0].time, "rng": nets[0].rng_state(), "delivered": delivered}
    for f in parity.SCALARS: got[f] = dig(grp.gather([net.read(f) for net in nets], nets))
    for f in parity.LEVELS: got[f] = dig(grp.gather([net.read_level(f) for net in nets], nets))
    for f in parity.BITS: got[f] = dig(grp.gather([net.read_bits(f) for net in nets], nets))
    bad = {k: (got.get(k), v) for k, v in gold["final"].items() if got.get(k) != v}
    per_shard = [net.device_bytes() for net in nets]
    out["config3_as_8_shards"] = {"bad": {k: [str(x) for x in v] for k, v in bad.items()}, "init_s": t_init,
                                  "run_s": time.time() - t0 - t_init, "device_bytes_per_shard": per_shard,
                                  "words_received_per_shard": [shards.traffic(net)[1] for net in nets],
                                  "snapshots": "owner-directed" if directed else "all-reduce image", "simulated_ms": nets[0].time,
                                  "by_exchange_shard0": shards.traffic_by_exchange(nets[0]),
                                  "by_exchange_shard7": shards.traffic_by_exchange(nets[7]),
                                  "same_rng": len({net.rng_state() for net in nets}) == 1}
def leg8():
    # 8. Casper IMD on logical shards of the one GPU: PT/CasperIMDTest.java:10-11's network (406 nodes) with 40 attesters
    # stop()ped as 4 shards, and ByzBlockProducerWF(-2000) as 3, both in lock-step with the oracle after every chunk
    import test_shards_casper as tc
    out["casper"] = []
    for k, params, byz, stopped, chunk, chunks in [(4, (5, False, 5, 80, 1000, 1), 0, 40, 4000, 12),
                                                   (3, (3, False, 3, 8, 1000, 1), -2000, 3, 1000, 40),
                                                   # randomOnTies with a fork (ByzBlockProducerWF(+7000)): ties draw, the ordered
                                                   # visit of the ms's blocks and tasks goes round the shards (k_casper_seq_shard)
                                                   (2, (2, True, 2, 6, 1000, 1), 7000, 2, 500, 160)]:
        c, traffic = tc.casper_loopback(k, params, seed=3, chunk=chunk, chunks=chunks, byz_delay=byz, stopped=stopped,
                                        device_memory=True)
        out["casper"].append({"k": k, "delivered": int(c.info()["delivered"]), "height": int(c.read("headHeight")[0]),
                              "same_collectives": len(set(traffic)) == 1, "calls": traffic[0][0]})
def leg9():
    # 9. Casper IMD, 2 x 1024 attesters (2 051 nodes, every vote a sendAll to all of them): 4 logical shards == the
    # unsharded engine after every chunk — a size the oracle does not reach inside a test
    from wittgenstein_amd import protocols as P
    import test_gpu_casper_resident as tcr
    params = (2, False, 2, 1024, 1000, 1)
    ref = P.CasperIMD(P.CasperParemeters(*params, None, None), seed=1, max_slots=8); ref.init()
    K = 4
    grp = shards.LoopbackGroup(K, device_memory=True)
    sims = [P.CasperIMD(P.CasperParemeters(*params, None, None), seed=1, max_slots=8, config=grp.config(s)) for s in range(K)]
    for g in sims: g.init()
    nets = [g.network() for g in sims]
    bad = []
    for step in range(6):
        grp.run(lambda s: nets[s].runMs(4000)); ref.network().runMs(4000)
        for f in tcr.FIELDS:
            a = nets[0].read(f) if f in ("x", "y") else grp.gather([net.read(f) for net in nets], nets)
            if not np.array_equal(a, ref.network().read(f)): bad.append((step, f))
        if (nets[0].time, nets[0].rng_state(), nets[0].msgs.size()) != (ref.network().time, ref.network().rng_state(), ref.network().msgs.size()):
            bad.append((step, "time / rd / msgs.size()"))
    out["casper_vs_unsharded"] = {"bad": [str(b) for b in bad[:6]], "delivered": int(ref.network().read("msgReceived").sum()),
                                  "height": int(ref.network().read("headHeight")[0])}
def leg10():
    # 10. P2PFlood on logical shards of the one GPU (shuffled MultipleDestWithDelayEnvelopes with explicit arrivals through the
    # replicated envelope creation), in lock-step with the oracle
    out["p2pflood"] = []
    for k, params, nl, seed, chunk, chunks in [(2, (100, 10, 50, 1, 1, 10, 30), "NetworkNoLatency", 0, 1000, 20),
                                               (4, (2000, 10, 50, 1, 1, 10, 30), None, 0, 250, 8)]:
        c, traffic = tl.p2pflood_loopback(k, params, nl, seed, chunk, chunks, device_memory=True)
        out["p2pflood"].append({"k": k, "delivered": int(c.info()["delivered"]), "same_collectives": len(set(traffic)) == 1})
def leg11():
    # 11. Casper IMD at 16 390 nodes (BASELINE config 5's shape with 256 attesters per round: 64 x 256 attesters, 5 producers, 1
    # observer; every vote a sendAll to all of them), a tenth of the attesters stop()ped, as 4 logical shards of the one GPU in
    # lock-step with the oracle after every chunk — the sharded pipeline against the oracle at forty times leg 8's node count
    import test_shards_casper as tc
    c, traffic = tc.casper_loopback(4, (64, False, 5, 256, 1000, 1), seed=2, chunk=4000, chunks=6, stopped=1639, max_slots=8,
                                    device_memory=True)
    out["casper_16390"] = {"delivered": int(c.info()["delivered"]), "height": int(c.read("headHeight")[0]),
                           "same_collectives": len(set(traffic)) == 1}
import traceback
out["errors"] = {}
for _name, _fn in [(k, v) for k, v in sorted(globals().items()) if k.startswith('leg') and callable(v)]:
    try:
        _fn()
    except Exception:
        out["errors"][_name] = traceback.format_exc()[-1500:]
print("RESULT " + json.dumps(out))
dist.destroy_process_group()
'''


@pytest.fixture(scope="module")
def result(tmp_path_factory):
    script = tmp_path_factory.mktemp("shards") / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):  # (kept as evidence: exchange volumes, shard sizes, run times)
        open(os.path.join(ROOT, "gpurun_out", "shards_result.json"), "w").write(line[len("RESULT "):])
    return json.loads(line[len("RESULT "):])


def test_allreduce_thunk_sums_device_words(result):
    assert "leg1" not in result["errors"], result["errors"]["leg1"]
    assert result["thunk_rc"] == 0 and result["thunk_ok"]


def test_sharded_pingpong_one_rank(result):
    assert "leg2" not in result["errors"], result["errors"]["leg2"]
    assert result["pingpong_bad"] == []
    assert result["pingpong_range"] == [0, 1000] and result["pong0"] == 1000
    calls, words = result["pingpong_traffic"]
    assert calls > 0 and words >= 2000 + 5 * 1000         # one packed word per event (round 4; two before) + one record image per Pong


def test_sharded_handel_one_rank(result):
    assert "leg3" not in result["errors"], result["errors"]["leg3"]
    assert len(result["handel"]) == 2
    for r in result["handel"]:
        assert r["bad"] == [] and r["done"] and r["by_level"], r
        assert r["traffic"][0] > 0


def test_sharded_gsf_one_rank(result):
    assert "leg4" not in result["errors"], result["errors"]["leg4"]
    r = result["gsf"]
    assert r["bad"] == [] and r["done"] and r["traffic"][0] > 0, r


def test_logical_shards_on_one_gpu(result):   # 2 and 4 shards of one Handel simulation on the one MI355X
    assert "leg5" not in result["errors"], result["errors"]["leg5"]
    assert [r["k"] for r in result["loopback"]] == [2, 4]
    for r in result["loopback"]:
        assert r["bad"] == [] and r["same_collectives"] and r["calls"] > 0, r


def test_four_logical_shards_equal_the_unsharded_engine_at_8192_nodes(result):
    assert "leg6" not in result["errors"], result["errors"]["leg6"]
    r = result["vs_unsharded_8192"]
    assert r["bad"] == [] and r["done"] == 8192 - 819 and r["delivered"] > 1000000, r
    # the owner-directed snapshot exchange (round 5) against the all-reduce image of rounds 1-4: the same run, and every
    # shard receives a fraction of the words (the snapshot rows were most of them)
    ri = result["vs_unsharded_8192_image"]
    assert ri["bad"] == [] and ri["done"] == r["done"] and ri["delivered"] == r["delivered"], ri
    # (8 192 nodes: rows of at most 64 words — the event words and the 5-word records of the replicated scheduler are most of
    # what is left; at config 3's / config 4's sizes the rows are 4 / 16 times wider: DESIGN.md §7.2 has the table)
    assert max(r["words"]) < 0.8 * min(ri["words"]), (r["words"], ri["words"])


def handel_shard_bytes_model(n, k, horizon=256, q=32):
    """device bytes of one of k node-range shards of an n-node Handel simulation: what wittgenstein_amd/csrc/engine.hip
    (ensure_device, HandelHost) allocates — per-node rows for the owned n/k nodes, the scheduler replicated"""
    L = n.bit_length()          # levels 0..log2(n)
    W = max(1, n // 64)
    own = n // k
    peer = 2 if n <= 65536 else 4   # emission lists: 16-bit ids up to 65 536 nodes
    # bit rows TI/LA/VI + the delivery-side pieces {SEEN, FP, BUMP}; receptionRanks + emission lists (a shard keeps the matrix
    # form); the header record; the queue records (head, valid mask, q entries, bad mask in whole lines); the cached evaluations
    rows = own * ((3 + 3) * W * 8 + n * 4 + (n - 1) * peer + (32 + 8 * (16 if L <= 16 else 32)) * 4 + L * ((3 + q + 1 + 7) // 8 * 8) * 8 + L * q * 4)
    nw = lambda l: max(1, (1 << (l - 1)) // 64)
    qsig = sum(own * (min(q, 16) if nw(l) >= 16 else q) * nw(l) * 8 for l in range(1, L))   # queue_cap_wide
    snap = (horizon // 20 + 2) * n * max(1, n // 128) * 8           # dissemination snapshots: replicated ring
    maxout = 24 * n
    sched = (max(1 << 20, 256 * n) + horizon * 1024) * 16 + maxout * (16 + 16 + 8 + 4 + 4 + 4 + 32 + 4 + 16 + 4) \
        + 16 * n * 32 + 2 * max(1 << 20, 256 * n) * 4 + (1 << 16) * 8 + maxout * 5 * 4 + maxout * 8
    # the owner-directed snapshot exchange (round 5): send and receive regions of 17-word chunks, one per shard each way
    stride = max(1, n // 128)
    xch = 2 * k * own * (stride // 8 + L) * 17 * 8
    return rows + qsig + snap + sched + xch


def test_config3_as_8_logical_shards_equals_the_oracle_trace(result):
    assert "leg7" not in result["errors"], result["errors"]["leg7"]
    r = result["config3_as_8_shards"]
    assert r["bad"] == {} and r["same_rng"], r
    # the capacity model reproduces what a shard of this run actually holds (within 10 %) ...
    measured = max(r["device_bytes_per_shard"])
    assert abs(handel_shard_bytes_model(32768, 8) - measured) < 0.10 * measured, (handel_shard_bytes_model(32768, 8), measured)
    # ... and says that BASELINE config 4 (Handel 131 072 nodes over the 8 GPUs of one box) fits an MI355X per shard
    assert handel_shard_bytes_model(131072, 8) < 0.92 * 288 * (1 << 30)


def test_casper_logical_shards_match_the_oracle(result):   # per-node rows by owner, tables by exchange, sendAll on every shard
    assert "leg8" not in result["errors"], result["errors"]["leg8"]
    assert [r["k"] for r in result["casper"]] == [4, 3, 2]   # (the last: randomOnTies with ties drawing, refused on shards until round 4)
    for r in result["casper"]:
        assert r["same_collectives"] and r["calls"] > 0 and r["height"] >= 3, r
    assert result["casper"][0]["delivered"] > 100000


def test_casper_16390_nodes_as_four_logical_shards_match_the_oracle(result):
    assert "leg11" not in result["errors"], result["errors"].get("leg11")
    r = result["casper_16390"]
    assert r["same_collectives"] and r["height"] >= 1 and r["delivered"] > 5000000, r


def test_casper_four_logical_shards_equal_the_unsharded_engine_at_2051_nodes(result):
    assert "leg9" not in result["errors"], result["errors"]["leg9"]
    r = result["casper_vs_unsharded"]
    assert r["bad"] == [] and r["height"] >= 1 and r["delivered"] > 1024 * 2051 * 0.9, r


def test_p2pflood_logical_shards_match_the_oracle(result):
    assert "leg10" not in result["errors"], result["errors"]["leg10"]
    assert [r["k"] for r in result["p2pflood"]] == [2, 4]
    assert all(r["same_collectives"] for r in result["p2pflood"]) and result["p2pflood"][1]["delivered"] > 10000
