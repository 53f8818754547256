"""bench.py's host logic without a GPU: the replica / thread planning arithmetic (wittgenstein_amd/replicas.py) and the
whole main loop — init once, wg_snapshot, per-step wg_restore + RunMultipleTimes pass, JSON line — driven with the
DRIVER'S literal argv (`--gpus 1 --steps 20 --warmup 5`) on a tiny network, the product's kernel sources on the CPU wave
emulator (tests/emu) and torch.cuda's four calls stubbed. Round 1's bench line died on exactly this argv."""
import json
import os
import subprocess
import sys

import pytest

from wittgenstein_amd import replicas

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_replicas_degrades_and_never_fails():
    GB = 1 << 30
    assert replicas.plan_replicas(16, 288 * GB, 10 * GB) == 16
    assert replicas.plan_replicas(32, 288 * GB, 10 * GB) == 27          # int(0.955 * 288 / 10)
    assert replicas.plan_replicas(16, 288 * GB, 16 * GB) == 16
    assert replicas.plan_replicas(16, 288 * GB, 17 * GB) == 16
    assert replicas.plan_replicas(16, 288 * GB, 18 * GB) == 15
    assert replicas.plan_replicas(16, 288 * GB, 400 * GB) == 1          # does not fit at all: still one copy, no error
    assert replicas.plan_replicas(4, 0, 0) == 4                         # nothing measured: the request stands
    # Handel's init() holds nodeCount^2 rank matrices only while it runs (at most two and at most 8 GiB of them at a time, one
    # alone may be larger): the batch leaves that beside the copies — 32 copies of config 3 fit (round 6: 4 per XCD), 8 of the
    # 65 536-node target size do not (round 5: they ran out of memory)
    GiB = 1 << 30
    assert replicas.handel_init_transient_bytes(32768) == 8 * GiB and replicas.handel_init_transient_bytes(65536) == 16 * GiB
    assert replicas.handel_init_transient_bytes(4096) == 2 * 4 * 4096 * 4096 and replicas.handel_init_transient_bytes(131072) == 0
    assert replicas.plan_replicas(32, 288 * GiB, int(8.56 * GiB), transient_bytes=8 * GiB) == 32
    assert replicas.plan_replicas(33, 288 * GiB, int(8.56 * GiB), transient_bytes=8 * GiB) == 32
    assert replicas.plan_replicas(8, 287 * GiB, 35 * GiB, transient_bytes=16 * GiB) == 7
    # a batch is sized by what a FURTHER copy takes: the first one also pays the process's one-time allocations
    assert replicas.marginal_copy_bytes(760 << 20, 590 << 20) == (590 << 20, 170 << 20)
    assert replicas.marginal_copy_bytes(590 << 20, 600 << 20) == (600 << 20, 0)
    assert replicas.plan_replicas(512, 288 * GiB - (170 << 20), 590 << 20) == 477
    # the number of steps plays no part (round 1: `fit // K` turned --steps 20 into rc=1)
    with pytest.raises(ValueError):
        replicas.plan_replicas(0, GB, GB)


def test_init_threads_bounds():
    GB = 1 << 30
    assert replicas.init_threads(0, 15, 3000 * GB, 16 * GB, 256) == 15      # one per copy
    assert replicas.init_threads(0, 31, 3000 * GB, 16 * GB, 16) == 16       # cores
    assert replicas.init_threads(0, 31, 62 * GB, 16 * GB, 256) == 2         # host memory
    assert replicas.init_threads(6, 31, 3000 * GB, 16 * GB, 256) == 6       # the request
    assert replicas.init_threads(0, 31, 3000 * GB, 16 * GB, 256, world=8) == 14  # this rank's share
    assert replicas.init_threads(0, 1, 0, 16 * GB, 1) == 1


def test_union_of_launch_intervals():
    u = replicas.union_ns
    assert u([]) == 0 and u([(0, 10)]) == 10
    assert u([(0, 10), (5, 15), (20, 30)]) == 25      # two overlapping launches of different streams count once
    assert u([(0, 5), (1, 2)]) == 5 and u([(3, 4), (0, 1)]) == 2


def _pmc_md(path, counter_rows):
    with open(path, "w") as f:
        f.write("| kernel | counter | dispatches | sum | mean per dispatch |\n|---|---|---|---|---|\n")
        for k, c, n, sm in counter_rows:
            f.write("| %s | %s | %d | %.1f | %.3f |\n" % (k, c, n, sm, sm / n))


def test_traffic_json_from_the_three_pmc_passes(tmp_path, monkeypatch):
    """tools/traffic_from_pmc.py (bytes read, bytes written, request counts -> profiles/traffic*.json) and bench.pmc_traffic
    (that file -> roofline.traffic / roofline.line_rate): per delivery PASS, a kernel that is not launched in every pass counted
    by its sum over the passes; whole-name patterns; a file of another copy count is not this line's workload"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import importlib
    tfp = importlib.import_module("traffic_from_pmc")
    K = [("k_handel_lane(EngineDev const*, HandelState const*)", 100), ("k_handel_lane2(EngineDev const*, HandelState const*, int)", 110),
         ("void k_handel_dissem<8>(EngineDev const*, HandelState const*)", 5), ("void k_scan2<ExpandF>(EngineDev const*, int const*)", 100),
         ("k_handel_init_sort(HandelState)", 4)]
    _pmc_md(tmp_path / "f.md", [(k, "FETCH_SIZE", n, 1000.0 * n) for k, n in K])        # KB
    _pmc_md(tmp_path / "w.md", [(k, "WRITE_SIZE", n, 500.0 * n) for k, n in K])
    _pmc_md(tmp_path / "r.md", [(k, c, n, q * n) for k, n in K for c, q in (("TCC_EA0_RDREQ_sum", 16000.0), ("TCC_EA0_WRREQ_sum", 8000.0), ("TCC_REQ_sum", 30000.0))])
    line = tmp_path / "bench.json"
    line.write_text("noise\n" + json.dumps({"config": {"replicas_per_gpu": 31}}) + "\n")
    out = tmp_path / "traffic.json"
    monkeypatch.setattr(sys, "argv", ["traffic_from_pmc.py", str(tmp_path / "f.md"), str(tmp_path / "w.md"), "32768", str(line), str(out),
                                      "k_handel_lane,k_handel_lane2,k_handel_dissem<", str(tmp_path / "r.md")])
    tfp.main()
    tj = json.load(open(out))
    assert tj["replicas"] == 31 and tj["nodes"] == 32768
    passes = 110  # (the kernel launched most often defines the passes; dissem's 5 launches count 5 / 110 of a launch each)
    assert abs(tj["fetch_bytes_per_launch_raw"] - 1024.0 * (1000.0 * 100 + 1000.0 * 110 + 1000.0 * 5) / passes) < 1e-6
    assert abs(tj["ea_requests_per_launch"] - 24000.0 * (100 + 110 + 5) / passes) < 1e-6
    assert abs(tj["whole_step_ea_requests"] - 24000.0 * (100 + 110 + 5 + 100)) < 1e-6  # init() kernels are not the step's
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    os.replace(out, tmp_path / "profiles" / "traffic.json")
    t, src, lr, _ = bench.pmc_traffic("traffic.json", 32768, 31, avg_launch_ns=250000.0, step_s=0.8)
    assert t == tj["hbm_bytes_per_launch"] and "traffic.json" in src
    assert abs(lr["requests_per_s"] - tj["ea_requests_per_launch"] / 250e-6) < 1.0 and abs(lr["frac"] - lr["requests_per_s"] / bench.LINE_RATE_CEILING) < 1e-12
    assert abs(lr["whole_step"]["requests_per_s"] - tj["whole_step_ea_requests"] / 0.8) < 1.0
    assert bench.pmc_traffic("traffic.json", 32768, 24, 250000.0)[0] is None    # another copy count: no traffic claimed
    # counters of OTHER code are never this code's traffic: the file carries a hash of wittgenstein_amd/csrc/* and bench.py
    # compares it with the tree it runs from (a file without the stamp is stale by definition)
    assert tj["csrc_sha"] == replicas.csrc_hash()
    for bad in ("0" * 16, None):
        stale = dict(tj)
        if bad is None:
            del stale["csrc_sha"]
        else:
            stale["csrc_sha"] = bad
        json.dump(stale, open(tmp_path / "profiles" / "traffic.json", "w"))
        t2, src2, lr2, _ = bench.pmc_traffic("traffic.json", 32768, 31, avg_launch_ns=250000.0, step_s=0.8)
        assert t2 is None and lr2 is None and src2.startswith("stale: ") and replicas.csrc_hash() in src2
    assert bench.pmc_traffic("traffic_gsf.json", 4096, 256, 1.0) == (None, None, None, None)  # no file


def test_rank_seeds_disjoint():
    seen = set()
    for rank in range(4):
        s = set(replicas.rank_seeds(rank, 4, 5))
        assert len(s) == 5 and not (s & seen)
        seen |= s
    with pytest.raises(ValueError):
        replicas.rank_seeds(4, 4, 1)


DRIVER = r'''
import os, sys, types, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import wittgenstein_amd._lib as L
L.LIB_PATH = os.path.join({root!r}, "tests", "emu", "libwittgpu_emu.so"); L._lib = None
import torch
free = [int({free})]
torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a: None
def mem_get_info(*a):
    import wittgenstein_amd  # every snapshot()ed copy "takes" 100 units of the pretend device
    free[0] -= 100
    return (free[0], int({free}))
torch.cuda.mem_get_info = mem_get_info
import bench
sys.argv = ["bench.py"] + {argv!r}
bench.main()
'''


def run_bench(argv, free=10**6):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")], check=True)
    p = subprocess.run([sys.executable, "-c", DRIVER.format(root=ROOT, argv=argv, free=free)], capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout  # ONE JSON line on stdout
    return json.loads(lines[0]), p.stderr


def test_bench_main_with_the_drivers_argv():
    # (the driver's flags with 8 / 2 instead of its 20 / 5 steps: the emulator runs a step in ten seconds; that the number of
    # steps plays no part in sizing the batch is test_plan_replicas_degrades_and_never_fails's)
    out, _ = run_bench(["--gpus", "1", "--steps", "8", "--warmup", "2", "--nodes", "16", "--replicas", "2", "--batches", "2",
                        "--no-cpu", "--no-second"])
    assert out["steps"] == 8 and out["warmup"] == 2 and out["n_gpus"] == 1
    assert out["config"]["replicas_per_gpu"] == 2 and out["config"]["nodes"] == 16
    assert out["value"] > 0 and out["unit"] == "delivered messages/s" and out["vs_baseline"] is None
    # the same 2 seeds are re-run from their init() image every step: 8 identical steps
    total = round(out["value"] * out["ms_per_step"] * 8 / 1000.0)
    assert total % 8 == 0 and total // 16 == out["config"]["delivered_per_simulation"]
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["launches"] > 0 and 0 < r["frac"] < 1
    assert out["config"]["workload"].startswith("Handel aggregation, 16 nodes")
    assert out["config"]["concurrent_batches"] == 2           # the two copies as two concurrent batches
    a = r["all_streams"]                                      # ... whose delivery launches are merged on one time axis
    assert a["launches"] > r["launches"]                      # (the other seed converges at its own time)
    assert 0 < a["union_ms"] <= a["sum_of_launch_ms"] * 1.0001


def test_bench_lowers_the_batch_instead_of_failing():
    # a pretend device on which one copy (100 units) fits 4 times into 92 % of the free memory: 8 requested -> 4 run
    out, err = run_bench(["--steps", "2", "--warmup", "1", "--nodes", "16", "--replicas", "8", "--no-cpu", "--no-second"],
                         free=600)
    assert out["config"]["replicas_requested"] == 8 and out["config"]["replicas_per_gpu"] == 4
    assert "copies per step instead" in err


SITE = r'''
# test infrastructure (tests/test_bench_cpu.py): every python process of the bench job — the parent, torch.distributed.run
# and its ranks — runs the engine's kernel sources on the CPU wave emulator and sees a pretend torch.cuda
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import wittgenstein_amd._lib as L
L.LIB_PATH = os.path.join({root!r}, "tests", "emu", "libwittgpu_emu.so"); L._lib = None
import torch
free = [10**6]
torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a: None
def mem_get_info(*a):
    free[0] -= 100
    return (free[0], 10**6)
torch.cuda.mem_get_info = mem_get_info
'''


def test_bench_gpus_2_starts_two_ranks_itself(tmp_path):
    """the driver's `python bench.py --gpus 2 ...` with no launcher around it: bench.py starts the two ranks itself
    (torch.distributed.run on 127.0.0.1), rank 0 prints ONE line with n_gpus == 2, the ranks ran disjoint seeds — here over
    gloo on the wave emulator (WG_BENCH_BACKEND=gloo, WG_EMU_DEVICES=2), the timing contract's barrier and reductions real"""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")], check=True)
    (tmp_path / "sitecustomize.py").write_text(SITE.format(root=ROOT))
    env = dict(os.environ, PYTHONPATH=str(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", ""), WG_BENCH_BACKEND="gloo",
               WG_EMU_DEVICES="2")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--nodes", "16", "--replicas", "2", "--no-cpu", "--no-second", "--shard-nodes", "16"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["config"]["replicas_per_gpu"] == 2
    # ... and, beside the replicas' value, ONE simulation split over the two ranks (`sharded_workload`):
    # it delivers what the oracle delivers for seed 0, by whatever split
    sw = out["sharded_workload"]
    assert "error" not in sw and "skipped" not in sw, sw
    assert sw["shards"] == 2 and sw["nodes"] == 16 and sw["scaling"] == "strong" and sw["rank0_node_range"] == [0, 8]
    assert sw["words_by_exchange"]["events"]["int32_words"] > 0 and sw["words_by_exchange"]["outbox"]["calls"] > 0
    assert sw["roofline"]["launches"] > 0 and sw["value"] > 0
    # one rank alone, same argv: seeds 0, 1; the two-rank job ran seeds 0..3 — twice the simulations per step
    one, _ = run_bench(["--gpus", "1"] + argv[2:])
    total2 = round(out["value"] * out["ms_per_step"] * 2 / 1000.0)
    total1 = round(one["value"] * one["ms_per_step"] * 2 / 1000.0)
    assert total2 > total1 and total2 % 2 == 0
    import oracle_lib as o
    import bench
    hp = bench.handel_params(16)
    want = 0
    for seed in range(4):  # what the oracle delivers for those four seeds (RunMultipleTimes loop, C/RunMultipleTimes.java:50-64)
        c = o.Handel(hp["nodeCount"], hp["threshold"], hp["pairingTime"], hp["levelWaitTime"], hp["extraCycle"],
                     hp["disseminationPeriodMs"], hp["fastPath"], hp["nodesDown"], bench.NB, bench.NL, 0, seed=seed)
        while True:
            did = c.run_ms(10)
            if c.info()["time"] >= 20000 or (did and not c.cont_if()):
                break
        want += c.info()["delivered"]
    assert total2 == 2 * want
    c = o.Handel(hp["nodeCount"], hp["threshold"], hp["pairingTime"], hp["levelWaitTime"], hp["extraCycle"],
                 hp["disseminationPeriodMs"], hp["fastPath"], hp["nodesDown"], bench.NB, bench.NL, 0, seed=0)
    while True:
        did = c.run_ms(10)
        if c.info()["time"] >= 20000 or (did and not c.cont_if()):
            break
    assert sw["delivered"] == c.info()["delivered"] and sw["simulated_ms"] == c.info()["time"]
