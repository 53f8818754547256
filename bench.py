#!/usr/bin/env python
"""bench.py — delivered messages/s + simulated ms/s of the core.Network hot path on MI355X.

Workload (BASELINE.json metric): Handel aggregation, 32 768 nodes, 10 % dead,
threshold 0.99 of the live nodes, pairing 4 ms, levelWait 50 ms, period 20 ms, fastPath 10, node builder
RANDOM/constant speed, NetworkLatencyByDistanceWJitter, run as the reference's RunMultipleTimes loop does
(runMs(10) while Handel.newContIf, C/RunMultipleTimes.java:50-64).

A "step" is ONE complete simulation (seed i, as rd.setSeed(i) of C/RunMultipleTimes.java:47) from the
state Protocol.init() leaves to the stop predicate. init() is host work outside the hot path and is done
for all steps before the timed region (the engines sit in HBM: ~17 GB each at 32 768 nodes). Delivered
messages = sum of Node.msgReceived increments (C/Network.java:607-613), simulated ms = network.time.

Multi-GPU (--gpus N, launched by torch.distributed.run): the path's natural parallelism is independent
simulations (RunMultipleTimes seeds), so every rank runs its own K simulations (seeds disjoint across
ranks), no data-path collective, "scaling": "weak".

One JSON line on stdout (rank 0). Everything else goes to stderr.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

NB = "RANDOM_SPEED=CONSTANT_TOR=0.00"
NL = "NetworkLatencyByDistanceWJitter"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def handel_params(n, dead=0.10):
    """HandelScenarios.defaultParams ratios (P/HandelScenarios.java:104-119) with dead ratio 0.10"""
    down = int(n * dead)
    return dict(nodeCount=n, threshold=int(n * (1 - dead) * 0.99), pairingTime=4, levelWaitTime=50, extraCycle=10,
                disseminationPeriodMs=20, fastPath=10, nodesDown=down)


def b_msg(level):
    """algorithmic bytes per delivered level-l SendSigs: 104 fixed + 32 + 3*ceil(2^(l-1)/8)"""
    bits = 1 if level == 0 else 1 << (level - 1)
    return 136 + 3 * ((bits + 7) // 8)


def make_sim(w, n, seed, device):
    hp = handel_params(n)
    p = w.HandelParameters(hp["nodeCount"], hp["threshold"], hp["pairingTime"], hp["levelWaitTime"], hp["extraCycle"],
                           hp["disseminationPeriodMs"], hp["fastPath"], hp["nodesDown"], NB, NL, 0)
    g = w.Handel(p, seed=seed, config={"device": device})
    g.init()
    return g


def run_sim(g, chunk=10, max_ms=20000):
    """RunMultipleTimes inner loop (C/RunMultipleTimes.java:50-64). returns (delivered, simulated_ms, device_wall_ns)"""
    net = g.network()
    delivered = ms = wall = launches = 0
    while g.cont_if() and ms < max_ms:
        net.runMs(chunk)
        st = net.last_stats
        delivered += st["delivered"]
        wall += st["wall_ns"]
        ms += chunk
        launches += chunk + 1
    return delivered, ms, wall


def cpu_baseline(n_sample):
    """the C++ oracle (event-for-event restatement of the single-threaded Java path) on one host core"""
    import oracle_lib as o
    o.build()
    hp = handel_params(n_sample)
    c = o.Handel(hp["nodeCount"], hp["threshold"], hp["pairingTime"], hp["levelWaitTime"], hp["extraCycle"],
                 hp["disseminationPeriodMs"], hp["fastPath"], hp["nodesDown"], NB, NL, 0, seed=0)
    t0 = time.perf_counter()
    while c.cont_if():
        c.run_ms(10)
    dt = time.perf_counter() - t0
    info = c.info(False)
    return {"value": info["delivered"] / dt, "unit": "delivered messages/s", "cores": 1, "kind": "port",
            "sample": "Handel %d nodes (same ratios as the GPU workload, seed 0), full run to the stop predicate: "
                      "%d delivered messages, %d simulated ms in %.2f s on one host core (C++ oracle, upper bound "
                      "on the JVM path)" % (n_sample, info["delivered"], info["time"], dt),
            "simulated_ms_per_s": info["time"] / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nodes", type=int, default=32768)
    ap.add_argument("--cpu-sample-nodes", type=int, default=8192)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import __graft_entry__
    from wittgenstein_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        __graft_entry__.build()
    import wittgenstein_amd as w

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    K, W, n = args.steps, args.warmup, args.nodes
    t_init = time.perf_counter()
    sims = [make_sim(w, n, rank * (K + W) + i, local) for i in range(K + W)]
    init_s = (time.perf_counter() - t_init) / max(1, K + W)
    log("[rank %d] init(): %.1f s per simulation (host, outside the timed region)" % (rank, init_s))

    for g in sims[:W]:
        run_sim(g)
    barrier()
    t0 = time.perf_counter()
    delivered = sim_ms = 0
    by_level = None
    for g in sims[W:]:
        d, ms, _ = run_sim(g)
        delivered += d
        sim_ms += ms
        bl = g.network().delivered_by_level()
        by_level = bl if by_level is None else by_level + bl
    barrier()
    elapsed = time.perf_counter() - t0
    for g in sims[W:]:
        assert not g.cont_if()
    check = int(sims[-1].network().read("msgReceived").sum())
    del sims

    if world > 1:
        tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cnt = torch.tensor([delivered, sim_ms], device="cuda", dtype=torch.int64)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        delivered, sim_ms = int(cnt[0].item()), int(cnt[1].item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    out = {
        "metric": "delivered messages/sec (Handel 32k nodes; simulated-ms/sec alongside)",
        "value": delivered / elapsed, "unit": "delivered messages/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": elapsed * 1000.0 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "simulated_ms_per_s": sim_ms / elapsed,
        "config": {"workload": "Handel aggregation, %d nodes, 10%% dead, threshold 0.99*live, pairing 4, levelWait 50, "
                               "period 20, fastPath 10, RANDOM nodes, NetworkLatencyByDistanceWJitter; runMs(10) until "
                               "Handel.newContIf is false; one simulation per step, seeds 0..K-1 per rank" % n,
                   "nodes": n, "simulations_per_rank": K, "parallelism": "independent simulations (replicas) per GPU",
                   "delivered_per_step": delivered // max(1, K * world), "init_s_per_simulation": init_s},
    }

    # ---- roofline of the dominant kernel: HIP events on the engine's stream around every kernel of the
    # per-ms pipeline, over one more simulation of seed 0 (same launches as timed step 0).
    if not args.no_profile:
        g = make_sim(w, n, 0, local)
        g.network().profile(True)
        d, ms, _ = run_sim(g)
        prof = g.network().profile_read()
        bl = g.network().delivered_by_level()
        del g
        alg_bytes = float(sum(int(c) * b_msg(l) for l, c in enumerate(bl)))
        dk = prof["deliver"]
        per_launch_bytes = alg_bytes / max(1, dk["spans"])
        avg_ns = dk["total_ns"] / max(1, dk["spans"])
        achieved = per_launch_bytes / avg_ns  # bytes/ns == GB/s
        total_ns = sum(v["total_ns"] for v in prof.values())
        out["roofline"] = {
            "bound": "hbm", "kernel": "k_deliver_handel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_launch": per_launch_bytes, "avg_launch_us": avg_ns / 1000.0,
            "launches": dk["spans"], "bytes_per_delivered_message": alg_bytes / max(1, d),
            "whole_pipeline_achieved_GBs": alg_bytes / total_ns,
            "phase_device_ms": {k: round(v["total_ns"] / 1e6, 3) for k, v in prof.items()},
        }
    if world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample_nodes)
    log("msgReceived sum of the last simulation: %d" % check)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
