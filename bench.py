#!/usr/bin/env python
"""bench.py — delivered messages/s + simulated ms/s of the core.Network hot path on MI355X.

Workload (BASELINE.json metric): Handel aggregation, 32 768 nodes, 10 % dead,
threshold 0.99 of the live nodes, pairing 4 ms, levelWait 50 ms, period 20 ms, fastPath 10, node builder
RANDOM/constant speed, NetworkLatencyByDistanceWJitter, run exactly as the reference runs simulations:
C/RunMultipleTimes.java:44-64 — for each of runCount copies rd.setSeed(i); init(); runMs(10) while
Handel.newContIf holds.

A "step" is ONE RunMultipleTimes pass over R = --replicas independent 32 768-node simulations (seeds
distinct), advanced in lock-step by a wg_batch (one launch sequence per simulated ms for all R, each copy
stopping at its own predicate). init() is host work outside the hot path: every copy is initialised ONCE
before the first step, its init() image kept on the device (wg_snapshot), and every step starts from the
restored image (wg_restore — the same R seeds every step), so `--steps 20 --warmup 5` costs 25 runs and R
init()s, not 25 R. Each step's pass is timed on its own (barrier + synchronize on both sides) and the K
brackets are summed; the restores sit between the brackets. If R copies (+ images) do not fit the free HBM
the batch is lowered to what fits and the line says so — a size check never aborts the run. Delivered
messages = sum of Node.msgReceived increments (C/Network.java:607-613), simulated ms = sum over copies of
network.time.

Multi-GPU (--gpus N): the copies shard across ranks with no data-path collective (every rank runs its own R copies,
seeds disjoint), "scaling": "weak". Under a launcher (torch.distributed.run: WORLD_SIZE set) every process is one rank;
a bare `python bench.py --gpus N` starts the N ranks itself (launch_ranks) and rank 0 prints the line with n_gpus = N.

--mode shard (not the default): a step is ONE simulation whose nodes are split by id range over
the N ranks (wg_shard_configure; RCCL all-reduces per simulated ms through wittgenstein_amd/shards.py),
"scaling": "strong". With --gpus 1 it measures what the sharded pipeline costs on one GPU.

One JSON line on stdout (rank 0). Everything else goes to stderr. At --gpus 1 the line also carries a
"second_workload" object, outside `value` and outside the timed region: BASELINE configs[4]'s protocol at its node
count — Casper IMD resident, 262 150 nodes, 10 % of the attesters stopped, 24 simulated seconds per step — with its
own metric / roofline / cpu_baseline fields (--workload casper prints that object alone; --no-second skips it).
"""
import argparse
import gc
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# stdout carries exactly one JSON line: native libraries print there too (RCCL's version banner at the first
# communicator), so file descriptor 1 points at stderr for the whole run and the line goes to a saved copy
_REAL_STDOUT = os.dup(1)
os.dup2(2, 1)


def _oracle():
    """the CPU oracle's binding (tests/oracle_lib.py): cpu_baseline legs only — the product path never loads it"""
    t = os.path.join(ROOT, "tests")
    if t not in sys.path:
        sys.path.append(t)
    import oracle_lib
    oracle_lib.build()
    return oracle_lib

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


def gsf_params(n):
    """BASELINE.json configs[1] / SURVEY.md §8d config 2: the GSFSignatureParameters() defaults
    (P/GSFSignature.java:47-57) scaled to n nodes"""
    return dict(nodeCount=n, threshold=int(n * 0.99), pairingTime=3, timeoutPerLevelMs=50, periodDurationMs=10,
                acceleratedCallsCount=10, nodesDown=0)


ENGINE_CONFIG = {}


def make_sim(w, n, seed, device, workload="handel"):
    if workload == "gsf":
        gp = gsf_params(n)
        g = w.GSFSignature(w.GSFSignatureParameters(gp["nodeCount"], gp["threshold"], gp["pairingTime"],
                                                    gp["timeoutPerLevelMs"], gp["periodDurationMs"],
                                                    gp["acceleratedCallsCount"], gp["nodesDown"], NB, NL),
                           seed=seed, config={"device": device})
        g.init()
        return g
    hp = handel_params(n)
    p = w.HandelParameters(hp["nodeCount"], hp["threshold"], hp["pairingTime"], hp["levelWaitTime"], hp["extraCycle"],
                           hp["disseminationPeriodMs"], hp["fastPath"], hp["nodesDown"], NB, NL, 0)
    cfg = {"device": device}
    cfg.update(ENGINE_CONFIG)  # (--engine-config: wg_config fields, e.g. queue_cap_wide=12)
    g = w.Handel(p, seed=seed, config=cfg)
    g.init()
    return g


def make_batch(w, n, seeds, device, threads, workload="handel"):
    """Protocol.copy() + rd.setSeed(i) + init() for every copy (host work; ctypes releases the GIL)."""
    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        sims = list(ex.map(lambda s: make_sim(w, n, s, device, workload), seeds))
    return sims, w.Batch([g.network() for g in sims])


def split_batches(w, sims, k):
    """the copies of one step as k batches (contiguous slices), each led by its first member's stream"""
    k = max(1, min(k, len(sims)))
    per = (len(sims) + k - 1) // k
    return [w.Batch([g.network() for g in sims[i:i + per]]) for i in range(0, len(sims), per)]


def _cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(n_sample, workload="handel", budget_s=25.0, all_cores=True):
    """the C++ oracle (event-for-event restatement of the single-threaded Java path) on one host core"""
    o = _oracle()
    if workload == "gsf":
        gp = gsf_params(n_sample)
        c = o.GSFSignature(gp["nodeCount"], gp["threshold"], gp["pairingTime"], gp["timeoutPerLevelMs"],
                           gp["periodDurationMs"], gp["acceleratedCallsCount"], gp["nodesDown"], NB, NL, seed=0)
        t0 = time.perf_counter()
        while c.cont_if():
            c.run_ms(10)
        dt = time.perf_counter() - t0
        info = c.info(False)
        return {"value": info["delivered"] / dt, "unit": "delivered messages/s", "cores": 1, "kind": "port",
                "sample": "GSFSignature %d nodes (seed 0), full run to the stop predicate: %d delivered messages, %d "
                          "simulated ms in %.2f s on one host core (C++ oracle, upper bound on the JVM path)"
                          % (n_sample, info["delivered"], info["time"], dt),
                "simulated_ms_per_s": info["time"] / dt, "nproc": len(os.sched_getaffinity(0)), "cpu_model": _cpu_model(),
                "all_cores": {"skipped": "the side workload's baseline is the one-core figure"}}
    hp = handel_params(n_sample)
    c = o.Handel(hp["nodeCount"], hp["threshold"], hp["pairingTime"], hp["levelWaitTime"], hp["extraCycle"],
                 hp["disseminationPeriodMs"], hp["fastPath"], hp["nodesDown"], NB, NL, 0, seed=0)
    # A BOUNDED sample of the metric's own configuration: the run from t = 0 for `budget_s` seconds of one core (or to the stop
    # predicate, whichever comes first) — the whole 32 768-node run is 8 minutes of one core (tests/golden/make_golden.py).
    # init() is outside the bracket, as it is on the GPU side.
    t0 = time.perf_counter()
    while c.cont_if() and time.perf_counter() - t0 < budget_s:
        c.run_ms(10)
    dt = time.perf_counter() - t0
    info = c.info(False)
    whole = not c.cont_if()
    out = {"value": info["delivered"] / dt, "unit": "delivered messages/s", "cores": 1, "kind": "port",
           "sample": "Handel %d nodes (the GPU workload's parameters, seed 0), %s: %d delivered messages, %d simulated ms "
                     "in %.2f s on one host core (C++ oracle, upper bound on the JVM path; init() outside the bracket)"
                     % (n_sample, "full run to the stop predicate" if whole else
                        "the first %d simulated ms of the run (bounded at %.0f s of CPU work)" % (info["time"], budget_s),
                        info["delivered"], info["time"], dt),
           "simulated_ms_per_s": info["time"] / dt, "sample_nodes": n_sample, "sample_simulated_ms": info["time"]}
    del c
    gc.collect()
    # the only parallelism the reference admits (C/RunMultipleTimes.java:44-48): independent seeds, one per core
    try:
        avail = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) * 1024
        # every host core, memory permitting — and bounded: one oracle copy of n nodes holds ~ 64 n^2 bytes at its peak (ranks,
        # emission lists as pointers, seven N-bit sets per level and node, the in-flight messages' clones: 18 GB at 16 384
        # nodes); 256 of them took a 3 TB box down. A quarter of what is available, 64 copies at most
        cores = max(1, min(len(os.sched_getaffinity(0)), 64, int(0.25 * avail / (64.0 * n_sample * n_sample + (1 << 30)))))
    except Exception:
        cores = 1
    out["nproc"] = len(os.sched_getaffinity(0))
    out["cpu_model"] = _cpu_model()
    if not (cores > 1 and all_cores):
        out["all_cores"] = {"skipped": "one oracle copy of %d nodes holds ~ %.0f GB at its peak and a quarter of the host's available "
                                       "memory holds %d of them" % (n_sample, 64.0 * n_sample * n_sample / 1e9, cores)
                            if all_cores else "not requested"}
    if cores > 1 and all_cores:
        def one(seed):
            cc = o.Handel(hp["nodeCount"], hp["threshold"], hp["pairingTime"], hp["levelWaitTime"], hp["extraCycle"],
                          hp["disseminationPeriodMs"], hp["fastPath"], hp["nodesDown"], NB, NL, 0, seed=seed)
            t1 = time.perf_counter()
            while cc.cont_if() and time.perf_counter() - t1 < budget_s:
                cc.run_ms(10)
            return cc.info(False)["delivered"], t1, time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:  # (ctypes releases the GIL inside the oracle)
            res = list(ex.map(one, range(1, cores + 1)))
        wall = max(r[2] for r in res) - min(r[1] for r in res)
        out["all_cores"] = {"value": sum(r[0] for r in res) / wall, "unit": "delivered messages/s", "cores": cores,
                            "sample": "%d independent seeds of the same %d-node run (each bounded at %.0f s), one per core, run loops only"
                                      % (cores, n_sample, budget_s)}
    return out


# Memory-side (EA) requests per second the chip sustains for SCATTERED lines mixed 53 : 47 reads : writes as the delivery pass
# mixes them — tools/micro/line_rate_probe (one lane = one random 64-byte line per access, 16-byte loads, 16-byte stores, full
# occupancy, 128 GiB footprint), counted by the same TCC_EA0_RDREQ / WRREQ counters: 36.2 G/s (19.3 G reads + 16.9 G writes);
# scattered reads alone 47.2 G/s, 16-byte writes alone 29.1 G/s (profiles/r24m_line_rate_probe.txt). Rounds 1-5 used mlp_probe's
# 21 G/s — eight lanes per line, reads only — which the pass exceeded: no ceiling.
LINE_RATE_CEILING = 36.2e9
LINE_RATE_CEILING_READS = 47.2e9


def pmc_traffic(name, n, R, avg_launch_ns, step_s=None):
    """(traffic, traffic_source, line_rate, json) of the delivery pass from profiles/<name> — the rocprofv3 PMC passes of the same
    workload (tools/gpu_final_round.sh); None where the file is of another node / copy count. `line_rate`: the memory-side
    REQUESTS per second of the pass (TCC_EA0_RDREQ + WRREQ of a request-count pass over the HIP-event duration measured here)
    against the scattered-line ceiling — `frac` near 1 says the pass runs at the rate the chip serves scattered lines, whatever
    its bytes are against 8 TB/s."""
    tpath = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(tpath):
        return None, None, None, None
    tj = json.load(open(tpath))
    if tj.get("replicas") != R or tj.get("nodes") != n:
        return None, "profiles/%s is of %s copies of %s nodes: not this line's workload" % (name, tj.get("replicas"), tj.get("nodes")), None, None
    # the counters must be of THIS code: traffic_from_pmc.py stamps a hash of wittgenstein_amd/csrc/* (a file without the stamp
    # is of an earlier round's code by definition)
    from wittgenstein_amd.replicas import csrc_hash
    have, mine = tj.get("csrc_sha"), csrc_hash()
    if have != mine:
        return None, "stale: profiles/%s was taken on csrc %s (commit %s), this tree is csrc %s — run tools/gpu_final_round.sh" % (
            name, have or "unstamped", tj.get("commit", "unknown"), mine), None, None
    src = "profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload's `bench.py --steps 1`, commit %s%s" % (
        name, tj.get("commit", "unknown"), " (this session)" if os.environ.get("WG_TRAFFIC_SESSION") else "")
    lr = None
    if tj.get("ea_requests_per_launch") and avg_launch_ns > 0:
        q = float(tj["ea_requests_per_launch"])
        rq, wq = float(tj.get("ea_read_requests_per_launch") or 0.0), float(tj.get("ea_write_requests_per_launch") or 0.0)
        lr = {"requests_per_launch": q, "read_requests_per_launch": rq, "write_requests_per_launch": wq,
              "requests_per_s": q / (avg_launch_ns * 1e-9), "ceiling_requests_per_s": LINE_RATE_CEILING,
              "frac": q / (avg_launch_ns * 1e-9) / LINE_RATE_CEILING,
              "read_requests_per_s": rq / (avg_launch_ns * 1e-9), "read_frac": rq / (avg_launch_ns * 1e-9) / LINE_RATE_CEILING_READS,
              "write_requests_per_s": wq / (avg_launch_ns * 1e-9),
              "bytes_per_request": float(tj["hbm_bytes_per_launch"]) / q,
              "source": "TCC_EA0_RDREQ_sum + TCC_EA0_WRREQ_sum (a third PMC pass) per delivery pass / this run's HIP-event duration of "
                        "the pass; ceiling: tools/micro/line_rate_probe under the same counters, scattered 64-byte lines at the pass's "
                        "53 : 47 read : write mix, full occupancy (profiles/r24m_line_rate_probe.txt: 36.2 G requests/s; reads alone "
                        "47.2 G/s, which `read_frac` sets the pass's reads against)"}
        if step_s and tj.get("whole_step_ea_requests"):
            lr["whole_step"] = {"requests_per_step": tj["whole_step_ea_requests"], "requests_per_s": tj["whole_step_ea_requests"] / step_s,
                                "frac": tj["whole_step_ea_requests"] / step_s / LINE_RATE_CEILING}
    return float(tj["hbm_bytes_per_launch"]), src, lr, tj


def replicas_line(workload, n, R_req, K, W, local=0):
    """a compact RunMultipleTimes line for ONE more workload inside the default run (outside `value` and its timed region):
    R copies initialised once, kept as init() images, W warm-up + K timed passes, the delivery pass bracketed by HIP events
    as in the main line. Used for Handel at the north star's target size (65 536 nodes) and GSFSignature (BASELINE configs[1])."""
    import torch
    import numpy as np
    import wittgenstein_amd as w
    from wittgenstein_amd import replicas
    free0 = torch.cuda.mem_get_info()[0]
    t_init = time.perf_counter()
    def init_one(sd):
        g = make_sim(w, n, sd, local, workload)
        g.network().snapshot()
        return g
    sims = [init_one(0)]
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    per_copy, once = max(1, free0 - free1), 0
    if R_req > 1:  # what a FURTHER copy takes (the first one also pays what the process allocates once: code objects, tables)
        sims.append(init_one(1))
        torch.cuda.synchronize()
        per_copy, once = replicas.marginal_copy_bytes(free0 - free1, free1 - torch.cuda.mem_get_info()[0])
    R = replicas.plan_replicas(R_req, free0 - once, per_copy, transient_bytes=0 if workload == "gsf" else replicas.handel_init_transient_bytes(n))
    if R > 2:
        with ThreadPoolExecutor(max_workers=min(R - 2, max(1, len(os.sched_getaffinity(0)) - 1))) as ex:
            sims += list(ex.map(init_one, range(2, R)))
    sims = sims[:R]
    batch = w.Batch([g.network() for g in sims])
    init_wall = time.perf_counter() - t_init
    delivered = sim_ms = 0
    elapsed = dk_ns = 0.0
    dk_spans = 0
    by_level = None
    for i in range(W + K):
        if i > 0:
            for g in sims:
                g.network().restore()
        sims[0].network().profile(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d, ms = batch.run_multiple_times(chunk=10, maxTime=20000)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if i < W:
            continue
        elapsed += dt
        delivered += sum(d)
        sim_ms += sum(ms)
        pr = sims[0].network().profile_read()["deliver"]
        dk_spans += pr["spans"]
        dk_ns += pr["total_ns"]
        for g in sims:
            bl = g.network().delivered_by_level()
            by_level = bl if by_level is None else by_level + bl
    del batch, sims
    gc.collect()
    gsf = workload == "gsf"
    alg = float(sum(int(c) * b_msg(l) for l, c in enumerate(by_level))) if by_level is not None else 0.0
    per_launch = alg / max(1, dk_spans)
    avg_ns = dk_ns / max(1, dk_spans)
    traffic, traffic_source, line_rate, _ = pmc_traffic("traffic_gsf.json" if gsf else "traffic_handel%d.json" % n, n, R, avg_ns,
                                                        elapsed / max(1, K))
    return {
        "metric": "delivered messages/sec (%s; simulated-ms/sec alongside)" % ("GSFSignature" if gsf else "Handel %dk nodes" % (n // 1024)),
        "value": delivered / max(elapsed, 1e-9), "unit": "delivered messages/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": elapsed * 1000.0 / max(1, K), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic", "simulated_ms_per_s": sim_ms / max(elapsed, 1e-9),
        "config": {"workload": ("GSFSignature, %d nodes, threshold 0.99, pairing 3, timeoutPerLevel 50, period 10, accelerated calls 10"
                                if gsf else "Handel aggregation, %d nodes, 10%% dead, threshold 0.99*live, pairing 4, levelWait 50, period 20, "
                                            "fastPath 10") % n + ", RANDOM nodes, NetworkLatencyByDistanceWJitter; RunMultipleTimes: %d independent "
                               "copies per step (seeds 0..%d), runMs(10) until each copy's continuation predicate is false" % (R, R - 1),
                   "nodes": n, "replicas_per_gpu": R, "replicas_requested": R_req, "hbm_bytes_per_copy_incl_init_image": int(per_copy),
                   "delivered_per_simulation": delivered // max(1, K * R), "init_wall_s": init_wall},
        "roofline": {"bound": "hbm", "kernel": "k_gsf_docycle + k_gsf_lane + k_deliver_inbox<GsfProto> (the delivery pass)" if gsf else
                     "k_handel_lane + k_handel_update + k_handel_lane2 + k_handel_copy + k_handel_dissem + k_handel_wave (the delivery pass)",
                     "achieved": per_launch / max(1.0, avg_ns), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": per_launch / max(1.0, avg_ns) / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "line_rate": line_rate,
                     "algorithmic_bytes_per_launch": per_launch, "avg_launch_us": avg_ns / 1000.0, "launches": dk_spans,
                     "bytes_per_delivered_message": alg / max(1, delivered),
                     "whole_step_frac": alg / (max(elapsed, 1e-9) * 1e9) / HBM_PEAK_GBS},
    }


def main_casper(args):
    emit(casper_line(args))


def casper_line(args):
    """BASELINE configs[4]'s protocol at its node count, on one GPU: Casper IMD, cycleLength 64, 5 block producers,
    --attesters-per-round (4096) attesters voting per slot = 262 150 nodes (every vote and block is a sendAll to all N
    nodes); --casper-stopped 0.1 stops 10 % of the attesters after init(). A step is one simulation of --casper-ms
    simulated ms. Not the BASELINE metric's workload (that is Handel): a second line."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("--workload casper is a one-GPU line")
    from wittgenstein_amd import protocols as P
    K, W, per, T = args.steps, args.warmup, args.attesters_per_round, args.casper_ms
    cl, bp = args.casper_cycle_length, args.casper_producers
    rot = bool(getattr(args, "casper_random_on_ties", False))
    params = (cl, rot, bp, per, 1000, 1)
    n = 1 + bp + cl * per
    delivered = 0
    elapsed = dk_ns = 0.0
    dk_spans = 0
    ks = getattr(args, "casper_shards", 0)
    shard_traffic = None
    for step in range(W + K):
        from wittgenstein_amd import shards
        grp = shards.LoopbackGroup(ks) if ks > 1 else None
        cfgs = [grp.config(sh) for sh in range(ks)] if grp else [shards.config_rccl() if ks == 1 else None]
        sims = []
        stopped_ids = []
        for cfg in cfgs:  # (k logical shards: k engines, each holding the rows of its node range; the same simulation)
            g = P.CasperIMD(P.CasperParemeters(*params, NB, NL), seed=step, max_slots=T // 8000 + 2, config=cfg)
            g.init()
            if args.casper_stopped > 0:  # config 5's "+10 %" (SURVEY.md §8d): attesters stop()ped after init()
                stopped_ids = g.stop_attesters(int(args.casper_stopped * cl * per), seed=step)
            sims.append(g)
        net = sims[0].network()
        net.profile(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if grp:
            grp.run(lambda sh: sims[sh].network().runMs(T))
        else:
            net.runMs(T)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if step >= W:
            delivered += net.last_stats["delivered"]
            elapsed += dt
            # (outside the timed region) the observables of PT/CasperIMDTest.java:263-274; sharded: a shard reports its own rows
            heights = sum(x.network().read("headHeight") for x in sims)
            if ks:
                shard_traffic = shards.traffic(net)
            import numpy as np
            live = np.ones(len(heights), bool)
            live[stopped_ids] = False  # (a stopped node receives nothing: its head stays the genesis block)
            observer_height, min_height = int(heights[0]), int(heights[live].min())
            pr = net.profile_read()["deliver"]
            dk_spans += pr["spans"]
            dk_ns += pr["total_ns"]
        del g, net, sims, grp
        gc.collect()
    ctraffic = None
    ctpath = os.path.join(ROOT, "profiles", "traffic_casper.json")  # per-launch HBM bytes of the delivery pass (rocprofv3 PMC passes)
    if os.path.exists(ctpath) and not ks:
        tj = json.load(open(ctpath))
        if tj.get("nodes") == n and tj.get("stopped_fraction", 0.0) == args.casper_stopped:
            ctraffic = tj.get("hbm_bytes_per_launch")
    bmsg = 104 + 24 + 8  # SURVEY.md §8d fixed part + the attestation's three bit-sets (8-byte RMW each) + attHead read
    alg = float(delivered) * bmsg
    avg_ns = dk_ns / max(1, dk_spans)
    out = {
        "metric": "delivered messages/sec (Casper IMD; simulated-ms/sec alongside)",
        "value": delivered / elapsed, "unit": "delivered messages/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": elapsed * 1000.0 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic", "simulated_ms_per_s": K * T / elapsed,
        "config": {"workload": ("Casper IMD, %d nodes (1 observer, %d block producers, %d x %d attesters), randomOnTies %s, "
                                "block / attestation construction 1000 / 1 ms, RANDOM nodes, NetworkLatencyByDistanceWJitter, "
                                "%d simulated ms per step%s") % (
                       n, bp, cl, per, "true (the events that can call best() are delivered by one wavefront in global event order, "
                                       "k_casper_seq)" if rot else "false", T,
                       "" if args.casper_stopped <= 0 else ", %d attesters stop()ped after init()" % int(args.casper_stopped * cl * per)),
                   "nodes": n, "observer_head_height_at_end": observer_height, "lowest_head_height_of_a_live_node_at_end": min_height,
                   "parallelism": "one simulation, unsharded" if not ks else
                                  ("one simulation on the node-range sharded pipeline: one rank through the engine's own RCCL communicator"
                                   if ks == 1 else "one simulation as %d logical node-range shards on this GPU (in-process loopback "
                                                   "all-reduce; the delivery-pass bracket is shard 0's)" % ks),
                   "allreduce_calls_and_int32_words_per_simulation": shard_traffic},
        "roofline": {"bound": "hbm", "kernel": "the delivery pass: k_casper_classify + k_casper_attestations + k_deliver<CasperProto> (one launch of each per simulated ms that is not skipped)", "achieved": (alg / max(1, dk_spans)) / max(1.0, avg_ns),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (alg / max(1, dk_spans)) / max(1.0, avg_ns) / HBM_PEAK_GBS,
                     "traffic": ctraffic, "algorithmic_bytes_per_launch": alg / max(1, dk_spans), "avg_launch_us": avg_ns / 1000.0,
                     "launches": dk_spans, "bytes_per_delivered_message": bmsg, "whole_run_achieved_GBs": alg / (elapsed * 1e9)},
    }
    if not args.no_cpu:
        o = _oracle()
        # the oracle keeps every attestation in every node's HashSet: a bounded sample (about 10 s of one core)
        sample = min(per, 512, max(16, 16384 // cl))
        c = o.CasperIMD((cl, False, bp, sample, 1000, 1), NB, NL, seed=0)
        if args.casper_stopped > 0:
            c.stop(P.choose_attesters(range(1 + bp, 1 + bp + cl * sample), int(args.casper_stopped * cl * sample), seed=0))
        t0 = time.perf_counter()
        c.run_ms(T)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": c.info()["delivered"] / dt, "unit": "delivered messages/s", "cores": 1, "kind": "port",
                               "sample": "Casper IMD with %d attesters per round (%d nodes), %d simulated ms: %d delivered "
                                         "messages in %.2f s on one host core (C++ oracle)" % (sample, 1 + bp + cl * sample, T, c.info()["delivered"], dt)}
    return out


def sharded_workload(args, dist, rank, world, local, on_gpu):
    """The `sharded_workload` object of a `--gpus N > 1` line: ONE Handel simulation split by node range
    over the N ranks — what BASELINE config 4 / the north star's "RCCL all-to-all per simulated ms" describe — beside the line's
    own value (independent replicas). Every rank calls this; rank 0 gets the object. One timed RunMultipleTimes pass, bracketed
    as the main line (barrier + synchronize on both sides, MAX over ranks), exchanges over the ENGINE'S OWN RCCL communicator
    (`--shard-callback`: torch.distributed through the callback hooks; the gloo CPU test always). Nodes: --shard-nodes, default
    32 768 — a sharded engine takes init() from the host, and config 4's 131 072 nodes are 279 s and 128 GiB of host memory per
    rank there (profiles/r08i_*): `--shard-nodes 131072` runs it, the default line must finish in minutes.
    The whole attempt runs under a watchdog (WG_BENCH_SHARD_TIMEOUT seconds, default 420): a collective that never completes
    must not take the replicas line with it — the object then says {"skipped": ...} and the caller leaves without tearing the
    process group down."""
    import threading
    import torch
    import wittgenstein_amd as w
    from wittgenstein_amd import shards
    n = args.shard_nodes
    box = {}

    def work():
        try:
            hp = handel_params(n)
            hparams = w.HandelParameters(hp["nodeCount"], hp["threshold"], hp["pairingTime"], hp["levelWaitTime"], hp["extraCycle"],
                                         hp["disseminationPeriodMs"], hp["fastPath"], hp["nodesDown"], NB, NL, 0)
            callback = args.shard_callback or not on_gpu
            scfg = shards.config(dist, device=local, device_memory=on_gpu) if callback else shards.config_rccl(dist, device=local)
            t0 = time.perf_counter()
            g = w.Handel(hparams, seed=0, config=scfg)
            g.init()
            init_s = time.perf_counter() - t0
            g.network().profile(2)
            dist.barrier()
            if on_gpu:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            d, ms = shards.run_multiple_times(dist, g, chunk=10, maxTime=20000)
            dist.barrier()
            if on_gpu:
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            tt = torch.tensor([dt], device="cuda" if on_gpu else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
            pr = g.network().profile_read()["deliver"]
            by_level = g.network().delivered_by_level()  # (replicated: the whole network's histogram)
            alg = float(sum(int(c) * b_msg(l) for l, c in enumerate(by_level)))
            avg_ns = pr["total_ns"] / max(1, pr["spans"])
            calls, words = shards.traffic(g.network())
            lo, hi = shards.shard_range(g.network())
            box["out"] = {
                "metric": "delivered messages/sec (Handel %s nodes, ONE simulation sharded by node range; simulated-ms/sec alongside)"
                          % ("32k" if n == 32768 else "%d" % n),
                "value": d / dt, "unit": "delivered messages/s", "n_gpus": world, "shards": world, "nodes": n, "steps": 1, "warmup": 0,
                "ms_per_step": dt * 1000.0, "higher_is_better": True, "scaling": "strong", "simulated_ms_per_s": ms / dt,
                "delivered": int(d), "simulated_ms": int(ms), "init_s": init_s, "rank0_node_range": [lo, hi],
                "transport": ("torch.distributed (%s) through wg_allreduce_fn / wg_alltoallv_fn" % dist.get_backend()) if callback
                             else "the engine's own RCCL communicator (wg_shard_configure_rccl): ncclAllReduce + grouped ncclSend / ncclRecv on the engine's stream",
                "words_by_exchange": {k: {"calls": int(c), "int32_words": int(wd)} for k, (c, wd) in shards.traffic_by_exchange(g.network()).items()},
                "collective_calls": int(calls), "int32_words_received_by_rank0": int(words),
                "roofline": {"bound": "hbm", "kernel": "the delivery pass (k_handel_lane + k_handel_update + k_handel_lane2 + k_handel_copy + "
                                                       "k_handel_dissem + k_handel_wave) on rank 0's node range",
                             "achieved": (alg / world / max(1, pr["spans"])) / max(1.0, avg_ns), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": (alg / world / max(1, pr["spans"])) / max(1.0, avg_ns) / HBM_PEAK_GBS, "traffic": None,
                             "avg_launch_us": avg_ns / 1000.0, "launches": pr["spans"], "whole_run_achieved_GBs": alg / (dt * 1e9)},
                "note": "not part of `value`: the line's value is N ranks of independent copies (weak scaling); this is one simulation's "
                        "rate when its nodes are split over the N GPUs (capacity, DESIGN.md §7.2)"}
            del g
        except BaseException as x:  # noqa: B036 — reported in the object, the replicas line stands on its own
            box["err"] = "%s: %s" % (type(x).__name__, x)

    limit = float(os.environ.get("WG_BENCH_SHARD_TIMEOUT", "420"))
    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(limit)
    if th.is_alive():
        return {"skipped": "no answer within %.0f s (a collective or an init() that did not complete)" % limit}, True
    if "err" in box:
        return {"error": box["err"]}, True  # (the other ranks may be waiting in a collective: leave without a teardown)
    return box["out"], False


def main_shard(args):
    """one simulation per step, sharded by node range over the ranks (strong scaling)"""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29581")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    import wittgenstein_amd as w
    from wittgenstein_amd import shards
    K, W, n = args.steps, args.warmup, args.nodes
    hp = handel_params(n)
    delivered = sim_ms = 0
    elapsed = 0.0
    dk_spans = dk_ns = 0
    by_level = None
    traffic = (0, 0)
    kl = args.logical_shards if world == 1 else 0
    hparams = w.HandelParameters(hp["nodeCount"], hp["threshold"], hp["pairingTime"], hp["levelWaitTime"],
                                 hp["extraCycle"], hp["disseminationPeriodMs"], hp["fastPath"], hp["nodesDown"], NB, NL, 0)
    for step in range(W + K):  # every rank builds the SAME simulation (seed = step) and owns a node range of it
        if kl > 0:  # k shards on this one GPU, one host thread each; RunMultipleTimes' loop driven from here
            grp = shards.LoopbackGroup(kl)
            sims = []
            for sh in range(kl):
                sims.append(w.Handel(hparams, seed=step, config=grp.config(sh, device=local)))
                sims[-1].init()
            g = sims[0]
            g.network().profile(2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d = ms = 0
            while True:
                did = grp.run(lambda sh: sims[sh].network().runMs(10))[0]
                d += g.network().last_stats["delivered"]
                ms += 10
                if not (g.network().time < 20000 and (not did or any(x.cont_if() for x in sims))):
                    break
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        else:
            sims = None
            # the engine's own RCCL communicator (wg_shard_configure_rccl): the unique id comes from rank 0's engine
            # library and travels once over torch.distributed; --shard-callback keeps the caller-supplied collective
            scfg = shards.config(dist, device=local) if args.shard_callback else shards.config_rccl(dist, device=local)
            g = w.Handel(hparams, seed=step, config=scfg)
            g.init()
            g.network().profile(2)
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d, ms = shards.run_multiple_times(dist, g, chunk=10, maxTime=20000)
            dist.barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        if step >= W:
            delivered += d
            sim_ms += ms
            elapsed += dt
            pr = g.network().profile_read()["deliver"]
            dk_spans += pr["spans"]
            dk_ns += pr["total_ns"]
            bl = g.network().delivered_by_level()   # replicated: the whole network's histogram
            by_level = bl if by_level is None else by_level + bl
            traffic = shards.traffic(g.network())
        del g, sims
        gc.collect()
    tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    if rank == 0:
        alg_bytes = float(sum(int(c) * b_msg(l) for l, c in enumerate(by_level)))
        avg_ns = dk_ns / max(1, dk_spans)
        out = {
            "metric": "delivered messages/sec (Handel %s nodes, ONE simulation sharded by node range; simulated-ms/sec alongside)"
                      % ("32k" if n == 32768 else "%d" % n),
            "value": delivered / elapsed, "unit": "delivered messages/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed * 1000.0 / K, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic", "simulated_ms_per_s": sim_ms / elapsed,
            "config": {"workload": "Handel aggregation, %d nodes, 10%% dead, threshold 0.99*live, pairing 4, levelWait 50, "
                                   "period 20, fastPath 10, RANDOM nodes, NetworkLatencyByDistanceWJitter; ONE simulation "
                                   "per step, nodes split by id range over %d shard(s)%s, runMs(10) until Handel.newContIf is "
                                   "false" % (n, kl if kl else world, " on one GPU (in-process loopback all-reduce)" if kl else ""),
                       "nodes": n, "parallelism": "node-range shards of one simulation (wg_shard_configure%s)" % ("" if (kl or args.shard_callback) else "_rccl: engine-owned RCCL communicator"), "shards": kl if kl else world,
                       "allreduce_calls_per_simulation": traffic[0], "allreduce_int32_words_per_simulation": traffic[1]},
            "roofline": {"bound": "hbm", "kernel": "k_handel_lane + k_handel_copy + k_handel_update + k_handel_wave on rank 0's node range",
                         "achieved": (alg_bytes / (kl if kl else world) / max(1, dk_spans)) / max(1.0, avg_ns), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": (alg_bytes / (kl if kl else world) / max(1, dk_spans)) / max(1.0, avg_ns) / HBM_PEAK_GBS,
                         "traffic": None, "avg_launch_us": avg_ns / 1000.0, "launches": dk_spans,
                         "whole_run_achieved_GBs": alg_bytes / (elapsed * 1e9)},
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_nodes or n, "handel", args.cpu_sample_s)
        emit(out)
    dist.destroy_process_group()


def dist_backend():
    """"nccl" (= RCCL on ROCm) on the GPUs; WG_BENCH_BACKEND=gloo is the CPU test hook of tests/test_bench_cpu.py (the
    timing contract's barrier and reductions over gloo, the engine's kernels on the wave emulator)"""
    return os.environ.get("WG_BENCH_BACKEND", "nccl")


def launch_ranks(n):
    """`python bench.py --gpus N` with no torchrun environment around it: start the N ranks ourselves — one process per
    GPU through torch.distributed.run on 127.0.0.1, the same command line — and hand their one JSON line (rank 0's)
    through. Under a launcher (WORLD_SIZE set) this is never reached."""
    import socket
    import subprocess
    if dist_backend() == "nccl":
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this box" % (n, have))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] --gpus %d without a launcher: %s" % (n, " ".join(cmd)))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    rc = subprocess.run(cmd, stdout=_REAL_STDOUT, env=env).returncode
    sys.exit(rc)


def emit(obj):
    """the ONE JSON line, on the process's real stdout"""
    os.write(_REAL_STDOUT, (json.dumps(obj) + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nodes", type=int, default=32768)
    ap.add_argument("--replicas", type=int, default=32, help="independent simulations per step and per GPU (lowered to what fits the free HBM). "
                    "32 copies of config 3 (9.19 GB each with their init() image) since round 6: a batch launch keeps an engine on ONE XCD "
                    "(csrc/engine_kernels.hip.h wg_place), so a step's time goes by ceil(copies / 8) and 4 per XCD is what the HBM holds "
                    "(24 / 28 / 31 / 32 copies: 551 / 541 / 584 / 592 M msgs/s same-box, profiles/r24b_*, r24c_*)")
    ap.add_argument("--engine-config", default="", help="wg_config capacities of the Handel copies as name=value,... "
                    "(include/wittgpu.h; overflow of any of them is a loud error, never a silent divergence)")
    ap.add_argument("--batches", type=int, default=0,
                    help="split a step's copies into this many concurrently running batches (one HIP stream and one host "
                         "thread each); 0 = 1: the whole step as one batch on one stream, so that a launch of the delivery kernels "
                         "has the chip to itself and its HIP-event / rocprofv3 duration is the kernel's own (the roofline "
                         "figure). --batches 2 overlaps two half-batches' chains of dependent latencies: +2.7 % delivered "
                         "messages/s on round 4's code (493.7 -> 506.9 M, profiles/r19f_sweep_batches.txt; it was +6 % before the "
                         "delivery pass's tiers filled the chip), at the price of launch durations inflated by "
                         "the sharing; three and more lose to their smaller launches")
    ap.add_argument("--init-threads", type=int, default=0, help="host threads for the copies' init() (0 = one per copy, "
                    "bounded by the box's cores and host memory)")
    ap.add_argument("--cpu-sample-nodes", type=int, default=0,
                    help="nodes of the cpu_baseline sample run (the oracle on host cores); 0 = the GPU workload's own node count: the "
                         "metric's configuration, run from t = 0 for --cpu-sample-s seconds of one core")
    ap.add_argument("--cpu-sample-s", type=float, default=25.0, help="CPU work of the cpu_baseline sample, seconds per core")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-second", action="store_true", help="skip the second_workload object (Casper IMD, config 5's node count)")
    ap.add_argument("--attesters-per-round", type=int, default=4096, help="--workload casper: attesters voting per slot")
    ap.add_argument("--casper-cycle-length", type=int, default=64, help="--workload casper: slots per cycle (BASELINE config 5: 64)")
    ap.add_argument("--casper-producers", type=int, default=5, help="--workload casper: block producers (BASELINE config 5: 5)")
    ap.add_argument("--casper-stopped", type=float, default=0.0,
                    help="--workload casper: fraction of the attesters stop()ped after init() (config 5's '+10 %%': 0.1)")
    ap.add_argument("--casper-ms", type=int, default=24000, help="--workload casper: simulated ms per step")
    ap.add_argument("--casper-random-on-ties", action="store_true",
                    help="--workload casper: CasperParemeters.randomOnTies = true (the reference's default): exact, through k_casper_seq")
    ap.add_argument("--casper-shards", type=int, default=0,
                    help="--workload casper: the step's ONE simulation on the node-range sharded pipeline — 1 = one rank through the "
                         "engine's own RCCL communicator, k > 1 = k logical shards on this GPU (in-process loopback all-reduce); "
                         "0 = unsharded (default)")
    ap.add_argument("--workload", choices=["handel", "gsf", "casper"], default="handel",
                    help="handel = the BASELINE metric's workload (default); gsf = BASELINE configs[1], GSFSignature "
                         "(use --nodes 4096)")
    ap.add_argument("--mode", choices=["replicas", "shard"], default="replicas",
                    help="replicas = independent copies per GPU (default, weak scaling); shard = one simulation per "
                         "step, its nodes split by id range over the ranks (strong scaling)")
    ap.add_argument("--shard-nodes", type=int, default=32768, help="nodes of the `sharded_workload` object of a --gpus N > 1 line (ONE "
                    "Handel simulation split over the N ranks); 131072 = BASELINE config 4 (init() on the host: ~ 5 min and 128 GiB per rank)")
    ap.add_argument("--no-shard-line", action="store_true", help="--gpus N > 1: leave the `sharded_workload` object out")
    ap.add_argument("--shard-callback", action="store_true",
                    help="--mode shard: the per-ms sums through the caller-supplied torch.distributed all-reduce "
                         "(wg_allreduce_fn) instead of the engine's own RCCL communicator")
    ap.add_argument("--logical-shards", type=int, default=0,
                    help="--mode shard on ONE GPU: k engines in this process, each owning a node range; the all-reduce "
                         "sums their buffers in place (shards.LoopbackGroup). 0 = one shard per rank over RCCL")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.logical_shards and not (args.workload == "casper"):
        return launch_ranks(args.gpus)  # (the driver's `python bench.py --gpus N`: N ranks, one per GPU; never returns)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        log("[bench] --gpus %d but the launcher started %s ranks: the launcher's count stands (n_gpus in the line)"
            % (args.gpus, os.environ["WORLD_SIZE"]))
    if args.workload == "casper":
        return main_casper(args)
    if args.mode == "shard":
        return main_shard(args)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local)
    rdev = "cuda" if dist_backend() == "nccl" else "cpu"  # where the contract's reductions live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rdev == "cuda":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(dist_backend())

    import __graft_entry__
    from wittgenstein_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        __graft_entry__.build()
    import wittgenstein_amd as w

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for kv in filter(None, args.engine_config.split(",")):
        ENGINE_CONFIG[kv.split("=")[0].strip()] = int(kv.split("=")[1])
    K, W, n, R_req = args.steps, args.warmup, args.nodes, args.replicas
    from wittgenstein_amd import replicas
    # ---- init(), ONCE per copy. RunMultipleTimes re-creates and re-initialises the protocol for every run
    # (`p.copy(); rd.setSeed(i); init()`, C/RunMultipleTimes.java:44-48); init() is sequential host work outside the hot
    # path ("wall time covers runMs only"), so every copy is initialised once, its init() image kept on
    # the device (wg_snapshot) and every step — warm-up or timed — starts from the restored image (wg_restore): the same
    # R seeds are re-run each step. Copies are created one wave of host threads at a time; the first copy measures what
    # a copy (plus its image) takes and the batch is lowered to what fits the free HBM (never an error).
    t_init = time.perf_counter()
    free0 = torch.cuda.mem_get_info()[0]
    seeds = list(replicas.rank_seeds(rank, world, R_req))
    first = make_sim(w, n, seeds[0], local, args.workload)
    first.network().snapshot()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    per_copy, once, second = max(1, free0 - free1), 0, None
    if R_req > 1:  # what a FURTHER copy takes: the first one also pays what the process allocates once (code objects, tables) —
        second = make_sim(w, n, seeds[1], local, args.workload)  # 0.17 GB, a quarter of a GSFSignature copy
        second.network().snapshot()
        torch.cuda.synchronize()
        per_copy, once = replicas.marginal_copy_bytes(free0 - free1, free1 - torch.cuda.mem_get_info()[0])
    R = replicas.plan_replicas(R_req, free0 - once, per_copy, transient_bytes=0 if args.workload == "gsf" else replicas.handel_init_transient_bytes(n))
    if world > 1:  # every rank runs the same batch size (weak scaling: fixed work per GPU)
        rt = torch.tensor([R], device=rdev, dtype=torch.int64)
        dist.all_reduce(rt, op=dist.ReduceOp.MIN)
        R = int(rt.item())
    if R < R_req:
        log("[rank %d] %d copies x %.1f GB (with init() image) exceed the %.0f GB free: %d copies per step instead"
            % (rank, R_req, per_copy / 1e9, free0 / 1e9, R))
    seeds = list(replicas.rank_seeds(rank, world, R))
    try:
        avail = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) * 1024
    except Exception:
        avail = 0
    # init() of one copy on the host holds ~3 N^2 int32 (ranks, their transpose, emission lists); with the rank shuffles and
    # the emission lists built on the device (wgh_handel_create: unsharded, 256 .. 131 072 nodes) a thread holds next to nothing
    on_device = bool(getattr(first, "init_on_device", False))
    threads = replicas.init_threads(args.init_threads, max(1, R - 1), avail, (1 << 28) if on_device else 3.5 * 4 * n * n + (1 << 30),
                                    len(os.sched_getaffinity(0)), world)

    def init_one(s):
        g = make_sim(w, n, s, local, args.workload)
        g.network().snapshot()
        return g
    sims = [first] + ([second] if second is not None and R > 1 else [])
    second = None
    if R > 2:
        with ThreadPoolExecutor(max_workers=threads) as ex:
            sims += list(ex.map(init_one, seeds[2:]))
    batch = w.Batch([g.network() for g in sims])
    init_wall = time.perf_counter() - t_init
    init_s = init_wall / R
    log("[rank %d] init(): %d copies in %.1f s (%.1f s per simulation amortised over %d host threads; rank shuffles and emission "
        "lists built on the %s; outside the timed region); %.2f GB per copy incl. its init() image"
        % (rank, R, init_wall, init_s, threads, "device" if on_device else "host", per_copy / 1e9))
    # WG_GRAPH=1: the device loop replayed as a hipGraph. The delivery pass is then bracketed by device clock stamps (one-lane
    # kernels on the engine's stream writing s_memrealtime, Engine::ProfScope / k_prof_stamp) instead of HIP events, which a
    # replayed graph would re-record; the spans-on-one-axis machinery of --batches > 1 (event based) is off there
    graph_mode = os.environ.get("WG_GRAPH", "0") not in ("", "0")

    def restore_all():
        for g in sims:
            g.network().restore()
        torch.cuda.synchronize()

    nb = args.batches if args.batches > 0 else 1
    nb = max(1, min(nb, R))
    # the step's copies as `nb` smaller batches, each on its own HIP stream and host thread
    subs = split_batches(w, sims, nb) if nb > 1 else [batch]
    n_first = (R + nb - 1) // nb  # copies of the first sub-batch: the one whose stream carries the HIP events

    def run_step(single=False):
        if nb > 1 and not single:
            with ThreadPoolExecutor(max_workers=len(subs)) as ex:
                res = list(ex.map(lambda b: b.run_multiple_times(chunk=10, maxTime=20000), subs))
            return sum(sum(d) for d, _ in res), sum(sum(ms) for _, ms in res)
        d, ms = batch.run_multiple_times(chunk=10, maxTime=20000)
        return sum(d), sum(ms)

    # ---- warm-up steps: same shape as a timed step; the first one brackets every phase with HIP events (the phase
    # breakdown costs the timed region nothing)
    prof_phase = None
    restore_s = 0.0
    restores = 0
    first_step = True
    for i in range(W):
        if not first_step:
            t_r = time.perf_counter()
            restore_all()
            restore_s += time.perf_counter() - t_r
            restores += 1
        first_step = False
        if i == 0:
            sims[0].network().profile(1)
        run_step(single=(i == 0))  # (the first warm-up step as ONE batch on one stream: a clean per-phase breakdown)
        if i == 0:
            prof_phase = sims[0].network().profile_read()
            sims[0].network().profile(0)
    # ---- K timed steps: each step's RunMultipleTimes pass is bracketed by barrier + synchronize on both sides and the
    # brackets are summed; between steps (outside the brackets) the copies go back to their init() image
    delivered = sim_ms = 0
    elapsed = 0.0
    step_s = []
    by_level = by_level_first = None
    dk_spans = dk_ns = 0
    chip_union_ns = chip_sum_ns = 0.0
    chip_launches = 0
    leads = [k * n_first for k in range(nb) if k * n_first < R]  # the first copy of every concurrent batch
    t_wall0 = time.perf_counter()
    for i in range(K):
        if not first_step:
            t_r = time.perf_counter()
            restore_all()
            restore_s += time.perf_counter() - t_r
            restores += 1
        first_step = False
        for q in leads:  # HIP events around the delivery kernels only, inside the timed region; every concurrent
            sims[q].network().profile(2)  # batch's lead, on one time axis (the first lead's reference event)
            if not graph_mode:
                sims[q].network().profile_reference(sims[0].network())
        barrier()
        t0 = time.perf_counter()
        d, ms = run_step()
        barrier()
        dt = time.perf_counter() - t0
        elapsed += dt
        step_s.append(dt)
        delivered += d
        sim_ms += ms
        assert not any(batch.cont_if())
        pr = sims[0].network().profile_read()["deliver"]
        dk_spans += pr["spans"]
        dk_ns += pr["total_ns"]
        if not graph_mode:
            iv = []
            for q in leads:
                a, b = sims[q].network().profile_spans(2)
                iv += list(zip(a.tolist(), b.tolist()))
                chip_sum_ns += float((b - a).sum())
            chip_union_ns += replicas.union_ns(iv)
            chip_launches += len(iv)
        for k, g in enumerate(sims):
            bl = g.network().delivered_by_level()  # (cumulative since the restored image: this step's)
            by_level = bl if by_level is None else by_level + bl
            if k < n_first:
                by_level_first = bl if by_level_first is None else by_level_first + bl
    wall_timed_loop = time.perf_counter() - t_wall0
    check = int(sum(int(g.network().read("msgReceived").sum()) for g in sims)) if K > 0 else 0
    del subs, batch, sims, first
    gc.collect()

    if world > 1:
        elapsed, delivered, sim_ms = replicas.reduce_job(dist, rdev, elapsed, delivered, sim_ms)
    sharded, hard_exit = None, False
    if world > 1 and args.workload == "handel" and not args.no_shard_line:
        gc.collect()
        if rdev == "cuda":
            torch.cuda.empty_cache()
        sharded, hard_exit = sharded_workload(args, dist, rank, world, local, rdev == "cuda")
    if rank != 0:
        if hard_exit:
            os._exit(0)
        if world > 1:
            dist.destroy_process_group()
        return

    gsf = args.workload == "gsf"
    out = {
        "metric": "delivered messages/sec (GSFSignature; simulated-ms/sec alongside)" if gsf else
                  "delivered messages/sec (Handel 32k nodes; simulated-ms/sec alongside)",
        "value": delivered / max(elapsed, 1e-9), "unit": "delivered messages/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": elapsed * 1000.0 / max(1, K), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "simulated_ms_per_s": sim_ms / max(elapsed, 1e-9),
        "config": {"workload": ("GSFSignature, %d nodes, threshold 0.99, pairing 3, timeoutPerLevel 50, period 10, "
                                "accelerated calls 10, RANDOM nodes, NetworkLatencyByDistanceWJitter; RunMultipleTimes: "
                                "%d independent copies per step and per GPU (seeds distinct), runMs(10) until each "
                                "copy's newConfIf is false" % (n, R)) if gsf else
                               ("Handel aggregation, %d nodes, 10%% dead, threshold 0.99*live, pairing 4, levelWait 50, "
                                "period 20, fastPath 10, RANDOM nodes, NetworkLatencyByDistanceWJitter; RunMultipleTimes: "
                                "%d independent copies per step and per GPU (seeds distinct), runMs(10) until each "
                                "copy's Handel.newContIf is false" % (n, R)),
                   "nodes": n, "replicas_per_gpu": R, "parallelism": "independent simulations batched per launch" +
                   ("" if nb <= 1 else ", as %d concurrent batches (one HIP stream each)" % nb),
                   "concurrent_batches": nb,
                   "replicas_requested": R_req, "hbm_bytes_per_copy_incl_init_image": int(per_copy),
                   "delivered_per_simulation": delivered // max(1, K * R * world),
                   "timing": "sum over the K steps of each step's RunMultipleTimes pass (barrier + synchronize on both "
                             "sides of every step); between steps the copies are put back to their init() image "
                             "(wg_restore, device-to-device) outside the brackets: the same R seeds every step",
                   "step_s_min_max": [min(step_s), max(step_s)] if step_s else None,
                   "restore_ms_per_step": 1000.0 * restore_s / max(1, restores),
                   "timed_loop_wall_s_incl_restores": wall_timed_loop,
                   "init_s_per_simulation": init_s, "init_wall_s": init_wall, "init_threads": threads,
                   "init_on_device": on_device,
                   "cpu_baseline_sample_nodes": None if args.no_cpu else (min(args.cpu_sample_nodes or n, n) if gsf else (args.cpu_sample_nodes or n)),
                   "tuning_env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("WG_")}},
    }
    # ---- roofline of the delivery pass (Handel: k_handel_lane + _update + _lane2 + _copy + _dissem + _wave): algorithmic bytes of everything delivered
    # in the timed region / its launches, over its average duration measured with HIP events in the timed region
    if by_level is None:
        import numpy as np
        by_level = np.zeros(32, np.int64)
    if by_level_first is None:
        by_level_first = by_level
    alg_bytes = float(sum(int(c) * b_msg(l) for l, c in enumerate(by_level)))
    # the HIP events bracket the FIRST sub-batch's launches (its stream): its copies' bytes, its launches
    per_launch_bytes = float(sum(int(c) * b_msg(l) for l, c in enumerate(by_level_first))) / max(1, dk_spans)
    avg_ns = dk_ns / max(1, dk_spans)
    achieved = per_launch_bytes / max(1.0, avg_ns)  # bytes/ns == GB/s
    whole_step_traffic = None
    # the counters cannot be read inside the timed run (rocprofv3 serialises kernels): they come from PMC passes of the same
    # command; tools/gpu_final_round.sh takes them in the session of the final line and stamps the commit
    traffic, traffic_source, line_rate, tj = pmc_traffic("traffic_gsf.json" if gsf else "traffic.json" if n == 32768 else "traffic_handel%d.json" % n,
                                                         n, R, avg_ns, elapsed / max(1, K))
    if tj is not None:  # (measured on one batch of R copies: per copy it scales)
        traffic = traffic * n_first / R
        if tj.get("whole_step_hbm_bytes") and K > 0:
            # every per-ms kernel of a step (not only the delivery pass) against the step's algorithmic bytes: the
            # conditional-task phase, the ordering / append chain and the scans move bytes SURVEY.md §8d does not price
            ws = float(tj["whole_step_hbm_bytes"])
            whole_step_traffic = {"hbm_bytes_per_step": ws, "algorithmic_bytes_per_step": alg_bytes / K,
                                  "ratio": ws / max(1.0, alg_bytes / K),
                                  "fetch_bytes_per_step": tj.get("whole_step_fetch_bytes"),
                                  "write_bytes_per_step": tj.get("whole_step_write_bytes")}
    out["roofline"] = {
        "bound": "hbm", "kernel": "k_gsf_docycle + k_gsf_lane + k_deliver_inbox<GsfProto> (the delivery pass, all inside the HIP-event bracket)" if gsf else "k_handel_lane + k_handel_update + k_handel_lane2 + k_handel_copy + k_handel_dissem + k_handel_wave "
                                                              "(the delivery pass: one launch of each per simulated ms, all six inside the HIP-event bracket and "
                                                              "inside `traffic`)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "line_rate": line_rate,
        "whole_step_traffic": whole_step_traffic,
        "whole_step_frac": alg_bytes / (max(elapsed, 1e-9) * 1e9) / HBM_PEAK_GBS,
        "algorithmic_bytes_per_launch": per_launch_bytes, "avg_launch_us": avg_ns / 1000.0, "launches": dk_spans,
        "bytes_per_delivered_message": alg_bytes / max(1, delivered if world == 1 else int(by_level.sum())),
        "whole_run_achieved_GBs": alg_bytes / (max(elapsed, 1e-9) * 1e9),
    }
    if graph_mode:
        out["roofline"]["note"] = ("WG_GRAPH=1: runMs(chunk) of the batch is captured once and replayed as a hipGraph; the delivery pass is "
                                   "bracketed by device clock stamps (one-lane kernels on the engine's stream writing s_memrealtime before "
                                   "and after the pass) — a HIP event inside a replayed graph keeps its last replay only. ")
    if nb > 1 and chip_union_ns > 0:
        # the delivery pass on the CHIP: all batches' delivery launches on one time axis; bytes of all copies over the time
        # during which at least one of them runs (their union) — the per-stream figure above counts the chip's sharing twice
        chip = alg_bytes / chip_union_ns
        out["roofline"]["all_streams"] = {"achieved": chip, "frac": chip / HBM_PEAK_GBS, "union_ms": chip_union_ns / 1e6,
                                          "sum_of_launch_ms": chip_sum_ns / 1e6, "launches": chip_launches,
                                          "overlap": 1.0 - chip_union_ns / max(1.0, chip_sum_ns)}
    if nb > 1:
        out["roofline"]["note"] = out["roofline"].get("note", "") + ("%d batches run concurrently: the bracketed launches are the first batch's (%d of the %d "
                                   "copies) and share the chip with the other batches' kernels while they run"
                                   % (nb, n_first, R))
    if prof_phase:
        out["roofline"]["warmup_phase_device_ms"] = {k: round(v["total_ns"] / 1e6, 3) for k, v in prof_phase.items()}
        dp = prof_phase.get("deliver")
        if dp and dp["spans"] and K > 0:  # the same pass alone on the chip (first warm-up step, all copies in one batch)
            one = alg_bytes / K / dp["spans"] / max(1.0, dp["total_ns"] / dp["spans"])
            out["roofline"]["single_stream"] = {"achieved": one, "frac": one / HBM_PEAK_GBS,
                                                "avg_launch_us": dp["total_ns"] / dp["spans"] / 1000.0,
                                                "launches": dp["spans"]}
    if world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(min(args.cpu_sample_nodes or n, n) if gsf else (args.cpu_sample_nodes or n), args.workload, args.cpu_sample_s)
    log("msgReceived sum of the last step's copies: %d" % check)
    if world == 1 and not args.no_cpu and not args.no_second and not gsf:
        # BASELINE configs[4] beside the metric's own workload, in the same driver-run line: Casper IMD resident at
        # config 5's node count (262 150 nodes, every vote a sendAll to all of them), one GPU, 10 % of the attesters
        # stopped; its own roofline / cpu_baseline objects inside. Not part of `value`.
        try:
            ca = argparse.Namespace(**vars(args))
            ca.steps, ca.warmup, ca.casper_stopped = 2, 1, 0.1
            out["second_workload"] = casper_line(ca)
        except Exception as x:  # the Handel line stands on its own
            out["second_workload"] = {"error": "%s: %s" % (type(x).__name__, x)}
        # the north star's TARGET SIZE (Handel 65 536 nodes, same ratios) and BASELINE configs[1]
        # (GSFSignature 4096 nodes), each as its own compact line — outside `value`, after the main line's copies are freed.
        # GSFSignature's copies are 0.6 GB each, so — as the Handel line takes the 32 copies that fit the HBM — it takes 480, 60 per
        # XCD (489 fit; 256 / 384 / 489 copies: 668 / 712 / 855 M msgs/s on round 6's code, profiles/r24b_*, r25b_*; the last with
        # the per-engine grids of engine.h grid_per_engine)
        for key, wl, nn, rr in (("target_size_workload", "handel", 65536, 8), ("third_workload", "gsf", 4096, 480)):
            if n == nn and args.workload == wl:
                continue
            try:
                gc.collect()
                torch.cuda.empty_cache()
                out[key] = replicas_line(wl, nn, rr, 3, 1, local)
            except Exception as x:
                out[key] = {"error": "%s: %s" % (type(x).__name__, x)}
    if sharded is not None:
        out["sharded_workload"] = sharded
    emit(out)
    if hard_exit:
        sys.stdout.flush()
        os._exit(0)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
