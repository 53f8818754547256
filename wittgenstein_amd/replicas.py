"""Multi-GPU form of the reference's only parallelism: independent copies (C/RunMultipleTimes.java:44-48 — each of
runCount copies gets rd.setSeed(i), init(), runs alone). One process per GPU, every rank owns a disjoint range of
seeds; there is no data-path collective — the only communication is the timing contract's barrier and the two
reductions below. Used by bench.py; covered on CPU by tests/test_replicas_gloo.py (gloo, world_size 2)."""


def seed_ranges(rank, world, steps, warmup, replicas):
    """(warm-up seed ranges, timed seed ranges) of `rank`: (steps + warmup) * replicas consecutive seeds per rank,
    disjoint across ranks, warm-up steps first."""
    if not (0 <= rank < world) or steps < 0 or warmup < 0 or replicas <= 0:
        raise ValueError("rank/world/steps/warmup/replicas")
    base = rank * (steps + warmup) * replicas
    warm = [range(base + i * replicas, base + (i + 1) * replicas) for i in range(warmup)]
    timed = [range(base + (warmup + i) * replicas, base + (warmup + i + 1) * replicas) for i in range(steps)]
    return warm, timed


def reduce_job(dist, device, elapsed, delivered, sim_ms):
    """whole-job figures: elapsed = MAX over ranks, delivered / simulated ms = SUM over ranks"""
    import torch
    tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    cnt = torch.tensor([delivered, sim_ms], device=device, dtype=torch.int64)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    return float(tt.item()), int(cnt[0].item()), int(cnt[1].item())


def rank_seeds(rank, world, replicas):
    """the seeds of `rank`'s copies when every step re-runs the SAME copies from their init() image (wg_snapshot /
    wg_restore): `replicas` consecutive seeds per rank, disjoint across ranks — rd.setSeed(i) of
    C/RunMultipleTimes.java:47 with i = rank * replicas + k."""
    if not (0 <= rank < world) or replicas <= 0:
        raise ValueError("rank/world/replicas")
    return range(rank * replicas, (rank + 1) * replicas)


def handel_init_transient_bytes(nodes):
    """device bytes Handel's init() holds only while it runs, over ALL the copies being initialised at once: with the
    reception ranks carried by the senders (unsharded, 256 .. 65 536 nodes, csrc/engine.hip HandelHost) the nodeCount^2
    int32 rank matrix lives for the length of init() only, and the library admits at most two of them and at most
    8 GiB between them (one alone may be larger) at a time — TmpMatrix there."""
    if nodes < 256 or nodes > 65536:
        return 0
    one = 4 * nodes * nodes
    return min(2 * one, max(one, 2 * 4 * 32768 * 32768))


def plan_replicas(requested, free_bytes, per_copy_bytes, headroom=0.955, transient_bytes=0):
    """How many resident copies a step runs on one GPU: the request, lowered to what fits `headroom` of the free HBM
    (per_copy_bytes = one copy incl. its init() image, measured on the first copy) — and leaves `transient_bytes` (what the
    copies' init() holds only while it runs, handel_init_transient_bytes) beside the copies. Never below 1 and never an
    error: a step that cannot hold the requested batch runs a smaller one and the bench line says so
    (config.replicas_per_gpu / config.replicas_requested)."""
    if requested <= 0:
        raise ValueError("replicas")
    if per_copy_bytes <= 0 or free_bytes <= 0:
        return requested
    fit = int(headroom * free_bytes) // int(per_copy_bytes)
    if transient_bytes > 0:
        fit = min(fit, max(0, int(free_bytes) - int(transient_bytes) - (1 << 30)) // int(per_copy_bytes))
    return max(1, min(requested, fit))


def marginal_copy_bytes(first_bytes, second_bytes):
    """(bytes per further copy, bytes the process pays once) from what the first and the second copy took of the free HBM: the
    first also pays the one-time allocations (code objects, latency tables, runtime pools), so sizing a batch by it alone gives
    up copies — a quarter of them for GSFSignature's 0.6 GB copies. Never below the second copy's own size, never negative."""
    per = max(1, int(second_bytes))
    if per > first_bytes:  # (allocator granularity can make the second look larger: then nothing was paid once)
        return max(1, int(first_bytes), per), 0
    return per, int(first_bytes) - per


def init_threads(requested, copies, host_avail_bytes, per_init_bytes, cores, world=1):
    """host threads for the copies' init() (sequential host work per copy, C/RunMultipleTimes.java:44-48): one per copy,
    bounded by the cores of this rank's share of the box and by host memory (per_init_bytes per running init())."""
    by_mem = int(0.6 * host_avail_bytes / max(1, world) / max(1, per_init_bytes)) if host_avail_bytes > 0 else 1
    by_cpu = max(1, cores // max(1, world))
    want = requested if requested > 0 else copies
    return max(1, min(want, copies, by_mem, by_cpu))


def union_ns(intervals):
    """total length of the union of [start, end) intervals (ns): the time during which AT LEAST ONE of them runs — the
    delivery kernels of concurrently running batches laid on one time axis (wg_profile_read_spans)"""
    total, cur_a, cur_b = 0.0, None, None
    for a, b in sorted(intervals):
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                total += cur_b - cur_a
            cur_a, cur_b = a, b
        elif b > cur_b:
            cur_b = b
    if cur_b is not None:
        total += cur_b - cur_a
    return total


def csrc_hash(root=None):
    """sha256 over the product's kernel sources (wittgenstein_amd/csrc/*, names and contents, sorted): what
    tools/traffic_from_pmc.py stamps into profiles/traffic*.json and bench.py compares with the tree it runs from, so that
    PMC figures of OTHER code are never published as this code's `roofline.traffic`."""
    import hashlib
    import os
    d = os.path.join(root or os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        p = os.path.join(d, name)
        if os.path.isfile(p):
            h.update(name.encode() + b"\0")
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]
