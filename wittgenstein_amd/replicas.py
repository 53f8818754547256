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
