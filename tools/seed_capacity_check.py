#!/usr/bin/env python
"""The seeds the driver's 8-GPU scaling run would use, on one GPU: rank r of `bench.py --gpus 8` runs the seeds
r * R .. (r + 1) * R - 1 of the default workload (replicas.rank_seeds), and a capacity of the engine that a seed overflows —
a verification queue (wg_config.queue_cap / queue_cap_wide), the rank-bump table (rank_bump_cap), the outbox — is a LOUD stop of
that rank's step, not a silent divergence. Only rank 0's seeds are exercised by a 1-GPU bench; this runs the other ranks'
batches one after the other and reports deliveries, simulated ms and any engine error per batch.

    python tools/seed_capacity_check.py [--world 8] [--replicas 31] [--nodes 32768] [--ranks 1,2,...] > gpurun_out/seed_capacity.json
"""
import argparse
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--replicas", type=int, default=31)
    ap.add_argument("--nodes", type=int, default=32768)
    ap.add_argument("--ranks", default="")
    args = ap.parse_args()
    import torch
    import bench
    import wittgenstein_amd as w
    from wittgenstein_amd import replicas
    ranks = [int(x) for x in args.ranks.split(",")] if args.ranks else list(range(1, args.world))
    out = {"nodes": args.nodes, "replicas": args.replicas, "world": args.world, "batches": []}
    for r in ranks:
        seeds = list(replicas.rank_seeds(r, args.world, args.replicas))
        t0 = time.time()
        row = {"rank": r, "seeds": [seeds[0], seeds[-1]]}
        try:
            sims, batch = bench.make_batch(w, args.nodes, seeds, 0, min(len(seeds), 7))
            d, ms = batch.run_multiple_times(chunk=10, maxTime=20000)
            torch.cuda.synchronize()
            row.update(delivered=int(sum(d)), simulated_ms=[int(min(ms)), int(max(ms))],
                       all_done=not any(g.cont_if() for g in sims), error=None)
            del batch, sims
        except Exception as x:  # an engine capacity error names the wg_config field to raise
            row.update(error="%s: %s" % (type(x).__name__, x))
        row["wall_s"] = round(time.time() - t0, 1)
        out["batches"].append(row)
        gc.collect()
        print(json.dumps(row), file=sys.stderr, flush=True)
    out["ok"] = all(b["error"] is None and b.get("all_done") for b in out["batches"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
