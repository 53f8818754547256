"""The N>1 path of bench.py on CPU: two processes over gloo (world_size 2). The path shards independent copies
(seeds) across ranks with no data-path collective (wittgenstein_amd/replicas.py), so what there is to check is that
the seed ranges are disjoint and complete, that the whole-job reductions are MAX(time) / SUM(counts), and that two
ranks each running their own copies on the engine (here: the kernel sources on the CPU wave emulator, test
infrastructure) deliver exactly what the CPU oracle delivers for those seeds."""
import os
import subprocess
import sys

import pytest


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, torch.distributed as dist
import wittgenstein_amd._lib as L
L.LIB_PATH = os.path.join(%(root)r, "tests", "emu", "libwittgpu_emu.so")   # test infrastructure: no GPU here
import wittgenstein_amd as w
from wittgenstein_amd import replicas
import oracle_lib as o
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
warm, timed = replicas.seed_ranges(rank, world, steps=1, warmup=1, replicas=2)
params = (64, 57, 4, 50, 10, 20, 10, 6)
NB, NL = "RANDOM_SPEED=CONSTANT_TOR=0.00", "NetworkLatencyByDistanceWJitter"
delivered = sim_ms = expect = 0
for seeds in timed:
    sims = []
    for sd in seeds:
        g = w.Handel(w.HandelParameters(*params, NB, NL, 0), seed=sd); g.init(); sims.append(g)
    d, ms = w.Batch([g.network() for g in sims]).run_multiple_times(chunk=10, maxTime=20000)
    delivered += sum(d); sim_ms += sum(ms)
    for sd in seeds:
        c = o.Handel(*params, NB, NL, 0, seed=sd)
        while c.cont_if(): c.run_ms(10)
        expect += c.info(False)["delivered"]
elapsed = 1.0 + rank  # rank 1 is "slower"
dist.barrier()
el, dl, sm = replicas.reduce_job(dist, "cpu", elapsed, delivered, sim_ms)
ex = torch.tensor([expect]); dist.all_reduce(ex)
all_seeds = [None] * world
dist.all_gather_object(all_seeds, [list(r) for r in warm + timed])
if rank == 0:
    print("RESULT " + json.dumps({"elapsed": el, "delivered": dl, "sim_ms": sm, "expect": int(ex.item()), "seeds": all_seeds}))
dist.destroy_process_group()
'''


def test_seed_ranges_are_disjoint_and_ordered():
    sys.path.insert(0, ROOT)
    from wittgenstein_amd import replicas
    seen = []
    for rank in range(4):
        warm, timed = replicas.seed_ranges(rank, 4, steps=2, warmup=1, replicas=3)
        assert len(warm) == 1 and len(timed) == 2 and all(len(r) == 3 for r in warm + timed)
        seen += [s for r in warm + timed for s in r]
    assert sorted(seen) == list(range(4 * 3 * 3))
    with pytest.raises(ValueError):
        replicas.seed_ranges(4, 4, 1, 1, 1)


def test_two_ranks_over_gloo(oracle, tmp_path):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")], check=True)
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    import json
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0]
    r = json.loads(line[len("RESULT "):])
    assert r["elapsed"] == 2.0                      # MAX over ranks
    assert r["delivered"] == r["expect"] > 0        # SUM over ranks == what the oracle delivers for those seeds
    flat = [s for rk in r["seeds"] for rng in rk for s in rng]
    assert sorted(flat) == list(range(8))           # 2 ranks x (1 warm-up + 1 timed) x 2 copies, disjoint
