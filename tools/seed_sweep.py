"""Extra parity evidence beyond the suite's fixed seeds: resident Handel and GSFSignature in lock-step with the oracle, to
convergence, over a range of seeds and a few shapes (one MI355X; TEST INFRASTRUCTURE — imports tests/ and the oracle).
    python tools/seed_sweep.py <first seed> <seeds> > gpurun_out/<tag>/seed_sweep.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_handel as th  # noqa: E402
import test_gpu_gsf as tg  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
out = {"handel": [], "gsf": [], "failures": []}
t0 = time.time()
for seed in range(first, first + count):
    for params, step in (((64, 57, 4, 50, 10, 20, 10, 6, 0), 1), ((256, 231, 4, 50, 10, 20, 10, 25, 0), 10),
                         ((1024, 922, 4, 50, 10, 20, 10, 102, 0), 10)):
        try:
            g, c = th.lockstep(params, step, seed=seed)
            out["handel"].append([params[0], seed, int(g.network().read("msgReceived").sum()), int(g.network().time)])
        except Exception as x:  # noqa: BLE001
            out["failures"].append(["handel", params[0], seed, str(x)[:300]])
    for params, step in (((32, 32, 3, 20, 10, 10, 0), 1), ((256, 250, 3, 50, 10, 10, 0), 10), ((512, 450, 3, 50, 10, 10, 20), 10)):
        try:
            g, c = tg.lockstep(params, seed=seed, step=step, total=0, to_convergence=True)
            out["gsf"].append([params[0], seed, int(g.network().read("msgReceived").sum()), int(g.network().time)])
        except Exception as x:  # noqa: BLE001
            out["failures"].append(["gsf", params[0], seed, str(x)[:300]])
if len(sys.argv) > 3 and sys.argv[3] == "attacks":  # ... and both attack scenarios + Casper IMD (with stopped attesters, with a byzantine delay)
    import test_gpu_casper_resident as tc
    out["suicide"], out["hidden"], out["casper"] = [], [], []
    P64 = (64, 50, 4, 50, 5, 20, 10, 6, 0)
    for seed in range(first, first + count):
        for key, kw in (("suicide", {"byzantine_suicide": True}), ("hidden", {"hidden_byzantine": True})):
            for params, step in ((P64, 5), (th.ratios(256, dead=0.25), 10)):
                try:
                    g, c = th.lockstep(params, step, max_ms=4000, seed=seed, **kw)
                    out[key].append([params[0], seed, int(g.network().read("msgReceived").sum()), int(g.network().time)])
                except Exception as x:  # noqa: BLE001
                    out["failures"].append([key, params[0], seed, str(x)[:300]])
        for params, byz, stopped in (((5, False, 5, 80, 1000, 1), 0, 40), ((3, True, 3, 8, 1000, 1), 7000, 2)):
            try:
                g, c = tc.lockstep(params, seed=seed, chunk=2000, chunks=16, byz_delay=byz, stopped=stopped)
                out["casper"].append([params[3], seed, int(g.network().read("msgReceived").sum()), int(g.network().time)])
            except Exception as x:  # noqa: BLE001
                out["failures"].append(["casper", params[3], seed, str(x)[:300]])
if len(sys.argv) > 3 and sys.argv[3] == "fuzz":  # the scheduler path itself through host-callback mode (tests/test_gpu_fuzz.py's run)
    import test_gpu_fuzz as tf
    out["fuzz"] = []
    OPS = {10: [("partition", 300)], 14: [("stop", 3), ("stop", 7)], 18: [("endPartition", 0), ("setMsgDiscardTime", 150)],
           24: [("start", 3)], 27: [("setMsgDiscardTime", 1 << 30)], 30: [("partition", 500), ("partition", 200)],
           36: [("endPartition", 0), ("start", 7)]}
    NLS = [None, "NetworkNoLatency", "NetworkFixedLatency(3)", "IC3NetworkLatency"]
    for seed in range(first, first + count):
        for nl, ops, n in ((NLS[seed % 4], (), 48), (None, OPS, 64)):
            try:
                g, c = tf.run(n, 12, nl, seed=seed, chunk=25, chunks=44, ops=ops)
                out["fuzz"].append([n, seed, str(nl), int(c.info()["delivered"]), int(c.info()["tasks"])])
            except Exception as x:  # noqa: BLE001
                out["failures"].append(["fuzz", n, seed, str(nl), str(x)[:300]])
out["wall_s"] = time.time() - t0
print(json.dumps(out))
