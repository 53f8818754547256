#!/usr/bin/env python
"""init() on the device vs on the host at sizes the oracle cannot hold: the same Handel run (tools/config4_unsharded.py's
checks) with WG_HOST_INIT unset and set; every figure of the run must agree.   python tools/gpu_init_equivalence.py [nodes] [seeds...]"""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
seeds = [int(x) for x in sys.argv[2:]] or [0, 1]
ok = True
for seed in seeds:
    outs = []
    for host in ("0", "1"):
        env = dict(os.environ, WG_HOST_INIT=host)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "config4_unsharded.py"), str(n), str(seed)],
                           capture_output=True, text=True, env=env)
        d = json.loads(p.stdout.strip().splitlines()[-1])
        outs.append(d)
    a, b = outs
    same = all(a[k] == b[k] for k in ("delivered", "time", "doneAt_max", "runMs10_calls", "checks"))
    ok &= same and a["ok"] and b["ok"] and a["init_on_device"] and not b["init_on_device"]
    print(json.dumps({"nodes": n, "seed": seed, "same_run": same, "device_init_s": a["init_s"], "host_init_s": b["init_s"],
                      "delivered": a["delivered"], "time": a["time"], "doneAt_max": a["doneAt_max"], "checks_ok": a["ok"] and b["ok"]}), flush=True)
sys.exit(0 if ok else 1)
