"""WG_GRAPH=1: wg_batch_run_multiple_times captures one runMs(chunk) of the batch into a hipGraph and replays it
(one graph launch instead of ~30 kernel launches per simulated ms). Same results as the plain launch sequence: Handel
and GSFSignature batches against per-seed oracle runs. The switch is read once per process, so the run happens in a
subprocess. (Off by default; measured A/B next to bench.py when a GPU is at hand: tools/next_round.sh.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
%(pre)s
import oracle_lib as o
o.build()
import test_gpu_batch as tb
tb.test_handel_batch_matches_oracle_per_seed(%(n)d, [0, 1])
import wittgenstein_amd as w, test_gpu_gsf as tg
seeds = [5, 6]
gs = [tg.pair((64, 63, 3, 50, 10, 10, 0), seed=s)[0] for s in seeds]
d, ms = w.Batch([g.network() for g in gs]).run_multiple_times(chunk=10, maxTime=20000)
for g, s, dd in zip(gs, seeds, d):
    c = tg.pair((64, 63, 3, 50, 10, 10, 0), seed=s)[1]
    while True:
        did = c.run_ms(10)
        if not (c.info(False)["time"] < 20000 and (not did or c.cont_if())): break
    assert not tg.diff(g, c), tg.diff(g, c)
    assert dd == c.info(False)["delivered"]
print("GRAPH OK")
'''


def run_worker(tmp_path, pre, n):
    script = tmp_path / "graph_worker.py"
    script.write_text(WORKER % {"root": ROOT, "pre": pre, "n": n})
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, WG_GRAPH="1"))
    assert p.returncode == 0 and "GRAPH OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


@pytest.mark.gpu
def test_device_loop_as_a_hipgraph(tmp_path):
    run_worker(tmp_path, "", 256)


def test_device_loop_as_a_graph_on_the_emulator(oracle, tmp_path):
    emu = os.path.join(ROOT, "tests", "emu")
    subprocess.run(["make", "-s", "-C", emu], check=True)
    pre = "import wittgenstein_amd._lib as L\nL.LIB_PATH = %r   # test infrastructure: no GPU here" % os.path.join(emu, "libwittgpu_emu.so")
    run_worker(tmp_path, pre, 64)
