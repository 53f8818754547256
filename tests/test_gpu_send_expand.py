"""k_send_expand_* — a host-issued list send with many destinations (a protocol's sendAll) resolved on the device:
latency per destination, drops, stable counting sort by arrival (C/Network.java:449-467). The engine takes the device
path from WG_SEND_EXPAND_MIN destinations on (default 4096); below that the subprocess case forces it (64) for
PingPong's 1000-destination sendAll and the scheduler fuzz (mid-run sendAlls under partitions, stopped nodes and
discard-time changes). Everything against the oracle."""
import os
import subprocess
import sys

import pytest

import oracle_lib as o
import parity
import wittgenstein_amd as w

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_sendall_expanded_on_the_device_many_tiles():
    """a sendAll of 5000 destinations — createMessageArrivals + its stable sort on the device (k_send_expand_*: five
    tiles of the counting sort) — against the oracle; drops at send time (partitions, stopped nodes, discard time) go
    through the same kernels in tests/test_gpu_fuzz.py"""
    g = w.PingPong(w.PingPongParameters(5000, parity.NB, parity.NL), seed=4)
    g.init()
    c = o.PingPong(5000, parity.NB, parity.NL, seed=4)
    for _ in range(8):
        g.network().runMs(60)
        c.run_ms(60)
        assert not parity.diff_pingpong(g, c)
    assert g.network().read("pong")[0] > 4000


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
%(pre)s
import oracle_lib as o
o.build()
import test_gpu_engine as te, test_gpu_fuzz as tf
te.test_pingpong_chunking_and_seeds(7)
tf.test_fuzz_partitions_stops_and_discard(2)
tf.run(64, 12, "NetworkNoLatency", seed=11, chunk=5, chunks=40)
print("EXPAND OK")
'''


def run_worker(tmp_path, pre):
    script = tmp_path / "expand_worker.py"
    script.write_text(WORKER % {"root": ROOT, "pre": pre})
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, WG_SEND_EXPAND_MIN="64"))
    assert p.returncode == 0 and "EXPAND OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


@pytest.mark.gpu
def test_forced_device_path_for_small_sends(tmp_path):
    run_worker(tmp_path, "")


def test_forced_device_path_on_the_emulator(oracle, tmp_path):
    emu = os.path.join(ROOT, "tests", "emu")
    subprocess.run(["make", "-s", "-C", emu], check=True)
    run_worker(tmp_path, "import wittgenstein_amd._lib as L\nL.LIB_PATH = %r   # test infrastructure: no GPU here" % os.path.join(emu, "libwittgpu_emu.so"))
