"""tests/c/test_abi_full.c — the C ABI driven from a plain C program (what the JNI shim jni/wittgpu_jni.c calls): resident
Handel 256 nodes to convergence, the same run from its init() image, one host-callback simulation through wg_step_begin /
wg_step_end. The digest the program prints is compared with the CPU oracle's run of the same parameters and seed.
On the MI355X against libwittgpu.so (-m gpu); here against the CPU wave-emulator build of the same sources."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as o
import parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")
FNV0, FNVP, M64 = 0xCBF29CE484222325, 0x100000001B3, (1 << 64) - 1


def fnv(a):
    h = FNV0
    for b in np.ascontiguousarray(a).tobytes():
        h = ((h ^ b) * FNVP) & M64
    return "%016x" % h


def build_and_run(tmp_path, libdir, libname):
    exe = str(tmp_path / "test_abi_full")
    subprocess.run(["gcc", "-O1", "-std=gnu11", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "c", "test_abi_full.c"),
                    os.path.join(libdir, libname), "-Wl,-rpath," + libdir], check=True)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = p.stdout.strip().splitlines()
    assert lines[-1] == "OK", p.stdout
    return dict(l.split("=", 1) for l in lines[:-1])


def check_against_oracle(out):
    n = 256
    down = n // 10
    c = o.Handel(n, int((n - down) * 0.99), 4, 50, 10, 20, 10, down, parity.NB, parity.NL, 0, seed=0)
    while c.cont_if():
        c.run_ms(10)
    info = c.info(False)
    for tag in ("handel", "restored"):
        assert int(out[tag + ".time"]) == info["time"]
        assert int(out[tag + ".rng"], 16) == info["rng"]
        for f in ["doneAt", "msgReceived", "msgSent", "bytesSent", "bytesReceived", "sigsChecked", "sigQueueSize", "msgFiltered",
                  "currWindowSize"]:
            v = c.read(f).astype(np.int64)
            assert int(out["%s.%s.sum" % (tag, f)]) == int(v.sum()), (tag, f)
            assert out["%s.%s.fnv" % (tag, f)] == fnv(v), (tag, f)
        assert out[tag + ".totalIncoming.fnv"] == fnv(c.read_bits("totalIncoming").astype(np.uint64)[:, :n // 64])
        assert out[tag + ".posInLevel.fnv"] == fnv(c.read_level("posInLevel").astype(np.int32))
    assert int(out["handel.delivered"]) == info["delivered"]
    assert (out["hostmode.pings"], out["hostmode.pongs"]) == ("4", "4")


def test_c_program_on_the_wave_emulator(tmp_path, oracle):
    subprocess.run(["make", "-s", "-C", EMU], check=True)
    check_against_oracle(build_and_run(tmp_path, EMU, "libwittgpu_emu.so"))


@pytest.mark.gpu
def test_c_program_on_the_mi355x(tmp_path, oracle):
    check_against_oracle(build_and_run(tmp_path, os.path.join(ROOT, "wittgenstein_amd"), "libwittgpu.so"))
