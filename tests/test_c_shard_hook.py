"""tests/c/test_shard_hook.c: the wg_allreduce_fn collective hook driven from plain C (two shards on two pthreads vs
the unsharded engine) — proof that node-range sharding needs nothing from Python or torch. Runs against the CPU
wave-emulator build of the product's sources."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")


def test_c_program_shards_pingpong_through_the_hook(tmp_path):
    subprocess.run(["make", "-s", "-C", EMU], check=True)
    exe = str(tmp_path / "test_shard_hook")
    subprocess.run(["gcc", "-O1", "-std=gnu11", "-D_GNU_SOURCE", "-o", exe, os.path.join(ROOT, "tests", "c", "test_shard_hook.c"),
                    os.path.join(EMU, "libwittgpu_emu.so"), "-lpthread", "-Wl,-rpath," + EMU], check=True)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.startswith("OK:"), p.stdout
