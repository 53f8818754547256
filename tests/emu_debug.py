"""Developer tool (not a test): run gpu_debug's lock-step comparisons with the CPU wave emulator build
of the kernels (tests/emu) standing in for libwittgpu.so."""
import os
import sys

import conftest  # noqa: F401
import wittgenstein_amd._lib as L

L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "libwittgpu_emu.so")
import gpu_debug  # noqa: E402

if __name__ == "__main__":
    ok = gpu_debug.pingpong(until=300)
    ok = gpu_debug.handel((64, 60, 6, 10, 5, 5, 10, 2, 100)) and ok
    sys.exit(0 if ok else 1)
