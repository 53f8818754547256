"""CPU-side checks of the product library: it loads, exports every symbol include/*.h declares,
refuses to run without a GPU, and its own java.util.Random (incl. LCG jump-ahead) matches the oracle's."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from wittgenstein_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.lib()


def test_exports_every_declared_symbol(lib):
    from wittgenstein_amd import _lib
    declared = set()
    for h in ("wittgpu.h", "wittgpu_host.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        declared |= set(re.findall(r"\b(wgh?_[a-z0-9_]+)\s*\(", src))
    declared -= {"wg_status"}
    assert declared == set(_lib.ABI_SYMBOLS), declared ^ set(_lib.ABI_SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import wittgenstein_amd as w
    with pytest.raises(w.HipError):
        w.PingPong().init()


def test_product_random_matches_oracle_and_jump_ahead(lib, oracle):
    for seed in (0, 1, 42, -7, 2**40 + 3):
        n = 2000
        a = np.zeros(n, np.int32)
        b = np.zeros(n, np.int32)
        lib.wgh_jrandom_ints(C.c_int64(seed), n, a.ctypes.data_as(C.POINTER(C.c_int32)))
        lib.wgh_jrandom_skip_ints(C.c_int64(seed), n, b.ctypes.data_as(C.POINTER(C.c_int32)))
        ref = o.jrandom_ints(seed, n)
        assert (a == ref).all() and (b == ref).all()
    for bound in (1, 2, 3, 10, 16, 17, 100, 32768, 32767, 2**30 + 1):
        a = np.zeros(500, np.int32)
        lib.wgh_jrandom_bounded(C.c_int64(5), bound, 500, a.ctypes.data_as(C.POINTER(C.c_int32)))
        assert (a == o.jrandom_bounded(5, bound, 500)).all()


def test_rccl_that_cannot_be_loaded_is_a_status_not_a_crash():
    """wg_rccl_unique_id on a box whose librccl cannot be dlopen'ed returns WG_EHIP with dlopen's message (include/wittgpu.h);
    it used to build that message from two dlerror() calls — the second returns NULL — and crash the host process. Run in
    a child process: the loader's outcome is cached per process."""
    import subprocess
    import sys
    code = ("import ctypes as C, sys\n"
            "sys.path.insert(0, %r)\n"
            "from wittgenstein_amd import _lib\n"
            "l = _lib.lib()\n"
            "buf = (C.c_uint8 * 128)()\n"
            "rc = l.wg_rccl_unique_id(buf)\n"
            "msg = l.wg_last_error(None).decode()\n"
            "assert rc == _lib.WG_EHIP, rc\n"
            "assert 'could not be loaded' in msg and '/nonexistent/librccl.so.1' in msg, msg\n"
            "print('ok')\n" % ROOT)
    env = dict(os.environ, WG_RCCL_LIB="/nonexistent/librccl.so.1", ROCM_PATH="")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
