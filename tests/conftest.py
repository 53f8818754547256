import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


def _cpu_suite_workers(config):
    """The `-m "not gpu"` suite runs the product's kernel sources on a single-threaded CPU wave emulator: spread it
    over a few xdist workers (WG_TEST_WORKERS=0 keeps one process). GPU runs (`-m gpu`) are never parallelised: one
    device, and the driver records which .so files the pytest process itself loaded."""
    if hasattr(config, "workerinput") or not config.pluginmanager.hasplugin("xdist"):
        return 0
    if getattr(config.option, "numprocesses", None) is not None or getattr(config.option, "usepdb", False):
        return 0
    if "not gpu" not in (getattr(config.option, "markexpr", "") or ""):
        return 0
    want = os.environ.get("WG_TEST_WORKERS")
    n = int(want) if want not in (None, "") else min(4, (os.cpu_count() or 1) // 2)
    return n if n > 1 else 0


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    n = _cpu_suite_workers(config)
    if n:
        # the checkers every worker would otherwise build at the same time
        import subprocess
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")], check=True)
        config.option.numprocesses = n
        config.option.dist = "load"
        config.option.tx = ["popen"] * n
