"""Runs oracle/test_network: the reference's engine + protocol unit tests restated against the oracle
(CT/NetworkTest, CT/EnvelopeStorageTest, CT/NetworkLatencyTest, PT/PingPongTest, PT/HandelTest), and
oracle/test_casper: PT/CasperIMDTest (11 tests), PT/CasperByzantineTest (2 tests) and PT/SanFerminTest (2 tests)
restated one for one."""
import os
import subprocess

import oracle_lib as o


def test_restated_reference_unit_tests(oracle):
    exe = os.path.join(o.ORACLE_DIR, "test_network")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    lines = r.stdout.strip().splitlines()
    failed = [l for l in lines if l.startswith("FAIL")]
    assert r.returncode == 0 and not failed, r.stdout
    assert sum(l.startswith("ok ") for l in lines) >= 29


def test_restated_casper_unit_tests(oracle):
    exe = os.path.join(o.ORACLE_DIR, "test_casper")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    lines = r.stdout.strip().splitlines()
    failed = [l for l in lines if l.startswith("FAIL")]
    assert r.returncode == 0 and not failed, r.stdout
    assert sum(l.startswith("ok ") for l in lines) == 17
