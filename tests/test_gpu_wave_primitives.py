"""The wave / block primitives the per-ms kernels are built from (wittgenstein_amd/csrc/engine_kernels.hip.h:95-217 — DPP
row_ror / row_bcast reductions and scans, v_readlane broadcasts —, the eight-lane group forms of proto_handel.hip.h and the
ballot multisplit rank, tile_rank), ONE at a time through wg_selftest against numpy: full wavefronts, partial ones (lanes
without an item contribute the operation's identity — the contract every call site keeps), all-zero and all-ones inputs.
On the MI355X this runs the DPP branches, which the CPU wave emulator compiles out (tests/test_emu_kernels.py runs the same
bodies on the shuffle forms)."""
import ctypes as C

import numpy as np
import pytest

import wittgenstein_amd._lib as L

pytestmark = pytest.mark.gpu

(ADD32, ADD64, MIN32, MAX32, SCAN32, SCAN64, BCAST, BCAST64, G8SUM, G8OR, G8MIN, G8MAX, BSCAN, TRANK, SHFL, BSUM) = range(16)
M64 = (1 << 64) - 1


def run(op, vals, aux=0, threads=64, extra=0):
    lib = L.lib()
    vals = [int(v) & M64 for v in vals]
    n = len(vals)
    a = (C.c_uint64 * max(1, n))(*vals)
    out = (C.c_uint64 * (threads + extra))()
    rc = lib.wg_selftest(op, aux, a, n, threads, out, threads + extra)
    assert rc == 0, lib.wg_last_error(None)
    return [int(x) for x in out]


def inputs(rng, n, bits):
    yield "random", [int(x) for x in rng.integers(0, 1 << bits, size=n, dtype=np.uint64)]
    yield "zeros", [0] * n
    yield "ones", [(1 << bits) - 1] * n
    yield "ramp", list(range(1, n + 1))


@pytest.mark.parametrize("n", [64, 63, 33, 17, 1, 0])
def test_wave_reductions_and_scans(n):
    rng = np.random.default_rng(n)
    for name, v in inputs(rng, n, 31):
        assert run(ADD32, v) == [sum(v) & 0xFFFFFFFF] * 64, name
        assert run(SCAN32, v) == [sum(v[:i + 1]) & 0xFFFFFFFF for i in range(64)], name
        sv = [x - (1 << 30) for x in v]  # signed values
        lo = min(sv) if sv else (1 << 31) - 1
        hi = max(sv) if sv else -(1 << 31)
        assert run(MIN32, [x & 0xFFFFFFFF for x in sv]) == [lo & 0xFFFFFFFF] * 64, name
        assert run(MAX32, [x & 0xFFFFFFFF for x in sv]) == [hi & 0xFFFFFFFF] * 64, name
    for name, v in inputs(rng, n, 63):
        assert run(ADD64, v) == [sum(v) & M64] * 64, name
        assert run(SCAN64, v) == [sum(v[:i + 1]) & M64 for i in range(64)], name


def test_lane_broadcasts_and_shuffles():
    rng = np.random.default_rng(7)
    v = [int(x) for x in rng.integers(0, 1 << 63, size=64, dtype=np.uint64)]
    for src in (0, 1, 15, 16, 31, 32, 47, 48, 63):
        assert run(BCAST, v, aux=src) == [v[src] & 0xFFFFFFFF] * 64
        assert run(BCAST64, v, aux=src) == [v[src]] * 64
    for d in (0, 1, 8, 17, 63):
        assert run(SHFL, v, aux=d) == [v[(i + d) & 63] for i in range(64)]


@pytest.mark.parametrize("n", [64, 60, 9, 8, 3])
def test_groups_of_eight_lanes(n):
    """the narrow levels' checkSigs items (k_handel_a1c, group form): three DPP steps inside a half-row of sixteen lanes"""
    rng = np.random.default_rng(100 + n)
    for name, v in inputs(rng, n, 61):
        pad = v + [0] * (64 - n)
        groups = [pad[g:g + 8] for g in range(0, 64, 8)]
        assert run(G8SUM, v) == [sum(g) & M64 for g in groups for _ in range(8)], name
        want_or = []
        for g in groups:
            o = 0
            for x in g:
                o |= x
            want_or += [o] * 8
        assert run(G8OR, v) == want_or, name
    for name, v in inputs(rng, n, 31):
        sv = [x - (1 << 30) for x in v]
        lo, hi = [], []
        for g0 in range(0, 64, 8):
            g = sv[g0:g0 + 8]  # (lanes beyond n: the identity)
            lo += [(min(g) if g else (1 << 31) - 1) & 0xFFFFFFFF] * 8
            hi += [(max(g) if g else -(1 << 31)) & 0xFFFFFFFF] * 8
        u = [x & 0xFFFFFFFF for x in sv]
        assert run(G8MIN, u) == lo, name
        assert run(G8MAX, u) == hi, name


@pytest.mark.parametrize("threads,n", [(64, 64), (256, 200), (1024, 1024), (1024, 777), (512, 1)])
def test_block_scan_and_sum(threads, n):
    rng = np.random.default_rng(threads + n)
    for name, v in inputs(rng, n, 20):
        out = run(BSCAN, v, threads=threads, extra=1)
        pad = v + [0] * (threads - n)
        assert out[:threads] == [sum(pad[:i]) & 0xFFFFFFFF for i in range(threads)], name
        assert out[threads] == sum(v) & 0xFFFFFFFF, name
    for name, v in inputs(rng, n, 50):
        assert run(BSUM, v, threads=threads) == [sum(v) & M64] * threads, name


@pytest.mark.parametrize("bits,threads,n", [(8, 1024, 1024), (8, 1024, 900), (3, 256, 256), (1, 64, 50), (5, 1024, 3)])
def test_ballot_multisplit_rank(bits, threads, n):
    """tile_rank (the append phase's stable multisplit by arrival ms, k_scatter): a record's rank among the records of its bin
    that precede it in the tile — wave-level ballot match per bin bit, then the waves in order through an LDS running count"""
    rng = np.random.default_rng(bits * 1000 + n)
    for density in (1.0, 0.6, 0.0):
        bins = rng.integers(0, 1 << bits, size=n)
        valid = rng.random(n) < density
        v = [int(b) | (int(ok) << 32) for b, ok in zip(bins, valid)]
        out = run(TRANK, v, aux=bits, threads=threads, extra=1 << bits)
        seen = [0] * (1 << bits)
        for i in range(n):
            if valid[i]:
                assert out[i] == seen[bins[i]], (density, i)
                seen[bins[i]] += 1
        assert out[threads:threads + (1 << bits)] == seen, density
    one = [5 | (1 << 32)] * n  # every record in one bin: the ranks are the positions
    out = run(TRANK, one, aux=max(bits, 3), threads=threads, extra=1 << max(bits, 3))
    assert out[:n] == list(range(n))
