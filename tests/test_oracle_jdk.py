"""The oracle's (and the product's) java.util.Random against published JDK known answers.
The reference's own tests pin no RNG-dependent value, so these anchor the stream."""
import numpy as np
import pytest

import oracle_lib as o


def test_known_answers_nextint(oracle):
    # new Random(0).nextInt() == -1155484576, second -723955400; new Random(42).nextInt() == -1170105035
    assert list(o.jrandom_ints(0, 2)) == [-1155484576, -723955400]
    assert o.jrandom_ints(42, 1)[0] == -1170105035


def test_known_answers_bounded(oracle):
    # new Random(0): nextInt(10) x5 = 0, 8, 9, 7, 5 (power-of-two and modulo paths)
    assert list(o.jrandom_bounded(0, 10, 5)) == [0, 8, 9, 7, 5]
    # new Random(42): nextInt(10) x10 = 0 3 8 4 0 5 5 8 9 3 (widely published JDK output)
    assert list(o.jrandom_bounded(42, 10, 10)) == [0, 3, 8, 4, 0, 5, 5, 8, 9, 3]
    # derived from the published nextInt(): (-1170105035 >>> 1) % 100 = 30 ; power-of-two path:
    # new Random(0).nextInt(16) = (16 * (-1155484576 >>> 1)) >> 31 = 11
    assert o.jrandom_bounded(42, 100, 1)[0] == 30
    assert o.jrandom_bounded(0, 16, 1)[0] == 11


def test_known_answers_double_boolean(oracle):
    # new Random(0).nextDouble() = 0.730967787376657 ; new Random(0).nextBoolean() = true
    assert abs(o.jrandom_doubles(0, 1)[0] - 0.730967787376657) < 1e-15
    assert o.jrandom_booleans(0, 1)[0] == 1


def test_first_node_position(oracle):
    # derived in SURVEY.md §8c: seed 0, RANDOM builder -> rdInt = -1155484576 -> (1633, 529)
    assert o.node_xy(-1155484576) == (1633, 529)


def test_shuffle_is_permutation_and_deterministic(oracle):
    a, b = o.jshuffle_iota(0, 1000, 3), o.jshuffle_iota(0, 1000, 3)
    assert (a == b).all() and sorted(a) == list(range(1000))
    # Collections.shuffle(Arrays.asList(0..9), new Random(0)) -> [4, 8, 9, 6, 3, 5, 2, 1, 7, 0]  (JDK)
    assert list(o.jshuffle_iota(0, 10, 1)) == [4, 8, 9, 6, 3, 5, 2, 1, 7, 0]


def test_pseudo_random_range_and_sign(oracle):
    # Network.getPseudoRandom: |x % 100| with Java's truncating % (C/Network.java:493-496)
    vals = [o.pseudo_random(i, s) for i in range(0, 3000, 7) for s in (-1155484576, 0, 2, 2147483647, -2147483648)]
    assert min(vals) >= 0 and max(vals) <= 99
    assert len(set(vals)) == 100


def test_latency_table_shape(oracle):
    # 1145 x 100 integer LUT, values 2..217
    t = o.latency_table()
    assert t.shape == (1145, 100)
    assert t.min() == 2 and t.max() == 217
    assert (np.diff(t, axis=1) >= 0).all() and (np.diff(t, axis=0) >= 0).all()


def test_host_side_java_random_of_the_package_matches(oracle):
    """wittgenstein_amd.protocols._JavaRandom (used by CasperIMD.stop_attesters to pick the stopped attesters) is
    java.util.Random.nextInt(bound): same stream as the oracle's restatement, power-of-two and rejection bounds"""
    from wittgenstein_amd.protocols import _JavaRandom, choose_attesters
    for seed in (0, 42, -7, 1 << 40):
        for bound in (1, 7, 64, 100, 1 << 30, (1 << 30) + 12345, (1 << 31) - 1):
            r = _JavaRandom(seed)
            assert [r.nextInt(bound) for _ in range(300)] == [int(v) for v in oracle.jrandom_bounded(seed, bound, 300)]
    ids = choose_attesters(range(6, 406), 40, seed=4)
    assert len(ids) == len(set(ids)) == 40 and all(6 <= i < 406 for i in ids)
    assert ids == choose_attesters(range(6, 406), 40, seed=4) and ids != choose_attesters(range(6, 406), 40, seed=5)
    with pytest.raises(ValueError):
        choose_attesters(range(6, 406), 400)
