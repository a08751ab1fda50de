"""numpy's legacy generator restated in the library (vb_legacy_rng.cpp): values and generator state
bit-identical to `numpy.random.RandomState`, call after call -- including the parallel evaluation of large `randn`
requests (attempt-indexed words, prefix sum over acceptances, rewind behind the last consumed attempt) and the
one-value cache of the polar method crossing calls.  Host code only: runs without a GPU."""
import os

import numpy as np
import pytest

from viabel_amd._legacy_rng import LegacyRandomState


def same_state(a, b):
    sa, sb = a.get_state(), b.get_state()
    return np.array_equal(sa[1], sb[1]) and tuple(sa[2:]) == tuple(sb[2:])


@pytest.mark.parametrize('seed', [0, 1, 12345, 2 ** 32 - 1])
def test_randn_values_and_state(seed):
    a, b = LegacyRandomState(seed), np.random.RandomState(seed)
    assert same_state(a, b)
    # odd lengths leave a cached normal behind; 32 768 is where the threaded path starts
    for n in (0, 1, 2, 3, 1001, 32767, 32768, 32769, 100003, 7, 250000):
        x, y = a.randn(n), b.randn(n)
        assert x.dtype == np.float64 and np.array_equal(x, y), (seed, n)
        assert same_state(a, b), (seed, n)
    assert a.randn() == b.randn()
    assert np.array_equal(a.randn(17, 5), b.randn(17, 5))
    assert np.array_equal(a.standard_normal((3, 4)), b.standard_normal((3, 4)))


@pytest.mark.parametrize('threads', ['1', '3', '8'])
def test_large_matrix_any_thread_count(threads, monkeypatch):
    monkeypatch.setenv('VIABEL_AMD_RNG_THREADS', threads)
    a, b = LegacyRandomState(42), np.random.RandomState(42)
    a.randn(3)                                   # start the big request with a cached value pending
    b.randn(3)
    x, y = a.randn(1500, 1001), b.randn(1500, 1001)
    assert np.array_equal(x, y) and same_state(a, b)
    assert np.array_equal(a.randn(5), b.randn(5))


@pytest.mark.parametrize('df', [0.5, 1.0, 2.0, 3.0, 7.0, 40.0, 2.5e3])
def test_chisquare_standard_t_interleaved_with_normals(df):
    """MultivariateT draws chisquare(df, N) and then randn(N, D) from the same stream (approximations.py:342-345)."""
    a, b = LegacyRandomState(3), np.random.RandomState(3)
    for _ in range(3):
        assert np.array_equal(a.chisquare(df, 257), b.chisquare(df, 257))
        assert np.array_equal(a.randn(257, 9), b.randn(257, 9))
        assert np.array_equal(a.standard_t(df, size=(31, 7)), b.standard_t(df, size=(31, 7)))
        assert np.array_equal(a.random_sample(11), b.random_sample(11))
        assert same_state(a, b)
    assert a.chisquare(df) == b.chisquare(df) and a.standard_t(df) == b.standard_t(df)


def test_state_round_trip_and_foreign_seeds():
    b = np.random.RandomState(9)
    b.randn(5)
    a = LegacyRandomState(0)
    a.set_state(b.get_state())
    assert np.array_equal(a.randn(40001), b.randn(40001)) and same_state(a, b)
    c = np.random.RandomState(1)
    c.set_state(a.get_state())
    assert np.array_equal(a.randn(10), c.randn(10))
    # seeds numpy does not treat as a plain 32-bit integer go through numpy's own seeding
    arr = LegacyRandomState([1, 2, 3])
    assert np.array_equal(arr.randn(9), np.random.RandomState([1, 2, 3]).randn(9))
    LegacyRandomState(None).randn(3)
    with pytest.raises(ValueError):
        LegacyRandomState(-1)
    with pytest.raises(ValueError):
        a.chisquare(0.0, 3)
    with pytest.raises(ValueError):
        a.randn(-1)


def test_families_draw_the_reference_stream():
    """The families' parity-mode noise is numpy's stream: same seed, same matrix (approximations.py:203)."""
    import viabel_amd as vb
    fam = vb.MFGaussian(7, seed=5)
    ref = np.random.RandomState(5)
    assert np.array_equal(fam._rs.randn(100, 7), ref.randn(100, 7))
    t = vb.MultivariateT(4, 6.0, seed=8)
    ref = np.random.RandomState(8)
    assert np.array_equal(t._rs.chisquare(6.0, 50), ref.chisquare(6.0, 50))
    assert np.array_equal(t._rs.randn(50, 4), ref.randn(50, 4))


def test_copies_and_pickles_carry_the_state():
    import copy
    import pickle
    a = LegacyRandomState(11)
    a.randn(7)                                   # a cached normal is pending
    b, c, d = copy.deepcopy(a), pickle.loads(pickle.dumps(a)), copy.copy(a)
    ref = a.randn(1001)
    for other in (b, c, d):
        assert np.array_equal(other.randn(1001), ref)
    import viabel_amd as vb
    fam = vb.MFGaussian(5, seed=3)
    twin = copy.deepcopy(fam)
    assert np.array_equal(fam._rs.randn(10, 5), twin._rs.randn(10, 5))


def test_host_libm_log_is_located_and_proven():
    """vb_glibc_log.h: the device draws' logarithm is the host libm's own operation sequence on its own table, found in the
    loaded libm and proven against log() on ~1.3 M arguments at first use.  On this image's glibc (x86-64, FMA) the proof
    succeeds; elsewhere 0 is a legitimate answer (the device paths then decline and the host generator draws)."""
    from viabel_amd import _lib
    lib = _lib.load()
    proven = lib.vb_legacy_rng_log_proven()
    assert proven in (0, 1)
    import platform
    if platform.machine() == 'x86_64' and 'fma' in open('/proc/cpuinfo').read() and platform.libc_ver()[0] == 'glibc':
        assert proven == 1
