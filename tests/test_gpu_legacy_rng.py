"""GPU: numpy's legacy normal stream generated ON THE DEVICE (vb_legacy_rng_randn_device) against numpy itself --
values AND generator state, bit for bit (the integer / exactly-rounded bar of this tier: array_equal, no tolerance).

What the reference's families draw (viabel/approximations.py:203, :213-216, :343-347) is
``RandomState(seed).randn(N, D)``: MT19937 + the polar method with its one-value cache.  Covered: the
headline shape (4096, 1024), odd counts (the cached value carried into the next call), a cached value carried IN,
starts in the middle of a generator block, consecutive calls, row blocks of a sharded job, the mixed host / device
sequence of MultivariateT.sample (chi-square draws first), and the objective-level default call."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import viabel_amd as vb
    from viabel_amd import _lib
    from viabel_amd._legacy_rng import LegacyRandomState
    return vb, _lib.default_engine(), LegacyRandomState


def _same_state(ours, ref):
    a, b = ours.get_state(), ref.get_state()
    np.testing.assert_array_equal(a[1], b[1])
    assert a[2:] == b[2:]


@pytest.mark.parametrize('n,d', [(4096, 1024), (4095, 1023), (1001, 77), (16384, 256), (3, 5), (1, 1), (257, 2049)])
@pytest.mark.parametrize('seed', [1, 851])
def test_device_randn_equals_numpy(env, n, d, seed):
    vb, eng, LegacyRandomState = env
    ours, ref = LegacyRandomState(seed), np.random.RandomState(seed)
    assert eng.noise_legacy_randn(7, ours._h, n, d)
    np.testing.assert_array_equal(eng.noise_get_host(7, n, d), ref.randn(n, d))
    _same_state(ours, ref)
    # the generator goes on where numpy's does (host draw after the device draw)
    np.testing.assert_array_equal(ours.randn(5), ref.randn(5))


def test_cached_value_and_mid_block_starts(env):
    vb, eng, LegacyRandomState = env
    ours, ref = LegacyRandomState(5), np.random.RandomState(5)
    for k, (n, d) in enumerate([(3, 1), (501, 129), (64, 1000), (7, 3), (1000, 333), (2, 2)]):
        if k % 2 == 0:        # an odd host draw first: a cached value is carried INTO the device draw
            np.testing.assert_array_equal(ours.randn(3 + 2 * k), ref.randn(3 + 2 * k))
        assert eng.noise_legacy_randn(6, ours._h, n, d)
        np.testing.assert_array_equal(eng.noise_get_host(6, n, d), ref.randn(n, d), err_msg=str((k, n, d)))
        _same_state(ours, ref)


def test_row_blocks_of_a_sharded_draw(env):
    vb, eng, LegacyRandomState = env
    n, d = 1003, 200
    want = np.random.RandomState(9).randn(n, d)
    for begin, rows in ((0, 1003), (0, 502), (502, 501), (1002, 1)):
        ours, ref = LegacyRandomState(9), np.random.RandomState(9)
        ref.randn(n, d)
        assert eng.noise_legacy_randn(5, ours._h, n, d, begin, rows)
        np.testing.assert_array_equal(eng.noise_get_host(5, rows, d), want[begin:begin + rows])
        _same_state(ours, ref)       # every rank's generator ends where the whole draw ends


def test_objective_default_mode_uses_the_device_stream(env):
    """ExclusiveKL(MFGaussian(1024)) with the DEFAULT rng='numpy': the noise the kernels stream is numpy's, drawn on the
    device; the families' generators stay in step with the reference's over consecutive calls."""
    vb, eng, LegacyRandomState = env
    from oracle import families as ofam, models as omod, objectives as oobj
    from viabel_amd.objectives import _NOISE_SLOT
    D, N = 1024, 4096
    for approx, ofamily in ((vb.MFGaussian(D, seed=1), ofam.MFGaussian(D)),
                            (vb.FullRankGaussian(D, seed=1), ofam.FullRankGaussian(D))):
        ref = np.random.RandomState(1)
        obj = vb.ExclusiveKL(approx, vb.GaussianModel(np.zeros(D), np.ones(D)), N)
        theta = approx.init_param() * 0.1
        for call in range(2):
            value, grad = obj(theta)
            noise = ref.randn(N, D)
            np.testing.assert_array_equal(eng.noise_get_host(_NOISE_SLOT, N, D), noise)
            ov, og = oobj.exclusive_kl(ofamily, omod.GaussDiag(np.zeros(D), np.ones(D)), theta, noise)
            assert abs(value - ov) <= 1e-12 * abs(ov)
            assert np.max(np.abs(grad - og)) <= 1e-11 * np.max(np.abs(og))
        _same_state(approx._rs, ref)


def test_multivariate_t_sequence_chi_square_then_device_normals(env):
    vb, eng, LegacyRandomState = env
    D, N, df = 96, 1024, 9.0
    approx = vb.MultivariateT(D, df, seed=3)
    ref = np.random.RandomState(3)
    chi = approx._stage_base_noise(eng, 4, N, 0, N)
    np.testing.assert_array_equal(chi, ref.chisquare(df, N))
    np.testing.assert_array_equal(eng.noise_get_host(4, N, D), ref.randn(N, D))
    _same_state(approx._rs, ref)
