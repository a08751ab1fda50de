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


# ---- chisquare / standard_t on the device (vb_legacy_gamma.hip): the t families' noise, approximations.py:273-274, :345 ----
def _need_log(env):
    from viabel_amd import _lib
    if not _lib.load().vb_legacy_rng_log_proven():
        pytest.skip("this host's libm log could not be restated: the device gamma path reports UNSUPPORTED by design")


@pytest.mark.parametrize('n,d', [(4096, 1024), (4095, 1023), (1001, 77), (3, 5), (1, 1), (257, 2049)])
@pytest.mark.parametrize('df', [2.5, 7, 100])
def test_device_standard_t_equals_numpy(env, n, d, df):
    _need_log(env)
    vb, eng, LegacyRandomState = env
    ours, ref = LegacyRandomState(226), np.random.RandomState(226)
    assert eng.noise_legacy_standard_t(7, ours._h, df, n, d)
    np.testing.assert_array_equal(eng.noise_get_host(7, n, d), ref.standard_t(df, (n, d)))
    _same_state(ours, ref)
    np.testing.assert_array_equal(ours.standard_t(df, 5), ref.standard_t(df, 5))     # the host generator goes on in step


@pytest.mark.parametrize('n', [1, 2, 17, 1000, 4096, 16384, 16385, 262144])
@pytest.mark.parametrize('df', [2.5, 7, 100])
def test_device_chisquare_equals_numpy(env, n, df):
    _need_log(env)
    vb, eng, LegacyRandomState = env
    ours, ref = LegacyRandomState(56), np.random.RandomState(56)
    got = eng.chisq_legacy(ours._h, df, n)
    assert got is not None
    np.testing.assert_array_equal(got, ref.chisquare(df, n))
    np.testing.assert_array_equal(eng.chisq_get_host(n), got)        # ... and they are the context's resident draws
    _same_state(ours, ref)


def test_device_gamma_draws_with_a_cached_normal_and_in_sequence(env):
    """A cached normal carried INTO a draw (the leading values come from the host generator until the cache is empty),
    draws that start in the middle of a generator block, and the three kinds of draw interleaved on one generator."""
    _need_log(env)
    vb, eng, LegacyRandomState = env
    ours, ref = LegacyRandomState(12), np.random.RandomState(12)
    for k, (kind, n, d, df) in enumerate([('t', 501, 129, 7), ('chi', 5000, 1, 3.5), ('n', 64, 1000, 0), ('t', 7, 3, 100),
                                          ('chi', 4097, 1, 41), ('t', 1000, 333, 2.5), ('chi', 2, 1, 9), ('n', 3, 33, 0),
                                          ('t', 300, 300, 4)]):
        if k % 2 == 0:            # an odd number of host normals first: a cached value enters the device draw
            np.testing.assert_array_equal(ours.randn(3 + 2 * k), ref.randn(3 + 2 * k))
        if kind == 't':
            assert eng.noise_legacy_standard_t(6, ours._h, df, n, d)
            np.testing.assert_array_equal(eng.noise_get_host(6, n, d), ref.standard_t(df, (n, d)), err_msg=str(k))
        elif kind == 'chi':
            np.testing.assert_array_equal(eng.chisq_legacy(ours._h, df, n), ref.chisquare(df, n), err_msg=str(k))
        else:
            assert eng.noise_legacy_randn(6, ours._h, n, d)
            np.testing.assert_array_equal(eng.noise_get_host(6, n, d), ref.randn(n, d), err_msg=str(k))
        _same_state(ours, ref)


def test_device_standard_t_row_blocks_of_a_sharded_draw(env):
    _need_log(env)
    vb, eng, LegacyRandomState = env
    n, d, df = 1003, 200, 6.0
    want = np.random.RandomState(9).standard_t(df, (n, d))
    for begin, rows in ((0, 1003), (0, 502), (502, 501), (1002, 1)):
        ours, ref = LegacyRandomState(9), np.random.RandomState(9)
        ref.standard_t(df, (n, d))
        assert eng.noise_legacy_standard_t(5, ours._h, df, n, d, begin, rows)
        np.testing.assert_array_equal(eng.noise_get_host(5, rows, d), want[begin:begin + rows])
        _same_state(ours, ref)


def test_device_gamma_out_of_range_leaves_the_generator_alone(env):
    """shape <= 1 (df <= 2) is numpy's other gamma algorithm: the device path declines and nothing has changed."""
    vb, eng, LegacyRandomState = env
    ours, ref = LegacyRandomState(4), np.random.RandomState(4)
    assert eng.chisq_legacy(ours._h, 2.0, 5000) is None
    assert not eng.noise_legacy_standard_t(5, ours._h, 1.5, 300, 300)
    _same_state(ours, ref)
    np.testing.assert_array_equal(ours.chisquare(2.0, 50), ref.chisquare(2.0, 50))


def test_mfstudentt_default_mode_draws_on_the_device(env):
    """ExclusiveKL(MFStudentT(1024, df)) with the DEFAULT rng='numpy': the noise the kernels stream is
    RandomState(seed).standard_t(df, (N, D)), drawn on the device; generator in step over consecutive calls."""
    _need_log(env)
    vb, eng, LegacyRandomState = env
    from oracle import families as ofam, models as omod, objectives as oobj
    from viabel_amd.objectives import _NOISE_SLOT
    D, N, df = 1024, 4096, 7
    approx = vb.MFStudentT(D, df, seed=1)
    ref = np.random.RandomState(1)
    obj = vb.ExclusiveKL(approx, vb.GaussianModel(np.zeros(D), np.ones(D)), N)
    theta = approx.init_param() * 0.1
    for call in range(2):
        value, grad = obj(theta)
        noise = ref.standard_t(df, (N, D))
        np.testing.assert_array_equal(eng.noise_get_host(_NOISE_SLOT, N, D), noise)
        ov, og = oobj.exclusive_kl(ofam.MFStudentT(D, df), omod.GaussDiag(np.zeros(D), np.ones(D)), theta, noise)
        assert abs(value - ov) <= 1e-12 * abs(ov)
        assert np.max(np.abs(grad - og)) <= 1e-11 * np.max(np.abs(og))
    _same_state(approx._rs, ref)


def test_multivariate_t_sequence_all_on_the_device(env):
    """MultivariateT.sample's order (approximations.py:345-347): N chi-square draws, then the N x D normals -- both on
    the device at the C3 shape, the generator where numpy's is afterwards."""
    _need_log(env)
    vb, eng, LegacyRandomState = env
    D, N, df = 256, 16384, 40.0
    approx = vb.MultivariateT(D, df, seed=3)
    ref = np.random.RandomState(3)
    for call in range(2):
        chi = approx._stage_base_noise(eng, 4, N, 0, N)
        np.testing.assert_array_equal(chi, ref.chisquare(df, N))
        np.testing.assert_array_equal(eng.noise_get_host(4, N, D), ref.randn(N, D))
        _same_state(approx._rs, ref)


def test_device_gamma_budget_short_falls_back_without_touching_the_generator(env, monkeypatch):
    """How much of the stream a draw consumes is random; the device path generates a budget of words (mean + 14 sigma)
    and declines -- generator untouched, nothing written that matters -- when the walk falls off its end before the
    request is complete.  Forced here with a budget cut to 60 %: the family then draws on the host, same values."""
    _need_log(env)
    vb, eng, LegacyRandomState = env
    import os
    ours, ref = LegacyRandomState(31), np.random.RandomState(31)
    monkeypatch.setenv('VB_LEGACY_BUDGET_SCALE', '0.6')
    assert not eng.noise_legacy_standard_t(5, ours._h, 7.0, 600, 500)
    assert eng.chisq_legacy(ours._h, 9.0, 20000) is None
    _same_state(ours, ref)
    approx = vb.MFStudentT(512, 7, seed=31)
    approx._stage_base_noise(eng, 5, 300, 0, 300)            # the family's staging: device declines, host draws
    np.testing.assert_array_equal(eng.noise_get_host(5, 300, 512), np.random.RandomState(31).standard_t(7, (300, 512)))
    monkeypatch.delenv('VB_LEGACY_BUDGET_SCALE')
    assert eng.noise_legacy_standard_t(5, ours._h, 7.0, 600, 500)
    np.testing.assert_array_equal(eng.noise_get_host(5, 600, 500), ref.standard_t(7.0, (600, 500)))
    _same_state(ours, ref)


def test_host_sample_draws_big_legacy_noise_on_the_device(env):
    """``family.sample(var_param, n)`` in the default mode (what vi_diagnostics calls with 10^5 draws): from 2^18 values on
    the base noise is generated on the device from the family's own generator and read back -- the same samples as the
    reference's ``sample`` (approximations.py:212-216, :270-274, :342-349) on the same seed, generator in step."""
    _need_log(env)
    vb, eng, LegacyRandomState = env
    n = 70000
    for fam, ref_draw in ((vb.MFGaussian(8, seed=4), lambda rs: rs.randn(n, 8)),
                          (vb.MFStudentT(8, 6.0, seed=4), lambda rs: rs.standard_t(6.0, size=(n, 8))),
                          (vb.FullRankGaussian(8, seed=4), lambda rs: rs.randn(n, 8))):
        theta = fam.init_param() * 0.1
        ref = np.random.RandomState(4)
        for call in range(2):
            x = fam.sample(theta, n)
            noise = ref_draw(ref)
            if isinstance(fam, vb.FullRankGaussian):
                mu, L = fam._unpack(theta)
                np.testing.assert_allclose(x, mu + noise @ L.T, rtol=1e-13, atol=1e-13)
            else:
                np.testing.assert_array_equal(x, theta[:8] + np.exp(theta[8:]) * noise)
        _same_state(fam._rs, ref)
    t = vb.MultivariateT(8, 9.0, seed=4)
    ref = np.random.RandomState(4)
    chi, z = t._base_noise(n)
    np.testing.assert_array_equal(chi, ref.chisquare(9.0, n))
    np.testing.assert_array_equal(z, ref.randn(n, 8))
    _same_state(t._rs, ref)


def test_device_gamma_random_requests(env):
    """Sixty random (seed, df, shape, preceding draws) requests, from one value to a few chunks of the stream (the
    single-chunk case has no summary tree at all; shapes near 1 reject most; huge df puts every log in its near-1
    branch): values and generator state against numpy every time."""
    _need_log(env)
    vb, eng, LegacyRandomState = env
    gen = np.random.RandomState(2026)
    for case in range(60):
        seed = int(gen.randint(0, 2 ** 31 - 1))
        df = float([2.0001, 2.3, 3.0, 7.5, 33.0, 1e3, 1e6][gen.randint(7)])
        n, d = int(gen.randint(1, 400)), int(gen.randint(1, 120))
        pre = int(gen.randint(0, 6))
        ours, ref = LegacyRandomState(seed), np.random.RandomState(seed)
        if pre:
            np.testing.assert_array_equal(ours.randn(pre), ref.randn(pre))
        if case % 2:
            assert eng.noise_legacy_standard_t(6, ours._h, df, n, d), (case, seed, df, n, d)
            np.testing.assert_array_equal(eng.noise_get_host(6, n, d), ref.standard_t(df, (n, d)), err_msg=str((case, seed, df, n, d)))
        else:
            got = eng.chisq_legacy(ours._h, df, n * d)
            assert got is not None, (case, seed, df, n, d)
            np.testing.assert_array_equal(got, ref.chisquare(df, n * d), err_msg=str((case, seed, df, n, d)))
        _same_state(ours, ref)
