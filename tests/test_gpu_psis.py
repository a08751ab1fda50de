"""GPU parity of the PSIS kernel (csrc/vb_psis.hip, via vb_psis_smooth / vb_log_weights_meanfield) against the
reference's own psislw outputs (tests/golden/psis.npz) and the oracle, and of vi_diagnostics against the
reference's test expectations (viabel/tests/test_convenience.py:49-77).

Tolerances: smoothed log weights 1e-10 absolute (they are O(1..20)), k-hat 1e-10 relative.
"""
import numpy as np
import pytest

import _golden as G
from oracle import families as ofam
from oracle import models as omod
from oracle import psis as opsis
from test_psis_cpu import assert_smoothed_equal

pytestmark = pytest.mark.gpu

FX = G.load(G.fixtures('psis')[0])
NAMES = [str(n) for n in FX['names']]


@pytest.fixture(scope='module')
def vb():
    import viabel_amd
    from viabel_amd import _lib
    _lib.default_engine()
    return viabel_amd


def _close_k(k, ref):
    return np.isinf(k) if np.isinf(ref) else abs(k - ref) <= 1e-10 * max(1.0, abs(ref))


@pytest.mark.parametrize('name', NAMES)
def test_psislw_matches_reference(vb, name):
    from viabel_amd._psis import psislw, sumlogs
    raw = FX[name + '_lw']
    sm, k = psislw(raw)
    assert _close_k(k, float(FX[name + '_khat'])), (k, float(FX[name + '_khat']))
    assert_smoothed_equal(raw, sm, FX[name + '_smoothed'])
    assert abs(sumlogs(sm)) < 1e-12
    assert sm is not raw and np.array_equal(raw, FX[name + '_lw'])      # input untouched without overwrite_lw


def test_psislw_reff_columns_overwrite(vb):
    from viabel_amd._psis import psislw, psisloo
    sm, k = psislw(FX['normal_heavy_lw'], Reff=float(FX['reff_value']))
    assert _close_k(k, float(FX['reff_khat']))
    np.testing.assert_allclose(sm, FX['reff_smoothed'], rtol=0, atol=1e-10)
    two = np.array(FX['two_lw'], order='F')
    sm2, k2 = psislw(two, overwrite_lw=True)
    assert sm2 is two
    np.testing.assert_allclose(k2, FX['two_khat'], rtol=1e-10)
    np.testing.assert_allclose(sm2, FX['two_smoothed'], rtol=0, atol=1e-10)
    # psisloo = sumlogs(smoothed(-log_lik) + log_lik) per column (_psis.py:70-110)
    rng = np.random.RandomState(3)
    log_lik = -0.5 * rng.randn(500, 3) ** 2
    loo, loos, ks = psisloo(log_lik)
    for j in range(3):
        s, kk = opsis.psis_smooth(-log_lik[:, j])
        assert abs(loos[j] - opsis.log_sum_exp(s + log_lik[:, j])) < 1e-10 and _close_k(ks[j], kk)
    assert abs(loo - loos.sum()) < 1e-12


def test_psislw_validation(vb):
    from viabel_amd._psis import psislw
    with pytest.raises(ValueError):
        psislw(np.zeros(1))
    with pytest.raises(ValueError):
        psislw(np.zeros((2, 2, 2)))
    with pytest.raises(NotImplementedError):            # tail larger than the on-chip sort capacity
        psislw(np.zeros(3000000))


@pytest.mark.parametrize('family', ['gaussian', 'student'])
@pytest.mark.parametrize('target', ['gauss_diag', 'funnel'])
def test_device_log_weights_and_psis(vb, family, target):
    """vb_log_weights_meanfield + vb_psis_smooth on device-resident weights == oracle on the same noise."""
    from viabel_amd import _lib
    D, N = 37, 5000
    rng = np.random.RandomState(5)
    if target == 'gauss_diag':
        mean, sd = rng.randn(D), np.exp(0.2 * rng.randn(D))
        model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    else:
        model, omodel = vb.FunnelModel(D, 3), omod.Funnel(D, 3)
    if family == 'gaussian':
        approx, ofamily = vb.MFGaussian(D, seed=11), ofam.MFGaussian(D)
    else:
        approx, ofamily = vb.MFStudentT(D, 7.0, seed=11), ofam.MFStudentT(D, 7.0)
    theta = np.concatenate([0.2 * rng.randn(D), -0.3 + 0.1 * rng.randn(D)])
    eng = _lib.default_engine()
    eng.set_model(model.device_spec())
    noise = approx._base_noise(N)
    eng.noise_set_host(2, noise)
    fam_id, df = approx._device_family()
    lw = eng.log_weights_meanfield(2, N, D, theta, fam_id, df=df)
    z = ofamily.sample_from_noise(theta, noise)
    lw_ref = omodel.logp(z) - ofamily.log_density(theta, z)
    np.testing.assert_allclose(lw, lw_ref, rtol=0, atol=1e-11 * np.max(np.abs(lw_ref)))
    sm, k = eng.psis_smooth(N)                          # device-resident weights
    sm_ref, k_ref = opsis.psis_smooth(lw_ref)
    assert _close_k(k, k_ref)
    np.testing.assert_allclose(sm, sm_ref, rtol=0, atol=1e-9)
    with pytest.raises(_lib.EngineError):               # consumed: a second smooth needs fresh weights
        eng.psis_smooth(N)


def test_vi_diagnostics_like_reference(vb, capsys):
    """viabel/tests/test_convenience.py:49-77 with q = N(0, I_2): a matching target passes, a wider target has
    k-hat > 0.7 (no further diagnostics), a narrower target has bounded weights (k-hat < 0) and d2 > 2."""
    np.random.seed(153)
    approx = vb.MFGaussian(2)
    opt_param = np.array([0.02, -0.01, 0.01, -0.02])      # a converged-but-not-exact fit, as bbvi returns
    objective = vb.ExclusiveKL(approx, vb.GaussianModel(np.zeros(2), np.ones(2)), 100)
    d1 = vb.vi_diagnostics(opt_param, objective=objective)
    assert d1['khat'] < .1 and d1['d2'] < 0.1
    assert d1['samples'].shape == (2, 100000) and d1['smoothed_log_weights'].shape == (100000,)
    for key in ('W1', 'W2', 'mean_error', 'std_error', 'cov_error', 'log_norm_bound'):
        assert np.isfinite(d1[key])
    assert 'All diagnostics pass.' in capsys.readouterr().out
    d2 = vb.vi_diagnostics(opt_param, approx=approx, model=vb.GaussianModel(np.zeros(2), 3 * np.ones(2)))
    assert d2['khat'] > 0.7 and 'd2' not in d2
    assert 'not running further diagnostics' in capsys.readouterr().out
    d3 = vb.vi_diagnostics(opt_param, approx=approx, model=vb.GaussianModel(np.zeros(2), .5 * np.ones(2)))
    assert d3['khat'] < 0 and d3['d2'] > 2
    with pytest.raises(ValueError):
        vb.vi_diagnostics(opt_param)
    with pytest.raises(ValueError):
        vb.vi_diagnostics(opt_param, objective=objective, model=objective.model)
    with pytest.raises(ValueError):
        vb.vi_diagnostics(opt_param, objective=objective, n_samples=0)


def test_vi_diagnostics_generic_path(vb, capsys):
    """A family / target pair without the fused row-statistics path goes through model(samples) on the GPU,
    log q on the host and vb_psis_smooth with uploaded weights."""
    D = 3
    rng = np.random.RandomState(2)
    A = rng.randn(D, D)
    S = A @ A.T / D + np.eye(D)
    model = vb.CorrelatedGaussianModel(np.zeros(D), covariance=S)
    approx = vb.FullRankGaussian(D)
    theta = approx.pack(np.zeros(D), np.linalg.cholesky(S))
    res = vb.vi_diagnostics(theta, approx=approx, model=model, n_samples=20000)
    capsys.readouterr()
    assert res['khat'] < 0.1 and res['d2'] < 0.01


@pytest.mark.parametrize('kind', ['gauss_full', 'logistic'])
def test_model_call_dense_targets(vb, kind):
    """Model.__call__ (models.py:27-39) for the targets whose density needs a GEMM."""
    rng = np.random.RandomState(8)
    D, N = 45, 777
    if kind == 'gauss_full':
        A = rng.randn(D, D)
        S = A @ A.T / D + np.eye(D)
        mean = rng.randn(D)
        model, omodel = vb.CorrelatedGaussianModel(mean, covariance=S), omod.GaussFull(mean, np.linalg.inv(S))
    else:
        X = rng.randn(300, D) / np.sqrt(D)
        y = (rng.rand(300) < 0.5).astype(float)
        model, omodel = vb.LogisticRegressionModel(X, y, prior_sd=3.0), omod.Logistic(X, y, prior_sd=3.0)
    x = rng.randn(N, D)
    f, fo = model(x), omodel.logp(x)
    np.testing.assert_allclose(f, fo, rtol=0, atol=1e-12 * np.max(np.abs(fo)))
    assert model(x[0]).shape == (1,)


def test_vi_diagnostics_philox_mode(vb, capsys):
    """Throughput mode: the diagnostics' 1e5 base draws come from the device generator (normal and Student-t)."""
    D = 6
    model = vb.GaussianModel(np.zeros(D), np.ones(D))
    for approx in (vb.MFGaussian(D, rng='philox'), vb.MFStudentT(D, 40.0, rng='philox')):
        theta = np.concatenate([0.01 * np.ones(D), 0.02 * np.ones(D)])
        res = vb.vi_diagnostics(theta, approx=approx, model=model)
        capsys.readouterr()
        assert res['khat'] < 0.7 and res['d2'] < 0.5
        assert res['samples'].shape == (D, 100000)
        assert abs(res['samples'].mean()) < 0.02


def _psis_both(lw, reff=1.0):
    import os
    from viabel_amd._psis import psislw
    saved = os.environ.get('VB_PSIS_GRID')
    try:
        os.environ['VB_PSIS_GRID'] = '1'
        grid = psislw(lw, Reff=reff)
        os.environ['VB_PSIS_GRID'] = '0'
        single = psislw(lw, Reff=reff)
    finally:
        os.environ.pop('VB_PSIS_GRID', None)
        if saved is not None:
            os.environ['VB_PSIS_GRID'] = saved
    return grid, single


@pytest.mark.parametrize('n', [1025, 2048, 5000, 16384, 16385, 40000, 65536, 100000, 262144])
@pytest.mark.parametrize('kind', ['student', 'clustered', 'ties', 'light'])
def test_grid_kernel_equals_single_workgroup_kernel(vb, n, kind):
    """The multi-workgroup smoothing (psis_grid_kernel: one slice per workgroup, grid barriers) against the
    single-workgroup kernel on the same weights: the tail selection and the GPD fit make the same comparisons and add
    in the same order (k-hat bit for bit); the normalising constant adds slice by slice instead of strided (1e-13)."""
    rng = np.random.RandomState(n % 1000 + len(kind))
    if kind == 'student':
        lw = 2.0 * rng.standard_t(3.0, n)
    elif kind == 'clustered':
        lw = -50.0 + 0.3 * rng.randn(n)
    elif kind == 'ties':
        lw = np.round(1.5 * rng.standard_t(4.0, n), 1)          # many equal weights, also at the cut-off
    else:
        lw = -0.5 * rng.randn(n) ** 2                            # bounded above: k-hat below 1/3, nothing is replaced
    (gs, gk), (ss, sk) = _psis_both(lw)
    assert gk == sk or (np.isinf(gk) and np.isinf(sk)), (gk, sk)
    np.testing.assert_allclose(gs, ss, rtol=0, atol=1e-12)
    ref, rk = opsis.psis_smooth(lw)
    assert _close_k(gk, rk)
    np.testing.assert_allclose(gs, ref, rtol=0, atol=1e-10)


def test_grid_kernel_repeated_launches_share_the_barrier_counter(vb):
    """The barrier counter runs on from launch to launch (no reset between them): sizes with different workgroup counts
    and pass counts in one process, twice."""
    rng = np.random.RandomState(5)
    for _ in range(2):
        for n in (3000, 16384, 70000, 2000):
            lw = 1.7 * rng.standard_t(3.5, n)
            (gs, gk), (ss, sk) = _psis_both(lw)
            assert gk == sk
            np.testing.assert_allclose(gs, ss, rtol=0, atol=1e-12)


@pytest.mark.parametrize('n', [1500, 16384, 70000])
def test_grid_kernel_degenerate_inputs(vb, n):
    """All weights equal (no element above the cut-off: nothing to fit, k-hat infinite), a block of -inf weights
    (zero importance weights stay zero), and a tail far narrower than a radix bin: the multi-workgroup kernel against
    the single-workgroup one."""
    rng = np.random.RandomState(n)
    cases = {
        'equal': np.full(n, -3.25),
        'minus_inf': np.where(rng.rand(n) < 0.3, -np.inf, rng.randn(n)),
        'narrow': -10.0 + 1e-9 * rng.rand(n),
        'two_values': np.where(rng.rand(n) < 0.01, 0.5, -0.5),
    }
    for name, lw in cases.items():
        (gs, gk), (ss, sk) = _psis_both(lw)
        assert gk == sk or (np.isnan(gk) and np.isnan(sk)) or (np.isinf(gk) and np.isinf(sk)), (name, gk, sk)
        fin = np.isfinite(ss)
        assert np.array_equal(fin, np.isfinite(gs)), name
        np.testing.assert_allclose(gs[fin], ss[fin], rtol=0, atol=1e-12, err_msg=name)
        assert np.array_equal(gs[~fin], ss[~fin], equal_nan=True), name


def test_grid_kernel_reff(vb):
    rng = np.random.RandomState(12)
    lw = 1.8 * rng.standard_t(3.0, 30000)
    for reff in (0.3, 1.0, 2.5):
        (gs, gk), (ss, sk) = _psis_both(lw, reff)
        assert gk == sk
        np.testing.assert_allclose(gs, ss, rtol=0, atol=1e-12)
        ref, rk = opsis.psis_smooth(lw, reff)
        assert _close_k(gk, rk)
