"""GPU parity of the full-rank Gaussian ExclusiveKL path (fp64 MFMA GEMMs) against the oracle.

No reference class exists for this family: the oracle's full-rank formulas are pinned
by reduction to MFGaussian and by torch.autograd fp64 (tests/test_oracle_golden.py); here the HIP
path must match the oracle on the same noise.  Tolerance: 1e-11 relative to max|grad| (K up to
4096 fp64 FMA chains in a different order than numpy's BLAS), value 1e-12.
"""
import numpy as np
import pytest

import _golden as G
from oracle import families as ofam
from oracle import models as omod
from oracle import objectives as oobj

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import viabel_amd
    from viabel_amd import _lib
    _lib.default_engine()
    return viabel_amd


def _models(vb, D, rng):
    mean, sd = rng.randn(D), np.exp(0.3 * rng.randn(D))
    A = rng.randn(D, D)
    S = A @ A.T / D + np.eye(D)
    P = np.linalg.inv(S)
    P = 0.5 * (P + P.T)
    m2 = rng.randn(D)
    out = [(vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)),
           (vb.CorrelatedGaussianModel(m2, precision=P), omod.GaussFull(m2, P))]
    if D >= 2:
        out.append((vb.FunnelModel(D, D // 2, 1.2), omod.Funnel(D, D // 2, 1.2)))
    return out


def _theta(fr, D, rng):
    L = np.tril(0.05 * rng.randn(D, D), -1) + np.diag(np.exp(-1.0 + 0.2 * rng.randn(D)))
    return fr.pack(0.3 * rng.randn(D), L)


# (48, 160) ... (272, 528): every k range a whole number of 16-deep slabs (the LDS-DMA GEMM kernels) with rows / columns
# that end inside a tile: clamped operand fetches, masked and paired 16-byte epilogue stores, skipped zero slabs
@pytest.mark.parametrize('D,N', [(3, 8), (5, 30), (64, 333), (130, 257), (200, 1000), (512, 4096), (1024, 4096),
                                 (48, 160), (112, 400), (144, 137), (80, 1024), (272, 528),
                                 # short shards of a wide family: both N x D x D products with their k range cut into
                                 # 4 / 4 / 2 / 3 pieces (fr_zsum_kernel, fr_gsum_kernel)
                                 # (the funnel's sampling product alone, the correlated Gaussian's two)
                                 (1024, 256), (1024, 512), (1024, 1024), (768, 384)])
def test_fullrank_against_oracle(vb, D, N):
    rng = np.random.RandomState(D + N)
    ofr = ofam.FullRankGaussian(D)
    theta = _theta(ofr, D, rng)
    for model, omodel in _models(vb, D, rng):
        approx = vb.FullRankGaussian(D, seed=4)
        value, grad = vb.ExclusiveKL(approx, model, N)(theta)
        noise = np.random.RandomState(4).randn(N, D)
        ov, og = oobj.exclusive_kl(ofr, omodel, theta, noise)
        assert G.rel_err(value, ov) < 1e-12, (type(omodel).__name__, value, ov)
        assert G.rel_err(grad, og) < 1e-11, (type(omodel).__name__, G.rel_err(grad, og))


@pytest.mark.parametrize('D,N', [(1, 4), (3, 8), (5, 30), (64, 333), (130, 257), (200, 1000), (512, 4096), (1024, 4096)])
def test_fullrank_path_derivative_against_oracle(vb, D, N):
    """use_path_deriv=True (objectives.py:156-159) for the dense Gaussian: the score L^-T eps is added to the rows of
    G on the device (one more N x D x D / 2 product with a blocked triangular inverse of L); the oracle solves the
    triangular system per sample.  Sizes below the fused path take the two-pass column sums."""
    rng = np.random.RandomState(7 * D + N)
    ofr = ofam.FullRankGaussian(D)
    theta = _theta(ofr, D, rng)
    for model, omodel in _models(vb, D, rng):
        approx = vb.FullRankGaussian(D, seed=4)
        value, grad = vb.ExclusiveKL(approx, model, N, use_path_deriv=True)(theta)
        noise = np.random.RandomState(4).randn(N, D)
        ov, og = oobj.exclusive_kl(ofr, omodel, theta, noise, use_path_deriv=True)
        assert G.rel_err(value, ov) < 1e-12, (type(omodel).__name__, value, ov)
        assert G.rel_err(grad, og) < 1e-11, (type(omodel).__name__, G.rel_err(grad, og))
        # and it is a different estimator from the entropy form on the same noise
        v2, _ = vb.ExclusiveKL(vb.FullRankGaussian(D, seed=4), model, N)(theta)
        assert v2 != value


def test_fullrank_path_derivative_ill_conditioned_factor(vb):
    """Row scales of L spanning six orders of magnitude (condition number ~1e7): the Newton inverse stays accurate."""
    D, N = 96, 512
    rng = np.random.RandomState(3)
    ofr = ofam.FullRankGaussian(D)
    L = np.exp(np.linspace(-7.0, 7.0, D))[:, None] * (np.eye(D) + np.tril(0.1 * rng.randn(D, D), -1))
    theta = ofr.pack(rng.randn(D), L)
    model, omodel = _models(vb, D, rng)[0]
    value, grad = vb.ExclusiveKL(vb.FullRankGaussian(D, seed=4), model, N, use_path_deriv=True)(theta)
    ov, og = oobj.exclusive_kl(ofr, omodel, theta, np.random.RandomState(4).randn(N, D), use_path_deriv=True)
    assert G.rel_err(value, ov) < 1e-11
    assert G.rel_err(grad, og) < 1e-9


def test_fullrank_reduces_to_meanfield_on_device(vb):
    """Diagonal L: the MFMA path must agree with the streaming mean-field kernels."""
    D, N = 96, 512
    rng = np.random.RandomState(1)
    mu, ls = 0.3 * rng.randn(D), -1 + 0.2 * rng.randn(D)
    model = vb.FunnelModel(D)
    v_mf, g_mf = vb.ExclusiveKL(vb.MFGaussian(D, seed=9), model, N)(np.concatenate([mu, ls]))
    fr = vb.FullRankGaussian(D, seed=9)
    v_fr, g_fr = vb.ExclusiveKL(fr, model, N)(fr.pack(mu, np.diag(np.exp(ls))))
    assert abs(v_mf - v_fr) < 1e-12 * abs(v_mf)
    diag_pos = D + np.cumsum(np.arange(1, D + 1)) - 1
    np.testing.assert_allclose(g_fr[:D], g_mf[:D], rtol=0, atol=1e-11 * np.max(np.abs(g_mf)))
    np.testing.assert_allclose(g_fr[diag_pos], g_mf[D:], rtol=0, atol=1e-11 * np.max(np.abs(g_mf)))


def test_fullrank_family_api(vb):
    D = 4
    fr = vb.FullRankGaussian(D)
    assert fr.var_param_dim == D + D * (D + 1) // 2
    th = fr.init_param()
    mean, cov = fr.mean_and_cov(th)
    np.testing.assert_allclose(cov, np.exp(4.0) * np.eye(D))
    assert fr.supports_kl and fr.supports_entropy and fr.supports_pth_moment(2)
    with pytest.raises(ValueError):
        fr.pth_moment(th, 3)
    with pytest.raises(NotImplementedError):
        vb.ExclusiveKL(fr, vb.GaussianModel(np.zeros(D), np.ones(D)), 8, hessian_approx_method='full')
    x = fr.sample(th, 7)
    assert x.shape == (7, D)
    ofr = ofam.FullRankGaussian(D)
    np.testing.assert_allclose(fr.log_density(th, x), ofr.log_density(th, x), rtol=1e-13)
    np.testing.assert_allclose(fr.entropy(th), ofr.entropy(th), rtol=1e-14)


@pytest.mark.parametrize('kind', ['logistic', 'poisson', 'linear'])
@pytest.mark.parametrize('D,n_data,N', [(6, 40, 64), (48, 300, 500), (130, 257, 333)])
def test_fullrank_regression_targets(vb, kind, D, n_data, N):
    """Dense Gaussian family on the regression targets: the sampling GEMM feeds the eta / gradient GEMMs of the
    GLM likelihood; entropy form and path derivative against the oracle."""
    rng = np.random.RandomState(5 * D + len(kind))
    X = rng.randn(n_data, D) / np.sqrt(D)
    beta = 0.5 * rng.randn(D)
    if kind == 'logistic':
        y = (rng.rand(n_data) < 1 / (1 + np.exp(-X @ beta))).astype(float)
        model, omodel = vb.LogisticRegressionModel(X, y, 4.0), omod.Logistic(X, y, 4.0)
    elif kind == 'poisson':
        y = rng.poisson(np.exp(X @ beta)).astype(float)
        model, omodel = vb.PoissonRegressionModel(X, y, 4.0), omod.Poisson(X, y, 4.0)
    else:
        y = X @ beta + 0.5 * rng.randn(n_data)
        model, omodel = vb.LinearRegressionModel(X, y, 4.0, noise_sd=0.5), omod.LinearRegression(X, y, 4.0, 0.5)
    ofr = ofam.FullRankGaussian(D)
    theta = _theta(ofr, D, rng)
    noise = np.random.RandomState(4).randn(N, D)
    for pd in (False, True):
        value, grad = vb.ExclusiveKL(vb.FullRankGaussian(D, seed=4), model, N, use_path_deriv=pd)(theta)
        ov, og = oobj.exclusive_kl(ofr, omodel, theta, noise, use_path_deriv=pd)
        assert G.rel_err(value, ov) < 1e-12, (kind, pd, value, ov)
        assert G.rel_err(grad, og) < 1e-11, (kind, pd, G.rel_err(grad, og))
