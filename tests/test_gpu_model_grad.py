"""Per-sample model gradients through the C ABI (vb_model_grad / DeviceModel.grad / check_gradient).

The reference differentiates its models with autograd and checks the result with check_vjp
(viabel/tests/test_models.py:13-15); here every target carries its own device gradient -- the one the objectives
use -- and this entry shows it to the caller.  Oracle: oracle/models.py (closed-form gradients)."""
import numpy as np
import pytest

from oracle import models as omod

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import viabel_amd
    from viabel_amd import _lib
    _lib.default_engine()
    return viabel_amd


def _targets(vb, D, rng):
    m, sd = rng.randn(D), np.exp(0.3 * rng.randn(D))
    A = rng.randn(D, D)
    S = A @ A.T / D + np.eye(D)
    n_data = 3 * D + 7
    X = rng.randn(n_data, D) / np.sqrt(D)
    yb = (rng.rand(n_data) < 0.5).astype(float)
    yc = rng.poisson(2.0, size=n_data).astype(float)
    yr = X @ rng.randn(D) + 0.1 * rng.randn(n_data)
    return [
        ('gauss_diag', vb.GaussianModel(m, sd), omod.GaussDiag(m, sd)),
        ('funnel', vb.FunnelModel(D, D // 2), omod.Funnel(D, D // 2)),
        ('gauss_full', vb.CorrelatedGaussianModel(m, covariance=S), omod.GaussFull(m, np.linalg.inv(S))),
        ('logistic', vb.LogisticRegressionModel(X, yb, prior_sd=3.0), omod.Logistic(X, yb, prior_sd=3.0)),
        ('poisson', vb.PoissonRegressionModel(X, yc, prior_sd=2.0), omod.Poisson(X, yc, prior_sd=2.0)),
        ('linear', vb.LinearRegressionModel(X, yr, prior_sd=4.0, noise_sd=0.5),
         omod.LinearRegression(X, yr, prior_sd=4.0, noise_sd=0.5)),
    ]


@pytest.mark.parametrize('D,N', [(2, 5), (45, 777), (130, 64), (257, 300)])
def test_model_grad_matches_oracle(vb, D, N):
    rng = np.random.RandomState(D + N)
    x = 0.4 * rng.randn(N, D)
    for name, model, omodel in _targets(vb, D, rng):
        g, go = model.grad(x), omodel.grad(x)
        assert g.shape == (N, D)
        np.testing.assert_allclose(g, go, rtol=0, atol=1e-12 * np.max(np.abs(go)), err_msg=name)
        assert model.grad(x[0]).shape == (D,)
        np.testing.assert_allclose(model.grad(x[0]), go[0], rtol=0, atol=1e-12 * np.max(np.abs(go)), err_msg=name)


def test_model_grad_returns_f_too(vb):
    """vb_model_grad's f output is the same number Model.__call__ gives (one pass for both)."""
    from viabel_amd import _lib
    rng = np.random.RandomState(2)
    D, N = 33, 100
    x = 0.3 * rng.randn(N, D)
    eng = _lib.default_engine()
    for name, model, omodel in _targets(vb, D, rng):
        eng.set_model(model.device_spec())
        f, g = eng.model_grad(x)
        fo = omodel.logp(x)
        np.testing.assert_allclose(f, fo, rtol=0, atol=1e-12 * np.max(np.abs(fo)), err_msg=name)
        np.testing.assert_allclose(f, model(x), rtol=0, atol=1e-13 * np.max(np.abs(fo)), err_msg=name)


def test_check_gradient_builtin_targets(vb):
    rng = np.random.RandomState(4)
    D = 12
    x = 0.3 * rng.randn(7, D)
    for name, model, _ in _targets(vb, D, rng):
        assert model.check_gradient(x) < 1e-6, name


GOOD = r"""
__device__ double vb_log_density(const double* z, int d, const double* p, double* g) {
  double f = 0.0;                                   // banana: z1 ~ N(0, s^2), z_j ~ N(b z1^2, 1)
  const double s = p[0], b = p[1];
  f -= 0.5 * z[0] * z[0] / (s * s);
  double g0 = -z[0] / (s * s);
  for (int j = 1; j < d; ++j) {
    const double r = z[j] - b * z[0] * z[0];
    f -= 0.5 * r * r;
    if (g) g[j] = -r;
    g0 += 2.0 * b * z[0] * r;
  }
  if (g) g[0] = g0;
  return f;
}
"""
BAD = GOOD.replace('g0 += 2.0 * b * z[0] * r;', 'g0 += b * z[0] * r;')          # a forgotten factor of two


def test_check_gradient_source_model(vb):
    """The use it is for: a hand-written gradient that does not match its density is caught."""
    D = 5
    x = np.random.RandomState(0).randn(9, D)
    good = vb.SourceModel(D, GOOD, [2.0, 0.3])
    bad = vb.SourceModel(D, BAD, [2.0, 0.3])
    assert good.check_gradient(x) < 1e-7
    assert bad.check_gradient(x) > 1e-2
    g = good.grad(x)
    r = x[:, 1:] - 0.3 * x[:, :1] ** 2
    expect = np.concatenate([(-x[:, :1] / 4.0 + 0.6 * x[:, :1] * r.sum(axis=1, keepdims=True)), -r], axis=1)
    np.testing.assert_allclose(g, expect, rtol=1e-13, atol=1e-13)


def test_model_grad_argument_checks(vb):
    model = vb.GaussianModel(np.zeros(3), np.ones(3))
    with pytest.raises(ValueError):
        model.grad(np.zeros((4, 2)))
    with pytest.raises(ValueError):
        model.grad(np.zeros((2, 2, 3)))


def test_model_grad_edge_shapes(vb):
    """One point, one dimension, ragged widths (row strides are padded to 16 doubles on the device)."""
    rng = np.random.RandomState(9)
    g = vb.GaussianModel([0.5], [2.0]).grad(np.array([[1.5]]))
    np.testing.assert_allclose(g, [[-(1.5 - 0.5) / 4.0]], rtol=1e-15)
    for D in (1, 15, 16, 17, 33):
        m, sd = rng.randn(D), np.exp(0.2 * rng.randn(D))
        x = rng.randn(3, D)
        np.testing.assert_allclose(vb.GaussianModel(m, sd).grad(x), omod.GaussDiag(m, sd).grad(x), rtol=1e-13, atol=1e-14)
    src = vb.SourceModel(17, GOOD, [1.5, -0.2])
    x = rng.randn(1, 17)
    assert src.check_gradient(x) < 1e-7
    assert src.grad(x).shape == (1, 17) and src.grad(x[0]).shape == (17,)
