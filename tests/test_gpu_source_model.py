"""SourceModel (VB_MODEL_SOURCE): a log density handed to the engine as HIP device code -- the adaptor for what the
reference does with an arbitrary Python callable and autograd (viabel/models.py:17-39, convenience.py:75).  The target
is the robust (Student-t) regression of the reference's docs (docs/source/robust-regression.ipynb), whose Stan model
is outside the built-in set; the oracle evaluates the same density and gradient in numpy."""
import numpy as np
import pytest

import _golden as G
from oracle import families as ofam, objectives as oobj

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import viabel_amd
    from viabel_amd import _lib
    _lib.default_engine()
    return viabel_amd

ROBUST_REGRESSION_SRC = r"""
// params = [n, nu, s, tau | X (n x d, row-major) | y (n)]
//   y_i ~ StudentT(nu, x_i' z, s),  z ~ N(0, tau^2 I)     (log density up to constants)
__device__ double vb_log_density(const double* z, int d, const double* p, double* g) {
  const int n = (int)p[0];
  const double nu = p[1], s = p[2], tau = p[3];
  const double* X = p + 4;
  const double* y = X + (long long)n * d;
  double f = 0.0;
  for (int j = 0; j < d; ++j) {
    f -= 0.5 * z[j] * z[j] / (tau * tau);
    if (g) g[j] = -z[j] / (tau * tau);
  }
  for (int i = 0; i < n; ++i) {
    double eta = 0.0;
    for (int j = 0; j < d; ++j) eta += X[(long long)i * d + j] * z[j];
    const double r = y[i] - eta, q = 1.0 + r * r / (nu * s * s);
    f -= 0.5 * (nu + 1.0) * log(q);
    if (g) {
      const double c = (nu + 1.0) * r / (nu * s * s * q);
      for (int j = 0; j < d; ++j) g[j] += c * X[(long long)i * d + j];
    }
  }
  return f;
}
"""


class RobustRegressionOracle:
    def __init__(self, X, y, nu, s, tau):
        self.X, self.y, self.nu, self.s, self.tau = X, y, nu, s, tau
        self.dim = X.shape[1]

    def logp(self, z):
        z = np.atleast_2d(z)
        r = self.y[None, :] - z @ self.X.T
        q = 1.0 + r * r / (self.nu * self.s ** 2)
        return -0.5 * np.sum(z * z, axis=1) / self.tau ** 2 - 0.5 * (self.nu + 1.0) * np.sum(np.log(q), axis=1)

    def grad(self, z):
        z = np.atleast_2d(z)
        r = self.y[None, :] - z @ self.X.T
        q = 1.0 + r * r / (self.nu * self.s ** 2)
        c = (self.nu + 1.0) * r / (self.nu * self.s ** 2 * q)
        return -z / self.tau ** 2 + c @ self.X


    def hessian(self, m):            # -I / tau^2 - X' diag(dc/dr) X at one point
        m = np.asarray(m, dtype=float).ravel()
        r = self.y - self.X @ m
        a = self.nu * self.s ** 2
        q = 1.0 + r * r / a
        dc = (self.nu + 1.0) / a * (1.0 - r * r / a) / q ** 2
        return -np.eye(self.dim) / self.tau ** 2 - (self.X * dc[:, None]).T @ self.X

    def hvp(self, m, V):             # rows of V times the Hessian at m
        return np.atleast_2d(V) @ self.hessian(m)


def _problem(vb, D, n_data, seed=3):
    rng = np.random.RandomState(seed)
    X = rng.randn(n_data, D)
    beta = rng.randn(D)
    y = X @ beta + 0.3 * rng.standard_t(3.0, size=n_data)
    nu, s, tau = 4.0, 0.5, 3.0
    params = np.concatenate([[n_data, nu, s, tau], X.ravel(), y])
    return vb.SourceModel(D, ROBUST_REGRESSION_SRC, params), RobustRegressionOracle(X, y, nu, s, tau)


def test_source_model_call_matches_oracle(vb):
    model, omodel = _problem(vb, 7, 40)
    x = np.random.RandomState(0).randn(33, 7)
    np.testing.assert_allclose(model(x), omodel.logp(x), rtol=1e-13, atol=1e-12)
    assert model(x[0]).shape == (1,)


@pytest.mark.parametrize('D,N,n_data', [(5, 64, 25), (24, 1000, 60), (130, 333, 40)])
@pytest.mark.parametrize('pd', [False, True])
def test_source_model_meanfield_against_oracle(vb, D, N, n_data, pd):
    model, omodel = _problem(vb, D, n_data, seed=D)
    rng = np.random.RandomState(D + N)
    theta = np.concatenate([0.2 * rng.randn(D), -1.0 + 0.1 * rng.randn(D)])
    for fam, ofamily in ((vb.MFGaussian(D, seed=5), ofam.MFGaussian(D)),
                         (vb.MFStudentT(D, 8.0, seed=5), ofam.MFStudentT(D, 8.0))):
        value, grad = vb.ExclusiveKL(fam, model, N, use_path_deriv=pd)(theta)
        noise = ofamily.draw_noise(np.random.RandomState(5), N)
        ov, og = oobj.exclusive_kl(ofamily, omodel, theta, noise, use_path_deriv=pd)
        assert G.rel_err(value, ov) < 1e-12, (type(fam).__name__, value, ov)
        assert G.rel_err(grad, og) < 1e-11, (type(fam).__name__, G.rel_err(grad, og))


@pytest.mark.parametrize('D,N,n_data', [(6, 100, 30), (48, 512, 50), (70, 257, 33)])
@pytest.mark.parametrize('pd', [False, True])
def test_source_model_fullrank_against_oracle(vb, D, N, n_data, pd):
    model, omodel = _problem(vb, D, n_data, seed=2 * D)
    rng = np.random.RandomState(D)
    ofr = ofam.FullRankGaussian(D)
    L = np.tril(0.05 * rng.randn(D, D), -1) + np.diag(np.exp(-1.0 + 0.2 * rng.randn(D)))
    theta = ofr.pack(0.1 * rng.randn(D), L)
    value, grad = vb.ExclusiveKL(vb.FullRankGaussian(D, seed=4), model, N, use_path_deriv=pd)(theta)
    noise = np.random.RandomState(4).randn(N, D)
    ov, og = oobj.exclusive_kl(ofr, omodel, theta, noise, use_path_deriv=pd)
    assert G.rel_err(value, ov) < 1e-12, (value, ov)
    assert G.rel_err(grad, og) < 1e-11, G.rel_err(grad, og)


def test_source_model_fit_recovers_coefficients(vb, capsys):
    """bbvi-style use: a mean-field fit on the device (Philox noise, vb_fit) of the robust regression."""
    from viabel_amd.optimization import RMSProp
    D, n_data = 4, 400
    rng = np.random.RandomState(1)
    X = rng.randn(n_data, D)
    beta = np.array([1.0, -2.0, 0.5, 3.0])
    y = X @ beta + 0.2 * rng.standard_t(4.0, size=n_data)
    params = np.concatenate([[n_data, 4.0, 0.2, 10.0], X.ravel(), y])
    model = vb.SourceModel(D, ROBUST_REGRESSION_SRC, params)
    obj = vb.ExclusiveKL(vb.MFGaussian(D, rng='philox', seed=2), model, 64)
    res = RMSProp(0.05).optimize(1500, obj, np.zeros(2 * D))
    capsys.readouterr()
    mean = res['opt_param'][:D] if 'opt_param' in res else res['variational_param_history'][-1][:D]
    assert np.max(np.abs(mean - beta)) < 0.1, mean


def test_source_model_errors(vb):
    bad = vb.SourceModel(3, '__device__ double vb_log_density(const double* z, int d) { return 0; }')   # wrong signature
    with pytest.raises(ValueError, match='does not compile'):
        bad(np.zeros(3))
    ok = vb.SourceModel(3, '__device__ double vb_log_density(const double* z, int d, const double* p, double* g) '
                           '{ double f = 0; for (int j = 0; j < d; ++j) { f -= 0.5 * z[j] * z[j]; if (g) g[j] = -z[j]; } return f; }')
    # control variates of a source model (round 3): for this quadratic f the `full` variate is exact -- the mean block of
    # the gradient is mu itself whatever the ten noise rows are
    th = np.array([0.3, -0.2, 0.1, -0.5, 0.0, 0.2])
    grad = vb.ExclusiveKL(vb.MFGaussian(3), ok, 10, hessian_approx_method='full')(th)[1]
    assert np.max(np.abs(grad[:3] - th[:3])) < 1e-9
    with pytest.raises(ValueError):
        vb.SourceModel(3, '')


def test_bbvi_with_source_string(vb, capsys):
    """`bbvi(dim, log_density=...)` -- the reference's primary entry (convenience.py:75) -- with the density as HIP source:
    a banana-shaped 2-D target, default RAABBVI path, device-resident chunks."""
    src = r'''
    __device__ double vb_log_density(const double* z, int d, const double* p, double* g) {
      const double b = 0.5, u = z[1] + b * (z[0] * z[0] - 1.0);        // N(0, 2^2) x N(0, 1) bent along z0^2
      if (g) { g[0] = -z[0] / 4.0 - 2.0 * b * z[0] * u; g[1] = -u; }
      return -0.5 * z[0] * z[0] / 4.0 - 0.5 * u * u;
    }'''
    res = vb.bbvi(2, log_density=src, n_iters=1500, num_mc_samples=64, learning_rate=0.05, RAABBVI_kwargs=dict(mcse_threshold=0.01))
    capsys.readouterr()
    assert isinstance(res['objective'].model, vb.SourceModel)
    mean = res['opt_param'][:2]
    assert abs(mean[0]) < 0.5 and np.all(np.isfinite(res['opt_param']))


@pytest.mark.parametrize('D,N,n_data', [(6, 100, 30), (48, 512, 50)])
def test_source_model_alpha_fullrank_against_oracle(vb, D, N, n_data):
    model, omodel = _problem(vb, D, n_data, seed=5 * D)
    rng = np.random.RandomState(D)
    ofr = ofam.FullRankGaussian(D)
    L = np.tril(0.05 * rng.randn(D, D), -1) + np.diag(np.exp(-1.0 + 0.2 * rng.randn(D)))
    theta = ofr.pack(0.1 * rng.randn(D), L)
    for alpha in (2.0, 0.5):
        np.random.seed(11)
        value, grad = vb.AlphaDivergence(vb.FullRankGaussian(D), model, N, alpha)(theta)
        np.random.seed(11)
        noise = np.random.RandomState(np.random.randint(2 ** 32)).randn(N, D)
        ov, og = oobj.alpha_divergence(ofr, omodel, theta, noise, alpha)
        assert G.rel_err(value, ov) < 1e-12, (alpha, value, ov)
        assert G.rel_err(grad, og) < 1e-11, (alpha, G.rel_err(grad, og))


@pytest.mark.parametrize('D,N,n_data', [(6, 100, 30), (70, 333, 40)])
@pytest.mark.parametrize('pd', [False, True])
def test_source_model_multivariate_t_against_oracle(vb, D, N, n_data, pd):
    model, omodel = _problem(vb, D, n_data, seed=7 * D)
    rng = np.random.RandomState(D)
    B = rng.randn(D, D)
    theta = np.concatenate([0.2 * rng.randn(D), ofam.psd_to_free(0.05 * (B @ B.T / D + 0.5 * np.eye(D)))])
    value, grad = vb.ExclusiveKL(vb.MultivariateT(D, 9.0, seed=6), model, N, use_path_deriv=pd)(theta)
    noise = ofam.MultivariateT(D, 9.0).draw_noise(np.random.RandomState(6), N)
    ov, og = oobj.exclusive_kl(ofam.MultivariateT(D, 9.0), omodel, theta, noise, pd)
    assert abs(value - ov) <= 1e-12 * abs(ov), (value, ov)
    np.testing.assert_allclose(grad, og, rtol=0, atol=1e-10 * np.max(np.abs(og)))


@pytest.mark.parametrize('family', ['multivariate_t', 'fullrank_gaussian'])
@pytest.mark.parametrize('use_resampling', [True, False])
def test_source_model_dis_dense_against_oracle(vb, family, use_resampling):
    """DISInclusiveKL over the dense families needs log p of the state samples only: the source model's row kernel with a
    NULL gradient.  Three calls with a moving theta, as tests/test_gpu_objectives.py does for the built-in targets."""
    D, N, df, n_data = 12, 900, 40, 30
    model, omodel = _problem(vb, D, n_data, seed=21)
    rng = np.random.RandomState(9)
    if family == 'multivariate_t':
        approx, ofamily = vb.MultivariateT(D, df, seed=6), ofam.MultivariateT(D, df)
    else:
        approx, ofamily = vb.FullRankGaussian(D, seed=6), ofam.FullRankGaussian(D)
    prior = np.concatenate([np.zeros(D), np.log(3.0) * np.ones(D)])
    kw = dict(use_resampling=use_resampling, num_resampling_batches=2)
    obj = vb.DISInclusiveKL(approx, model, N, ess_target=200, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, **kw)
    ref = oobj.DISInclusiveKL(ofamily, omodel, N, 200, ofam.MFGaussian(D), prior, **kw)
    A = rng.randn(D, D)
    theta = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(A @ A.T / D + 0.7 * np.eye(D))])
    rs = np.random.RandomState(6)
    np.random.seed(12)
    for step in range(3):
        state = np.random.get_state()
        value, grad = obj(theta)
        np.random.set_state(state)
        noise = ofamily.draw_noise(rs, N) if ref.needs_refresh() else None
        if use_resampling:
            if ref.needs_refresh():
                ref.refresh(theta, noise)
            idx = np.random.choice(N, size=ref._resampling_batch_size, p=ref._state_w_normalized)
            ref._objective_step += 1
            xs = ref._state_samples[idx]
            scale = ref._state_w_sum / N
            ov = np.mean(-ofamily.log_density(theta, xs)) * scale
            og = -ofamily.log_density_grad_weighted(theta, xs, np.ones(len(idx))) / len(idx) * scale
        else:
            ov, og = ref(theta, noise=noise)
        assert G.rel_err(obj._eps, ref._eps) < 1e-10
        assert G.rel_err(value, ov) < 1e-10, (step, value, ov)
        assert G.rel_err(grad, og) < 1e-9, (step, G.rel_err(grad, og))
        theta = theta - 0.01 * grad / (1 + np.abs(grad))


@pytest.mark.parametrize('student', [False, True])
@pytest.mark.parametrize('use_resampling', [True, False])
def test_source_model_dis_meanfield_against_oracle(vb, student, use_resampling):
    """Mean-field DIS: f of the state samples from the source model's row kernel (samples materialised for it), the base
    log density from the row-statistics kernel as for the built-in targets."""
    D, N, n_data = 10, 800, 30
    model, omodel = _problem(vb, D, n_data, seed=31)
    rng = np.random.RandomState(4)
    if student:
        approx, ofamily = vb.MFStudentT(D, 12.0, seed=6), ofam.MFStudentT(D, 12.0)
    else:
        approx, ofamily = vb.MFGaussian(D, seed=6), ofam.MFGaussian(D)
    prior = np.concatenate([np.zeros(D), np.log(3.0) * np.ones(D)])
    kw = dict(use_resampling=use_resampling, num_resampling_batches=2)
    obj = vb.DISInclusiveKL(approx, model, N, ess_target=200, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, **kw)
    ref = oobj.DISInclusiveKL(ofamily, omodel, N, 200, ofam.MFGaussian(D), prior, **kw)
    theta = np.concatenate([0.1 * rng.randn(D), -0.5 + 0.1 * rng.randn(D)])
    rs = np.random.RandomState(6)
    np.random.seed(12)
    for step in range(3):
        state = np.random.get_state()
        value, grad = obj(theta)
        np.random.set_state(state)
        noise = ofamily.draw_noise(rs, N) if ref.needs_refresh() else None
        if use_resampling:
            if ref.needs_refresh():
                ref.refresh(theta, noise)
            idx = np.random.choice(N, size=ref._resampling_batch_size, p=ref._state_w_normalized)
            ref._objective_step += 1
            xs = ref._state_samples[idx]
            scale = ref._state_w_sum / N
            ov = np.mean(-ofamily.log_density(theta, xs)) * scale
            og = -ofamily.log_density_grad_weighted(theta, xs, np.ones(len(idx))) / len(idx) * scale
        else:
            ov, og = ref(theta, noise=noise)
        assert G.rel_err(obj._eps, ref._eps) < 1e-10
        assert G.rel_err(value, ov) < 1e-10, (step, value, ov)
        assert G.rel_err(grad, og) < 1e-9, (step, G.rel_err(grad, og))
        theta = theta - 0.01 * grad / (1 + np.abs(grad))


def test_source_model_device_log_weights(vb):
    """vi_diagnostics' importance log weights for a mean-field family on the device (vb_log_weights_meanfield) against
    model(samples) - log q(samples) formed on the host."""
    from viabel_amd import convenience
    D = 9
    model, omodel = _problem(vb, D, 25, seed=8)
    approx = vb.MFGaussian(D, seed=3)
    theta = np.concatenate([0.1 * np.arange(D), -0.7 * np.ones(D)])
    assert convenience._on_device_weights(model, approx)
    samples, lw, khat = convenience.psis_correction(theta, model, approx, 4000)
    assert np.isfinite(khat) and lw.shape == (4000,)
    x = samples.T
    raw = omodel.logp(x) - ofam.MFGaussian(D).log_density(theta, x)
    from viabel_amd._psis import psislw
    want, _ = psislw(raw.copy())
    assert G.rel_err(lw, want) < 1e-9


def test_two_source_models_take_turns(vb):
    """Binding a source model again (two objectives evaluated alternately) reuses the compiled module and re-uploads its
    parameters: results stay those of each model."""
    m1, o1 = _problem(vb, 6, 20, seed=1)
    m2, o2 = _problem(vb, 6, 35, seed=2)
    x = np.random.RandomState(0).randn(11, 6)
    for _ in range(3):
        np.testing.assert_allclose(m1(x), o1.logp(x), rtol=1e-13, atol=1e-12)
        np.testing.assert_allclose(m2(x), o2.logp(x), rtol=1e-13, atol=1e-12)


def test_source_model_fullrank_device_fit_matches_host_loop(vb, capsys):
    """The dense family's device-resident loop (vb_fit) with a source model: same trajectory as the host loop."""
    from viabel_amd.optimization import RMSProp
    D = 6
    model, _ = _problem(vb, D, 40, seed=13)
    hist = {}
    for on_device in (False, True):
        fam = vb.FullRankGaussian(D, rng='philox', seed=3)
        obj = vb.ExclusiveKL(fam, model, 128)
        res = RMSProp(0.02).optimize(60, obj, fam.init_param(), on_device=on_device)
        hist[on_device] = np.asarray(res['value_history'])
    capsys.readouterr()
    np.testing.assert_array_equal(hist[False], hist[True])


@pytest.mark.parametrize('use_resampling', [True, False])
def test_source_model_dis_lowrank_against_oracle(vb, use_resampling):
    D, k, N, n_data = 24, 4, 800, 30
    model, omodel = _problem(vb, D, n_data, seed=41)
    rng = np.random.RandomState(3 * D + k)
    approx, ofamily = vb.LRGaussian(D, seed=5, k=k), ofam.LRGaussian(D, k)
    prior = np.concatenate([np.zeros(D), np.log(3.0) * np.ones(D)])
    kw = dict(use_resampling=use_resampling, num_resampling_batches=2)
    obj = vb.DISInclusiveKL(approx, model, N, ess_target=N // 5, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, **kw)
    ref = oobj.DISInclusiveKL(ofamily, omodel, N, N // 5, ofam.MFGaussian(D), prior, **kw)
    theta = np.concatenate([0.1 * rng.randn(D), -0.5 + 0.1 * rng.randn(D), 0.2 * rng.randn(D * k) / np.sqrt(k)])
    rs = np.random.RandomState(5)
    np.random.seed(13)
    for step in range(3):
        state = np.random.get_state()
        value, grad = obj(theta)
        np.random.set_state(state)
        noise = ofamily.draw_noise(rs, N) if ref.needs_refresh() else None
        if use_resampling:
            if ref.needs_refresh():
                ref.refresh(theta, noise)
            idx = np.random.choice(N, size=ref._resampling_batch_size, p=ref._state_w_normalized)
            ref._objective_step += 1
            xs = ref._state_samples[idx]
            scale = ref._state_w_sum / N
            ov = np.mean(-ofamily.log_density(theta, xs)) * scale
            og = -ofamily.log_density_grad_weighted(theta, xs, np.ones(len(idx))) / len(idx) * scale
        else:
            ov, og = ref(theta, noise=noise)
        assert G.rel_err(obj._eps, ref._eps) < 1e-10
        assert G.rel_err(value, ov) < 1e-10, (step, value, ov)
        assert G.rel_err(grad, og) < 1e-9, (step, G.rel_err(grad, og))
        theta = theta - 0.005 * grad / (1 + np.abs(grad))


@pytest.mark.parametrize('D,N,n_data', [(8, 200, 30), (77, 333, 40)])
@pytest.mark.parametrize('family', ['gauss', 't'])
def test_source_model_alpha_meanfield_against_oracle(vb, D, N, n_data, family):
    """Mean-field AlphaDivergence: weights from the row kernel's f, weighted gradient from the row-scaled loaded G."""
    model, omodel = _problem(vb, D, n_data, seed=3 * D)
    rng = np.random.RandomState(D + N)
    theta = np.concatenate([0.2 * rng.randn(D), -1.0 + 0.1 * rng.randn(D)])
    if family == 'gauss':
        approx, ofamily = vb.MFGaussian(D), ofam.MFGaussian(D)
    else:
        approx, ofamily = vb.MFStudentT(D, 12), ofam.MFStudentT(D, 12)
    for alpha in (2.0, 0.5):
        np.random.seed(7)
        value, grad = vb.AlphaDivergence(approx, model, N, alpha)(theta)
        np.random.seed(7)
        seed = np.random.randint(2 ** 32)
        noise = ofamily.draw_noise(np.random.RandomState(seed), N)
        ov, og = oobj.alpha_divergence(ofamily, omodel, theta, noise, alpha)
        assert G.rel_err(value, ov) < 1e-12, (alpha, value, ov)
        assert G.rel_err(grad, og) < 1e-11, (alpha, G.rel_err(grad, og))


@pytest.mark.parametrize('D,k,N,n_data', [(6, 1, 100, 30), (24, 4, 800, 30), (130, 7, 333, 40), (200, 16, 1030, 20)])
@pytest.mark.parametrize('pd', [False, True])
def test_source_model_lowrank_against_oracle(vb, D, k, N, n_data, pd):
    """ExclusiveKL + LRGaussian: the streaming pass loads the user kernel's G beside the noise (vb_lowrank.hip)."""
    model, omodel = _problem(vb, D, n_data, seed=D + k)
    rng = np.random.RandomState(D + N)
    fam, ofamily = vb.LRGaussian(D, seed=7, k=k), ofam.LRGaussian(D, k)
    theta = fam.pack(0.2 * rng.randn(D), -1.0 + 0.1 * rng.randn(D), 0.2 * rng.randn(D, k) / np.sqrt(k))
    value, grad = vb.ExclusiveKL(fam, model, N, use_path_deriv=pd)(theta)
    noise = ofamily.draw_noise(np.random.RandomState(7), N)
    ov, og = oobj.exclusive_kl(ofamily, omodel, theta, noise, pd)
    assert G.rel_err(value, ov) < (1e-10 if pd else 1e-12), (value, ov)
    assert G.rel_err(grad, og) < (1e-9 if pd else 1e-11), G.rel_err(grad, og)


def test_source_model_lowrank_device_fit_matches_host_loop(vb, capsys):
    """vb_fit with the low-rank family and a source model: the device-resident loop == the host loop (Philox noise)."""
    from viabel_amd.optimization import RMSProp
    D, k, N = 12, 3, 96
    model, _ = _problem(vb, D, 50, seed=8)
    init = vb.LRGaussian(D, k=k).pack(np.zeros(D), np.zeros(D), 0.05 * np.random.RandomState(0).randn(D, k))
    hist = {}
    for on_device in (False, True):
        obj = vb.ExclusiveKL(vb.LRGaussian(D, seed=11, k=k, rng='philox'), model, N)
        res = RMSProp(0.02).optimize(40, obj, init.copy(), on_device=on_device)
        hist[on_device] = np.asarray(res['value_history'])
    capsys.readouterr()
    np.testing.assert_array_equal(hist[False], hist[True])


@pytest.mark.parametrize('D,N,n_data,df', [(6, 100, 30, 40.0), (70, 333, 40, 7.0)])
def test_source_model_alpha_multivariate_t_against_oracle(vb, D, N, n_data, df):
    model, omodel = _problem(vb, D, n_data, seed=7 * D)
    rng = np.random.RandomState(D)
    omvt = ofam.MultivariateT(D, df)
    L = np.tril(0.05 * rng.randn(D, D), -1) + np.diag(np.exp(-1.0 + 0.2 * rng.randn(D)))
    theta = np.concatenate([0.1 * rng.randn(D), ofam.chol_to_free(L)])
    for alpha in (2.0, 0.5):
        np.random.seed(17)
        value, grad = vb.AlphaDivergence(vb.MultivariateT(D, df), model, N, alpha)(theta)
        np.random.seed(17)
        noise = omvt.draw_noise(np.random.RandomState(np.random.randint(2 ** 32)), N)
        ov, og = oobj.alpha_divergence(omvt, omodel, theta, noise, alpha)
        assert G.rel_err(value, ov) < 1e-12, (alpha, value, ov)
        assert G.rel_err(grad, og) < 1e-10, (alpha, G.rel_err(grad, og))


@pytest.mark.parametrize('D,k,N,n_data', [(6, 1, 100, 30), (64, 4, 1000, 40), (130, 16, 777, 25)])
def test_source_model_alpha_lowrank_against_oracle(vb, D, k, N, n_data):
    model, omodel = _problem(vb, D, n_data, seed=D + 3 * k)
    rng = np.random.RandomState(D + k)
    ofamily = ofam.LRGaussian(D, k)
    theta = np.concatenate([0.1 * rng.randn(D), -0.7 + 0.2 * rng.randn(D), 0.3 * rng.randn(D * k) / np.sqrt(k)])
    for alpha in (2.0, 0.5):
        np.random.seed(77)
        value, grad = vb.AlphaDivergence(vb.LRGaussian(D, seed=2, k=k), model, N, alpha)(theta)
        np.random.seed(77)
        noise = ofamily.draw_noise(np.random.RandomState(np.random.randint(2 ** 32)), N)
        ov, og = oobj.alpha_divergence(ofamily, omodel, theta, noise, alpha)
        assert G.rel_err(value, ov) < 1e-11, (alpha, value, ov)
        assert G.rel_err(grad, og) < 1e-9, (alpha, G.rel_err(grad, og))


# the same density with VB_LOG_DENSITY_PARTS: K threads per sample, each summing its share of the observations
ROBUST_REGRESSION_PARTS_SRC = r"""
#define VB_LOG_DENSITY_PARTS %d
__device__ double vb_log_density_part(const double* z, int d, const double* p, double* g, int part, int n_parts) {
  const int n = (int)p[0];
  const double nu = p[1], s = p[2], tau = p[3];
  const double* X = p + 4;
  const double* y = X + (long long)n * d;
  double f = 0.0;
  if (part == 0)                       // the prior belongs to one part (the wrapper zeroed g)
    for (int j = 0; j < d; ++j) {
      f -= 0.5 * z[j] * z[j] / (tau * tau);
      if (g) g[j] = -z[j] / (tau * tau);
    }
  for (int i = part; i < n; i += n_parts) {
    double eta = 0.0;
    for (int j = 0; j < d; ++j) eta += X[(long long)i * d + j] * z[j];
    const double r = y[i] - eta, q = 1.0 + r * r / (nu * s * s);
    f -= 0.5 * (nu + 1.0) * log(q);
    if (g) {
      const double c = (nu + 1.0) * r / (nu * s * s * q);
      for (int j = 0; j < d; ++j) g[j] += c * X[(long long)i * d + j];
    }
  }
  return f;
}
"""


@pytest.mark.parametrize('K', [2, 16, 64])
@pytest.mark.parametrize('D,N,n_data', [(5, 64, 25), (24, 1000, 60), (128, 333, 40)])
def test_source_model_parts_against_oracle(vb, K, D, N, n_data):
    """K threads per sample: f, the gradient matrix and the objectives built on them against the numpy oracle (the K
    shares are added in a fixed butterfly order, so the tolerance is rounding, not bit equality with K = 1)."""
    rng = np.random.RandomState(D + K)
    X = rng.randn(n_data, D)
    y = X @ rng.randn(D) + 0.3 * rng.standard_t(3.0, size=n_data)
    params = np.concatenate([[n_data, 4.0, 0.5, 3.0], X.ravel(), y])
    model = vb.SourceModel(D, ROBUST_REGRESSION_PARTS_SRC % K, params)
    omodel = RobustRegressionOracle(X, y, 4.0, 0.5, 3.0)
    x = 0.3 * rng.randn(N, D)
    fo, go = omodel.logp(x), omodel.grad(x)
    np.testing.assert_allclose(model(x), fo, rtol=0, atol=1e-12 * np.max(np.abs(fo)))
    np.testing.assert_allclose(model.grad(x), go, rtol=0, atol=1e-12 * np.max(np.abs(go)))
    assert model.check_gradient(x[:3]) < 1e-6
    theta = np.concatenate([0.2 * rng.randn(D), -1.0 + 0.1 * rng.randn(D)])
    value, grad = vb.ExclusiveKL(vb.MFGaussian(D, seed=5), model, N)(theta)
    noise = np.random.RandomState(5).randn(N, D)
    ov, og = oobj.exclusive_kl(ofam.MFGaussian(D), omodel, theta, noise)
    assert G.rel_err(value, ov) < 1e-12 and G.rel_err(grad, og) < 1e-11
    again = vb.ExclusiveKL(vb.MFGaussian(D, seed=5), model, N)(theta)
    assert again[0] == value and np.array_equal(again[1], grad)          # reproducible to the bit


def test_source_model_parts_errors(vb):
    bad = vb.SourceModel(3, ROBUST_REGRESSION_PARTS_SRC % 3, np.zeros(4))          # not a power of two
    with pytest.raises(ValueError, match='power of two'):
        bad(np.zeros(3))
    wide = vb.SourceModel(200, ROBUST_REGRESSION_PARTS_SRC % 4, np.zeros(4))      # private arrays stop at 128
    with pytest.raises(NotImplementedError):
        wide(np.zeros(200))


@pytest.mark.parametrize('method', ['full', 'mean_only', 'loo_diag_approx', 'loo_direct_approx'])
@pytest.mark.parametrize('student,pd', [(False, False), (False, True), (True, False)])
def test_source_model_control_variates_against_literal_rge(vb, method, student, pd):
    """The four RGE control variates (objectives.py:200-268) for a model given as device source: the reference needs
    autograd Hessian-vector products of the callable; here the model's derivatives at the mean come from central
    differences of its own device gradient and the noise moments from vb_noise_moments.  Checked against the literal
    per-sample restatement of the reference code with the oracle model's ANALYTIC gradient / Hessian."""
    D, N, n_data = 24, 1024, 96
    model, omodel = _problem(vb, D, n_data)
    rng = np.random.RandomState(17)
    theta = np.concatenate([0.2 * rng.randn(D), -1.2 + 0.2 * rng.randn(D)])
    if student:
        approx, ofamily = vb.MFStudentT(D, 7.0, seed=5), ofam.MFStudentT(D, 7.0)
    else:
        approx, ofamily = vb.MFGaussian(D, seed=5), ofam.MFGaussian(D)
    obj = vb.ExclusiveKL(approx, model, N, use_path_deriv=pd, hessian_approx_method=method)
    value, grad = obj(theta)
    noise = ofamily.draw_noise(np.random.RandomState(5), N)
    ov, og = oobj.rge_literal(ofamily, omodel, theta, noise, method, use_path_deriv=pd)
    assert G.rel_err(value, ov) < 1e-12, (value, ov)
    assert G.rel_err(grad, og) < 1e-8, G.rel_err(grad, og)
    plain = vb.ExclusiveKL(type(approx)(*((D, 7.0) if student else (D,)), seed=5), model, N)(theta)[1]
    assert G.rel_err(grad, plain) > 1e-4                     # the control variate really changed the estimate
    assert not obj.supports_device_fit()


def test_noise_moments_against_numpy(vb):
    from viabel_amd import _lib
    eng = _lib.default_engine()
    for n, d in ((1000, 37), (4096, 200), (130, 16)):
        e = np.random.RandomState(n).randn(n, d)
        eng.noise_set_host(5, e)
        cs, gram = eng.noise_moments(5, n, d, want_gram=True)
        assert G.rel_err(cs, e.sum(0)) < 1e-12
        assert G.rel_err(gram, e.T @ e) < 1e-12
        assert np.array_equal(gram, gram.T)
        assert eng.noise_moments(5, n, d)[1] is None
    with pytest.raises(ValueError):
        eng.noise_moments(5, 131, 16)


def test_source_model_hessian_vector_product(vb):
    """ExclusiveKL._hessian_vector_product (objectives.py:166, :275-277) with a source model, against the second
    derivative of the oracle objective on the same noise (analytic model Hessian, chain rule by hand)."""
    D, N, n_data = 12, 512, 64
    model, omodel = _problem(vb, D, n_data)
    rng = np.random.RandomState(18)
    theta = np.concatenate([0.2 * rng.randn(D), -1.0 + 0.1 * rng.randn(D)])
    obj = vb.ExclusiveKL(vb.MFGaussian(D, seed=9), model, N)
    x = rng.randn(2 * D)
    hv = obj._hessian_vector_product(theta, x)
    noise = np.random.RandomState(9).randn(N, D)
    h = 1e-5

    def og(t):
        return oobj.exclusive_kl(ofam.MFGaussian(D), omodel, t, noise)[1]
    u = x / np.linalg.norm(x)
    ref = (8 * (og(theta + h * u) - og(theta - h * u)) - (og(theta + 2 * h * u) - og(theta - 2 * h * u))) / (12 * h)
    ref *= np.linalg.norm(x)
    assert G.rel_err(hv, ref) < 1e-6, G.rel_err(hv, ref)


# ---- grad='auto': the density alone, differentiated on the device (forward-mode dual numbers) ---------------------------
ROBUST_REGRESSION_AUTO_SRC = r"""
// robust regression, density only: params = [n, nu, s, tau | X (n x d) | y (n)]
template <class T>
__device__ T vb_log_density(vb::vec<T> z, int d, const double* p) {
  const int n = (int)p[0];
  const double nu = p[1], s = p[2], tau = p[3];
  const double* X = p + 4;
  const double* y = X + (long long)n * d;
  T f = 0.0;
  for (int j = 0; j < d; ++j) f -= 0.5 * z[j] * z[j] / (tau * tau);
  for (int i = 0; i < n; ++i) {
    T eta = 0.0;
    for (int j = 0; j < d; ++j) eta += X[(long long)i * d + j] * z[j];
    const T r = y[i] - eta;
    f -= 0.5 * (nu + 1.0) * log(1.0 + r * r / (nu * s * s));
  }
  return f;
}
"""


@pytest.mark.parametrize('D,n_data', [(7, 40), (24, 96), (130, 64)])
def test_source_model_auto_gradient_matches_hand_written(vb, D, n_data):
    """SourceModel(grad='auto') (models.py:17-39: the reference differentiates the callable): the gradient the device
    derives from the density alone against the hand-written one and against the numpy oracle's analytic gradient."""
    model, omodel = _problem(vb, D, n_data)
    auto = vb.SourceModel(D, ROBUST_REGRESSION_AUTO_SRC, model.params, grad='auto')
    x = np.random.RandomState(D).randn(50, D)
    assert G.rel_err(auto(x), omodel.logp(x)) < 1e-13
    assert G.rel_err(auto.grad(x), omodel.grad(x)) < 1e-12
    assert G.rel_err(auto.grad(x), model.grad(x)) < 1e-12
    assert auto.check_gradient(x[:4]) < 1e-6


@pytest.mark.parametrize('D,n_data', [(7, 40), (16, 40), (24, 96), (30, 64), (64, 96), (130, 64)])
def test_source_model_auto_gradient_with_dot_helper(vb, D, n_data):
    """`vb::dot(row, z, d)`: the linear predictor as ONE operation of the dual arithmetic -- its derivative with respect
    to the thread's 8 coordinates is the row itself.  Same density, same gradient.  D = 16, 30, 64: 2 / 4 / 8 threads per
    sample, a power of two -- the threads share the product through a lane butterfly (D = 30: ragged last window);
    D = 7, 24, 130: 1 / 3 / 17 threads, every thread forms the whole product."""
    model, omodel = _problem(vb, D, n_data)
    src = ROBUST_REGRESSION_AUTO_SRC.replace('''    T eta = 0.0;
    for (int j = 0; j < d; ++j) eta += X[(long long)i * d + j] * z[j];
    const T r = y[i] - eta;''', '    const T r = y[i] - vb::dot(X + (long long)i * d, z, d);')
    assert 'vb::dot' in src
    auto = vb.SourceModel(D, src, model.params, grad='auto')
    x = np.random.RandomState(D + 1).randn(50, D)
    assert G.rel_err(auto(x), omodel.logp(x)) < 1e-13
    assert G.rel_err(auto.grad(x), omodel.grad(x)) < 1e-12
    assert auto.check_gradient(x[:4]) < 1e-6


@pytest.mark.parametrize('family', ['mf_gaussian', 'fullrank'])
def test_source_model_auto_gradient_under_exclusive_kl(vb, family):
    D, N, n_data = 20, 1024, 64
    model, omodel = _problem(vb, D, n_data)
    auto = vb.SourceModel(D, ROBUST_REGRESSION_AUTO_SRC, model.params, grad='auto')
    rng = np.random.RandomState(2)
    if family == 'mf_gaussian':
        approx, ofamily = vb.MFGaussian(D, seed=4), ofam.MFGaussian(D)
        theta = np.concatenate([0.2 * rng.randn(D), -1.0 + 0.1 * rng.randn(D)])
    else:
        approx, ofamily = vb.FullRankGaussian(D, seed=4), ofam.FullRankGaussian(D)
        L = np.tril(0.05 * rng.randn(D, D), -1) + np.diag(np.exp(-1.0 + 0.1 * rng.randn(D)))
        theta = approx.pack(0.2 * rng.randn(D), L)
    value, grad = vb.ExclusiveKL(approx, auto, N)(theta)
    ov, og = oobj.exclusive_kl(ofamily, omodel, theta, np.random.RandomState(4).randn(N, D))
    assert G.rel_err(value, ov) < 1e-12 and G.rel_err(grad, og) < 1e-11


FUNCTION_ZOO_SRC = r"""
template <class T>
__device__ T vb_log_density(vb::vec<T> z, int d, const double* p) {
  T f = 0.0;
  for (int j = 0; j < d; ++j) {
    const T x = z[j];
    T t = log1p(exp(x)) + tanh(x) * sin(x) - sqrt(1.0 + x * x) + lgamma(2.0 + x * x) - pow(1.0 + x * x, 1.5) / 7.0;
    t += erf(x) + atan(x) * cos(x) + expm1(-x * x) - fabs(x - 0.25) + fmax(x, 0.1) * fmin(x, 2.0);
    t -= p[0] / (2.0 + x * x) + (3.0 - x) / (1.0 + exp(-x)) + pow(2.0 + sin(x), x);
    if (x > 0.5) t += x * x * x; else t -= 2.0 * x;
    f += t * (1.0 + 0.1 * j);
  }
  return f;
}
"""


def test_source_model_auto_gradient_function_zoo(vb):
    """Every overloaded operation of the dual-number header, against numpy / scipy closed forms."""
    from scipy.special import digamma, erf, gammaln
    D = 11
    p0 = 0.7
    model = vb.SourceModel(D, FUNCTION_ZOO_SRC, np.array([p0]), grad='auto')
    x = np.random.RandomState(1).uniform(-1.5, 1.5, size=(64, D))
    w = 1.0 + 0.1 * np.arange(D)

    def f_and_g(x):
        q = 1.0 + x * x
        sg = 1.0 / (1.0 + np.exp(-x))
        t = np.log1p(np.exp(x)) + np.tanh(x) * np.sin(x) - np.sqrt(q) + gammaln(2.0 + x * x) - q ** 1.5 / 7.0
        g = sg + (1 - np.tanh(x) ** 2) * np.sin(x) + np.tanh(x) * np.cos(x) - x / np.sqrt(q) + digamma(2.0 + x * x) * 2 * x \
            - 1.5 * np.sqrt(q) * 2 * x / 7.0
        t = t + erf(x) + np.arctan(x) * np.cos(x) + np.expm1(-x * x) - np.abs(x - 0.25) + np.maximum(x, 0.1) * np.minimum(x, 2.0)
        g = g + 2 / np.sqrt(np.pi) * np.exp(-x * x) + np.cos(x) / q - np.arctan(x) * np.sin(x) - 2 * x * np.exp(-x * x) \
            - np.sign(x - 0.25) + (x >= 0.1) * np.minimum(x, 2.0) + np.maximum(x, 0.1) * (x <= 2.0)
        b = 2.0 + np.sin(x)
        t = t - (p0 / (2.0 + x * x) + (3.0 - x) * sg + b ** x)
        g = g - (-p0 * 2 * x / (2.0 + x * x) ** 2 - sg + (3.0 - x) * sg * (1 - sg) + b ** x * (np.log(b) + x * np.cos(x) / b))
        t = t + np.where(x > 0.5, x ** 3, -2.0 * x)
        g = g + np.where(x > 0.5, 3 * x * x, -2.0)
        return (t * w).sum(1), g * w
    f, g = f_and_g(x)
    assert G.rel_err(model(x), f) < 1e-13
    assert G.rel_err(model.grad(x), g) < 1e-12, G.rel_err(model.grad(x), g)


def test_bbvi_with_auto_gradient_source(vb, capsys):
    """bbvi(dim, log_density=<density-only source>): the reference's primary entry (convenience.py:75) with nothing but
    the log density written down, as with autograd."""
    src = r"""
    template <class T>
    __device__ T vb_log_density(vb::vec<T> z, int d, const double* p) {       // banana: N(z0; 0, 2^2) N(z1; z0^2 / 4, 1)
      const T m = z[1] - 0.25 * z[0] * z[0];
      return -0.125 * z[0] * z[0] - 0.5 * m * m;
    }
    """
    np.random.seed(0)
    res = vb.bbvi(2, log_density=src, n_iters=2000, num_mc_samples=64, adaptive=False, fixed_lr=True, learning_rate=0.05)
    assert res['objective'].model.grad_mode == 'auto'
    mu = res['opt_param'][:2]
    assert abs(mu[0]) < 0.4 and 0.0 < mu[1] < 2.0
    with pytest.raises(ValueError):
        vb.SourceModel(2, src, grad='finite-differences')
