"""GPU parity of AlphaDivergence and DISInclusiveKL against the reference-derived golden vectors
and the oracle, through the product's Python classes (which call the C ABI).

Tolerances: value 1e-12 relative, gradient 1e-11 relative to max|grad|; 2e-7 / 2e-6 against the
reference's finite-difference gradients (as in tests/test_oracle_golden.py).
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


def product_family(vb, fx, seed=1):
    kind, D = str(fx['family_kind']), int(fx['dim'])
    if kind == 'mf_gaussian':
        return vb.MFGaussian(D, seed=seed)
    if kind == 'mf_student_t':
        return vb.MFStudentT(D, float(fx['df']), seed=seed)
    if kind == 'multivariate_t':
        return vb.MultivariateT(D, float(fx['df']), seed=seed)
    if kind == 'lr_gaussian':
        return vb.LRGaussian(D, seed=seed, k=int(fx['rank']))
    raise ValueError(kind)


def product_model(vb, fx):
    if str(fx['model_kind']) == 'gauss_diag':
        return vb.GaussianModel(fx['model_mean'], fx['model_stdev'])
    return vb.FunnelModel(int(fx['dim']), int(fx['model_scale_index']),
                          float(fx['model_log_sigma_stdev']))


@pytest.mark.parametrize('path', G.fixtures('alpha_'), ids=lambda p: p.split('/')[-1][:-4])
def test_alpha_golden(vb, path):
    fx = G.load(path)
    obj = vb.AlphaDivergence(product_family(vb, fx), product_model(vb, fx), int(fx['n']), float(fx['alpha']))
    np.random.seed(int(fx['np_seed']))          # the objective draws its noise seed from the global RNG
    value, grad = obj(fx['theta'])
    assert G.rel_err(value, fx['value']) < 1e-12
    assert G.rel_err(grad, fx['grad']) < 1e-11
    assert G.rel_err(grad, fx['grad_fd']) < 2e-7


@pytest.mark.parametrize('path', G.fixtures('dis_'), ids=lambda p: p.split('/')[-1][:-4])
def test_dis_golden(vb, path):
    fx = G.load(path)
    D = int(fx['dim'])
    obj = vb.DISInclusiveKL(product_family(vb, fx, int(fx['seed'])), product_model(vb, fx), int(fx['n']),
                            ess_target=int(fx['ess_target']), temper_prior=vb.MFGaussian(D),
                            temper_prior_params=fx['prior_params'], use_resampling=bool(fx['use_resampling']))
    np.random.seed(int(fx['np_seed']))
    value, grad = obj(fx['theta'])
    assert G.rel_err(obj._eps, fx['eps']) < 1e-12
    assert G.rel_err(obj._state_log_p_unnormalized, fx['log_p']) < 1e-12
    assert G.rel_err(obj._state_log_q, fx['log_q']) < 1e-11
    assert G.rel_err(obj._state_w_clipped, fx['w_clipped']) < 1e-10
    assert G.rel_err(value, fx['value']) < 1e-11
    assert G.rel_err(grad, fx['grad']) < 1e-11
    assert G.rel_err(grad, fx['grad_fd']) < 2e-6


@pytest.mark.parametrize('family', ['gauss', 't'])
@pytest.mark.parametrize('use_resampling', [True, False])
def test_dis_against_oracle_multi_step(vb, family, use_resampling):
    """Several calls with a moving theta and num_resampling_batches = 2: refreshes on even steps,
    reuses the state samples (with the NEW theta in log q) on odd steps."""
    D, N = 130, 2048
    rng = np.random.RandomState(5)
    if family == 'gauss':
        approx, ofamily = vb.MFGaussian(D, seed=3), ofam.MFGaussian(D)
    else:
        approx, ofamily = vb.MFStudentT(D, 7, seed=3), ofam.MFStudentT(D, 7)
    mean, sd = 0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D))
    model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    prior = np.concatenate([np.zeros(D), 0.5 * np.ones(D)])
    kw = dict(use_resampling=use_resampling, num_resampling_batches=2)
    obj = vb.DISInclusiveKL(approx, model, N, ess_target=400, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, **kw)
    ref = oobj.DISInclusiveKL(ofamily, omodel, N, 400, ofam.MFGaussian(D), prior, **kw)
    rs = np.random.RandomState(3)
    theta = np.concatenate([0.1 * rng.randn(D), -0.2 + 0.1 * rng.randn(D)])
    np.random.seed(11)
    for step in range(4):
        state = np.random.get_state()
        value, grad = obj(theta)
        np.random.set_state(state)
        noise = ofamily.draw_noise(rs, N) if ref.needs_refresh() else None
        if use_resampling:
            if ref.needs_refresh():
                ref.refresh(theta, noise)      # what __call__ does first (objectives.py:392-401)
            idx = np.random.choice(N, size=ref._resampling_batch_size, p=ref._state_w_normalized)
            ref._objective_step += 1
            xs = ref._state_samples[idx]
            scale = ref._state_w_sum / N
            ov = np.mean(-ofamily.log_density(theta, xs)) * scale
            og = -ofamily.log_density_grad_weighted(theta, xs, np.ones(len(idx))) / len(idx) * scale
        else:
            ov, og = ref(theta, noise=noise)
        assert G.rel_err(obj._eps, ref._eps) < 1e-11
        assert G.rel_err(value, ov) < 1e-11, (step, value, ov)
        assert G.rel_err(grad, og) < 1e-10, (step, G.rel_err(grad, og))
        theta = theta - 0.01 * grad / (1 + np.abs(grad))


@pytest.mark.parametrize('family', ['multivariate_t', 'fullrank_gaussian'])
@pytest.mark.parametrize('use_resampling', [True, False])
def test_dis_multivariate_t_against_oracle(vb, use_resampling, family):
    """MultivariateT + DIS (BASELINE configs[3] family) at a size where the MFMA GEMMs tile: D=200, N=3000,
    three calls with a moving theta and num_resampling_batches = 2; and the dense Gaussian, which runs through the
    same kernels as their df -> infinity member."""
    D, N, df = 200, 3000, 40
    rng = np.random.RandomState(9)
    if family == 'multivariate_t':
        approx, ofamily = vb.MultivariateT(D, df, seed=6), ofam.MultivariateT(D, df)
    else:
        approx, ofamily = vb.FullRankGaussian(D, seed=6), ofam.FullRankGaussian(D)
    mean, sd = 0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D))
    model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    prior = np.concatenate([np.zeros(D), 0.5 * np.ones(D)])
    kw = dict(use_resampling=use_resampling, num_resampling_batches=2)
    obj = vb.DISInclusiveKL(approx, model, N, ess_target=600, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, **kw)
    ref = oobj.DISInclusiveKL(ofamily, omodel, N, 600, ofam.MFGaussian(D), prior, **kw)
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
        assert G.rel_err(obj._state_log_q, ref._state_log_q) < 1e-11
        assert G.rel_err(value, ov) < 1e-10, (step, value, ov)
        assert G.rel_err(grad, og) < 1e-9, (step, G.rel_err(grad, og))
        theta = theta - 0.002 * grad / (1 + np.abs(grad))


def test_dis_all_weights_zero_raises(vb):
    D, N = 4, 16
    # (z - 1e200)^2 overflows: log p = -inf for every sample, so every log weight is -inf
    obj = vb.DISInclusiveKL(vb.MFGaussian(D), vb.GaussianModel(1e200 * np.ones(D), 1e-100 * np.ones(D)), N,
                            ess_target=8, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=np.zeros(2 * D))
    with pytest.raises(ValueError) as info:
        obj(np.concatenate([np.zeros(D), np.zeros(D)]))
    assert str(info.value).startswith('All weights zero!')


@pytest.mark.parametrize('D,N', [(1024, 4096), (77, 333), (300, 1000)])
@pytest.mark.parametrize('family', ['gauss', 't'])
def test_alpha_against_oracle(vb, D, N, family):
    rng = np.random.RandomState(D + N)
    theta = np.concatenate([0.3 * rng.randn(D), -1.0 + 0.2 * rng.randn(D)])
    for model, omodel in ((vb.FunnelModel(D, 5), omod.Funnel(D, 5)),
                          (vb.GaussianModel(np.ones(D), 2 * np.ones(D)), omod.GaussDiag(np.ones(D), 2 * np.ones(D)))):
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
            assert G.rel_err(value, ov) < 1e-12
            assert G.rel_err(grad, og) < 1e-11


@pytest.mark.parametrize('D,N', [(3, 9), (70, 333), (200, 1000)])
def test_alpha_fullrank_against_oracle(vb, D, N):
    """AlphaDivergence for the dense Gaussian: per-row f on the samples, weights as for the mean-field families,
    the entropy-form pipeline with the rows of G weighted; every target the pipeline knows."""
    rng = np.random.RandomState(3 * D + N)
    ofr = ofam.FullRankGaussian(D)
    L = np.tril(0.05 * rng.randn(D, D), -1) + np.diag(np.exp(-1.0 + 0.2 * rng.randn(D)))
    theta = ofr.pack(0.3 * rng.randn(D), L)
    A = rng.randn(D, D)
    S = A @ A.T / D + np.eye(D)
    m2 = rng.randn(D)
    X = rng.randn(2 * D + 5, D) / np.sqrt(D)
    y = (rng.rand(2 * D + 5) < 0.5).astype(float)
    models = [(vb.GaussianModel(np.ones(D), 2 * np.ones(D)), omod.GaussDiag(np.ones(D), 2 * np.ones(D))),
              (vb.CorrelatedGaussianModel(m2, covariance=S), omod.GaussFull(m2, np.linalg.inv(S))),
              (vb.LogisticRegressionModel(X, y, 3.0), omod.Logistic(X, y, 3.0))]
    if D >= 2:
        models.append((vb.FunnelModel(D, D // 2), omod.Funnel(D, D // 2)))
    for model, omodel in models:
        for alpha in (2.0, 0.5):
            np.random.seed(11)
            value, grad = vb.AlphaDivergence(vb.FullRankGaussian(D), model, N, alpha)(theta)
            np.random.seed(11)
            noise = np.random.RandomState(np.random.randint(2 ** 32)).randn(N, D)
            ov, og = oobj.alpha_divergence(ofr, omodel, theta, noise, alpha)
            assert G.rel_err(value, ov) < 1e-12, (type(omodel).__name__, alpha, value, ov)
            assert G.rel_err(grad, og) < 1e-11, (type(omodel).__name__, alpha, G.rel_err(grad, og))


@pytest.mark.parametrize('D,N,df', [(3, 9, 100), (70, 333, 7), (200, 1000, 30)])
def test_alpha_multivariate_t_against_oracle(vb, D, N, df):
    """AlphaDivergence for the multivariate t (beyond the golden sizes): weights from the t density of the samples,
    weighted sums through the t family's pipeline, chain rule through the symmetric root."""
    rng = np.random.RandomState(5 * D + N)
    omvt = ofam.MultivariateT(D, df)
    L = np.tril(0.05 * rng.randn(D, D), -1) + np.diag(np.exp(-1.0 + 0.2 * rng.randn(D)))
    theta = np.concatenate([0.3 * rng.randn(D), ofam.chol_to_free(L)])
    A = rng.randn(D, D)
    S = A @ A.T / D + np.eye(D)
    m2 = rng.randn(D)
    models = [(vb.GaussianModel(np.ones(D), 2 * np.ones(D)), omod.GaussDiag(np.ones(D), 2 * np.ones(D))),
              (vb.CorrelatedGaussianModel(m2, covariance=S), omod.GaussFull(m2, np.linalg.inv(S))),
              (vb.FunnelModel(D, D // 2), omod.Funnel(D, D // 2))]
    for model, omodel in models:
        for alpha in (2.0, 0.5):
            np.random.seed(13)
            value, grad = vb.AlphaDivergence(vb.MultivariateT(D, df), model, N, alpha)(theta)
            np.random.seed(13)
            noise = omvt.draw_noise(np.random.RandomState(np.random.randint(2 ** 32)), N)
            ov, og = oobj.alpha_divergence(omvt, omodel, theta, noise, alpha)
            assert G.rel_err(value, ov) < 1e-12, (type(omodel).__name__, alpha, value, ov)
            assert G.rel_err(grad, og) < 1e-10, (type(omodel).__name__, alpha, G.rel_err(grad, og))


@pytest.mark.parametrize('D,n_data,N', [(7, 33, 50), (50, 200, 300), (200, 500, 1000)])
def test_logistic_regression_target(vb, D, n_data, N):
    """New target against the oracle's logistic model: plain and path-derivative ELBO,
    Gaussian and Student-t mean-field families."""
    rng = np.random.RandomState(D)
    X = rng.randn(n_data, D) / np.sqrt(D)
    beta = rng.randn(D)
    y = (rng.rand(n_data) < 1 / (1 + np.exp(-X @ beta))).astype(float)
    model, omodel = vb.LogisticRegressionModel(X, y, 10.0), omod.Logistic(X, y, 10.0)
    theta = np.concatenate([0.3 * rng.randn(D), -1.0 + 0.2 * rng.randn(D)])
    for approx, ofamily in ((vb.MFGaussian(D, seed=2), ofam.MFGaussian(D)),
                            (vb.MFStudentT(D, 9, seed=2), ofam.MFStudentT(D, 9))):
        for pd in (False, True):
            value, grad = vb.ExclusiveKL(approx, model, N, use_path_deriv=pd)(theta)
            rs = np.random.RandomState(2)
            if pd:
                ofamily.draw_noise(rs, N)
            noise = ofamily.draw_noise(rs, N)
            ov, og = oobj.exclusive_kl(ofamily, omodel, theta, noise, pd)
            assert G.rel_err(value, ov) < 1e-12, (pd, value, ov)
            assert G.rel_err(grad, og) < 1e-11, (pd, G.rel_err(grad, og))
    with pytest.raises(NotImplementedError):
        vb.ExclusiveKL(vb.MFGaussian(D), model, N, hessian_approx_method='full')(theta)


@pytest.mark.parametrize('kind', ['poisson', 'linear'])
@pytest.mark.parametrize('D,n_data,N', [(7, 33, 50), (60, 250, 400)])
def test_glm_regression_targets(vb, kind, D, n_data, N):
    """Poisson (log link) and Gaussian (identity link) members of the regression target: same two-GEMM pipeline as
    the logistic model, different likelihood in the GEMM epilogue; value, gradient and Model.__call__ against the
    numpy oracle."""
    rng = np.random.RandomState(3 * D + len(kind))
    X = rng.randn(n_data, D) / np.sqrt(D)
    beta = 0.5 * rng.randn(D)
    if kind == 'poisson':
        y = rng.poisson(np.exp(X @ beta)).astype(float)
        model, omodel = vb.PoissonRegressionModel(X, y, 5.0), omod.Poisson(X, y, 5.0)
    else:
        y = X @ beta + 0.7 * rng.randn(n_data)
        model, omodel = vb.LinearRegressionModel(X, y, 5.0, noise_sd=0.7), omod.LinearRegression(X, y, 5.0, 0.7)
    theta = np.concatenate([0.3 * rng.randn(D), -1.5 + 0.2 * rng.randn(D)])
    for pd in (False, True):
        approx, ofamily = vb.MFGaussian(D, seed=2), ofam.MFGaussian(D)
        value, grad = vb.ExclusiveKL(approx, model, N, use_path_deriv=pd)(theta)
        noise = ofamily.draw_noise(np.random.RandomState(2), N)
        ov, og = oobj.exclusive_kl(ofamily, omodel, theta, noise, pd)
        assert G.rel_err(value, ov) < 1e-12, (pd, value, ov)
        assert G.rel_err(grad, og) < 1e-11, (pd, G.rel_err(grad, og))
    x = 0.3 * rng.randn(123, D)
    fo = omodel.logp(x)
    np.testing.assert_allclose(model(x), fo, rtol=0, atol=1e-12 * np.max(np.abs(fo)))
    with pytest.raises(ValueError):
        vb.LinearRegressionModel(X, y, 5.0, noise_sd=0.0)
    if kind == 'poisson':
        with pytest.raises(ValueError):
            vb.PoissonRegressionModel(X, -np.ones(n_data))


@pytest.mark.parametrize('df', [2.5, 9.0, 100.0])
def test_device_chisquare_draws(vb, df):
    """The chi-square draws of MultivariateT in throughput mode (vb_chisq_generate: Philox + Marsaglia-Tsang):
    right distribution, pure function of (seed, stream, global row) -- so sharding does not change them."""
    from scipy import stats
    from viabel_amd import _lib
    eng = _lib.default_engine()
    n = 200000
    eng.chisq_generate(df, n, seed=11, stream=3)
    x = eng.chisq_get_host(n)
    assert np.all(x > 0) and np.all(np.isfinite(x))
    assert stats.kstest(x, 'chi2', args=(df,)).pvalue > 1e-3
    assert abs(x.mean() - df) < 6 * np.sqrt(2 * df / n)
    eng.chisq_generate(df, 1000, seed=11, stream=3, row_offset=150000)
    assert np.array_equal(eng.chisq_get_host(1000), x[150000:151000])
    eng.chisq_generate(df, 1000, seed=11, stream=4)
    assert not np.array_equal(eng.chisq_get_host(1000), x[:1000])
    with pytest.raises(ValueError):
        eng.chisq_generate(2.0, 10, seed=1)


@pytest.mark.parametrize('use_resampling', [True, False])
def test_dis_multivariate_t_philox_mode_against_oracle(vb, use_resampling):
    """rng='philox': the normals AND the chi-square draws of the state refresh are generated on the device; read
    both back and the oracle must reproduce the step on them.  In this mode the state samples are
    x = mu + (z L') / s with the Cholesky factor (same distribution as the reference's symmetric root, no D^3 root
    on the hot path), so the oracle family used here samples the same way; everything downstream of the samples
    (log q, log p, tempering, weights, gradient) is the reference's arithmetic."""
    from viabel_amd import _lib
    from viabel_amd.objectives import _DIS_SLOT

    class CholeskySampledT(ofam.MultivariateT):
        def sample_from_noise(self, theta, noise):
            chi, z = noise
            mu, S = self.split(theta)
            return mu + (z @ np.linalg.cholesky(S).T) / np.sqrt(chi / self.df)[:, None]

    D, N, df = 130, 2048, 12.0
    rng = np.random.RandomState(19)
    approx, ofamily = vb.MultivariateT(D, df, seed=8, rng='philox'), CholeskySampledT(D, df)
    mean, sd = 0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D))
    model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    prior = np.concatenate([np.zeros(D), 0.3 * np.ones(D)])
    obj = vb.DISInclusiveKL(approx, model, N, ess_target=300, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, use_resampling=use_resampling)
    ref = oobj.DISInclusiveKL(ofamily, omodel, N, 300, ofam.MFGaussian(D), prior, use_resampling=use_resampling)
    A = rng.randn(D, D)
    theta = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(A @ A.T / D + 0.7 * np.eye(D))])
    np.random.seed(4)
    state = np.random.get_state()
    value, grad = obj(theta)
    np.random.set_state(state)
    eng = _lib.default_engine()
    noise = (eng.chisq_get_host(N), eng.noise_get_host(_DIS_SLOT, N, D))
    if use_resampling:
        ref.refresh(theta, noise)
        counts = eng.dis_weights_get(N, resampled=True)       # the device's multinomial draw (objectives.py:408)
        M = ref._resampling_batch_size
        assert counts.sum() == M
        scale = ref._state_w_sum / N / M
        ov = -np.sum(counts * ofamily.log_density(theta, ref._state_samples)) * scale
        og = -ofamily.log_density_grad_weighted(theta, ref._state_samples, counts) * scale
    else:
        ov, og = ref(theta, noise=noise)
    assert G.rel_err(obj._eps, ref._eps) < 1e-10
    assert G.rel_err(value, ov) < 1e-10, (value, ov)
    assert G.rel_err(grad, og) < 1e-9, G.rel_err(grad, og)
    # the per-sample logs stay on the device in this mode and are fetched on first access
    assert G.rel_err(obj._state_log_q, ref._state_log_q) < 1e-11
    assert G.rel_err(obj._state_log_p_unnormalized, ref._state_log_p) < 1e-11


@pytest.mark.parametrize('pd', [False, True], ids=['entropy', 'path_deriv'])
@pytest.mark.parametrize('family', ['mf_gaussian', 'mf_student_t', 'fullrank'])
def test_exclusive_kl_hessian_vector_product(vb, family, pd):
    """ExclusiveKL._hessian_vector_product (objectives.py:166, :275-277) against torch.autograd's exact
    Hessian-vector product of the same objective on the same noise (fp64, CPU).  With use_path_deriv the reference's
    objective is mean(f(z) - log q(z; stopped theta)) (objectives.py:156-159): the stopped copy is a constant of both
    differentiations."""
    import torch
    D, N = 24, 300
    rng = np.random.RandomState(2)
    mean, sd = rng.randn(D), np.exp(0.3 * rng.randn(D))
    model = vb.GaussianModel(mean, sd)
    if family == 'fullrank':
        approx = vb.FullRankGaussian(D, seed=5)
        L = np.tril(0.2 * rng.randn(D, D), -1) + np.diag(np.exp(-0.5 + 0.2 * rng.randn(D)))
        theta = approx.pack(0.3 * rng.randn(D), L)
        noise = np.random.RandomState(5).randn(N, D)
    elif family == 'mf_student_t':
        approx = vb.MFStudentT(D, 7.0, seed=5)
        theta = np.concatenate([0.3 * rng.randn(D), -0.5 + 0.2 * rng.randn(D)])
        noise = np.random.RandomState(5).standard_t(7.0, (N, D))
    else:
        approx = vb.MFGaussian(D, seed=5)
        theta = np.concatenate([0.3 * rng.randn(D), -0.5 + 0.2 * rng.randn(D)])
        noise = np.random.RandomState(5).randn(N, D)
    x = rng.randn(theta.size)
    obj = vb.ExclusiveKL(approx, model, N, use_path_deriv=pd)
    hv = obj._hessian_vector_product(theta, x)
    ts = torch.from_numpy(theta.copy())           # the stopped parameter

    E = torch.from_numpy(noise)
    tm, tiv = torch.from_numpy(mean), torch.from_numpy(1.0 / sd ** 2)
    tril = np.tril_indices(D)

    def objective(t):
        mu = t[:D]
        if family == 'fullrank':
            Lf = torch.zeros(D, D, dtype=torch.float64)
            Lf[tril[0], tril[1]] = t[D:]
            Lm = torch.tril(Lf, -1) + torch.diag(torch.exp(torch.diagonal(Lf)))
            z = mu + E @ Lm.T
            logdet = torch.sum(torch.diagonal(Lf))
        else:
            z = mu + torch.exp(t[D:]) * E
            logdet = torch.sum(t[D:])
        f = -0.5 * torch.sum((z - tm) ** 2 * tiv, 1)
        if pd:                                    # log q(z; stopped theta) up to constants
            if family == 'fullrank':
                Ls = torch.zeros(D, D, dtype=torch.float64)
                Ls[tril[0], tril[1]] = ts[D:]
                Ls = torch.tril(Ls, -1) + torch.diag(torch.exp(torch.diagonal(Ls)))
                w = torch.linalg.solve_triangular(Ls, (z - ts[:D]).T, upper=False).T
                logq = -0.5 * torch.sum(w * w, 1)
            else:
                u = (z - ts[:D]) / torch.exp(ts[D:])
                logq = (torch.sum(-0.5 * (7.0 + 1.0) * torch.log1p(u * u / 7.0), 1) if family == 'mf_student_t'
                        else -0.5 * torch.sum(u * u, 1))
            return -torch.mean(f - logq)
        return -(torch.mean(f) + logdet)          # the entropy's theta-independent terms do not matter here

    _, ref = torch.autograd.functional.hvp(objective, torch.from_numpy(theta), torch.from_numpy(x))
    ref = ref.numpy()
    assert np.max(np.abs(hv - ref)) < 1e-7 * np.max(np.abs(ref)), np.max(np.abs(hv - ref)) / np.max(np.abs(ref))
    assert np.array_equal(obj._hessian_vector_product(theta, np.zeros_like(x)), np.zeros_like(x))
    with pytest.raises(AttributeError):
        vb.ExclusiveKL(vb.MFGaussian(D), model, N, hessian_approx_method='full')._hessian_vector_product(
            theta[:2 * D], x[:2 * D])
    with pytest.raises(NotImplementedError):
        t = vb.MultivariateT(D, 5.0)
        vb.ExclusiveKL(t, model, N)._hessian_vector_product(np.zeros(t.var_param_dim), np.ones(t.var_param_dim))


@pytest.mark.parametrize('family', ['mf_gaussian', 'multivariate_t'])
def test_dis_psis_smoothed_weights(vb, family):
    """psis_smooth=True (BASELINE configs[3]: 'DISInclusiveKL with PSIS reweighting'): the tempered weights go
    through the reference's psislw (viabel/_psis.py:113-209, pinned by tests/golden/psis.npz) before clipping and
    keep their sum; everything downstream uses the smoothed weights."""
    from oracle import psis as opsis
    D, N = 40, 4096
    rng = np.random.RandomState(8)
    mean, sd = 0.4 * rng.randn(D), np.exp(0.1 * rng.randn(D))
    model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    prior = np.concatenate([np.zeros(D), 0.3 * np.ones(D)])
    if family == 'mf_gaussian':
        approx, ofamily = vb.MFGaussian(D, seed=3), ofam.MFGaussian(D)
        theta = np.concatenate([0.1 * rng.randn(D), 0.2 + 0.1 * rng.randn(D)])
    else:
        approx, ofamily = vb.MultivariateT(D, 9.0, seed=3), ofam.MultivariateT(D, 9.0)
        A = rng.randn(D, D)
        theta = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(A @ A.T / D + np.eye(D))])
    kw = dict(ess_target=500, use_resampling=False)
    obj = vb.DISInclusiveKL(approx, model, N, temper_prior=vb.MFGaussian(D), temper_prior_params=prior,
                            psis_smooth=True, **kw)
    raw = vb.DISInclusiveKL(type(approx)(*((D,) if family == 'mf_gaussian' else (D, 9.0)), seed=3), model, N,
                            temper_prior=vb.MFGaussian(D), temper_prior_params=prior, **kw)
    value, grad = obj(theta)
    raw(theta)
    ref = oobj.DISInclusiveKL(ofamily, omodel, N, 500, ofam.MFGaussian(D), prior, use_resampling=False)
    ref.refresh(theta, ofamily.draw_noise(np.random.RandomState(3), N))
    w = ref._state_w_clipped                       # the reference's tempered weights (no clipping happens at thr = 10)
    smoothed, khat = opsis.psis_smooth(np.log(w))
    w_s = np.sum(w) * np.exp(smoothed)
    assert G.rel_err(raw._state_w_clipped, w) < 1e-10
    assert G.rel_err(obj._state_w_clipped, w_s) < 1e-9
    assert abs(obj._khat - khat) < 1e-8
    assert abs(np.sum(obj._state_w_clipped) - np.sum(w)) < 1e-9 * np.sum(w)
    assert not np.allclose(obj._state_w_clipped, w)          # the tail really was replaced
    lq = ofamily.log_density(theta, ref._state_samples)
    ov = -np.inner(w_s, lq) / N
    og = -ofamily.log_density_grad_weighted(theta, ref._state_samples, w_s) / N
    assert G.rel_err(value, ov) < 1e-9
    assert G.rel_err(grad, og) < 1e-8


@pytest.mark.parametrize('D,k,N', [(64, 4, 1000), (1024, 8, 4096), (130, 16, 777), (96, 17, 500), (256, 32, 2048),
                                   (130, 64, 777)])
@pytest.mark.parametrize('rng_kind', ['numpy', 'philox'])
def test_lowrank_alpha_against_oracle(vb, D, k, N, rng_kind):
    """LRGaussian + AlphaDivergence (the reference's objective is family-generic, objectives.py:443-463 over
    approximations.py:636-707) at sizes where the skinny GEMMs tile and split."""
    from viabel_amd import _lib
    from viabel_amd.objectives import _NOISE_SLOT, _LR_SLOT
    rng = np.random.RandomState(D + k)
    theta = np.concatenate([0.2 * rng.randn(D), -0.7 + 0.2 * rng.randn(D), 0.3 * rng.randn(D * k) / np.sqrt(k)])
    ofamily = ofam.LRGaussian(D, k)
    for model, omodel in ((vb.FunnelModel(D, 3), omod.Funnel(D, 3)),
                          (vb.GaussianModel(0.3 * np.ones(D), 1.5 * np.ones(D)), omod.GaussDiag(0.3 * np.ones(D), 1.5 * np.ones(D)))):
        for alpha in (2.0, 0.5):
            approx = vb.LRGaussian(D, seed=2, k=k, rng=rng_kind)
            obj = vb.AlphaDivergence(approx, model, N, alpha)
            np.random.seed(77)
            value, grad = obj(theta)
            np.random.seed(77)
            seed = np.random.randint(2 ** 32)
            if rng_kind == 'numpy':
                noise = ofamily.draw_noise(np.random.RandomState(seed), N)
            else:
                eng = _lib.default_engine()
                noise = (eng.noise_get_host(_LR_SLOT, N, k), eng.noise_get_host(_NOISE_SLOT, N, D))
            ov, og = oobj.alpha_divergence(ofamily, omodel, theta, noise, alpha)
            assert G.rel_err(value, ov) < 1e-11, (value, ov)
            assert G.rel_err(grad, og) < 1e-9, G.rel_err(grad, og)


@pytest.mark.parametrize('D,k,N', [(64, 4, 1024), (256, 8, 4096), (130, 16, 800), (96, 17, 512), (256, 32, 2048),
                                   (130, 64, 800)])
@pytest.mark.parametrize('use_resampling', [True, False])
def test_lowrank_dis_against_oracle_multi_step(vb, D, k, N, use_resampling):
    """LRGaussian + DISInclusiveKL: refresh on even steps, state samples reused with the NEW theta on odd steps."""
    rng = np.random.RandomState(3 * D + k)
    approx, ofamily = vb.LRGaussian(D, seed=5, k=k), ofam.LRGaussian(D, k)
    mean, sd = 0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D))
    model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    prior = np.concatenate([np.zeros(D), 0.2 * np.ones(D)])
    kw = dict(use_resampling=use_resampling, num_resampling_batches=2)
    obj = vb.DISInclusiveKL(approx, model, N, ess_target=N // 5, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, **kw)
    ref = oobj.DISInclusiveKL(ofamily, omodel, N, N // 5, ofam.MFGaussian(D), prior, **kw)
    theta = np.concatenate([0.1 * rng.randn(D), 0.1 + 0.1 * rng.randn(D), 0.3 * rng.randn(D * k) / np.sqrt(k)])
    rs = np.random.RandomState(5)
    np.random.seed(13)
    for step in range(4):
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
        assert G.rel_err(obj._state_log_q, ref._state_log_q) < 1e-11
        assert G.rel_err(value, ov) < 1e-10, (step, value, ov)
        assert G.rel_err(grad, og) < 1e-9, (step, G.rel_err(grad, og))
        theta = theta - 0.005 * grad / (1 + np.abs(grad))


@pytest.mark.parametrize('kind', ['mf', 'mvt_numpy', 'mvt_philox', 'fr_philox', 'lr'])
def test_interleaved_dis_objectives_keep_their_own_states(vb, kind):
    """Two DISInclusiveKL objectives with num_resampling_batches > 1 taking turns on ONE engine.  The reference keeps the
    state samples per object (objectives.py:391-403); here they live in the engine, one set per family kind -- until round 6
    the second objective's refresh invalidated the first one's kept weights and the call raised.  Now the state
    leaves the context with its owner (vb_dis_state_park: buffers detached) and comes back for the owner's next
    kept-weights step: each objective's values and gradients are exactly what it computes running alone -- with a third
    objective (no kept weights) refreshing in between as well."""
    D, N = (16, 600) if kind in ('mf', 'lr') else (48, 4200)
    rng = np.random.RandomState(3)
    model = vb.GaussianModel(0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D)))
    prior = np.concatenate([np.zeros(D), 0.3 * np.ones(D)])
    A = rng.randn(D, D)
    th_ch = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(0.7 * (A @ A.T / D + np.eye(D)))])
    th_mf = np.concatenate([0.1 * rng.randn(D), -0.5 + 0.1 * rng.randn(D)])

    def make(seed, batches):
        kw = dict(ess_target=N // 6, temper_prior=vb.MFGaussian(D), temper_prior_params=prior, use_resampling=True,
                  num_resampling_batches=batches)
        if kind == 'mf':
            return vb.DISInclusiveKL(vb.MFGaussian(D, seed=seed), model, N, **kw), th_mf
        if kind == 'lr':
            fam = vb.LRGaussian(D, seed=seed, k=3)
            return vb.DISInclusiveKL(fam, model, N, **kw), fam.pack(np.zeros(D), -0.5 * np.ones(D), 0.1 * np.ones((D, 3)))
        if kind == 'fr_philox':
            return vb.DISInclusiveKL(vb.FullRankGaussian(D, seed=seed, rng='philox'), model, N, **kw), th_ch
        return vb.DISInclusiveKL(vb.MultivariateT(D, 9.0, seed=seed, rng='numpy' if kind == 'mvt_numpy' else 'philox'),
                                 model, N, **kw), th_ch

    def alone(seed, batches, calls, np_seed):
        obj, th = make(seed, batches)
        np.random.seed(np_seed)
        out = []
        for _ in range(calls):
            v, g = obj(th)
            out.append((v, g.copy()))
            th = th - 0.01 * g / (1.0 + np.abs(g))
        return out

    want_a, want_b = alone(1, 3, 7, 100), alone(2, 2, 7, 200)
    a, tha = make(1, 3)
    b, thb = make(2, 2)
    c, thc = make(5, 1)
    sa, sb = np.random.RandomState(100).get_state(), np.random.RandomState(200).get_state()
    got_a, got_b = [], []
    for it in range(7):
        np.random.set_state(sa)                 # (each objective sees the global generator it would see alone)
        v, g = a(tha)
        sa = np.random.get_state()
        got_a.append((v, g.copy()))
        tha = tha - 0.01 * g / (1.0 + np.abs(g))
        if it % 2:
            c(thc)                              # a refresh without kept weights in between
        np.random.set_state(sb)
        v, g = b(thb)
        sb = np.random.get_state()
        got_b.append((v, g.copy()))
        thb = thb - 0.01 * g / (1.0 + np.abs(g))
    for want, got in ((want_a, got_a), (want_b, got_b)):
        for (v0, g0), (v1, g1) in zip(want, got):
            assert v0 == v1
            np.testing.assert_array_equal(g0, g1)


def test_more_than_65535_samples(vb):
    """Row-indexed kernels put the sample index on gridDim.x: gridDim.y stops at 65 535 (the RNG launcher broke on that
    limit in round 1).  N = 70 000 through the row-scaling (alpha, dense family), low-rank sampling, logistic sampling
    and row-centring kernels, against the oracle."""
    N, D, k = 70000, 6, 2
    rng = np.random.RandomState(1)
    mean, sd = 0.3 * rng.randn(D), np.exp(0.1 * rng.randn(D))
    model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    # AlphaDivergence over the dense Gaussian (fr_rowscale_kernel)
    fr, ofr = vb.FullRankGaussian(D, seed=3), ofam.FullRankGaussian(D)
    L = np.tril(0.1 * rng.randn(D, D), -1) + np.diag(np.exp(-0.3 + 0.1 * rng.randn(D)))
    theta = ofr.pack(0.1 * rng.randn(D), L)
    np.random.seed(5)
    v, g = vb.AlphaDivergence(fr, model, N, 2.0)(theta)
    np.random.seed(5)
    noise = np.random.RandomState(np.random.randint(2 ** 32)).randn(N, D)
    ov, og = oobj.alpha_divergence(ofr, omodel, theta, noise, 2.0)
    assert G.rel_err(v, ov) < 1e-11 and G.rel_err(g, og) < 1e-9
    # AlphaDivergence over the low-rank Gaussian (lro_sample_kernel)
    olr = ofam.LRGaussian(D, k)
    th = np.concatenate([0.1 * rng.randn(D), -0.3 + 0.1 * rng.randn(D), 0.2 * rng.randn(D * k)])
    np.random.seed(6)
    v, g = vb.AlphaDivergence(vb.LRGaussian(D, seed=3, k=k), model, N, 0.5)(th)
    np.random.seed(6)
    ov, og = oobj.alpha_divergence(olr, omodel, th, olr.draw_noise(np.random.RandomState(np.random.randint(2 ** 32)), N), 0.5)
    assert G.rel_err(v, ov) < 1e-11 and G.rel_err(g, og) < 1e-9
    # logistic target under the mean-field family (lg_sample_kernel) and the correlated-Gaussian row kernel
    X = rng.randn(50, D)
    y = (rng.rand(50) < 0.5).astype(float)
    thm = np.concatenate([0.1 * rng.randn(D), -0.5 + 0.1 * rng.randn(D)])
    v, g = vb.ExclusiveKL(vb.MFGaussian(D, seed=4), vb.LogisticRegressionModel(X, y, 3.0), N)(thm)
    ov, og = oobj.exclusive_kl(ofam.MFGaussian(D), omod.Logistic(X, y, 3.0), thm, np.random.RandomState(4).randn(N, D))
    assert G.rel_err(v, ov) < 1e-12 and G.rel_err(g, og) < 1e-10
    A = rng.randn(D, D)
    S = A @ A.T / D + np.eye(D)
    cg = vb.CorrelatedGaussianModel(mean, covariance=S)
    x = rng.randn(N, D)
    np.testing.assert_allclose(cg(x), omod.GaussFull(cg.mean, cg.precision).logp(x), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('target', ['gauss_diag', 'funnel', 'gauss_full', 'logistic'])
@pytest.mark.parametrize('D,N', [(256, 4096), (130, 1000), (33, 77)])
def test_exclusive_kl_multivariate_t_throughput_mode_against_oracle(vb, target, D, N):
    """MultivariateT + ExclusiveKL with rng='philox' (the family / objective pair of the reference's robust-regression
    notebook, docs/source/robust-regression.ipynb:324): chi-square draws and normals on the device, samples through the
    Cholesky factor, d/dL = tril(sum g (z / s)') -- no matrix root.  The device noise is read back and the same
    estimator is written out in numpy: x = mu + (z L') / s, value = -(mean f + sum log L_ii) (approximations.py:351-354
    drops the df-only constants), chain rule through L by hand."""
    from viabel_amd import _lib
    from viabel_amd.objectives import _NOISE_SLOT
    df = 9.0
    rng = np.random.RandomState(D + N)
    if target == 'gauss_diag':
        mean, sd = 0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D))
        model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    elif target == 'funnel':
        model, omodel = vb.FunnelModel(D, D // 2), omod.Funnel(D, D // 2)
    elif target == 'logistic':
        X = rng.randn(3 * D, D) / np.sqrt(D)
        y = (rng.rand(3 * D) < 0.5).astype(float)
        model, omodel = vb.LogisticRegressionModel(X, y, 3.0), omod.Logistic(X, y, 3.0)
    else:
        A = rng.randn(D, D)
        model = vb.CorrelatedGaussianModel(0.2 * rng.randn(D), covariance=A @ A.T / D + np.eye(D))
        omodel = omod.GaussFull(model.mean, model.precision)
    approx = vb.MultivariateT(D, df, seed=3, rng='philox')
    A = rng.randn(D, D)
    scale = 0.05 if target == 'funnel' else 0.7
    theta = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(scale * (A @ A.T / D + np.eye(D)))])
    value, grad = vb.ExclusiveKL(approx, model, N)(theta)
    eng = _lib.default_engine()
    chi, z = eng.chisq_get_host(N), eng.noise_get_host(_NOISE_SLOT, N, D)
    mu, L = theta[:D], ofam.free_to_chol(theta[D:], D)
    zs = z / np.sqrt(chi / df)[:, None]
    x = mu + zs @ L.T
    g = omodel.grad(x)
    ov = -(np.mean(omodel.logp(x)) + np.sum(np.log(np.diag(L))))
    dL = np.tril(g.T @ zs) / N
    dL[np.diag_indices(D)] = np.diag(dL) * np.diag(L) + 1.0
    og = -np.concatenate([g.mean(0), dL[np.tril_indices(D)]])
    assert G.rel_err(value, ov) < 1e-12, (value, ov)
    assert G.rel_err(grad, og) < 1e-11, G.rel_err(grad, og)
    # a second call draws fresh noise (the family's Philox stream advances)
    v2, _ = vb.ExclusiveKL(approx, model, N)(theta)
    assert v2 != value


def test_dis_throughput_state_survives_a_rewritten_noise_slot(vb):
    """In throughput mode the residuals of freshly drawn state samples are read straight out of the DIS noise slot.
    A gradient WITHOUT a refresh (num_resampling_batches = 2: every second call) after something else rewrote that
    slot must not look at it: the residuals are then formed from the state samples themselves."""
    from viabel_amd import _lib
    from viabel_amd.objectives import _DIS_SLOT
    D, N, df = 64, 2048, 9.0
    rng = np.random.RandomState(5)
    mean, sd = 0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D))
    model = vb.GaussianModel(mean, sd)
    prior = np.concatenate([np.zeros(D), 0.3 * np.ones(D)])
    A = rng.randn(D, D)
    theta = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(A @ A.T / D + 0.7 * np.eye(D))])
    results = []
    for clobber in (False, True):
        approx = vb.MultivariateT(D, df, seed=8, rng='philox')
        obj = vb.DISInclusiveKL(approx, model, N, ess_target=300, temper_prior=vb.MFGaussian(D),
                                temper_prior_params=prior, use_resampling=True, num_resampling_batches=2)
        np.random.seed(4)
        first = obj(theta)                      # refresh + gradient
        if clobber:
            _lib.default_engine().noise_generate(_DIS_SLOT, N, D, 12345, 99)
        second = obj(theta)                     # gradient only, on the state of the first call
        results.append((first, second))
    (f0, s0), (f1, s1) = results
    assert f0[0] == f1[0] and np.array_equal(f0[1], f1[1])
    assert G.rel_err(s1[0], s0[0]) < 1e-11 and G.rel_err(s1[1], s0[1]) < 1e-10


@pytest.mark.parametrize('alpha', [0.5, 2.0])
@pytest.mark.parametrize('target', ['gauss_diag', 'funnel', 'gauss_full'])
def test_alpha_multivariate_t_throughput_mode_against_numpy(vb, target, alpha):
    """AlphaDivergence over MultivariateT with rng='philox' (objectives.py:443-463): chi-square draws and normals on the
    device, samples through the Cholesky factor x = mu + L z / s.  The device noise is read back and the estimator is
    written out in numpy: lw = f(x) - log q(x), s = exp(lw - max)^alpha, value = log(mean s) / alpha + max (:457-459),
    gradient alpha / N sum s d(lw)/dtheta with d/dL = tril(sum s g (z / s)') and -d log q / d log L_ii = 1."""
    from viabel_amd import _lib
    from viabel_amd.objectives import _NOISE_SLOT
    D, N, df = 96, 1024, 9.0
    rng = np.random.RandomState(D)
    if target == 'gauss_diag':
        mean, sd = 0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D))
        model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    elif target == 'funnel':
        model, omodel = vb.FunnelModel(D, D // 2), omod.Funnel(D, D // 2)
    else:
        A = rng.randn(D, D)
        model = vb.CorrelatedGaussianModel(0.2 * rng.randn(D), covariance=A @ A.T / D + np.eye(D))
        omodel = omod.GaussFull(model.mean, model.precision)
    approx, ofamily = vb.MultivariateT(D, df, seed=3, rng='philox'), ofam.MultivariateT(D, df)
    A = rng.randn(D, D)
    scale = 0.05 if target == 'funnel' else 0.7
    theta = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(scale * (A @ A.T / D + np.eye(D)))])
    np.random.seed(21)
    value, grad = vb.AlphaDivergence(approx, model, N, alpha)(theta)
    eng = _lib.default_engine()
    chi, z = eng.chisq_get_host(N), eng.noise_get_host(_NOISE_SLOT, N, D)
    mu, L = theta[:D], ofam.free_to_chol(theta[D:], D)
    zs = z / np.sqrt(chi / df)[:, None]
    x = mu + zs @ L.T
    lw = omodel.logp(x) - ofamily.log_density(theta, x)
    mx = np.max(lw)
    sv = np.exp(lw - mx) ** alpha
    ov = np.log(np.mean(sv)) / alpha + mx
    g = omodel.grad(x)
    dL = np.tril((sv[:, None] * g).T @ zs)
    dL[np.diag_indices(D)] = np.diag(dL) * np.diag(L) + np.sum(sv)
    og = alpha * np.concatenate([(sv[:, None] * g).sum(0), dL[np.tril_indices(D)]]) / N
    assert G.rel_err(value, ov) < 1e-12, (value, ov)
    assert G.rel_err(grad, og) < 1e-10, G.rel_err(grad, og)


@pytest.mark.parametrize('use_resampling', [False, True])
def test_dis_fullrank_philox_mode_against_oracle(vb, use_resampling):
    """DISInclusiveKL over the dense Gaussian family with rng='philox': the device-resident step of the t family with
    df = 0 (s_n = 1): refresh enqueued, weights / eps / ESS stay on the device, one synchronisation per call.  The noise
    is read back and the oracle (objectives.py:391-414 restated) must reproduce two consecutive steps on it."""
    from viabel_amd import _lib
    from viabel_amd.objectives import _DIS_SLOT
    D, N = 96, 2048
    rng = np.random.RandomState(23)
    approx, ofamily = vb.FullRankGaussian(D, seed=8, rng='philox'), ofam.FullRankGaussian(D)
    mean, sd = 0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D))
    model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    prior = np.concatenate([np.zeros(D), 0.3 * np.ones(D)])
    obj = vb.DISInclusiveKL(approx, model, N, ess_target=300, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, use_resampling=use_resampling)
    ref = oobj.DISInclusiveKL(ofamily, omodel, N, 300, ofam.MFGaussian(D), prior, use_resampling=use_resampling)
    L = np.tril(0.05 * rng.randn(D, D), -1) + np.diag(np.exp(-0.2 + 0.1 * rng.randn(D)))
    theta = approx.pack(0.1 * rng.randn(D), L)
    eng = _lib.default_engine()
    np.random.seed(4)
    for step in range(2):
        value, grad = obj(theta)
        noise = eng.noise_get_host(_DIS_SLOT, N, D)
        if use_resampling:
            ref.refresh(theta, noise)
            counts = eng.dis_weights_get(N, resampled=True)       # the device's multinomial draw
            M = ref._resampling_batch_size
            assert counts.sum() == M
            scale = ref._state_w_sum / N / M
            ov = -np.sum(counts * ofamily.log_density(theta, ref._state_samples)) * scale
            og = -ofamily.log_density_grad_weighted(theta, ref._state_samples, counts) * scale
        else:
            ov, og = ref(theta, noise=noise)
        assert G.rel_err(obj._eps, ref._eps) < 1e-10, (obj._eps, ref._eps)
        assert G.rel_err(value, ov) < 1e-10, (step, value, ov)
        assert G.rel_err(grad, og) < 1e-9, (step, G.rel_err(grad, og))
        assert G.rel_err(obj._state_log_q, ref._state_log_q) < 1e-11
        theta = theta - 0.002 * grad / (1 + np.abs(grad))


def test_dis_prior_copy_follows_the_layout_across_sample_counts(vb):
    """Round-3 ADVICE (high): the device copy of the tempering prior sits at an offset that moves with
    num_mc_samples.  Two DISInclusiveKL objectives on one engine, same D and same prior, N = 4096 then N = 1024
    (the smaller job reuses the allocation): both must match the oracle on the read-back noise."""
    from viabel_amd import _lib
    from viabel_amd.objectives import _DIS_SLOT
    D = 64
    rng = np.random.RandomState(29)
    mean, sd = 0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D))
    model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    prior = np.concatenate([0.1 * rng.randn(D), 0.3 + 0.05 * rng.randn(D)])
    ofamily = ofam.FullRankGaussian(D)
    eng = _lib.default_engine()
    L = np.tril(0.05 * rng.randn(D, D), -1) + np.diag(np.exp(-0.2 + 0.1 * rng.randn(D)))
    for N in (4096, 1024, 2048):
        approx = vb.FullRankGaussian(D, seed=8, rng='philox')
        theta = approx.pack(0.1 * rng.randn(D), L)
        obj = vb.DISInclusiveKL(approx, model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                                temper_prior_params=prior, use_resampling=False)
        ref = oobj.DISInclusiveKL(ofamily, omodel, N, N // 8, ofam.MFGaussian(D), prior, use_resampling=False)
        value, grad = obj(theta)
        noise = eng.noise_get_host(_DIS_SLOT, N, D)
        ov, og = ref(theta, noise=noise)
        assert G.rel_err(obj._eps, ref._eps) < 1e-10, (N, obj._eps, ref._eps)
        assert G.rel_err(value, ov) < 1e-10, (N, value, ov)
        assert G.rel_err(grad, og) < 1e-9, (N, G.rel_err(grad, og))


def _product_prior(vb, fx):
    kind, D = str(fx['prior_kind']), int(fx['dim'])
    if kind == 'mf_student_t':
        return vb.MFStudentT(D, float(fx['prior_df']))
    if kind == 'multivariate_t':
        return vb.MultivariateT(D, float(fx['prior_df']))
    return vb.LRGaussian(D, k=int(fx['prior_rank']))


@pytest.mark.parametrize('path', G.fixtures('disprior_'), ids=G.ids(G.fixtures('disprior_')))
def test_dis_general_tempering_prior_golden(vb, path):
    """objectives.py:283-285: temper_prior may be any family -- fixtures from the reference's own DISInclusiveKL with
    MFStudentT / MultivariateT / LRGaussian priors (vb_dis_set_temper_prior)."""
    fx = G.load(path)
    obj = vb.DISInclusiveKL(product_family(vb, fx, int(fx['seed'])), product_model(vb, fx), int(fx['n']),
                            ess_target=int(fx['ess_target']), temper_prior=_product_prior(vb, fx),
                            temper_prior_params=fx['prior_params'], use_resampling=bool(fx['use_resampling']))
    np.random.seed(int(fx['np_seed']))
    value, grad = obj(fx['theta'])
    assert G.rel_err(obj._eps, fx['eps']) < 1e-11
    assert G.rel_err(obj._state_w_clipped, fx['w_clipped']) < 1e-10
    assert G.rel_err(value, fx['value']) < 1e-11
    assert G.rel_err(grad, fx['grad']) < 1e-11
    assert G.rel_err(grad, fx['grad_fd']) < 2e-6


@pytest.mark.parametrize('prior_kind', ['mf_student_t', 'fullrank', 'multivariate_t'])
@pytest.mark.parametrize('family', ['mf_gaussian', 'fullrank_philox', 'multivariate_t', 'lr_gaussian'])
def test_dis_general_tempering_prior_against_oracle(vb, family, prior_kind):
    """The same at sizes where the dense prior's N x D x D product runs real MFMA tiles; the MFGaussian prior on the
    same engine before and after (the installed prior must not leak into an objective that does not use it)."""
    from viabel_amd import _lib
    from viabel_amd.objectives import _DIS_SLOT
    D, N = 70, 1500
    rng = np.random.RandomState(41)
    mean, sd = 0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D))
    model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    A = rng.randn(D, D)
    S = A @ A.T / D + np.eye(D)
    if prior_kind == 'mf_student_t':
        prior, oprior = vb.MFStudentT(D, 6.0), ofam.MFStudentT(D, 6.0)
        pp = np.concatenate([0.1 * rng.randn(D), 0.5 + 0.1 * rng.randn(D)])
    elif prior_kind == 'fullrank':
        prior, oprior = vb.FullRankGaussian(D), ofam.FullRankGaussian(D)
        pp = prior.pack(0.1 * rng.randn(D), np.linalg.cholesky(2.0 * S))
    else:
        prior, oprior = vb.MultivariateT(D, 8.0), ofam.MultivariateT(D, 8.0)
        pp = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(2.0 * S)])
    eng = _lib.default_engine()
    if family == 'mf_gaussian':
        approx, ofamily = vb.MFGaussian(D, seed=5), ofam.MFGaussian(D)
        theta = np.concatenate([0.1 * rng.randn(D), -0.3 + 0.1 * rng.randn(D)])
    elif family == 'fullrank_philox':
        approx, ofamily = vb.FullRankGaussian(D, seed=5, rng='philox'), ofam.FullRankGaussian(D)
        theta = approx.pack(0.1 * rng.randn(D), np.linalg.cholesky(0.6 * S))
    elif family == 'multivariate_t':
        approx, ofamily = vb.MultivariateT(D, 12.0, seed=5), ofam.MultivariateT(D, 12.0)
        theta = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(0.6 * S)])
    else:
        approx, ofamily = vb.LRGaussian(D, seed=5, k=3), ofam.LRGaussian(D, 3)
        theta = np.concatenate([0.1 * rng.randn(D), -0.3 + 0.1 * rng.randn(D), 0.2 * rng.randn(D * 3)])
    for p_fam, p_ofam, p_par in ((vb.MFGaussian(D), ofam.MFGaussian(D), np.concatenate([np.zeros(D), 0.3 * np.ones(D)])),
                                 (prior, oprior, pp),
                                 (vb.MFGaussian(D), ofam.MFGaussian(D), np.concatenate([np.zeros(D), 0.3 * np.ones(D)]))):
        obj = vb.DISInclusiveKL(approx, model, N, ess_target=200, temper_prior=p_fam, temper_prior_params=p_par,
                                use_resampling=False)
        ref = oobj.DISInclusiveKL(ofamily, omodel, N, 200, p_ofam, p_par, use_resampling=False)
        if family == 'fullrank_philox':
            value, grad = obj(theta)
            noise = eng.noise_get_host(_DIS_SLOT, N, D)
        else:
            rs = np.random.RandomState(0)
            rs.set_state(approx._rs.get_state())         # the draws this call is about to consume
            value, grad = obj(theta)
            noise = ofamily.draw_noise(rs, N)
        ov, og = ref(theta, noise=noise)
        assert ref._eps > 0.0                            # the prior takes part in the weights
        assert G.rel_err(obj._eps, ref._eps) < 1e-9, (obj._eps, ref._eps)
        assert G.rel_err(value, ov) < 1e-9, (value, ov)
        assert G.rel_err(grad, og) < 1e-8, G.rel_err(grad, og)


@pytest.mark.parametrize('thr', [0.02, 0.005])
@pytest.mark.parametrize('use_resampling', [False, True])
@pytest.mark.parametrize('family', ['fullrank_philox', 'multivariate_t_philox', 'mf_gaussian', 'multivariate_t'])
def test_dis_clip_active_branch(vb, family, use_resampling, thr):
    """objectives.py:370-386 with w_clip_threshold < 1 (the branch the default threshold 10 never takes): on the
    device-resident step (vb_dis_clip_mvt) and on the host path, against the oracle's fixed-point restatement (the
    reference's own line :385 cannot run: pinned by the oracle, see its _clip)."""
    from viabel_amd import _lib
    from viabel_amd.objectives import _DIS_SLOT
    D, N = 40, 1024
    rng = np.random.RandomState(43)
    mean, sd = 0.5 * rng.randn(D), np.exp(0.2 * rng.randn(D))
    model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    prior = np.concatenate([np.zeros(D), 0.3 * np.ones(D)])
    A = rng.randn(D, D)
    S = A @ A.T / D + np.eye(D)
    eng = _lib.default_engine()
    philox = family.endswith('philox')
    if family == 'fullrank_philox':
        approx, ofamily = vb.FullRankGaussian(D, seed=9, rng='philox'), ofam.FullRankGaussian(D)
        theta = approx.pack(0.1 * rng.randn(D), np.linalg.cholesky(0.5 * S))
    elif family.startswith('multivariate_t'):
        approx = vb.MultivariateT(D, 10.0, seed=9, rng='philox' if philox else 'numpy')
        theta = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(0.5 * S)])

        class CholeskySampledT(ofam.MultivariateT):
            def sample_from_noise(self, theta, noise):
                chi, z = noise
                mu, Sg = self.split(theta)
                return mu + (z @ np.linalg.cholesky(Sg).T) / np.sqrt(chi / self.df)[:, None]
        ofamily = CholeskySampledT(D, 10.0) if philox else ofam.MultivariateT(D, 10.0)
    else:
        approx, ofamily = vb.MFGaussian(D, seed=9), ofam.MFGaussian(D)
        theta = np.concatenate([0.1 * rng.randn(D), -0.2 + 0.1 * rng.randn(D)])
    # a large ESS target keeps eps at 1 (weights = prior / q: heavy-tailed) so that the clipping has work to do
    obj = vb.DISInclusiveKL(approx, model, N, ess_target=N // 2, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, use_resampling=use_resampling, w_clip_threshold=thr)
    ref = oobj.DISInclusiveKL(ofamily, omodel, N, N // 2, ofam.MFGaussian(D), prior, use_resampling=use_resampling,
                              w_clip_threshold=thr)
    np.random.seed(6)
    rs_state = None if philox else approx._rs.get_state()
    np_state = np.random.get_state()
    value, grad = obj(theta)
    if philox:
        z = eng.noise_get_host(_DIS_SLOT, N, D)
        noise = (eng.chisq_get_host(N), z) if family.startswith('multivariate_t') else z
    else:
        rs = np.random.RandomState(0)
        rs.set_state(rs_state)
        noise = ofamily.draw_noise(rs, N)
    ref.refresh(theta, noise)
    n_clipped = int(np.sum(ref._state_w_clipped != ref._weights(ref._eps, ofam.MFGaussian(D).log_density(
        prior, ref._state_samples), ref._state_log_p, ref._state_log_q)))
    assert n_clipped >= 1, 'the test needs weights above the threshold'
    assert G.rel_err(obj._state_w_clipped, ref._state_w_clipped) < 1e-10
    if not use_resampling:
        w = ref._state_w_clipped
        ov = -np.sum(w * ofamily.log_density(theta, ref._state_samples)) / N
        og = -ofamily.log_density_grad_weighted(theta, ref._state_samples, w) / N
    else:
        M = ref._resampling_batch_size
        if philox:
            counts = eng.dis_weights_get(N, resampled=True)
        else:
            np.random.set_state(np_state)
            counts = np.bincount(np.random.choice(N, size=M, p=ref._state_w_clipped / np.sum(ref._state_w_clipped)),
                                 minlength=N).astype(float)
        assert counts.sum() == M
        scale = np.sum(ref._state_w_clipped) / N / M
        ov = -np.sum(counts * ofamily.log_density(theta, ref._state_samples)) * scale
        og = -ofamily.log_density_grad_weighted(theta, ref._state_samples, counts) * scale
    assert G.rel_err(value, ov) < 1e-10, (value, ov)
    assert G.rel_err(grad, og) < 1e-9, G.rel_err(grad, og)


def test_lazy_dis_state_survives_another_objective(vb):
    """A device-resident step leaves its weights / per-sample logs on the device until they are read.  Another objective
    refreshing on the same engine -- or the t family's resident ExclusiveKL, which reuses the buffers -- must not change
    what the first one reports afterwards (the reference's ``_state_*`` are plain arrays of their own refresh)."""
    D, N = 12, 4096
    rng = np.random.RandomState(21)
    model = vb.GaussianModel(0.3 * rng.randn(D), np.exp(0.1 * rng.randn(D)))
    prior = np.concatenate([np.zeros(D), 0.3 * np.ones(D)])
    A = rng.randn(D, D)
    theta = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(A @ A.T / D + np.eye(D))])

    def make(seed, **kw):
        return vb.DISInclusiveKL(vb.MultivariateT(D, 9.0, seed=seed, rng=kw.pop('rng', 'numpy')), model, N, ess_target=500,
                                 temper_prior=vb.MFGaussian(D), temper_prior_params=prior, use_resampling=False, **kw)
    for rng_kind in ('numpy', 'philox'):
        a, b = make(3, rng=rng_kind), make(3, rng=rng_kind)
        a(theta)
        want_w, want_lp = np.array(a._state_w_clipped), np.array(a._state_log_p_unnormalized)
        b(theta)                                    # read at once: the reference values
        c = make(4, rng=rng_kind)                   # another stream: another state
        b2 = make(3, rng=rng_kind)
        b2(theta)                                   # same draws as `a`; nothing read yet
        c(theta)                                    # overwrites the engine's state
        vb.ExclusiveKL(vb.MultivariateT(D, 9.0, seed=8), model, N)(theta)      # reuses the buffers (resident route)
        np.testing.assert_array_equal(b2._state_w_clipped, want_w)
        np.testing.assert_array_equal(b2._state_log_p_unnormalized, want_lp)


@pytest.mark.parametrize('rng_kind', ['numpy', 'philox'])
def test_resident_elbo_monitor_beside_a_dis_fit_with_kept_weights(vb, rng_kind):
    """the resident ExclusiveKL / AlphaDivergence routes of the t family used the DIS state's buffer as scratch
    without bumping its generation, so a DISInclusiveKL with num_resampling_batches > 1 on the same engine failed its
    next kept-weights step with a bare 'no multivariate-t DIS state'.  They have a buffer of their own now: an ELBO /
    alpha monitor evaluated between the steps leaves the fit's values and gradients exactly what they are without it."""
    D, N = 64, 4096
    rng = np.random.RandomState(23)
    model = vb.GaussianModel(0.3 * rng.randn(D), np.exp(0.1 * rng.randn(D)))
    prior = np.concatenate([np.zeros(D), 0.3 * np.ones(D)])
    A = rng.randn(D, D)
    theta0 = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(A @ A.T / D + np.eye(D))])

    def fit(monitor):
        obj = vb.DISInclusiveKL(vb.MultivariateT(D, 9.0, seed=3, rng=rng_kind), model, N, ess_target=500,
                                temper_prior=vb.MFGaussian(D), temper_prior_params=prior, use_resampling=True,
                                num_resampling_batches=3)
        elbo = vb.ExclusiveKL(vb.MultivariateT(D, 9.0, seed=8), model, N)
        alpha = vb.AlphaDivergence(vb.MultivariateT(D, 9.0, seed=9), model, N, 0.5)
        np.random.seed(5)
        theta, out = theta0.copy(), []
        for it in range(7):
            state = np.random.get_state()
            v, g = obj(theta)
            if monitor:
                after = np.random.get_state()
                elbo(theta)
                alpha(theta)
                np.random.set_state(after)       # (the alpha monitor draws its seed from the global generator)
            del state
            out.append((v, g))
            theta = theta - 0.01 * g / (1.0 + np.abs(g))
        return out
    plain, watched = fit(False), fit(True)
    for (v0, g0), (v1, g1) in zip(plain, watched):
        assert v0 == v1
        np.testing.assert_array_equal(g0, g1)
