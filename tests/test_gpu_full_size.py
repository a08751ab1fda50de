"""Parity at the sizes BASELINE.json states for configs[3] and configs[4] (the smaller shapes are covered in
test_gpu_objectives.py / test_gpu_meanfield.py / test_gpu_convergence.py):

  configs[3]  MultivariateT(256, df=100) + DISInclusiveKL, N_mc = 16 384, resampling on and off
  configs[4]  MFGaussian + ExclusiveKL on Bayesian logistic regression, D = 2000, n_data = 8192, N_mc = 8192,
              one evaluation against the oracle and the RAABBVI / FASO optimiser loop (device-resident chunks
              against the host loop, bit for bit)

The oracle (numpy, oracle/) needs a few seconds per evaluation at these sizes.  Tolerances as in the small-shape
tests: values 1e-10..1e-11 relative, gradients 1e-9..1e-10 relative to max |grad|.

The variational parameter of the configs[3] test is NOT `init_param()` (Sigma = 10 I): in 256 dimensions its
importance weights against any unit-scale target collapse to a single sample (ESS = 1).  Here q sits
on the tempering prior (mean 0, log sigma 0.5) up to a small correlated perturbation, and the target is shifted
away from it, so ESS(eps = 0) is far below the target, ESS(eps = 1) far above, and the bisection has to find an
interior eps: asserted below.
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


def c3_problem(rng, D):
    mean = 0.3 * rng.randn(D)
    sd = np.exp(0.5 + 0.02 * rng.randn(D))
    prior = np.concatenate([np.zeros(D), 0.5 * np.ones(D)])
    A = rng.randn(D, D)
    Sigma = np.e * np.eye(D) + 0.04 * (A @ A.T / D - np.eye(D))
    theta = np.concatenate([0.02 * rng.randn(D), ofam.psd_to_free(Sigma)])
    return mean, sd, prior, theta


@pytest.mark.parametrize('use_resampling', [True, False])
def test_c3_multivariate_t_dis_full_size(vb, use_resampling):
    D, N, df, ess_target = 256, 16384, 100, 2048
    rng = np.random.RandomState(33)
    mean, sd, prior, theta = c3_problem(rng, D)
    approx, ofamily = vb.MultivariateT(D, df, seed=6), ofam.MultivariateT(D, df)
    model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    kw = dict(use_resampling=use_resampling, num_resampling_batches=2)
    obj = vb.DISInclusiveKL(approx, model, N, ess_target=ess_target, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, **kw)
    ref = oobj.DISInclusiveKL(ofamily, omodel, N, ess_target, ofam.MFGaussian(D), prior, **kw)
    rs = np.random.RandomState(6)
    np.random.seed(12)
    for step in range(2):
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
        if step == 0:
            # the tempering had work to do: interior eps, effective sample size on target (not the ESS = 1 collapse)
            assert 0.0 < ref._eps < 1.0, ref._eps
            assert abs(ref._ess_val - ess_target) < 0.02 * ess_target, ref._ess_val
            assert abs(obj._ess - ess_target) < 0.02 * ess_target, obj._ess
        assert G.rel_err(obj._eps, ref._eps) < 1e-10, (obj._eps, ref._eps)
        assert G.rel_err(obj._state_log_q, ref._state_log_q) < 1e-11
        assert G.rel_err(obj._state_log_p_unnormalized, ref._state_log_p) < 1e-11
        assert G.rel_err(obj._state_w_clipped, ref._state_w_clipped) < 1e-8     # exp of O(100) log weights
        assert G.rel_err(value, ov) < 1e-10, (step, value, ov)
        assert G.rel_err(grad, og) < 1e-9, (step, G.rel_err(grad, og))
        theta = theta - 0.002 * grad / (1 + np.abs(grad))


class CholeskySampledT(ofam.MultivariateT):
    """rng='philox' draws the DIS state samples as x = mu + (z L') / s with the Cholesky factor; the oracle
    family samples the same way, everything downstream of the samples is the reference's arithmetic."""

    def sample_from_noise(self, theta, noise):
        chi, z = noise
        mu, S = self.split(theta)
        return mu + (z @ np.linalg.cholesky(S).T) / np.sqrt(chi / self.df)[:, None]


@pytest.mark.parametrize('use_resampling,psis_smooth', [(False, False), (True, False), (False, True), (True, True)])
def test_c3_multivariate_t_dis_full_size_throughput_mode(vb, use_resampling, psis_smooth):
    """BASELINE configs[3] at full size through the path bench.py's c3_mvt_dis leg TIMES: rng='philox' (normals and
    chi-square draws on the device), Cholesky sampling, device factor algebra and the packed chain rule
    (vb_dis_grad_mvt_packed); the device noise is read back and the oracle must reproduce the step on it.  With
    psis_smooth the tempered weights additionally go through psislw (_psis.py:113-209) at this size."""
    from viabel_amd import _lib
    from viabel_amd.objectives import _DIS_SLOT
    from oracle import psis as opsis
    D, N, df, ess_target = 256, 16384, 100, 2048
    rng = np.random.RandomState(33)
    mean, sd, prior, theta = c3_problem(rng, D)
    approx, ofamily = vb.MultivariateT(D, df, seed=6, rng='philox'), CholeskySampledT(D, df)
    model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    obj = vb.DISInclusiveKL(approx, model, N, ess_target=ess_target, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, use_resampling=use_resampling, psis_smooth=psis_smooth)
    ref = oobj.DISInclusiveKL(ofamily, omodel, N, ess_target, ofam.MFGaussian(D), prior, use_resampling=use_resampling)
    eng = _lib.default_engine()
    np.random.seed(12)
    for step in range(2):
        state = np.random.get_state()
        value, grad = obj(theta)
        np.random.set_state(state)
        noise = (eng.chisq_get_host(N), eng.noise_get_host(_DIS_SLOT, N, D))     # what the device drew for this refresh
        ref.refresh(theta, noise)
        if step == 0:      # the tempering had work to do: interior eps, effective sample size on target
            assert 0.0 < ref._eps < 1.0 and abs(ref._ess_val - ess_target) < 0.02 * ess_target, (ref._eps, ref._ess_val)
        assert G.rel_err(obj._eps, ref._eps) < 1e-10, (obj._eps, ref._eps)
        assert G.rel_err(obj._state_log_q, ref._state_log_q) < 1e-11
        assert G.rel_err(obj._state_log_p_unnormalized, ref._state_log_p) < 1e-11
        w = ref._state_w_clipped
        if psis_smooth:
            smoothed, khat = opsis.psis_smooth(np.log(w))
            w = np.sum(w) * np.exp(smoothed)
            assert abs(obj._khat - khat) < 1e-7, (obj._khat, khat)
            assert G.rel_err(obj._state_w_clipped, w) < 1e-8
        xs = ref._state_samples
        if use_resampling:
            # the multinomial draw happens on the device in this mode (Philox uniforms inverted through the running
            # sums of the weights): read the counts back; the objective is objectives.py:410-414 on those indices
            counts = eng.dis_weights_get(N, resampled=True)
            M = ref._resampling_batch_size
            assert counts.sum() == M and np.all(counts == np.round(counts)) and counts.min() >= 0
            top = np.argsort(w)[-200:]            # the draw follows the weights: the 200 heaviest samples take their share
            assert abs(counts[top].sum() / M - w[top].sum() / w.sum()) < 0.05
            # ... and block by block (32 index ranges, across the chunk boundaries of the two-level running sums):
            # Pearson's statistic against M w / sum w has 31 degrees of freedom (mean 31, sd 7.9)
            expect = M * w.reshape(32, -1).sum(axis=1) / w.sum()
            chi2 = float(np.sum((counts.reshape(32, -1).sum(axis=1) - expect) ** 2 / expect))
            assert chi2 < 80.0, chi2
            scale = ref._state_w_sum / N / M
            ov = -np.sum(counts * ofamily.log_density(theta, xs)) * scale
            og = -ofamily.log_density_grad_weighted(theta, xs, counts) * scale
        else:
            ov = -np.sum(w * ofamily.log_density(theta, xs)) / N
            og = -ofamily.log_density_grad_weighted(theta, xs, w) / N
            assert G.rel_err(obj._state_w_clipped, w) < 1e-8          # fetched from the device on first access
        assert G.rel_err(value, ov) < 1e-10, (step, value, ov)
        assert G.rel_err(grad, og) < 1e-9, (step, G.rel_err(grad, og))
        theta = theta - 0.002 * grad / (1 + np.abs(grad))


def c4_problem(D=2000, n_data=8192, seed=4):
    rng = np.random.RandomState(seed)
    X = rng.randn(n_data, D) / np.sqrt(D)
    beta = rng.randn(D)
    y = (rng.rand(n_data) < 1 / (1 + np.exp(-X @ beta))).astype(float)
    theta = np.concatenate([0.05 * rng.randn(D), -2.0 + 0.1 * rng.randn(D)])
    return X, y, theta


def test_c4_logistic_full_size_against_oracle(vb):
    D, n_data, N = 2000, 8192, 8192
    X, y, theta = c4_problem(D, n_data)
    model, omodel = vb.LogisticRegressionModel(X, y, 10.0), omod.Logistic(X, y, 10.0)
    for pd in (False, True):
        approx = vb.MFGaussian(D, seed=9)
        value, grad = vb.ExclusiveKL(approx, model, N, use_path_deriv=pd)(theta)
        noise = np.random.RandomState(9).randn(N, D)
        ov, og = oobj.exclusive_kl(ofam.MFGaussian(D), omodel, theta, noise, use_path_deriv=pd)
        assert G.rel_err(value, ov) < 1e-12, (pd, value, ov)
        assert G.rel_err(grad, og) < 1e-10, (pd, G.rel_err(grad, og))


def test_logistic_ragged_large_tiles_against_oracle(vb):
    """eta = Z X' of 2900 x 2950 x 48: 23 x 24 tiles of 128 x 128 -- the two-stage large-tile GEMM kernel with rows and
    columns that end inside a tile (C4 itself only has a ragged column edge)."""
    D, n_data, N = 48, 2950, 2900
    X, y, theta = c4_problem(D, n_data, seed=12)
    model, omodel = vb.LogisticRegressionModel(X, y, 10.0), omod.Logistic(X, y, 10.0)
    approx = vb.MFGaussian(D, seed=9)
    value, grad = vb.ExclusiveKL(approx, model, N)(theta)
    noise = np.random.RandomState(9).randn(N, D)
    ov, og = oobj.exclusive_kl(ofam.MFGaussian(D), omodel, theta, noise)
    assert G.rel_err(value, ov) < 1e-12, (value, ov)
    assert G.rel_err(grad, og) < 1e-10, G.rel_err(grad, og)


def test_c4_logistic_optimiser_loop_full_size(vb, capsys):
    """The optimiser loop of configs[4] at full size: RMSProp iterations with fresh Philox noise through the host
    loop (one blocking objective call + numpy step per iteration, optimization.py:91-112) and through the
    device-resident loop that FASO / RAABBVI run between two convergence checks -- same iterates bit for bit; and
    the objective goes down."""
    from viabel_amd.optimization import RMSProp
    D, n_data, N = 2000, 8192, 8192
    X, y, _ = c4_problem(D, n_data)
    model = vb.LogisticRegressionModel(X, y, 10.0)
    theta0 = np.concatenate([np.zeros(D), -2.0 * np.ones(D)])
    hist = {}
    for mode, on_device in (('host', False), ('device', True)):
        obj = vb.ExclusiveKL(vb.MFGaussian(D, rng='philox', seed=2), model, N)
        res = RMSProp(0.02).optimize(12, obj, theta0, on_device=on_device)
        hist[mode] = res
    capsys.readouterr()
    assert np.array_equal(hist['host']['value_history'], hist['device']['value_history'])
    assert np.array_equal(hist['host']['opt_param'], hist['device']['opt_param'])
    v = hist['device']['value_history']
    assert v[-1] < v[0]


def test_c4_logistic_raabbvi_runs_full_size(vb, capsys):
    """`bbvi` with its default RAABBVI step-size adaptation on the full-size logistic target, a bounded number of
    iterations: the adaptive loop runs its device-resident chunks and the ELBO estimate improves."""
    D, n_data, N = 2000, 8192, 8192
    X, y, _ = c4_problem(D, n_data)
    model = vb.LogisticRegressionModel(X, y, 10.0)
    np.random.seed(3)
    approx = vb.MFGaussian(D, rng='philox', seed=2)
    objective = vb.ExclusiveKL(approx, model, N)
    init = np.concatenate([np.zeros(D), -2.0 * np.ones(D)])
    v0, _ = objective(init)
    results = vb.bbvi(D, objective=objective, init_var_param=init, n_iters=300, learning_rate=0.05)
    capsys.readouterr()
    v1, _ = objective(results['opt_param'])
    assert np.isfinite(v1) and v1 < v0, (v0, v1)


@pytest.mark.parametrize('path_deriv', [False, True])
@pytest.mark.parametrize('target', ['gauss_diag', 'funnel', 'gauss_full'])
def test_multivariate_t_exclusive_kl_reference_mode_resident(vb, target, path_deriv):
    """MultivariateT(256) + ExclusiveKL with the DEFAULT rng='numpy' at N = 16 384: the reference's chi-square and normal
    streams generated on the device, its symmetric root (approximations.py:348) and the root's Frechet derivative by
    device iterations, the chain rule to the free Cholesky parameters on the device (vb_elbo_grad_mvt_symroot) -- against
    the oracle's eigen-decomposition route on numpy's own draws, and against the host-root route it replaces.
    path_deriv: objectives.py:156-159 (vb_elbo_grad_mvt_symroot_path: the score's noise-only sums and the inverse root on
    the device as well)."""
    import os
    from viabel_amd import objectives as vobj
    D, N, df = 256, 16384, 9.0
    rng = np.random.RandomState(41)
    if target == 'gauss_diag':
        mean, sd = 0.3 * rng.randn(D), np.exp(0.2 * rng.randn(D))
        model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    elif target == 'funnel':
        model, omodel = vb.FunnelModel(D, D // 2), omod.Funnel(D, D // 2)
    else:
        A = rng.randn(D, D)
        S = A @ A.T / D + np.eye(D)
        mean = rng.randn(D)
        model, omodel = vb.CorrelatedGaussianModel(mean, covariance=S), omod.GaussFull(mean, np.linalg.inv(S))
    B = rng.randn(D, D)
    theta = np.concatenate([0.2 * rng.randn(D), ofam.psd_to_free(0.05 * (B @ B.T / D + 0.5 * np.eye(D)))])
    approx = vb.MultivariateT(D, df, seed=6)
    obj = vb.ExclusiveKL(approx, model, N, use_path_deriv=path_deriv)
    calls = []
    from viabel_amd import _lib
    eng = _lib.default_engine()
    real = eng.elbo_grad_mvt_symroot
    eng.elbo_grad_mvt_symroot = lambda *a, **k: (calls.append(k.get('path_deriv', False)), real(*a, **k))[1]
    ref = np.random.RandomState(6)
    for call in range(2):
        value, grad = obj(theta)
        noise = ofam.MultivariateT(D, df).draw_noise(ref, N)
        ov, og = oobj.exclusive_kl(ofam.MultivariateT(D, df), omodel, theta, noise, path_deriv)
        assert abs(value - ov) <= 1e-11 * abs(ov), (call, value, ov)
        np.testing.assert_allclose(grad, og, rtol=0, atol=1e-9 * np.max(np.abs(og)))
    eng.elbo_grad_mvt_symroot = real
    assert calls == [path_deriv, path_deriv]      # (the resident entry point ran, in the right form)
    # the host-root route on the same draws (the resident path switched off through its dimension gate)
    keep = vobj._HOST_ROOT_MAX_DIM, vobj._RESIDENT_GATE
    try:
        vobj._HOST_ROOT_MAX_DIM = vobj._RESIDENT_GATE = 10 ** 6
        approx2 = vb.MultivariateT(D, df, seed=6)
        v2, g2 = vb.ExclusiveKL(approx2, model, N, use_path_deriv=path_deriv)(theta)
    finally:
        vobj._HOST_ROOT_MAX_DIM, vobj._RESIDENT_GATE = keep
    approx3 = vb.MultivariateT(D, df, seed=6)
    v3, g3 = vb.ExclusiveKL(approx3, model, N, use_path_deriv=path_deriv)(theta)
    assert abs(v3 - v2) <= 1e-11 * abs(v2)
    np.testing.assert_allclose(g3, g2, rtol=0, atol=1e-9 * np.max(np.abs(g2)))


@pytest.mark.parametrize('alpha', [2.0, 0.5])
@pytest.mark.parametrize('target', ['gauss_diag', 'funnel', 'gauss_full'])
def test_multivariate_t_alpha_reference_mode_resident(vb, target, alpha):
    """MultivariateT(256) + AlphaDivergence with the DEFAULT rng='numpy' at N = 16 384, resident on the device
    (vb_alpha_grad_mvt_symroot: the call's fresh RandomState(seed) of objectives.py:455-456 drawn there, the symmetric
    root and its Frechet derivative by device iterations) -- against the oracle on numpy's own draws and against the
    host-root route it replaces."""
    from viabel_amd import _lib
    from viabel_amd import objectives as vobj
    D, N, df = 256, 16384, 9.0
    rng = np.random.RandomState(43)
    if target == 'gauss_diag':
        mean, sd = 0.3 * rng.randn(D), np.exp(0.2 * rng.randn(D))
        model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    elif target == 'funnel':
        model, omodel = vb.FunnelModel(D, D // 2), omod.Funnel(D, D // 2)
    else:
        A = rng.randn(D, D)
        S = A @ A.T / D + np.eye(D)
        mean = rng.randn(D)
        model, omodel = vb.CorrelatedGaussianModel(mean, covariance=S), omod.GaussFull(mean, np.linalg.inv(S))
    B = rng.randn(D, D)
    theta = np.concatenate([0.2 * rng.randn(D), ofam.psd_to_free(0.05 * (B @ B.T / D + 0.5 * np.eye(D)))])
    omvt = ofam.MultivariateT(D, df)
    eng = _lib.default_engine()
    calls = []
    real = eng.alpha_grad_mvt_symroot
    eng.alpha_grad_mvt_symroot = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        np.random.seed(17)
        value, grad = vb.AlphaDivergence(vb.MultivariateT(D, df), model, N, alpha)(theta)
    finally:
        eng.alpha_grad_mvt_symroot = real
    assert calls == [1]
    np.random.seed(17)
    noise = omvt.draw_noise(np.random.RandomState(np.random.randint(2 ** 32)), N)
    ov, og = oobj.alpha_divergence(omvt, omodel, theta, noise, alpha)
    assert abs(value - ov) <= 1e-11 * abs(ov), (value, ov)
    np.testing.assert_allclose(grad, og, rtol=0, atol=1e-9 * np.max(np.abs(og)))
    keep = vobj._HOST_ROOT_MAX_DIM, vobj._RESIDENT_GATE
    try:
        vobj._HOST_ROOT_MAX_DIM = vobj._RESIDENT_GATE = 10 ** 6
        np.random.seed(17)
        v2, g2 = vb.AlphaDivergence(vb.MultivariateT(D, df), model, N, alpha)(theta)
    finally:
        vobj._HOST_ROOT_MAX_DIM, vobj._RESIDENT_GATE = keep
    assert abs(value - v2) <= 1e-11 * abs(v2)
    np.testing.assert_allclose(grad, g2, rtol=0, atol=1e-9 * np.max(np.abs(g2)))


@pytest.mark.parametrize('D,N', [(2, 4096), (5, 4096), (16, 4096), (33, 4096), (100, 4096), (160, 4096), (48, 100), (64, 300),
                                 (130, 777), (256, 1000)])
def test_resident_reference_mode_at_small_dimensions(vb, D, N):
    """The resident reference-identical routes of the t family are taken for every D once the chi-square draws are on the
    device (N >= 4096; objectives._RESIDENT_GATE, measured faster than the host-root route from D = 2 up): ExclusiveKL in
    both forms, AlphaDivergence and the DIS step against the oracle on numpy's own draws, at dimensions where rounds 3-4
    went through LAPACK on the host.  Below N = 4096 the chi-square draws move to the device for the resident route's sake
    from D = 48 on (objectives._RESIDENT_SMALL_N_MIN_DIM)."""
    from viabel_amd import _lib
    df = 9.0
    rng = np.random.RandomState(100 + D)
    mean, sd = 0.3 * rng.randn(D), np.exp(0.2 * rng.randn(D))
    model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    B = rng.randn(D, D)
    theta = np.concatenate([0.2 * rng.randn(D), ofam.psd_to_free(0.3 * (B @ B.T / D + 0.5 * np.eye(D)))])
    omvt = ofam.MultivariateT(D, df)
    eng = _lib.default_engine()
    used = []
    real = {k: getattr(eng, k) for k in ('elbo_grad_mvt_symroot', 'alpha_grad_mvt_symroot', 'dis_refresh_mvt_symroot')}
    for k, f in real.items():
        setattr(eng, k, (lambda name, fn: lambda *a, **kw: (used.append(name), fn(*a, **kw))[1])(k, f))
    try:
        for pd in (False, True):
            v, g = vb.ExclusiveKL(vb.MultivariateT(D, df, seed=6), model, N, use_path_deriv=pd)(theta)
            ov, og = oobj.exclusive_kl(omvt, omodel, theta, omvt.draw_noise(np.random.RandomState(6), N), pd)
            assert abs(v - ov) <= 1e-10 * abs(ov), (pd, v, ov)
            np.testing.assert_allclose(g, og, rtol=0, atol=1e-8 * np.max(np.abs(og)))
        np.random.seed(17)
        v, g = vb.AlphaDivergence(vb.MultivariateT(D, df), model, N, 0.5)(theta)
        np.random.seed(17)
        noise = omvt.draw_noise(np.random.RandomState(np.random.randint(2 ** 32)), N)
        ov, og = oobj.alpha_divergence(omvt, omodel, theta, noise, 0.5)
        assert abs(v - ov) <= 1e-10 * abs(ov)
        np.testing.assert_allclose(g, og, rtol=0, atol=1e-8 * np.max(np.abs(og)))
        prior = np.concatenate([np.zeros(D), 0.5 * np.ones(D)])
        obj = vb.DISInclusiveKL(vb.MultivariateT(D, df, seed=6), model, N, ess_target=max(2, N // 8), temper_prior=vb.MFGaussian(D),
                                temper_prior_params=prior, use_resampling=False)
        ref = oobj.DISInclusiveKL(omvt, omodel, N, max(2, N // 8), ofam.MFGaussian(D), prior, use_resampling=False)
        v, g = obj(theta)
        ov, og = ref(theta, noise=omvt.draw_noise(np.random.RandomState(6), N))
        assert abs(obj._eps - ref._eps) <= 1e-10 and abs(v - ov) <= 1e-9 * abs(ov), (obj._eps, ref._eps, v, ov)
        np.testing.assert_allclose(g, og, rtol=0, atol=1e-8 * np.max(np.abs(og)))
    finally:
        for k, f in real.items():
            setattr(eng, k, f)
    assert used.count('elbo_grad_mvt_symroot') == 2 and 'alpha_grad_mvt_symroot' in used and 'dis_refresh_mvt_symroot' in used, used


def test_reference_mode_falls_back_when_the_device_root_does_not_resolve(vb):
    """A scale matrix beyond the Newton-Schulz iteration's reach (condition number ~1e10: the accuracy check
    ||R R - Sigma|| / ||Sigma||_inf < 1e-12 fails): vb_elbo_grad_mvt_symroot / vb_dis_refresh_mvt_symroot decline and the
    objectives take the LAPACK route -- same call, same results as the oracle's eigen-decomposition to the accuracy such
    a matrix allows."""
    from viabel_amd import _lib
    D, N, df = 192, 4096, 9.0           # (N at the threshold from which the chi-square draws are generated on the device)
    rng = np.random.RandomState(3)
    Q, _ = np.linalg.qr(rng.randn(D, D))
    Sigma = (Q * np.logspace(-7, 3, D)) @ Q.T
    Sigma = 0.5 * (Sigma + Sigma.T)
    theta = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(Sigma)])
    mean, sd = 0.3 * rng.randn(D), np.exp(0.2 * rng.randn(D))
    model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    eng = _lib.default_engine()
    approx = vb.MultivariateT(D, df, seed=8)
    value, grad = vb.ExclusiveKL(approx, model, N)(theta)
    # the resident entry point itself: declines (None), nothing raised
    approx._stage_base_noise(eng, 9, N, 0, N)
    eng.set_model(model.device_spec())
    assert eng.elbo_grad_mvt_symroot(9, N, D, df, theta) is None
    noise = ofam.MultivariateT(D, df).draw_noise(np.random.RandomState(8), N)
    ov, og = oobj.exclusive_kl(ofam.MultivariateT(D, df), omodel, theta, noise, False)
    assert abs(value - ov) <= 1e-7 * abs(ov), (value, ov)
    assert np.max(np.abs(grad - og)) <= 1e-5 * np.max(np.abs(og))
