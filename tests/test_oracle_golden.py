"""The numpy oracle against the golden vectors produced from the reference's own code.

Fixtures come from ``tests/golden/make_golden.py`` (reference forward code + FD /
analytic derivatives, see its header).  Tolerances: values 1e-12 relative (same fp64
arithmetic, different summation order), gradients 1e-12 against the stored analytic
gradient and 2e-7 against the reference finite differences.
"""
import os

import numpy as np
import pytest

from oracle import families as ofam
from oracle import objectives as oobj

import _golden as G


@pytest.mark.parametrize('path', G.fixtures('family_'), ids=lambda p: p.split('/')[-1][:-4])
def test_family_forward(path):
    fx = G.load(path)
    fam = G.oracle_family(fx)
    noise = fam.draw_noise(np.random.RandomState(int(fx['seed'])), int(fx['n']))
    if isinstance(noise, tuple):
        np.testing.assert_array_equal(noise[0], fx['noise_chi'])
        np.testing.assert_array_equal(noise[1], fx['noise_z'])
    else:
        np.testing.assert_array_equal(noise, fx['noise'])
    th0, th1 = fx['theta0'], fx['theta1']
    assert G.rel_err(fam.sample_from_noise(th0, noise), fx['samples']) < 1e-12
    assert G.rel_err(fam.log_density(th1, fx['samples']), fx['log_density']) < 1e-12
    assert G.rel_err(fam.entropy(th0), fx['entropy']) < 1e-12
    assert G.rel_err(fam.init_param(), fx['init_param']) < 1e-15
    mean, cov = fam.mean_and_cov(th0)
    assert G.rel_err(mean, fx['mean']) < 1e-15 and G.rel_err(cov, fx['cov']) < 1e-12
    assert G.rel_err(fam.pth_moment(th0, 2), fx['pth2']) < 1e-12
    assert G.rel_err(fam.pth_moment(th0, 4), fx['pth4']) < 1e-12
    if 'kl' in fx:
        assert G.rel_err(fam.kl(th0, th1), fx['kl']) < 1e-12


@pytest.mark.parametrize('path', G.fixtures('ekl_'), ids=lambda p: p.split('/')[-1][:-4])
def test_exclusive_kl(path):
    fx = G.load(path)
    v, g = oobj.exclusive_kl(G.oracle_family(fx), G.oracle_model(fx), fx['theta'], G.noise_of(fx),
                             use_path_deriv=bool(fx['use_path_deriv']))
    assert G.rel_err(v, fx['value']) < 1e-12
    assert G.rel_err(g, fx['grad']) < 1e-12
    # the MultivariateT reference closure differentiates scipy's sqrtm: its finite differences are noisier
    assert G.rel_err(g, fx['grad_fd']) < (2e-6 if str(fx['family_kind']) == 'multivariate_t' else 2e-7)


@pytest.mark.parametrize('path', G.fixtures('rge_'), ids=lambda p: p.split('/')[-1][:-4])
def test_rge(path):
    fx = G.load(path)
    fam, model = G.oracle_family(fx), G.oracle_model(fx)
    args = (fam, model, fx['theta'], fx['noise'], str(fx['method']), bool(fx['use_path_deriv']))
    lv, lg = oobj.rge_literal(*args)
    rv, rg = oobj.rge_reduced(*args)
    assert G.rel_err(lv, fx['value']) < 1e-12 and G.rel_err(lg, fx['grad']) < 1e-12
    assert G.rel_err(rv, fx['value']) < 1e-12 and G.rel_err(rg, fx['grad']) < 1e-10


@pytest.mark.parametrize('path', G.fixtures('alpha_'), ids=lambda p: p.split('/')[-1][:-4])
def test_alpha(path):
    fx = G.load(path)
    v, g = oobj.alpha_divergence(G.oracle_family(fx), G.oracle_model(fx), fx['theta'],
                                 G.noise_of(fx), float(fx['alpha']))
    assert G.rel_err(v, fx['value']) < 1e-12
    assert G.rel_err(g, fx['grad']) < 1e-12
    assert G.rel_err(g, fx['grad_fd']) < 2e-7


@pytest.mark.parametrize('path', G.fixtures('dis_'), ids=lambda p: p.split('/')[-1][:-4])
def test_dis(path):
    fx = G.load(path)
    fam, model = G.oracle_family(fx), G.oracle_model(fx)
    D = int(fx['dim'])
    dis = oobj.DISInclusiveKL(fam, model, int(fx['n']), int(fx['ess_target']), ofam.MFGaussian(D),
                              fx['prior_params'], use_resampling=bool(fx['use_resampling']))
    v, g = dis(fx['theta'], noise=G.noise_of(fx), indices=fx.get('indices'))
    assert G.rel_err(dis._state_samples, fx['samples']) < 1e-12
    assert G.rel_err(dis._state_log_q, fx['log_q']) < 1e-11
    assert G.rel_err(dis._state_log_p, fx['log_p']) < 1e-12
    assert G.rel_err(dis._eps, fx['eps']) < 1e-12
    assert G.rel_err(dis._state_w_clipped, fx['w_clipped']) < 1e-10
    assert G.rel_err(v, fx['value']) < 1e-11
    assert G.rel_err(g, fx['grad']) < 1e-11
    assert G.rel_err(g, fx['grad_fd']) < 2e-6


@pytest.mark.parametrize('path', G.fixtures('disprior_'), ids=G.ids(G.fixtures('disprior_')))
def test_dis_general_tempering_prior(path):
    """objectives.py:283-285 / :317-319: the tempering prior may be any family (fixtures: the reference's MFStudentT,
    MultivariateT and LRGaussian as priors)."""
    fx = G.load(path)
    fam, model = G.oracle_family(fx), G.oracle_model(fx)
    dis = oobj.DISInclusiveKL(fam, model, int(fx['n']), int(fx['ess_target']), G.oracle_prior_family(fx),
                              fx['prior_params'], use_resampling=bool(fx['use_resampling']))
    v, g = dis(fx['theta'], noise=G.noise_of(fx), indices=fx.get('indices'))
    assert G.rel_err(dis._state_samples, fx['samples']) < 1e-12
    assert G.rel_err(dis._eps, fx['eps']) < 1e-12
    assert G.rel_err(dis._state_w_clipped, fx['w_clipped']) < 1e-10
    assert G.rel_err(v, fx['value']) < 1e-11
    assert G.rel_err(g, fx['grad']) < 1e-11
    assert G.rel_err(g, fx['grad_fd']) < 2e-6


def test_clip_is_the_fixed_point_of_the_reference_recursion():
    """objectives.py:370-386 with :385's evident intent: where the literal recursion terminates the fixed-point
    restatement returns the same numbers; where it does not (rounding re-triggers it) the restatement still does, and
    every clipped weight sits at threshold * sum."""
    def literal(w, thr, depth=0):
        S = np.sum(w)
        if not np.any(w > S * thr):
            return w
        to_clip = (w >= S * thr)
        n_to_clip = np.sum(to_clip)
        sum_unclipped = np.sum(w[~to_clip])
        if sum_unclipped == 0:
            return w
        w = w.copy()
        w[to_clip] = thr * sum_unclipped / (1. - thr * n_to_clip)
        if depth > 100:
            raise RecursionError
        return literal(w, thr, depth + 1)

    terminated = 0
    for seed in range(120):
        rng = np.random.RandomState(seed)
        N = int(rng.choice([64, 500, 4096]))
        w = np.exp(rng.randn(N) * rng.choice([1, 2, 4]))
        for thr in (0.05, 0.2):
            dis = oobj.DISInclusiveKL.__new__(oobj.DISInclusiveKL)
            dis._w_clip_threshold = thr
            out = dis._clip(w)
            assert np.all(out <= thr * out.sum() * (1 + 1e-12))
            try:
                np.testing.assert_array_equal(out, literal(w, thr))
                terminated += 1
            except RecursionError:
                pass
    assert terminated > 150


def test_fullrank_reduces_to_meanfield():
    """FullRankGaussian has no reference class: pin it by reduction to MFGaussian."""
    from oracle import models as omod
    rng = np.random.RandomState(3)
    D, N = 6, 40
    mf, fr = ofam.MFGaussian(D), ofam.FullRankGaussian(D)
    mu, ls = rng.randn(D), 0.3 * rng.randn(D)
    th_mf = np.concatenate([mu, ls])
    th_fr = fr.pack(mu, np.diag(np.exp(ls)))
    eps = rng.randn(N, D)
    diag_pos = np.cumsum(np.arange(1, D + 1)) - 1
    for model in (omod.GaussDiag(rng.randn(D), np.exp(rng.randn(D))), omod.Funnel(D)):
        for pd in (False, True):
            v0, g0 = oobj.exclusive_kl(mf, model, th_mf, eps, pd)
            v1, g1 = oobj.exclusive_kl(fr, model, th_fr, eps, pd)
            assert abs(v0 - v1) < 1e-12 * max(1, abs(v0))
            np.testing.assert_allclose(g1[:D], g0[:D], rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(g1[D:][diag_pos], g0[D:], rtol=1e-12, atol=1e-13)
    assert abs(fr.entropy(th_fr) - mf.entropy(th_mf)) < 1e-12
    x = rng.randn(5, D)
    np.testing.assert_allclose(fr.log_density(th_fr, x), mf.log_density(th_mf, x), rtol=1e-12)
    assert abs(fr.kl(th_fr, fr.pack(mu + 1, np.diag(np.exp(ls + .2))))
               - mf.kl(th_mf, np.concatenate([mu + 1, ls + .2]))) < 1e-12


def test_fullrank_gradient_torch_fp64():
    """Analytic full-rank ELBO gradient against torch.autograd (fp64)."""
    import torch
    from oracle import models as omod
    rng = np.random.RandomState(4)
    D, N = 5, 30
    fr = ofam.FullRankGaussian(D)
    A = rng.randn(D, D)
    P = A @ A.T / D + np.eye(D)
    model = omod.GaussFull(rng.randn(D), P)
    L = np.tril(0.2 * rng.randn(D, D)) + np.diag(np.exp(0.2 * rng.randn(D)))
    theta = fr.pack(rng.randn(D), L)
    eps = rng.randn(N, D)
    tril = np.tril_indices(D)
    for pd in (False, True):
        v, g = oobj.exclusive_kl(fr, model, theta, eps, pd)
        th = torch.tensor(theta, dtype=torch.float64, requires_grad=True)

        def build(t):
            Lt = torch.zeros(D, D, dtype=torch.float64)
            Lt[tril[0], tril[1]] = t[D:]
            d = torch.diagonal(Lt)
            return t[:D], Lt - torch.diag(d) + torch.diag(torch.exp(d))
        mu_t, L_t = build(th)
        e_t = torch.tensor(eps)
        z = mu_t + e_t @ L_t.T
        dz = z - torch.tensor(model.mean)
        f = -0.5 * ((dz @ torch.tensor(model.P)) * dz).sum(1) + model.const
        if pd:
            mu_s, L_s = build(th.detach())
            ee = torch.linalg.solve_triangular(L_s, (z - mu_s).T, upper=False).T
            logq = -0.5 * (ee * ee).sum(1) - torch.log(torch.diagonal(L_s)).sum() \
                - 0.5 * D * np.log(2 * np.pi)
            val = -(f - logq).mean()
        else:
            ent = 0.5 * D * (1 + np.log(2 * np.pi)) + torch.log(torch.diagonal(L_t)).sum()
            val = -(f.mean() + ent)
        gt, = torch.autograd.grad(val, th)
        assert abs(val.item() - v) < 1e-12 * max(1, abs(v))
        np.testing.assert_allclose(g, gt.numpy(), rtol=1e-11, atol=1e-12)


@pytest.mark.skipif(not os.path.isdir('/root/reference/viabel'), reason='the reference tree exists in the build container only')
def test_committed_fixtures_match_a_fresh_run_of_the_reference():
    """Drift guard: `make_golden.py --check` regenerates every fixture from the reference's own code into a scratch
    directory and compares it with the committed file (same keys, every array bit for bit)."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'make_golden.py')
    res = subprocess.run([sys.executable, script, '--check'], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert 'no drift' in res.stdout
