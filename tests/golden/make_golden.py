#!/usr/bin/env python3
"""Generate golden vectors for the BBVI hot path from the upstream reference itself.

Runs ONLY in the build container (needs the read-only reference tree); writes
``tests/golden/*.npz``.  The reference's Python is imported through the
container-only shims of ``_ref_stubs.py`` (autograd/paragami/pystan are absent
here); see that file for exactly what is real reference code (all forward
arithmetic, the RGE control-variate algebra, DIS tempering/bisection,
alpha-divergence weights) and what is supplied (finite-difference / analytic
derivatives in place of autograd).

Each fixture stores inputs (theta, seed, the noise the reference drew) and the
reference's outputs (samples, log densities, objective value, gradient), plus a
``provenance`` string.  ``tests/test_oracle_golden.py`` checks the numpy oracle
against them; the GPU parity tests check the HIP engine against the same files.

Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import scipy.stats

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _ref_stubs  # noqa: E402

_ref_stubs.install('/root/reference')

from viabel import approximations as ref_approx  # noqa: E402  (the reference)
from viabel import objectives as ref_obj         # noqa: E402

from oracle import families as ofam              # noqa: E402
from oracle import models as omod                # noqa: E402
from oracle import objectives as oobj            # noqa: E402

norm = scipy.stats.norm


# ----------------------------------------------------------------------------
# models: the reference-side callable is written the way the reference's own
# tests / docs write it (scipy logpdf calls), independent of the oracle formula
# ----------------------------------------------------------------------------
def make_model(spec):
    kind = spec['kind']
    if kind == 'gauss_diag':
        mean = np.asarray(spec['mean'], dtype=float)[np.newaxis, :]
        stdev = np.asarray(spec['stdev'], dtype=float)[np.newaxis, :]

        def log_p(x):      # viabel/tests/test_objectives.py:18-19
            x = np.atleast_2d(x)
            return np.sum(norm.logpdf(x, loc=mean, scale=stdev), axis=1)
        return log_p, omod.GaussDiag(spec['mean'], spec['stdev'])
    if kind == 'funnel':
        D, k, tau = spec['dim'], spec['scale_index'], spec['log_sigma_stdev']

        def log_p(x):      # docs/source/quickstart.ipynb:23-29 generalised to D dims
            x = np.atleast_2d(x)
            log_sigma = x[:, k]
            out = norm.logpdf(log_sigma, 0, tau)
            for d in range(D):
                if d != k:
                    out = out + norm.logpdf(x[:, d], 0, np.exp(log_sigma))
            return out
        return log_p, omod.Funnel(D, k, tau)
    raise ValueError(kind)


def make_family(spec, seed):
    kind, D = spec['kind'], spec['dim']
    if kind == 'mf_gaussian':
        return ref_approx.MFGaussian(D, seed=seed), ofam.MFGaussian(D)
    if kind == 'mf_student_t':
        return ref_approx.MFStudentT(D, spec['df'], seed=seed), ofam.MFStudentT(D, spec['df'])
    if kind == 'multivariate_t':
        return ref_approx.MultivariateT(D, spec['df'], seed=seed), ofam.MultivariateT(D, spec['df'])
    if kind == 'lr_gaussian':
        return ref_approx.LRGaussian(D, seed=seed, k=spec['k']), ofam.LRGaussian(D, spec['k'])
    raise ValueError(kind)


def rel_err(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def snapshot_hook(approx, objective=None, np_seed=None):
    rs_state = approx._rs.get_state()
    obj_state = None
    if objective is not None:
        obj_state = {k: (v.copy() if isinstance(v, np.ndarray) else v)
                     for k, v in objective.__dict__.items()
                     if k.startswith('_state') or k in ('_eps', '_objective_step')}

    def hook():
        approx._rs.set_state(rs_state)
        if np_seed is not None:
            np.random.seed(np_seed)
        if obj_state is not None:
            for k in [k for k in objective.__dict__ if k.startswith('_state')]:
                del objective.__dict__[k]
            objective.__dict__.update(
                {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in obj_state.items()})
    return hook


def theta_for(fspec, rng, kind='random'):
    D = fspec['dim']
    if fspec['kind'] in ('mf_gaussian', 'mf_student_t'):
        return np.concatenate([0.3 * rng.randn(D), -0.5 + 0.3 * rng.randn(D)])
    if fspec['kind'] == 'lr_gaussian':
        return np.concatenate([0.3 * rng.randn(D), -0.5 + 0.3 * rng.randn(D), 0.4 * rng.randn(D * fspec['k'])])
    A = rng.randn(D, D)
    S = A @ A.T / D + 0.5 * np.eye(D)
    return np.concatenate([0.3 * rng.randn(D), ofam.psd_to_free(S)])


SAVED = []


OUT_DIR = HERE          # --check writes into a scratch directory instead and compares with the committed files


def save(name, **arrs):
    path = os.path.join(OUT_DIR, name + '.npz')
    np.savez(path, **arrs)
    SAVED.append(name)


def compare_with_committed(fresh_dir, names):
    """Every regenerated fixture against the committed file of the same name: same keys, every array equal bit for bit
    (strings compared as strings).  Returns a list of human-readable differences (empty = no drift)."""
    problems = []
    for name in names:
        old_path = os.path.join(HERE, name + '.npz')
        if not os.path.exists(old_path):
            problems.append('%s: not committed' % name)
            continue
        a, b = np.load(os.path.join(fresh_dir, name + '.npz'), allow_pickle=False), np.load(old_path, allow_pickle=False)
        if set(a.files) != set(b.files):
            problems.append('%s: keys differ (regenerated only: %s; committed only: %s)'
                            % (name, sorted(set(a.files) - set(b.files)), sorted(set(b.files) - set(a.files))))
            continue
        for k in a.files:
            x, y = a[k], b[k]
            same = x.shape == y.shape and (np.array_equal(x, y, equal_nan=True) if x.dtype.kind in 'fc'
                                            else np.array_equal(x, y))
            if not same:
                problems.append('%s[%s]: values differ' % (name, k))
    committed = {f[:-4] for f in os.listdir(HERE) if f.endswith('.npz')}
    for extra in sorted(committed - set(names)):
        problems.append('%s: committed but no generator writes it' % extra)
    return problems


def spec_arrays(fspec, mspec):
    out = {'family_kind': fspec['kind'], 'dim': fspec['dim'], 'df': fspec.get('df', 0.0),
           'rank': fspec.get('k', 0), 'model_kind': mspec['kind']}
    if mspec['kind'] == 'gauss_diag':
        out['model_mean'] = np.asarray(mspec['mean'], dtype=float)
        out['model_stdev'] = np.asarray(mspec['stdev'], dtype=float)
    else:
        out['model_scale_index'] = mspec['scale_index']
        out['model_log_sigma_stdev'] = mspec['log_sigma_stdev']
    return out


def model_specs(D, rng):
    return [
        {'kind': 'gauss_diag', 'mean': rng.randn(D), 'stdev': np.exp(0.5 * rng.randn(D))},
        {'kind': 'funnel', 'dim': D, 'scale_index': D - 1, 'log_sigma_stdev': 1.0},
    ]


# ----------------------------------------------------------------------------
def gen_family_forward():
    """sample / log_density / entropy / kl / mean_and_cov / pth_moment of the reference."""
    rng = np.random.RandomState(11)
    for fspec in ({'kind': 'mf_gaussian', 'dim': 3}, {'kind': 'mf_gaussian', 'dim': 10},
                  {'kind': 'mf_student_t', 'dim': 3, 'df': 8},
                  {'kind': 'mf_student_t', 'dim': 4, 'df': 100},
                  {'kind': 'multivariate_t', 'dim': 3, 'df': 100},
                  {'kind': 'multivariate_t', 'dim': 5, 'df': 7}):
        seed, N = 5, 12
        ref, orc = make_family(fspec, seed)
        th0, th1 = theta_for(fspec, rng), theta_for(fspec, rng)
        x = ref.sample(th0, N)
        noise = orc.draw_noise(np.random.RandomState(seed), N)
        xo = orc.sample_from_noise(th0, noise)
        lq = ref.log_density(th1, x)
        ent = ref.entropy(th0)
        mean, cov = ref.mean_and_cov(th0)
        out = dict(spec_arrays(fspec, {'kind': 'gauss_diag', 'mean': [0], 'stdev': [1]}),
                   seed=seed, n=N, theta0=th0, theta1=th1, init_param=ref.init_param(),
                   samples=x, log_density=lq, entropy=ent, mean=mean, cov=cov,
                   pth2=ref.pth_moment(th0, 2), pth4=ref.pth_moment(th0, 4),
                   provenance='reference forward code via autograd->numpy alias')
        if ref.supports_kl:
            out['kl'] = ref.kl(th0, th1)
            assert rel_err(orc.kl(th0, th1), out['kl']) < 1e-13
        if fspec['kind'] == 'multivariate_t':
            out['noise_chi'], out['noise_z'] = noise
        else:
            out['noise'] = noise
        assert rel_err(xo, x) < 1e-12, (fspec, rel_err(xo, x))
        assert rel_err(orc.log_density(th1, x), lq) < 1e-12
        assert rel_err(orc.entropy(th0), ent) < 1e-12
        assert rel_err(orc.init_param(), ref.init_param()) < 1e-15
        assert rel_err(orc.pth_moment(th0, 2), out['pth2']) < 1e-12
        assert rel_err(orc.pth_moment(th0, 4), out['pth4']) < 1e-12
        assert rel_err(orc.mean_and_cov(th0)[1], cov) < 1e-12
        save('family_%s_d%d_df%s' % (fspec['kind'], fspec['dim'], fspec.get('df', 0)), **out)


def gen_exclusive_kl():
    rng = np.random.RandomState(21)
    worst = 0.0
    for fspec in ({'kind': 'mf_gaussian', 'dim': 2}, {'kind': 'mf_gaussian', 'dim': 10},
                  {'kind': 'mf_gaussian', 'dim': 16}, {'kind': 'mf_student_t', 'dim': 3, 'df': 8},
                  {'kind': 'mf_student_t', 'dim': 2, 'df': 100}):
        D = fspec['dim']
        for mspec in model_specs(D, rng):
            for pd in (False, True):
                for N in (8, 100):
                    seed = 1
                    ref, orc = make_family(fspec, seed)
                    log_p, omodel = make_model(mspec)
                    theta = theta_for(fspec, rng)
                    objective = ref_obj.ExclusiveKL(ref, log_p, N, use_path_deriv=pd)
                    _ref_stubs.STATE['before_eval'] = snapshot_hook(ref)
                    value, grad_fd = objective(theta)
                    _ref_stubs.STATE['before_eval'] = None
                    noise = orc.draw_noise(np.random.RandomState(seed), N)
                    ov, og = oobj.exclusive_kl(orc, omodel, theta, noise, use_path_deriv=pd)
                    assert rel_err(ov, value) < 1e-12, (fspec, mspec['kind'], pd, ov, value)
                    e = rel_err(og, grad_fd)
                    worst = max(worst, e)
                    assert e < 2e-7, (fspec, mspec['kind'], pd, N, e)
                    name = 'ekl_%s_d%d_%s_pd%d_n%d' % (fspec['kind'], D, mspec['kind'], pd, N)
                    save(name, **spec_arrays(fspec, mspec), seed=seed, n=N, theta=theta,
                         noise=noise, use_path_deriv=pd, value=value, grad_fd=grad_fd,
                         grad=og,
                         provenance='value: reference closure; grad_fd: Richardson central '
                                    'differences of the reference closure (getval replayed); '
                                    'grad: analytic (oracle), agrees with grad_fd')
    print('ExclusiveKL plain: worst analytic-vs-FD(reference) grad rel err %.2e' % worst)


def gen_rge():
    rng = np.random.RandomState(31)
    worst = 0.0
    for fspec in ({'kind': 'mf_gaussian', 'dim': 3}, {'kind': 'mf_student_t', 'dim': 2, 'df': 100},
                  {'kind': 'mf_student_t', 'dim': 4, 'df': 8}, {'kind': 'mf_gaussian', 'dim': 10}):
        D = fspec['dim']
        for mspec in model_specs(D, rng):
            for method in ('full', 'mean_only', 'loo_diag_approx', 'loo_direct_approx'):
                for pd in (False, True):
                    seed, N = 1, 24
                    ref, orc = make_family(fspec, seed)
                    log_p, omodel = make_model(mspec)
                    theta = theta_for(fspec, rng)
                    _ref_stubs.STATE['model'] = omodel
                    objective = ref_obj.ExclusiveKL(ref, log_p, N, use_path_deriv=pd,
                                                    hessian_approx_method=method)
                    value, grad = objective(theta)     # reference RGE code, literally
                    _ref_stubs.STATE['model'] = None
                    noise = orc.draw_noise(np.random.RandomState(seed), N)
                    lv, lg = oobj.rge_literal(orc, omodel, theta, noise, method, pd)
                    rv, rg = oobj.rge_reduced(orc, omodel, theta, noise, method, pd)
                    assert rel_err(lv, value) < 1e-12 and rel_err(lg, grad) < 1e-12
                    e = rel_err(rg, grad)
                    worst = max(worst, e)
                    assert rel_err(rv, value) < 1e-12 and e < 1e-10, (fspec, mspec['kind'], method, e)
                    name = 'rge_%s_d%d_%s_%s_pd%d' % (fspec['kind'], D, mspec['kind'], method, pd)
                    save(name, **spec_arrays(fspec, mspec), seed=seed, n=N, theta=theta,
                         noise=noise, use_path_deriv=pd, method=method, value=value, grad=grad,
                         provenance='reference RGE code (objectives.py:170-271) run literally; '
                                    'model derivatives analytic')
    print('RGE: worst reduced-vs-reference grad rel err %.2e' % worst)


def gen_alpha():
    rng = np.random.RandomState(41)
    worst = 0.0
    for fspec in ({'kind': 'mf_gaussian', 'dim': 2}, {'kind': 'mf_student_t', 'dim': 3, 'df': 100},
                  {'kind': 'mf_gaussian', 'dim': 10}, {'kind': 'multivariate_t', 'dim': 3, 'df': 100},
                  {'kind': 'multivariate_t', 'dim': 4, 'df': 7}):
        D = fspec['dim']
        for mspec in model_specs(D, rng):
            for alpha in (2.0, 0.5):
                N, np_seed = 32, 851
                ref, orc = make_family(fspec, 1)
                log_p, omodel = make_model(mspec)
                theta = theta_for(fspec, rng)
                objective = ref_obj.AlphaDivergence(ref, log_p, N, alpha)
                np.random.seed(np_seed)
                value, grad_fd = objective(theta)
                np.random.seed(np_seed)
                seed = np.random.randint(2 ** 32)                      # objectives.py:455
                noise = orc.draw_noise(np.random.RandomState(seed), N)
                ov, og = oobj.alpha_divergence(orc, omodel, theta, noise, alpha)
                assert rel_err(ov, value) < 1e-12
                e = rel_err(og, grad_fd)
                worst = max(worst, e)
                assert e < 2e-7, (fspec, mspec['kind'], alpha, e)
                name = 'alpha_%s_d%d_%s_a%g' % (fspec['kind'], D, mspec['kind'], alpha)
                if fspec['kind'] == 'multivariate_t':
                    noise_kw = dict(noise_chi=noise[0], noise_z=noise[1])
                else:
                    noise_kw = dict(noise=noise)
                save(name, **spec_arrays(fspec, mspec), np_seed=np_seed, seed=seed, n=N,
                     theta=theta, alpha=alpha, value=value, grad_fd=grad_fd, grad=og, **noise_kw,
                     provenance='value: reference; grad_fd: FD of the reference log-weights '
                                'closure contracted as in objectives.py:460; grad: analytic')
    print('AlphaDivergence: worst analytic-vs-FD(reference) grad rel err %.2e' % worst)


def gen_dis():
    rng = np.random.RandomState(51)
    worst = 0.0
    for fspec in ({'kind': 'mf_student_t', 'dim': 2, 'df': 100}, {'kind': 'mf_gaussian', 'dim': 3},
                  {'kind': 'multivariate_t', 'dim': 3, 'df': 100},
                  {'kind': 'multivariate_t', 'dim': 4, 'df': 7}):
        D = fspec['dim']
        mspec = model_specs(D, rng)[0]
        for use_resampling in (True, False):
            N, ess_target, np_seed = 64, 20, 851
            ref, orc = make_family(fspec, 1)
            log_p, omodel = make_model(mspec)
            theta = theta_for(fspec, rng)
            prior_params = np.concatenate([[0] * D, [1] * D]).astype(float)   # test_objectives.py:85
            objective = ref_obj.DISInclusiveKL(
                ref, log_p, N, ess_target=ess_target, temper_prior=ref_approx.MFGaussian(D),
                temper_prior_params=prior_params, use_resampling=use_resampling)
            chosen = []
            real_choice = np.random.choice

            def rec_choice(*a, **k):
                idx = real_choice(*a, **k)
                chosen.append(np.array(idx))
                return idx
            np.random.choice = rec_choice
            try:
                np.random.seed(np_seed)
                _ref_stubs.STATE['before_eval'] = snapshot_hook(ref, objective, np_seed)
                value, grad_fd = objective(theta)
            finally:
                np.random.choice = real_choice
                _ref_stubs.STATE['before_eval'] = None
            noise = orc.draw_noise(np.random.RandomState(1), N)
            od = oobj.DISInclusiveKL(orc, omodel, N, ess_target, ofam.MFGaussian(D), prior_params,
                                     use_resampling=use_resampling)
            indices = chosen[0] if use_resampling else None
            ov, og = od(theta, noise=noise, indices=indices)
            assert rel_err(ov, value) < 1e-11, (fspec, use_resampling, ov, value)
            assert rel_err(od._eps, objective._eps) < 1e-12
            assert rel_err(od._state_w_clipped, objective._state_w_clipped) < 1e-10
            e = rel_err(og, grad_fd)
            worst = max(worst, e)
            assert e < 2e-6, (fspec, use_resampling, e)
            out = dict(spec_arrays(fspec, mspec), np_seed=np_seed, seed=1, n=N,
                       ess_target=ess_target, use_resampling=use_resampling, theta=theta,
                       prior_params=prior_params, value=value, grad_fd=grad_fd, grad=og,
                       eps=objective._eps, w_clipped=objective._state_w_clipped,
                       log_q=objective._state_log_q, log_p=objective._state_log_p_unnormalized,
                       samples=objective._state_samples,
                       provenance='reference DISInclusiveKL code; grad_fd by FD with getval replay')
            if use_resampling:
                out['indices'] = indices
            if fspec['kind'] == 'multivariate_t':
                out['noise_chi'], out['noise_z'] = noise
            else:
                out['noise'] = noise
            save('dis_%s_d%d_rs%d' % (fspec['kind'], D, use_resampling), **out)
    print('DISInclusiveKL: worst analytic-vs-FD(reference) grad rel err %.2e' % worst)


def gen_torch_crosscheck():
    """Independent check of the analytic gradients with torch.autograd (fp64)."""
    import torch
    torch.set_default_dtype(torch.float64)
    rng = np.random.RandomState(61)
    worst = 0.0
    # model derivatives
    for D in (3, 7):
        for mspec in model_specs(D, rng):
            _, om = make_model(mspec)
            x = rng.randn(5, D)
            xt = torch.tensor(x, requires_grad=True)
            if mspec['kind'] == 'gauss_diag':
                f = torch.distributions.Normal(torch.tensor(om.mean), torch.tensor(om.stdev)) \
                    .log_prob(xt).sum(1)
            else:
                v = xt[:, om.k]
                f = torch.distributions.Normal(0., om.tau).log_prob(v)
                for d in range(D):
                    if d != om.k:
                        f = f + torch.distributions.Normal(0., torch.exp(v)).log_prob(xt[:, d])
            g, = torch.autograd.grad(f.sum(), xt, create_graph=True)
            assert rel_err(om.logp(x), f.detach().numpy()) < 1e-13
            assert rel_err(om.grad(x), g.detach().numpy()) < 1e-12
            m = torch.tensor(x[0], requires_grad=True)

            def fm(mm):
                if mspec['kind'] == 'gauss_diag':
                    return torch.distributions.Normal(torch.tensor(om.mean), torch.tensor(om.stdev)) \
                        .log_prob(mm).sum()
                v = mm[om.k]
                out = torch.distributions.Normal(0., om.tau).log_prob(v)
                for d in range(D):
                    if d != om.k:
                        out = out + torch.distributions.Normal(0., torch.exp(v)).log_prob(mm[d])
                return out
            H = torch.autograd.functional.hessian(fm, m).numpy()
            e = rel_err(om.hessian(x[0]), H)
            worst = max(worst, e)
            assert e < 1e-12
    print('model derivatives vs torch.autograd fp64: worst hessian rel err %.2e' % worst)


def gen_chain_stats():
    """ess / MCSE / R-hat of the reference (viabel/_mc_diagnostics.py, pure numpy) on seeded chains."""
    from viabel import _mc_diagnostics as ref
    rng = np.random.RandomState(71)
    chains, ess_ref, mcse_ref, rhat_ref = [], [], [], []
    for n, phi in ((200, 0.0), (200, 0.9), (401, 0.5), (57, -0.4), (1000, 0.99), (16, 0.3), (9, 0.0)):
        x = np.zeros((n, 3))
        e = rng.randn(n, 3)
        for t in range(1, n):
            x[t] = phi * x[t - 1] + e[t]
        x[:, 2] += np.linspace(0, 3, n)          # a drifting coordinate
        chains.append(x)
        ess_ref.append(np.array([ref.ess(x[:, i].reshape(1, n)) for i in range(3)]))
        mcse_ref.append(ref.MCSE(x)[1])
        rhat_ref.append(ref.compute_R_hat(x))
    windows = np.array([50, 100, 150, 200])
    ok, best = ref.R_hat_convergence_check(list(chains[1]), windows)
    save('chainstats', n_chains=len(chains), windows=windows, rhat_ok=ok, rhat_best=best,
         **{'chain%d' % i: c for i, c in enumerate(chains)},
         **{'ess%d' % i: c for i, c in enumerate(ess_ref)},
         **{'mcse%d' % i: c for i, c in enumerate(mcse_ref)},
         **{'rhat%d' % i: c for i, c in enumerate(rhat_ref)},
         provenance='reference viabel/_mc_diagnostics.py functions (pure numpy) run as they are')


def gen_optimizers():
    """Iterate histories of the reference optimisers (viabel/optimization.py:51-518) on a seeded noisy
    quadratic -- the same dummy objective tests/test_host_logic.py builds (modelled on the reference's
    tests/test_optimization.py:20-32)."""
    from viabel import optimization as ref_opt

    class Family:
        supports_kl = True

    class Objective:
        def __init__(self, target, noise=0.3, seed=3):
            self.target, self.noise = target, noise
            self.rs = np.random.RandomState(seed)
            self.approx = Family()

        def __call__(self, x):
            g = (x - self.target) + self.noise * self.rs.randn(*x.shape)
            return 0.5 * np.sum((x - self.target) ** 2), g

        def update(self, x, d):
            return x - d

    target = np.array([1.0, -2.0, 0.5, 3.0])
    ctors = {
        'sgd': lambda: ref_opt.StochasticGradientOptimizer(0.05, diagnostics=True),
        'rmsprop': lambda: ref_opt.RMSProp(0.05, diagnostics=True),
        'avgrmsprop': lambda: ref_opt.AveragedRMSProp(0.05, diagnostics=True),
        'adam': lambda: ref_opt.Adam(0.05, diagnostics=True),
        'avgadam': lambda: ref_opt.AveragedAdam(0.05, diagnostics=True),
        'adagrad': lambda: ref_opt.Adagrad(0.5, diagnostics=True),
        'wadagrad': lambda: ref_opt.WindowedAdagrad(0.05, diagnostics=True),
    }
    out = {}
    for name, ctor in ctors.items():
        res = ctor().optimize(300, Objective(target), np.zeros(4))
        out[name + '_opt_param'] = np.asarray(res['opt_param'])
        out[name + '_last'] = np.asarray(res['variational_param_history'][-1])
        out[name + '_values'] = np.asarray(res['value_history'])
    save('optimizers', target=target, provenance='reference viabel/optimization.py optimisers on a seeded noisy quadratic',
         **out)


def gen_psis():
    """psislw / gpdfitnew of the reference (viabel/_psis.py, pure numpy) and the bounds of
    viabel/diagnostics.py on seeded log-weight vectors: heavy, moderate, light and bounded tails, ties,
    tiny n, a 2-column input and a non-default Reff."""
    from viabel import _psis as ref
    from viabel import diagnostics as ref_diag
    rng = np.random.RandomState(1234)
    cases = {
        'normal_heavy': 2.5 * rng.randn(2000),                    # log-normal weights, k-hat > 1/3
        'normal_moderate': 1.0 * rng.randn(4000),
        'normal_light': 0.2 * rng.randn(1500),                    # k < 1/3: no smoothing
        'bounded': -rng.gamma(2.0, 1.0, size=3000),               # weights bounded above
        'student': 1.5 * rng.standard_t(3, size=16384),           # the C3 sample count
        'ties': np.round(1.5 * rng.randn(800), 1),
        'tiny': rng.randn(12),
        'five': rng.randn(5),
        'big': 2.0 * rng.randn(100000),                           # vi_diagnostics' default n_samples
    }
    out = {'names': np.array(sorted(cases))}
    for name, lw in cases.items():
        sm, k = ref.psislw(lw.copy())
        out[name + '_lw'] = lw
        out[name + '_smoothed'] = sm
        out[name + '_khat'] = np.float64(k)
    sm, k = ref.psislw(cases['normal_heavy'].copy(), Reff=0.37)
    out['reff_smoothed'], out['reff_khat'], out['reff_value'] = sm, np.float64(k), np.float64(0.37)
    two = np.stack([cases['normal_heavy'], cases['normal_moderate'][:2000]], axis=1)
    sm2, k2 = ref.psislw(two.copy())
    out['two_lw'], out['two_smoothed'], out['two_khat'] = two, sm2, np.asarray(k2)
    # GPD fit by itself
    x = np.sort(rng.pareto(2.0, size=500))
    kf, sf = ref.gpdfitnew(x, sort=False)
    out['gpd_x'], out['gpd_k'], out['gpd_sigma'] = x, np.float64(kf), np.float64(sf)
    # diagnostics.py on Gaussian q / p
    samples = 3.0 * rng.randn(50000, 2)
    lwd = scipy.stats.norm.logpdf(samples, scale=2.0).sum(1) - scipy.stats.norm.logpdf(samples, scale=3.0).sum(1)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        res = ref_diag.all_diagnostics(lwd, samples=samples)
        res_q = ref_diag.all_diagnostics(lwd, samples=samples, q_var=9.0 * np.eye(2), p_var=4.0 * np.eye(2),
                                         log_norm_bound=0.0)
        d3 = ref_diag.divergence_bound(lwd, alpha=3.0)
    out['diag_samples'], out['diag_lw'] = samples, lwd
    for key in ('d2', 'log_norm_bound', 'W1', 'W2', 'mean_error', 'std_error', 'cov_error'):
        out['diag_' + key] = np.float64(res[key])
        out['diagq_' + key] = np.float64(res_q[key])
    out['diag_d3'] = np.float64(d3)
    save('psis', provenance='reference viabel/_psis.py and viabel/diagnostics.py functions (pure numpy) run as they are',
         **out)


def gen_lowrank():
    """LRGaussian (viabel/approximations.py:610-731): forward methods of the reference and its ExclusiveKL
    closure (value exact, gradient by Richardson differences of the reference closure)."""
    rng = np.random.RandomState(61)
    worst = 0.0
    for D, k in ((3, 1), (6, 2), (10, 4), (17, 5), (12, 9), (24, 32), (12, 64), (21, 17)):
        seed, N = 7, 40
        ref = ref_approx.LRGaussian(D, seed=seed, k=k)
        orc = ofam.LRGaussian(D, k)
        init = ref.init_param()                      # consumes D k draws of the family's stream
        th0 = np.concatenate([0.3 * rng.randn(D), -0.4 + 0.3 * rng.randn(D), 0.4 * rng.randn(D * k)])
        th1 = np.concatenate([0.3 * rng.randn(D), -0.2 + 0.3 * rng.randn(D), 0.4 * rng.randn(D * k)])
        state = ref._rs.get_state()
        x = ref.sample(th0, N)
        rs = np.random.RandomState(seed)
        rs.set_state(state)
        noise = orc.draw_noise(rs, N)
        assert rel_err(orc.sample_from_noise(th0, noise), x) < 1e-13
        lq, ent, kl = ref.log_density(th1, x), ref.entropy(th0), ref.kl(th0, th1)
        mean, cov = ref.mean_and_cov(th0)
        assert rel_err(orc.log_density(th1, x), lq) < 1e-12 and rel_err(orc.entropy(th0), ent) < 1e-12
        assert rel_err(orc.kl(th0, th1), kl) < 1e-11 and rel_err(orc.cov(th0), cov) < 1e-13
        out = dict(dim=D, k=k, seed=seed, n=N, init_param=init, theta0=th0, theta1=th1, noise_z=noise[0],
                   noise_eps=noise[1], samples=x, log_density=lq, entropy=ent, kl=kl, mean=mean, cov=cov,
                   pth2=ref.pth_moment(th0, 2), pth4=ref.pth_moment(th0, 4))
        for mi, mspec in enumerate(model_specs(D, rng)):
            log_p, omodel = make_model(mspec)
            ref2 = ref_approx.LRGaussian(D, seed=seed, k=k)
            objective = ref_obj.ExclusiveKL(ref2, log_p, N)
            _ref_stubs.STATE['before_eval'] = snapshot_hook(ref2)
            value, grad_fd = objective(th0)
            _ref_stubs.STATE['before_eval'] = None
            noise2 = orc.draw_noise(np.random.RandomState(seed), N)
            ov, og = oobj.exclusive_kl(orc, omodel, th0, noise2)
            assert rel_err(ov, value) < 1e-12, (D, k, ov, value)
            e = rel_err(og, grad_fd)
            worst = max(worst, e)
            assert e < 2e-7, (D, k, mspec['kind'], e)
            tag = 'm%d_' % mi
            out.update({tag + key: val for key, val in spec_arrays({'kind': 'lr_gaussian', 'dim': D}, mspec).items()})
            out.update({tag + 'noise_z': noise2[0], tag + 'noise_eps': noise2[1], tag + 'value': value,
                        tag + 'grad_fd': grad_fd, tag + 'grad': og})
            # the same closure with use_path_deriv=True (objectives.py:156-159)
            ref3 = ref_approx.LRGaussian(D, seed=seed, k=k)
            objective = ref_obj.ExclusiveKL(ref3, log_p, N, use_path_deriv=True)
            _ref_stubs.STATE['before_eval'] = snapshot_hook(ref3)
            pd_value, pd_grad_fd = objective(th0)
            _ref_stubs.STATE['before_eval'] = None
            pv, pg = oobj.exclusive_kl(orc, omodel, th0, noise2, True)
            assert rel_err(pv, pd_value) < 1e-12, (D, k, pv, pd_value)
            e = rel_err(pg, pd_grad_fd)
            worst = max(worst, e)
            assert e < 2e-7, (D, k, mspec['kind'], 'path_deriv', e)
            out.update({tag + 'pd_value': pd_value, tag + 'pd_grad_fd': pd_grad_fd, tag + 'pd_grad': pg})
        save('lowrank_d%d_k%d' % (D, k), **out,
             provenance='reference LRGaussian forward code and ExclusiveKL closure via the autograd->numpy alias; '
                        'grad_fd: Richardson differences of the reference closure; grad: analytic (oracle)')
    print('LRGaussian ExclusiveKL: worst analytic-vs-FD(reference) grad rel err %.2e' % worst)


def gen_exclusive_kl_mvt():
    """ExclusiveKL with the MultivariateT family (the reference differentiates scipy's sqrtm with autograd,
    approximations.py:348): value from the reference closure, gradient by Richardson differences of it."""
    rng = np.random.RandomState(81)
    worst = 0.0
    for fspec in ({'kind': 'multivariate_t', 'dim': 3, 'df': 100}, {'kind': 'multivariate_t', 'dim': 6, 'df': 7},
                  {'kind': 'multivariate_t', 'dim': 10, 'df': 30}):
        D = fspec['dim']
        for mspec in model_specs(D, rng):
          for pd in (False, True):
            N, seed = 40, 1
            ref, orc = make_family(fspec, seed)
            log_p, omodel = make_model(mspec)
            theta = theta_for(fspec, rng) if not pd else theta
            objective = ref_obj.ExclusiveKL(ref, log_p, N, use_path_deriv=pd)
            _ref_stubs.STATE['before_eval'] = snapshot_hook(ref)
            value, grad_fd = objective(theta)
            _ref_stubs.STATE['before_eval'] = None
            noise = orc.draw_noise(np.random.RandomState(seed), N)
            ov, og = oobj.exclusive_kl(orc, omodel, theta, noise, pd)
            assert rel_err(ov, value) < 1e-12, (fspec, mspec['kind'], pd, ov, value)
            e = rel_err(og, grad_fd)
            worst = max(worst, e)
            assert e < 2e-6, (fspec, mspec['kind'], pd, e)
            save('ekl_%s_d%d_%s_pd%d_n%d' % (fspec['kind'], D, mspec['kind'], int(pd), N), **spec_arrays(fspec, mspec),
                 seed=seed, n=N, theta=theta, noise_chi=noise[0], noise_z=noise[1], use_path_deriv=pd,
                 value=value, grad_fd=grad_fd, grad=og,
                 provenance='value: reference closure; grad_fd: Richardson central differences of the reference '
                            'closure (through scipy sqrtm); grad: analytic (oracle, Sylvester solve), agrees with grad_fd')
    print('ExclusiveKL MultivariateT: worst analytic-vs-FD(reference) grad rel err %.2e' % worst)


def gen_lowrank_alpha_dis():
    """LRGaussian under AlphaDivergence and DISInclusiveKL: the reference's objectives are family-generic
    (objectives.py:391-416, :443-463 over approximations.py:636-644, :685-707)."""
    rng = np.random.RandomState(61)
    worst_a = worst_d = 0.0
    for fspec in ({'kind': 'lr_gaussian', 'dim': 4, 'k': 2}, {'kind': 'lr_gaussian', 'dim': 5, 'k': 1},
                  {'kind': 'lr_gaussian', 'dim': 7, 'k': 3}):
        D = fspec['dim']
        for mspec in model_specs(D, rng):
            for alpha in (2.0, 0.5):
                N, np_seed = 32, 851
                ref, orc = make_family(fspec, 1)
                log_p, omodel = make_model(mspec)
                theta = theta_for(fspec, rng)
                objective = ref_obj.AlphaDivergence(ref, log_p, N, alpha)
                np.random.seed(np_seed)
                value, grad_fd = objective(theta)
                np.random.seed(np_seed)
                seed = np.random.randint(2 ** 32)                      # objectives.py:455
                noise = orc.draw_noise(np.random.RandomState(seed), N)
                ov, og = oobj.alpha_divergence(orc, omodel, theta, noise, alpha)
                assert rel_err(ov, value) < 1e-12, (ov, value)
                e = rel_err(og, grad_fd)
                worst_a = max(worst_a, e)
                assert e < 2e-7, (fspec, mspec['kind'], alpha, e)
                save('alpha_lr_gaussian_d%d_k%d_%s_a%g' % (D, fspec['k'], mspec['kind'], alpha),
                     **spec_arrays(fspec, mspec), np_seed=np_seed, seed=seed, n=N, theta=theta, alpha=alpha,
                     value=value, grad_fd=grad_fd, grad=og, noise_z=noise[0], noise_eps=noise[1],
                     provenance='value: reference; grad_fd: FD of the reference log-weights closure contracted as in '
                                'objectives.py:460; grad: analytic (oracle)')
        mspec = model_specs(D, rng)[0]
        for use_resampling in (True, False):
            N, ess_target, np_seed = 64, 20, 851
            ref, orc = make_family(fspec, 1)
            log_p, omodel = make_model(mspec)
            theta = theta_for(fspec, rng)
            prior_params = np.concatenate([[0] * D, [1] * D]).astype(float)   # test_objectives.py:85
            objective = ref_obj.DISInclusiveKL(
                ref, log_p, N, ess_target=ess_target, temper_prior=ref_approx.MFGaussian(D),
                temper_prior_params=prior_params, use_resampling=use_resampling)
            chosen = []
            real_choice = np.random.choice

            def rec_choice(*a, **k):
                idx = real_choice(*a, **k)
                chosen.append(np.array(idx))
                return idx
            np.random.choice = rec_choice
            try:
                np.random.seed(np_seed)
                _ref_stubs.STATE['before_eval'] = snapshot_hook(ref, objective, np_seed)
                value, grad_fd = objective(theta)
            finally:
                np.random.choice = real_choice
                _ref_stubs.STATE['before_eval'] = None
            noise = orc.draw_noise(np.random.RandomState(1), N)
            od = oobj.DISInclusiveKL(orc, omodel, N, ess_target, ofam.MFGaussian(D), prior_params,
                                     use_resampling=use_resampling)
            indices = chosen[0] if use_resampling else None
            ov, og = od(theta, noise=noise, indices=indices)
            assert rel_err(ov, value) < 1e-11, (fspec, use_resampling, ov, value)
            assert rel_err(od._eps, objective._eps) < 1e-12
            assert rel_err(od._state_w_clipped, objective._state_w_clipped) < 1e-10
            e = rel_err(og, grad_fd)
            worst_d = max(worst_d, e)
            assert e < 2e-6, (fspec, use_resampling, e)
            out = dict(spec_arrays(fspec, mspec), np_seed=np_seed, seed=1, n=N,
                       ess_target=ess_target, use_resampling=use_resampling, theta=theta,
                       prior_params=prior_params, value=value, grad_fd=grad_fd, grad=og,
                       eps=objective._eps, w_clipped=objective._state_w_clipped,
                       log_q=objective._state_log_q, log_p=objective._state_log_p_unnormalized,
                       samples=objective._state_samples, noise_z=noise[0], noise_eps=noise[1],
                       provenance='reference DISInclusiveKL code over the reference LRGaussian; grad_fd by FD with '
                                  'getval replay')
            if use_resampling:
                out['indices'] = indices
            save('dis_lr_gaussian_d%d_k%d_rs%d' % (D, fspec['k'], use_resampling), **out)
    print('LRGaussian AlphaDivergence / DISInclusiveKL: worst analytic-vs-FD(reference) grad rel err %.2e / %.2e'
          % (worst_a, worst_d))


def gen_dis_priors():
    """DISInclusiveKL with a tempering prior that is NOT an MFGaussian (objectives.py:283-285 takes any family;
    :317-319 calls its log_density): the reference's own MFStudentT, MultivariateT and LRGaussian as priors."""
    rng = np.random.RandomState(57)
    worst = 0.0
    priors = ({'kind': 'mf_student_t', 'df': 5.0}, {'kind': 'multivariate_t', 'df': 7.0},
              {'kind': 'lr_gaussian', 'k': 1})
    for fspec in ({'kind': 'mf_gaussian', 'dim': 3}, {'kind': 'multivariate_t', 'dim': 3, 'df': 9},
                  {'kind': 'lr_gaussian', 'dim': 3, 'k': 2}):
        D = fspec['dim']
        mspec = model_specs(D, rng)[0]
        for pspec in priors:
            pspec = dict(pspec, dim=D)
            use_resampling = pspec['kind'] == 'multivariate_t'      # one resampled case per family
            N, ess_target, np_seed = 64, 20, 851
            ref, orc = make_family(fspec, 1)
            ref_prior, orc_prior = make_family(pspec, 3)
            log_p, omodel = make_model(mspec)
            theta = theta_for(fspec, rng)
            prior_params = theta_for(pspec, rng)
            prior_params[:D] *= 0.3                       # a broad prior around the origin
            prior_params[D:2 * D] += 0.8
            objective = ref_obj.DISInclusiveKL(
                ref, log_p, N, ess_target=ess_target, temper_prior=ref_prior,
                temper_prior_params=prior_params, use_resampling=use_resampling)
            chosen = []
            real_choice = np.random.choice

            def rec_choice(*a, **k):
                idx = real_choice(*a, **k)
                chosen.append(np.array(idx))
                return idx
            np.random.choice = rec_choice
            try:
                np.random.seed(np_seed)
                _ref_stubs.STATE['before_eval'] = snapshot_hook(ref, objective, np_seed)
                value, grad_fd = objective(theta)
            finally:
                np.random.choice = real_choice
                _ref_stubs.STATE['before_eval'] = None
            noise = orc.draw_noise(np.random.RandomState(1), N)
            od = oobj.DISInclusiveKL(orc, omodel, N, ess_target, orc_prior, prior_params, use_resampling=use_resampling)
            indices = chosen[0] if use_resampling else None
            ov, og = od(theta, noise=noise, indices=indices)
            assert rel_err(ov, value) < 1e-11, (fspec, pspec, ov, value)
            assert rel_err(od._eps, objective._eps) < 1e-12, (fspec, pspec, od._eps, objective._eps)
            assert rel_err(od._state_w_clipped, objective._state_w_clipped) < 1e-10
            e = rel_err(og, grad_fd)
            worst = max(worst, e)
            assert e < 2e-6, (fspec, pspec, e)
            out = dict(spec_arrays(fspec, mspec), np_seed=np_seed, seed=1, n=N,
                       ess_target=ess_target, use_resampling=use_resampling, theta=theta,
                       prior_kind=pspec['kind'], prior_df=pspec.get('df', 0.0), prior_rank=pspec.get('k', 0),
                       prior_params=prior_params, value=value, grad_fd=grad_fd, grad=og,
                       eps=objective._eps, w_clipped=objective._state_w_clipped,
                       log_q=objective._state_log_q, log_p=objective._state_log_p_unnormalized,
                       samples=objective._state_samples,
                       provenance='reference DISInclusiveKL code with a non-MFGaussian temper_prior; grad_fd by FD with getval replay')
            if use_resampling:
                out['indices'] = indices
            if fspec['kind'] == 'multivariate_t':
                out['noise_chi'], out['noise_z'] = noise
            elif fspec['kind'] == 'lr_gaussian':
                out['noise_z'], out['noise_eps'] = noise
            else:
                out['noise'] = noise
            save('disprior_%s_%s_d%d_rs%d' % (fspec['kind'], pspec['kind'], D, use_resampling), **out)
    print('DISInclusiveKL, general tempering priors: worst analytic-vs-FD(reference) grad rel err %.2e' % worst)


GENERATORS = {}

if __name__ == '__main__':
    GENERATORS.update(torch=gen_torch_crosscheck, family=gen_family_forward, ekl=gen_exclusive_kl, rge=gen_rge,
                      alpha=gen_alpha, dis=gen_dis, chainstats=gen_chain_stats, optimizers=gen_optimizers,
                      psis=gen_psis, lowrank=gen_lowrank, ekl_mvt=gen_exclusive_kl_mvt,
                      lowrank_alpha_dis=gen_lowrank_alpha_dis, dis_priors=gen_dis_priors)
    picked = sys.argv[1:]          # e.g. `make_golden.py psis optimizers` regenerates only those fixtures
    if picked and picked[0] == '--check':
        # regenerate everything into a scratch directory and diff against the committed fixtures (drift guard, run by
        # tests/test_oracle_golden.py wherever the reference tree exists)
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            OUT_DIR = tmp
            for name in (picked[1:] or list(GENERATORS)):
                GENERATORS[name]()
            diffs = compare_with_committed(tmp, SAVED) if not picked[1:] else [
                d for d in compare_with_committed(tmp, SAVED) if 'no generator' not in d]
        for d in diffs:
            print('DRIFT ' + d)
        print('checked %d fixtures: %s' % (len(SAVED), 'no drift' if not diffs else '%d differences' % len(diffs)))
        sys.exit(1 if diffs else 0)
    if not picked:
        for f in os.listdir(HERE):
            if f.endswith('.npz'):
                os.remove(os.path.join(HERE, f))
        picked = list(GENERATORS)
    for name in picked:
        GENERATORS[name]()
    print('wrote %d fixtures to %s' % (len(SAVED), HERE))
