"""CPU tests of the host-side logic around the hot path: optimisers, FASO / RAABBVI, chain statistics,
families' parameter-space methods, argument validation.  Modelled on the reference's
viabel/tests/test_optimization.py (dummy objective + dummy family) and test_convenience.py."""
import numpy as np
import pytest

import _golden as G
import viabel_amd as vb
from viabel_amd import _chain_stats as cs
from viabel_amd import optimization as opt_mod


class DummyApproximationFamily:
    """Minimal duck-typed family (tests/test_optimization.py:12-17)."""
    supports_kl = True

    def kl(self, a, b):
        return float(np.sum((np.asarray(a) - np.asarray(b)) ** 2))


class DummyObjective:
    """Noisy quadratic with a seeded noise stream (tests/test_optimization.py:20-32)."""

    def __init__(self, target, noise=0.3, seed=3, scale=1.0):
        self.target = np.asarray(target, dtype=float)
        self.rs = np.random.RandomState(seed)
        self.noise = noise
        self.scale = scale
        self.approx = DummyApproximationFamily()

    def __call__(self, x):
        g = self.scale * (x - self.target) + self.noise * self.rs.randn(*x.shape)
        return 0.5 * self.scale * np.sum((x - self.target) ** 2), g

    def update(self, x, d):
        return x - d


CTORS = {
    'sgd': lambda: vb.StochasticGradientOptimizer(0.05, diagnostics=True),
    'rmsprop': lambda: vb.RMSProp(0.05, diagnostics=True),
    'avgrmsprop': lambda: vb.AveragedRMSProp(0.05, diagnostics=True),
    'adam': lambda: vb.Adam(0.05, diagnostics=True),
    'avgadam': lambda: vb.AveragedAdam(0.05, diagnostics=True),
    'adagrad': lambda: vb.Adagrad(0.5, diagnostics=True),
    'wadagrad': lambda: vb.WindowedAdagrad(0.05, diagnostics=True),
}


@pytest.mark.parametrize('name', sorted(CTORS))
def test_optimizer_trajectory_matches_reference(name):
    """Same seeded objective as the fixture generator: iterates must match the reference's to rounding."""
    fx = G.load(G.fixtures('optimizers')[0])
    res = CTORS[name]().optimize(300, DummyObjective(fx['target']), np.zeros(4))
    np.testing.assert_allclose(res['variational_param_history'][-1], fx[name + '_last'], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(res['opt_param'], fx[name + '_opt_param'], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(res['value_history'], fx[name + '_values'], rtol=1e-11, atol=1e-13)


def test_chain_stats_match_reference():
    fx = G.load(G.fixtures('chainstats')[0])
    for i in range(int(fx['n_chains'])):
        x = fx['chain%d' % i]
        n = x.shape[0]
        e = np.array([cs.ess(x[:, k].reshape(1, n)) for k in range(3)])
        np.testing.assert_allclose(e, fx['ess%d' % i], rtol=1e-12)
        np.testing.assert_allclose(cs.MCSE(x)[1], fx['mcse%d' % i], rtol=1e-12)
        np.testing.assert_allclose(cs.compute_R_hat(x), fx['rhat%d' % i], rtol=1e-13)
    ok, best = cs.R_hat_convergence_check(list(fx['chain1']), fx['windows'])
    assert bool(ok) == bool(fx['rhat_ok']) and int(best) == int(fx['rhat_best'])
    assert np.isnan(cs.ess(np.ones((1, 50))))


def test_constructor_validation():
    with pytest.raises(ValueError):
        vb.StochasticGradientOptimizer(0.1, iterate_avg_prop=1.5)
    with pytest.raises(ValueError):
        vb.StochasticGradientOptimizer(0.1, iterate_avg_prop=0.0)
    with pytest.raises(ValueError):
        vb.FASO(object())
    sgo = vb.RMSProp(0.1)
    for kw in (dict(mcse_threshold=0), dict(W_min=0), dict(k_check=0), dict(ESS_min=0)):
        with pytest.raises(ValueError):
            vb.FASO(sgo, **kw)
    with pytest.raises(ValueError):
        vb.RAABBVI(sgo, rho=1.5)


def test_faso_converges_and_stops(capsys):
    target = np.array([1.0, -2.0, 0.5])
    sgo = vb.RMSProp(0.01, diagnostics=True)
    res = vb.FASO(sgo, mcse_threshold=0.02).optimize(20000, DummyObjective(target, noise=0.5), np.zeros(3))
    assert res['k_stopped'] is not None and res['k_conv'] is not None
    np.testing.assert_allclose(res['opt_param'], target, atol=0.05)
    assert 'Convergence reached at iteration' in capsys.readouterr().out


def test_raabbvi_terminates_and_is_accurate(capsys):
    target = np.array([1.0, -2.0, 0.5])
    sgo = vb.AveragedRMSProp(0.1, diagnostics=True)
    res = vb.RAABBVI(sgo, mcse_threshold=0.05, accuracy_threshold=0.05).optimize(
        40000, DummyObjective(target, noise=0.5), np.zeros(3))
    np.testing.assert_allclose(res['opt_param'], target, atol=0.05)
    assert len(res['learning_rate_hist']) >= 2
    out = capsys.readouterr().out
    assert 'Termination rule reached' in out or 'maximum number of iterations' in out


def test_raabbvi_falls_back_to_faso_without_kl(capsys):
    obj = DummyObjective(np.ones(2))
    obj.approx.supports_kl = False
    res = vb.RAABBVI(vb.RMSProp(0.01, diagnostics=True)).optimize(3000, obj, np.zeros(2))
    assert 'does not support KL. Using FASO' in capsys.readouterr().out
    assert 'k_stopped' in res


def test_weighted_regression_posterior_recovers_parameters():
    """y = log c + 2 log(rho^-kappa - 1) + 2 kappa x + small noise: posterior means near the truth."""
    rho, kappa, log_c = 0.5, 0.7, -1.0
    x = np.log(0.1 * rho ** np.arange(8))
    rs = np.random.RandomState(0)
    y = log_c + 2 * np.log(rho ** (-kappa) - 1) + 2 * kappa * x + 0.01 * rs.randn(8)
    opt = vb.RAABBVI(vb.RMSProp(0.1))
    fit, k_hat, c_hat = opt.weighted_linear_regression(None, y, x)
    assert abs(k_hat - kappa) < 0.05 and abs(np.log(c_hat) - log_c) < 0.3
    fit2, k2, c2 = opt.weighted_linear_regression(None, y, x)
    assert k_hat == k2 and c_hat == c2          # seeded: deterministic
    opt_avg = vb.RAABBVI(vb.AveragedRMSProp(0.1))
    _, k_fixed, _ = opt_avg.weighted_linear_regression(None, y, x)
    assert k_fixed == 1
    b0, b1 = opt.wls(np.arange(5.0), 2.0 + 3.0 * np.arange(5.0))
    assert abs(b0 - 2) < 1e-9 and abs(b1 - 3) < 1e-9


def test_bbvi_argument_validation():
    """viabel/tests/test_convenience.py:39-46 (no GPU needed: validation happens first)."""
    with pytest.raises(ValueError):
        vb.bbvi(2)
    with pytest.raises(ValueError):
        vb.bbvi(2, objective=True, fit=True)
    with pytest.raises(ValueError):
        vb.bbvi(2, log_density=True, fit=True)
    with pytest.raises(ValueError):
        vb.bbvi(2, objective=True, log_density=True)
    with pytest.raises(TypeError):
        vb.bbvi(2, log_density=3.0)
    with pytest.raises(ValueError):
        vb.bbvi(2, log_density=vb.GaussianModel([0, 0], [1, 1]), grad_log_density=lambda x: -x)


def test_family_parameter_space_methods_match_golden():
    """Host methods of the product families against the reference-derived family fixtures."""
    for path in G.fixtures('family_'):
        fx = G.load(path)
        kind, D = str(fx['family_kind']), int(fx['dim'])
        fam = {'mf_gaussian': lambda: vb.MFGaussian(D, seed=int(fx['seed'])),
               'mf_student_t': lambda: vb.MFStudentT(D, float(fx['df']), seed=int(fx['seed'])),
               'multivariate_t': lambda: vb.MultivariateT(D, float(fx['df']), seed=int(fx['seed']))}[kind]()
        th0, th1 = fx['theta0'], fx['theta1']
        np.testing.assert_allclose(fam.init_param(), fx['init_param'], rtol=1e-14)
        np.testing.assert_allclose(fam.sample(th0, int(fx['n'])), fx['samples'], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(fam.log_density(th1, fx['samples']), fx['log_density'], rtol=1e-11)
        np.testing.assert_allclose(fam.entropy(th0), fx['entropy'], rtol=1e-12)
        mean, cov = fam.mean_and_cov(th0)
        np.testing.assert_allclose(mean, fx['mean'], rtol=1e-14)
        np.testing.assert_allclose(cov, fx['cov'], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(fam.pth_moment(th0, 2), fx['pth2'], rtol=1e-12)
        np.testing.assert_allclose(fam.pth_moment(th0, 4), fx['pth4'], rtol=1e-12)
        if 'kl' in fx:
            np.testing.assert_allclose(fam.kl(th0, th1), fx['kl'], rtol=1e-12)
        else:
            with pytest.raises(NotImplementedError):
                fam.kl(th0, th1)
        with pytest.raises(ValueError):
            fam.pth_moment(th0, 3)


def test_family_errors():
    with pytest.raises(ValueError, match='df must be greater than 2'):
        vb.MFStudentT(2, 2)
    with pytest.raises(ValueError, match='df must be greater than 2'):
        vb.MultivariateT(2, 1.5)
    assert not vb.MFStudentT(2, 3).supports_pth_moment(4)
    assert vb.MFGaussian(3).var_param_dim == 6 and vb.MultivariateT(3, 5).var_param_dim == 9


def test_host_blas_thread_policy(monkeypatch):
    """The dense-covariance objectives pin the host BLAS pool (viabel_amd._lib.apply_host_blas_policy): default one
    thread, VIABEL_AMD_HOST_BLAS_THREADS=0 leaves it alone, an explicit set_host_blas_threads wins."""
    threadpoolctl = pytest.importorskip('threadpoolctl')
    from viabel_amd import _lib

    def blas_threads():
        return sorted({p['num_threads'] for p in threadpoolctl.threadpool_info() if p['user_api'] == 'blas'})

    np.ones((4, 4)) @ np.ones((4, 4))            # make sure the BLAS library is loaded
    before = blas_threads()
    monkeypatch.setattr(_lib, '_blas_policy_done', False)
    monkeypatch.setattr(_lib, '_blas_sticky', None)
    monkeypatch.setenv('VIABEL_AMD_HOST_BLAS_THREADS', '0')
    _lib.apply_host_blas_policy()
    assert blas_threads() == before
    monkeypatch.setattr(_lib, '_blas_policy_done', False)
    monkeypatch.setenv('VIABEL_AMD_HOST_BLAS_THREADS', '1')
    _lib.apply_host_blas_policy()
    assert blas_threads() == [1]
    _lib.apply_host_blas_policy()                # idempotent
    assert _lib.set_host_blas_threads(2) and blas_threads() == [min(2, max(before))]
    with _lib.small_lapack(8):
        assert blas_threads() == [1]
    threadpoolctl.threadpool_limits(limits=max(before), user_api='blas')      # back to the session's setting
    assert blas_threads() == before


def test_shared_choice_is_numpy_choice():
    """The resampling draw of DISInclusiveKL (objectives.py:408) skips np.random.choice's argument checks but must
    consume the global stream and pick the indices exactly as it does."""
    from viabel_amd.objectives import _shared_choice

    class _OneRank:
        n_ranks = 1
    rng = np.random.RandomState(3)
    for n, size in ((16384, 2048), (100, 7), (5, 50)):
        w = rng.rand(n) ** 4
        p = w / w.sum()
        np.random.seed(11)
        want = np.random.choice(n, size=size, p=p)
        after_want = np.random.random_sample()
        np.random.seed(11)
        got = _shared_choice(_OneRank(), n, size, p)
        after_got = np.random.random_sample()
        np.testing.assert_array_equal(got, want)
        assert after_got == after_want
    with pytest.raises(ValueError):
        _shared_choice(_OneRank(), 4, 2, np.ones(3) / 3)


def test_source_model_host_side():
    """SourceModel is plain data until it is bound to an engine: construction, spec layout and argument checks need no GPU."""
    from viabel_amd import _lib
    src = '__device__ double vb_log_density(const double* z, int d, const double* p, double* g) { return 0.0; }'
    m = vb.SourceModel(4, src, params=[1, 2, 3])
    spec = m.device_spec()
    assert spec[0] == _lib.MODEL_SOURCE and spec[1] == 4 and spec[4] == src.encode()
    np.testing.assert_array_equal(spec[2], [1.0, 2.0, 3.0])
    assert m.device_spec() is spec                      # cached: the engine keys its model cache on identity
    assert vb.SourceModel(2, src.encode()).params.size == 0
    for bad in ('', None, 3):
        with pytest.raises(ValueError):
            vb.SourceModel(2, bad)
    for fam in (vb.MultivariateT(4, 10), vb.LRGaussian(4, k=1)):    # every objective x family takes a source model
        vb.AlphaDivergence(fam, m, 10, 2.0)
        vb.ExclusiveKL(fam, m, 10)
    with pytest.raises(ValueError):
        vb.ExclusiveKL(vb.MFGaussian(3), m, 10)         # dimension mismatch
