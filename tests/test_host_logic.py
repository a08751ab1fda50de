"""CPU tests of the host-side logic around the hot path: optimisers, FASO / RAABBVI, chain statistics,
families' parameter-space methods, argument validation.  Modelled on the reference's
viabel/tests/test_optimization.py (dummy objective + dummy family) and test_convenience.py."""
import os
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


def test_callable_model_host_side():
    """CallableModel without a GPU: the trampoline the C library calls (vb_model_callback) fills f and grad through raw
    pointers, a missing gradient is differenced numerically (and announced once), a raising callable returns non-zero
    with its exception parked for the engine to re-raise."""
    import ctypes
    import warnings
    rng = np.random.RandomState(0)
    D, N = 4, 7
    A = rng.randn(D, D)
    P = A @ A.T + np.eye(D)

    def f(z):
        return -0.5 * np.einsum('ni,ij,nj->n', z, P, z)

    def g(z):
        return -z @ P
    z = np.ascontiguousarray(rng.randn(N, D))
    dp = ctypes.POINTER(ctypes.c_double)
    for model, tol in ((vb.CallableModel(D, f, g), 0.0), (vb.CallableModel(D, value_and_grad=lambda x: (f(x), g(x))), 0.0),
                       (vb.CallableModel(D, f), 1e-8)):
        fo, go = np.full(N, np.nan), np.full((N, D), np.nan)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            assert model._trampoline(None, z.ctypes.data_as(dp), N, D, fo.ctypes.data_as(dp), go.ctypes.data_as(dp)) == 0
            assert model._trampoline(None, z.ctypes.data_as(dp), N, D, fo.ctypes.data_as(dp), go.ctypes.data_as(dp)) == 0
        assert len([x for x in w if 'central differences' in str(x.message)]) == (1 if tol else 0)
        np.testing.assert_array_equal(fo, f(z))
        assert np.max(np.abs(go - g(z))) <= tol * np.max(np.abs(g(z))) + (0.0 if tol else 0.0)
        fo[:] = np.nan                                   # value-only call: grad pointer NULL
        assert model._trampoline(None, z.ctypes.data_as(dp), N, D, fo.ctypes.data_as(dp), dp()) == 0
        np.testing.assert_array_equal(fo, f(z))
    bad = vb.CallableModel(D, lambda x: 1 / 0, g)
    fo = np.zeros(N)
    assert bad._trampoline(None, z.ctypes.data_as(dp), N, D, fo.ctypes.data_as(dp), dp()) == 1
    assert isinstance(bad._error[0], ZeroDivisionError)
    from viabel_amd.models import as_device_model
    assert as_device_model(vb.GaussianModel([0, 0], [1, 1]), 2).dim == 2
    assert isinstance(as_device_model(vb.Model(f), D), vb.CallableModel)
    assert isinstance(as_device_model(f, D), vb.CallableModel)
    with pytest.raises(TypeError):
        as_device_model(3.0, D)
    with pytest.raises(ValueError):
        vb.CallableModel(D)


def test_dis_tempering_prior_specs_and_clip_fixed_point():
    """DISInclusiveKL host logic without a GPU: any family as tempering prior becomes the engine's spec
    (objectives.py:283-285), and the clipping (``:370-386``) is the oracle's fixed point."""
    from oracle import families as ofam
    from oracle import objectives as oobj
    from viabel_amd import _lib
    from viabel_amd.objectives import DISInclusiveKL
    D = 5
    rng = np.random.RandomState(3)
    dis = DISInclusiveKL.__new__(DISInclusiveKL)
    A = rng.randn(D, D)
    S = A @ A.T / D + np.eye(D)
    x = rng.randn(6, D)
    for prior, oprior, params in (
            (vb.MFStudentT(D, 7.0), ofam.MFStudentT(D, 7.0), np.concatenate([rng.randn(D), 0.1 * rng.randn(D)])),
            (vb.FullRankGaussian(D), ofam.FullRankGaussian(D), vb.FullRankGaussian(D).pack(rng.randn(D), np.linalg.cholesky(S))),
            (vb.MultivariateT(D, 9.0), ofam.MultivariateT(D, 9.0), np.concatenate([rng.randn(D), ofam.psd_to_free(S)])),
            (vb.LRGaussian(D, k=2), ofam.LRGaussian(D, 2), np.concatenate([rng.randn(D), 0.1 * rng.randn(D), rng.randn(2 * D)]))):
        dis._temper_prior, dis._temper_prior_params = prior, np.asarray(params, dtype=float)
        spec, arg = dis._build_prior_spec(D)
        assert arg.shape == (2 * D,) and spec is not None
        kind, df, loc, scale, logdet = spec
        want = oprior.log_density(params, x)
        if kind == _lib.PRIOR_DIAG_STUDENT_T:
            from scipy import stats
            got = np.sum(stats.t.logpdf((x - loc) * np.exp(-scale), df) - scale, axis=1)
        else:
            u = (x - loc) @ scale.T
            maha = np.sum(u * u, axis=1)
            if df > 0:
                from scipy.special import gammaln
                got = (gammaln(0.5 * (df + D)) - gammaln(0.5 * df) - 0.5 * D * np.log(np.pi * df) - logdet
                       - 0.5 * (df + D) * np.log1p(maha / df))
            else:
                got = -0.5 * D * np.log(2 * np.pi) - logdet - 0.5 * maha
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    dis._temper_prior, dis._temper_prior_params = vb.MFGaussian(D), np.zeros(2 * D)
    assert dis._build_prior_spec(D)[0] is None
    dis._temper_prior_params = np.zeros(3)
    with pytest.raises(ValueError):
        dis._build_prior_spec(D)
    ref = oobj.DISInclusiveKL.__new__(oobj.DISInclusiveKL)
    for seed in range(40):
        w = np.exp(np.random.RandomState(seed).randn(300) * 3)
        for thr in (0.02, 0.1, 10):
            dis._w_clip_threshold = ref._w_clip_threshold = thr
            np.testing.assert_array_equal(dis._clip_weights(w), ref._clip(w))


def test_mt19937_jump_table_against_numpy():
    """The committed jump-ahead polynomials (viabel_amd/csrc/vb_mt_jump.h, tools/make_mt_jump.py): the correlation of
    a polynomial with the sequence generated from a block equals the block that many words later, as numpy's own
    generator reaches it (three rows here; the generator script checks five more when it writes the table)."""
    import re
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('make_mt_jump', os.path.join(root, 'tools', 'make_mt_jump.py'))
    mj = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mj)
    text = open(os.path.join(root, 'viabel_amd', 'csrc', 'vb_mt_jump.h')).read()
    assert 'kMtBlocksPerStream = %d' % mj.BLOCKS_PER_STREAM in text
    rows = re.findall(r'\{((?:0x[0-9a-f]{8}u,?)+)\}', text)
    assert len(rows) == 3 * mj.R
    for row in (0, 4, 6):                                # jumps of 1, 2 x 4 and 1 x 16 streams
        words = [int(x[:-1], 16) for x in rows[row].split(',') if x]
        assert len(words) == mj.N
        g = sum(w << (32 * i) for i, w in enumerate(words))
        m = mj.BLOCKS_PER_STREAM * (row % 3 + 1) * 4 ** (row // 3)
        rs = np.random.RandomState(4242 + row)
        b1 = mj.refresh(rs.get_state()[1])
        rs.random_sample(mj.N * m // 2)                  # two words per draw: 624 m words
        st = rs.get_state()
        assert st[2] == mj.N
        assert mj.jump_by_correlation(b1, g) == [int(x) for x in mj.refresh(st[1])]


def test_dptr_call_sites_pass_names():
    """``_lib._dptr`` returns a bare address (ctypes ``c_void_p``): the array behind it must outlive the call, so every
    call site passes a plain local name -- never a temporary such as ``_dptr(_f64(x))`` or ``_dptr(a[1:].copy())``, which
    would be freed before the C function reads it."""
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'viabel_amd')
    bad = []
    for name in sorted(os.listdir(root)):
        if not name.endswith('.py'):
            continue
        text = open(os.path.join(root, name)).read()
        for m in re.finditer(r'_dptr\(([^()]*(?:\([^()]*\))?[^()]*)\)', text):
            arg = m.group(1).strip()
            if name == '_lib.py' and arg == 'a':          # the definition itself
                continue
            if not re.fullmatch(r'[A-Za-z_][A-Za-z_0-9]*', arg):
                bad.append((name, arg))
    assert not bad, bad


def test_peek_of_numpys_next_randint():
    """objectives._peek_next_randint: the seed AlphaDivergence will draw next (objectives.py:455), read off the global
    generator's state without consuming it -- across block boundaries (the twist), after other draws, after reseeding and
    set_state -- and the peek itself leaves the stream untouched."""
    from viabel_amd.objectives import _peek_next_randint
    saved = np.random.get_state()
    try:
        for seed in (0, 1, 12345, 2 ** 32 - 1):
            np.random.seed(seed)
            for i in range(1500):                       # more than two 624-word blocks
                peek = _peek_next_randint()
                assert peek == _peek_next_randint()     # no side effect
                assert peek == int(np.random.randint(2 ** 32))
                if i % 5 == 0:
                    np.random.randn(3)
                if i % 11 == 0:
                    np.random.standard_t(7, 4)
        st = np.random.get_state()
        a = _peek_next_randint()
        np.random.randint(2 ** 32, size=1000)
        np.random.set_state(st)
        assert _peek_next_randint() == a == int(np.random.randint(2 ** 32))
    finally:
        np.random.set_state(saved)


def test_peek_declines_another_bit_generator():
    """np.random.set_bit_generator can put a generator with another state layout behind the global functions: no hint."""
    from viabel_amd.objectives import _peek_next_randint
    if not hasattr(np.random, 'set_bit_generator'):
        pytest.skip('numpy without set_bit_generator')
    old = np.random.get_bit_generator()
    try:
        np.random.set_bit_generator(np.random.PCG64(1))
        assert _peek_next_randint() is None
    finally:
        np.random.set_bit_generator(old)
    np.random.seed(4)
    assert _peek_next_randint() == int(np.random.randint(2 ** 32))
