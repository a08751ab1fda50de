"""GPU parity: the HIP mean-field ExclusiveKL path against the reference-derived golden
vectors and against the numpy oracle, all through the C ABI (ctypes).

Tolerances (fp64 kernels, different summation order than numpy): value 1e-12 relative;
gradient 1e-11 relative to max|grad| (control-variate variants subtract nearly equal sums,
so 1e-10 there); 2e-7 against the reference's finite-difference gradients.
north_star asks for 1e-5 relative on the ELBO.
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
    _lib.default_engine()          # raises if no GPU / no library: no silent fallback
    return viabel_amd


def product_family(vb, fx, seed):
    kind, D = str(fx['family_kind']), int(fx['dim'])
    if kind == 'mf_gaussian':
        return vb.MFGaussian(D, seed=seed)
    if kind == 'mf_student_t':
        return vb.MFStudentT(D, float(fx['df']), seed=seed)
    if kind == 'multivariate_t':
        return vb.MultivariateT(D, float(fx['df']), seed=seed)
    raise ValueError(kind)


def product_model(vb, fx):
    if str(fx['model_kind']) == 'gauss_diag':
        return vb.GaussianModel(fx['model_mean'], fx['model_stdev'])
    return vb.FunnelModel(int(fx['dim']), int(fx['model_scale_index']),
                          float(fx['model_log_sigma_stdev']))


@pytest.mark.parametrize('path', G.fixtures('ekl_'), ids=lambda p: p.split('/')[-1][:-4])
def test_exclusive_kl_golden(vb, path):
    fx = G.load(path)
    approx = product_family(vb, fx, int(fx['seed']))
    obj = vb.ExclusiveKL(approx, product_model(vb, fx), int(fx['n']),
                         use_path_deriv=bool(fx['use_path_deriv']))
    value, grad = obj(fx['theta'])
    assert G.rel_err(value, fx['value']) < 1e-12
    assert G.rel_err(grad, fx['grad']) < 1e-11
    assert G.rel_err(grad, fx['grad_fd']) < (2e-6 if str(fx['family_kind']) == 'multivariate_t' else 2e-7)


@pytest.mark.parametrize('pd', [False, True])
@pytest.mark.parametrize('model_kind', ['gauss_diag', 'funnel', 'gauss_full', 'logistic'])
@pytest.mark.parametrize('D,N,rng_kind', [(256, 2048, 'numpy'), (70, 333, 'numpy'), (129, 1000, 'philox')])
def test_multivariate_t_exclusive_kl_matches_oracle(vb, D, N, rng_kind, model_kind, pd):
    """MultivariateT + ExclusiveKL (sampling, model gradient and the D x D contraction on the device, chain rule
    through the symmetric root on the host) against the oracle on the same draws; C3's D = 256."""
    from viabel_amd import _lib
    rng = np.random.RandomState(D)
    if model_kind == 'gauss_diag':
        mean, sd = rng.randn(D), np.exp(0.3 * rng.randn(D))
        model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    elif model_kind == 'funnel':
        model, omodel = vb.FunnelModel(D, D // 2), omod.Funnel(D, D // 2)
    elif model_kind == 'logistic':
        X = rng.randn(3 * D, D) / np.sqrt(D)
        y = (rng.rand(3 * D) < 0.5).astype(float)
        model, omodel = vb.LogisticRegressionModel(X, y, 3.0), omod.Logistic(X, y, 3.0)
    else:
        A = rng.randn(D, D)
        S = A @ A.T / D + np.eye(D)
        mean = rng.randn(D)
        model, omodel = vb.CorrelatedGaussianModel(mean, covariance=S), omod.GaussFull(mean, np.linalg.inv(S))
    if rng_kind == 'philox' and not pd:
        pytest.skip("rng='philox' without the path derivative is the throughput mode (Cholesky sampling, device "
                    "chi-square): tests/test_gpu_objectives.py::test_exclusive_kl_multivariate_t_throughput_mode_against_oracle")
    approx = vb.MultivariateT(D, 9.0, seed=6, rng=rng_kind)
    B = rng.randn(D, D)
    theta = np.concatenate([0.2 * rng.randn(D), ofam.psd_to_free(0.05 * (B @ B.T / D + 0.5 * np.eye(D)))])
    value, grad = vb.ExclusiveKL(approx, model, N, use_path_deriv=pd)(theta)
    if rng_kind == 'numpy':
        noise = ofam.MultivariateT(D, 9.0).draw_noise(np.random.RandomState(6), N)
    else:
        eng = _lib.default_engine()
        eng.noise_generate(30, N, D, seed=6, stream=0)
        noise = (np.random.RandomState(6).chisquare(9.0, N), eng.noise_get_host(30, N, D))
    ov, og = oobj.exclusive_kl(ofam.MultivariateT(D, 9.0), omodel, theta, noise, pd)
    assert abs(value - ov) <= 1e-12 * abs(ov), (value, ov)
    np.testing.assert_allclose(grad, og, rtol=0, atol=1e-10 * np.max(np.abs(og)))


@pytest.mark.parametrize('path', G.fixtures('rge_'), ids=lambda p: p.split('/')[-1][:-4])
def test_rge_golden(vb, path):
    fx = G.load(path)
    approx = product_family(vb, fx, int(fx['seed']))
    obj = vb.ExclusiveKL(approx, product_model(vb, fx), int(fx['n']),
                         use_path_deriv=bool(fx['use_path_deriv']),
                         hessian_approx_method=str(fx['method']))
    value, grad = obj(fx['theta'])
    assert G.rel_err(value, fx['value']) < 1e-12
    assert G.rel_err(grad, fx['grad']) < 1e-10


def _theta(D, rng):
    return np.concatenate([0.3 * rng.randn(D), -1.0 + 0.2 * rng.randn(D)])


@pytest.mark.parametrize('D,N', [(1024, 4096), (10, 100), (1000, 333), (129, 7), (2, 1), (257, 4097)])
@pytest.mark.parametrize('model_kind', ['gauss_diag', 'funnel'])
def test_against_oracle_shapes(vb, D, N, model_kind):
    """BASELINE configs C0 (D=10, N=100) and C1 (D=1024, N=4096) plus ragged shapes."""
    rng = np.random.RandomState(D * 7 + N)
    if model_kind == 'gauss_diag':
        mean, sd = rng.randn(D), np.exp(0.3 * rng.randn(D))
        model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    else:
        k = D // 3
        model, omodel = vb.FunnelModel(D, k, 1.3), omod.Funnel(D, k, 1.3)
    theta = _theta(D, rng)
    for pd in (False, True):
        approx = vb.MFGaussian(D, seed=5)
        value, grad = vb.ExclusiveKL(approx, model, N, use_path_deriv=pd)(theta)
        noise = np.random.RandomState(5).randn(N, D)
        ov, og = oobj.exclusive_kl(ofam.MFGaussian(D), omodel, theta, noise, pd)
        assert G.rel_err(value, ov) < 1e-12, (pd, value, ov)
        assert G.rel_err(grad, og) < 1e-11, (pd, G.rel_err(grad, og))


@pytest.mark.parametrize('method', ['full', 'mean_only', 'loo_diag_approx', 'loo_direct_approx'])
@pytest.mark.parametrize('family', ['gauss', 't'])
def test_rge_against_oracle_large(vb, method, family):
    D, N = 300, 513
    rng = np.random.RandomState(17)
    theta = _theta(D, rng)
    for model, omodel in ((vb.FunnelModel(D, 7), omod.Funnel(D, 7)),
                          (vb.GaussianModel(np.ones(D), 2 * np.ones(D)), omod.GaussDiag(np.ones(D), 2 * np.ones(D)))):
        if family == 'gauss':
            approx, ofamily = vb.MFGaussian(D, seed=3), ofam.MFGaussian(D)
        else:
            approx, ofamily = vb.MFStudentT(D, 9, seed=3), ofam.MFStudentT(D, 9)
        for pd in (False, True):
            value, grad = vb.ExclusiveKL(approx, model, N, use_path_deriv=pd,
                                         hessian_approx_method=method)(theta)
            noise = ofamily.draw_noise(np.random.RandomState(3), N)
            if pd:   # the family's RandomState advanced: replay both calls
                noise = ofamily.draw_noise(_advance(ofamily, 3, N), N)
            ov, og = oobj.rge_reduced(ofamily, omodel, theta, noise, method, pd)
            assert G.rel_err(value, ov) < 1e-12
            assert G.rel_err(grad, og) < 1e-10


def _advance(ofamily, seed, n):
    rs = np.random.RandomState(seed)
    ofamily.draw_noise(rs, n)
    return rs


def test_rng_stream_advances_like_reference(vb):
    """Second call consumes draws [N*D, 2*N*D) of RandomState(seed)."""
    D, N = 16, 50
    approx = vb.MFGaussian(D, seed=1)
    model = vb.GaussianModel(np.zeros(D), np.ones(D))
    obj = vb.ExclusiveKL(approx, model, N)
    theta = np.concatenate([np.zeros(D), np.zeros(D)])
    rs = np.random.RandomState(1)
    for _ in range(3):
        value, grad = obj(theta)
        ov, og = oobj.exclusive_kl(ofam.MFGaussian(D), omod.GaussDiag(np.zeros(D), np.ones(D)),
                                   theta, rs.randn(N, D))
        # the ELBO is ~0 here (q == p): compare on the scale of its two terms (entropy ~ 22.7)
        assert abs(value - ov) < 1e-12 * 22.7 and G.rel_err(grad, og) < 1e-11


def test_float32_theta_accepted(vb):
    """tests/test_objectives.py:24 passes a float32 init; computation stays fp64."""
    D = 2
    obj = vb.ExclusiveKL(vb.MFStudentT(D, 100), vb.GaussianModel([1., -1.], [2., 5.]), 100)
    v, g = obj(np.array([0, 0, 1, 1], dtype=np.float32))
    assert np.isfinite(v) and g.dtype == np.float64 and g.shape == (4,)


def test_model_call_on_device(vb):
    rng = np.random.RandomState(0)
    for D in (2, 77, 1024):
        x = rng.randn(33, D)
        m, sd = rng.randn(D), np.exp(rng.randn(D))
        np.testing.assert_allclose(vb.GaussianModel(m, sd)(x), omod.GaussDiag(m, sd).logp(x), rtol=1e-12)
        np.testing.assert_allclose(vb.FunnelModel(D)(0.3 * x), omod.Funnel(D).logp(0.3 * x), rtol=1e-12)
        assert vb.FunnelModel(D)(0.3 * x[0]).shape == (1,)


def test_invalid_hessian_approx_method(vb):
    with pytest.raises(ValueError) as info:
        vb.ExclusiveKL(vb.MFGaussian(2), vb.GaussianModel([0, 0], [1, 1]), 10,
                       hessian_approx_method='invalid method')
    assert str(info.value) == ("Name of approximation must be one of 'full', 'mean_only', "
                               "'loo_diag_approx', 'loo_direct_approx' or None object.")


def test_host_callable_is_bound_as_callable_model(vb):
    obj = vb.ExclusiveKL(vb.MFGaussian(2), lambda x: -0.5 * np.sum(x ** 2, axis=1), 10)
    assert isinstance(obj.model, vb.CallableModel) and obj.model.dim == 2
    with pytest.raises(TypeError):
        vb.ExclusiveKL(vb.MFGaussian(2), 'not a model', 10).model


def test_philox_noise_is_standard_normal_and_shard_invariant(vb):
    from viabel_amd import _lib
    eng = _lib.default_engine()
    N, D = 4096, 257
    eng.noise_generate(5, N, D, seed=9, stream=3)
    full = eng.noise_get_host(5, N, D)
    assert abs(full.mean()) < 5 / np.sqrt(N * D)
    assert abs(full.var() - 1) < 0.01
    assert abs(np.mean(full ** 4) - 3) < 0.05
    # rows [1000, 1500) generated alone equal the same rows of the full matrix: sharding-invariant
    eng.noise_generate(6, 500, D, seed=9, stream=3, row_offset=1000)
    np.testing.assert_array_equal(eng.noise_get_host(6, 500, D), full[1000:1500])
    eng.noise_generate(6, N, D, seed=9, stream=4)
    assert np.abs(np.corrcoef(full.ravel(), eng.noise_get_host(6, N, D).ravel())[0, 1]) < 0.01


def test_philox_objective_matches_oracle_on_same_noise(vb):
    from viabel_amd import _lib
    D, N = 512, 2048
    approx = vb.MFGaussian(D, seed=7, rng='philox')
    model = vb.FunnelModel(D)
    theta = _theta(D, np.random.RandomState(2))
    eng = _lib.default_engine()
    for call in range(2):
        value, grad = vb.ExclusiveKL(approx, model, N)(theta)
        # the kernel generated its noise in registers (call c of the family = Philox stream c); the generator
        # kernel produces the same matrix
        eng.noise_generate(9, N, D, seed=7, stream=call)
        noise = eng.noise_get_host(9, N, D)
        ov, og = oobj.exclusive_kl(ofam.MFGaussian(D), omod.Funnel(D), theta, noise)
        assert G.rel_err(value, ov) < 1e-12 and G.rel_err(grad, og) < 1e-11
    # control-variate estimators take the same route
    value, grad = vb.ExclusiveKL(approx, model, N, hessian_approx_method='loo_diag_approx')(theta)
    eng.noise_generate(9, N, D, seed=7, stream=2)
    ov, og = oobj.rge_reduced(ofam.MFGaussian(D), omod.Funnel(D), theta, eng.noise_get_host(9, N, D), 'loo_diag_approx')
    assert G.rel_err(value, ov) < 1e-12 and G.rel_err(grad, og) < 1e-10


def test_async_pipeline_matches_sync(vb):
    from viabel_amd import _lib
    eng = _lib.default_engine()
    D, N = 1024, 4096
    model = vb.FunnelModel(D)
    eng.set_model(model.device_spec())
    theta = _theta(D, np.random.RandomState(4))
    for s in range(4):
        eng.noise_generate(10 + s, N, D, seed=1, stream=s)
    sync = [eng.elbo_grad_meanfield(10 + s, N, D, theta, _lib.FAMILY_MF_GAUSSIAN) for s in range(4)]
    for s in range(4):
        eng.elbo_grad_meanfield_async(10 + s, N, D, theta, _lib.FAMILY_MF_GAUSSIAN, rslot=s)
    eng.sync()
    for s in range(4):
        v, g = eng.result_get(s, 2 * D)
        assert v == sync[s][0]
        np.testing.assert_array_equal(g, sync[s][1])     # deterministic reductions: bitwise


@pytest.mark.parametrize('count,D,N', [(16, 1024, 4096), (5, 130, 77), (33, 64, 256)])
def test_batch_matches_single(vb, count, D, N):
    """`count` evaluations sharing one launch of each kernel == the same evaluations one by one."""
    from viabel_amd import _lib
    eng = _lib.default_engine()
    model = vb.FunnelModel(D, 3)
    eng.set_model(model.device_spec())
    rng = np.random.RandomState(count)
    thetas = np.stack([_theta(D, rng) for _ in range(count)])
    slots = [30 + (b % 20) for b in range(count)]
    for s in sorted(set(slots)):
        eng.noise_generate(s, N, D, seed=11, stream=s)
    single = [eng.elbo_grad_meanfield(slots[b], N, D, thetas[b], _lib.FAMILY_MF_GAUSSIAN,
                                      flags=_lib.FLAG_PATH_DERIV) for b in range(count)]
    eng.elbo_grad_meanfield_batch_async(slots, N, D, thetas, _lib.FAMILY_MF_GAUSSIAN,
                                        list(range(count)), flags=_lib.FLAG_PATH_DERIV)
    eng.sync()
    for b in range(count):
        v, g = eng.result_get(b, 2 * D)
        # a batch may pick a different row-block split than a single call: not bitwise
        assert abs(v - single[b][0]) < 1e-12 * abs(single[b][0])
        np.testing.assert_allclose(g, single[b][1], rtol=0, atol=1e-12 * np.max(np.abs(single[b][1])))
    noise = eng.noise_get_host(slots[0], N, D)
    ov, og = oobj.exclusive_kl(ofam.MFGaussian(D), omod.Funnel(D, 3), thetas[0], noise, True)
    v, g = eng.result_get(0, 2 * D)
    assert G.rel_err(v, ov) < 1e-12 and G.rel_err(g, og) < 1e-11


def test_result_slot_reuse_is_safe(vb):
    """Re-staging a result slot waits for the evaluation that last used it (different thetas)."""
    from viabel_amd import _lib
    eng = _lib.default_engine()
    D, N = 1024, 4096
    eng.set_model(vb.FunnelModel(D).device_spec())
    eng.noise_generate(40, N, D, seed=2, stream=0)
    rng = np.random.RandomState(8)
    thetas = [_theta(D, rng) for _ in range(6)]
    want = [eng.elbo_grad_meanfield(40, N, D, t, _lib.FAMILY_MF_GAUSSIAN) for t in thetas]
    got = []
    for t in thetas:            # same result slot every time, no explicit sync in between
        eng.elbo_grad_meanfield_async(40, N, D, t, _lib.FAMILY_MF_GAUSSIAN, rslot=7)
    v, g = eng.result_get(7, 2 * D)
    assert v == want[-1][0] and np.array_equal(g, want[-1][1])


def test_full_size_properties(vb):
    """C1 at full size: linearity in the sample axis (mean of halves = whole) and determinism."""
    from viabel_amd import _lib
    eng = _lib.default_engine()
    D, N = 1024, 4096
    model = vb.FunnelModel(D)
    eng.set_model(model.device_spec())
    theta = _theta(D, np.random.RandomState(6))
    eng.noise_generate(20, N, D, seed=3, stream=0)
    full = eng.noise_get_host(20, N, D)
    v, g = eng.elbo_grad_meanfield(20, N, D, theta, _lib.FAMILY_MF_GAUSSIAN)
    v2, g2 = eng.elbo_grad_meanfield(20, N, D, theta, _lib.FAMILY_MF_GAUSSIAN)
    assert v == v2 and np.array_equal(g, g2)
    eng.noise_set_host(21, full[:N // 2])
    eng.noise_set_host(22, full[N // 2:])
    va, ga = eng.elbo_grad_meanfield(21, N // 2, D, theta, _lib.FAMILY_MF_GAUSSIAN)
    vb_, gb = eng.elbo_grad_meanfield(22, N // 2, D, theta, _lib.FAMILY_MF_GAUSSIAN)
    assert abs(0.5 * (va + vb_) - v) < 1e-12 * abs(v)
    np.testing.assert_allclose(0.5 * (ga + gb), g, rtol=0, atol=1e-11 * np.max(np.abs(g)))


@pytest.mark.parametrize('df', [3.5, 8.0, 100.0])
def test_philox_student_t_noise(vb, df):
    """Device Student-t base noise (Philox + Bailey's polar method): distribution, shard invariance, and the
    MFStudentT objective on it equals the oracle on the same draws."""
    from scipy import stats
    from viabel_amd import _lib
    eng = _lib.default_engine()
    N, D = 4000, 101
    eng.noise_generate(5, N, D, seed=13, stream=2, kind=_lib.NOISE_STUDENT_T, df=df)
    full = eng.noise_get_host(5, N, D)
    assert np.all(np.isfinite(full))
    ks = stats.kstest(full.ravel()[::7], stats.t(df).cdf)
    assert ks.pvalue > 1e-4, ks
    assert abs(np.median(full)) < 0.01
    eng.noise_generate(6, 300, D, seed=13, stream=2, row_offset=700, kind=_lib.NOISE_STUDENT_T, df=df)
    np.testing.assert_array_equal(eng.noise_get_host(6, 300, D), full[700:1000])
    # objective in throughput mode == oracle on the read-back noise
    approx = vb.MFStudentT(D, df, seed=13, rng='philox')
    model = vb.FunnelModel(D, 5)
    theta = np.concatenate([0.1 * np.sin(np.arange(D)), -1.0 + 0.05 * np.cos(np.arange(D))])
    value, grad = vb.ExclusiveKL(approx, model, N)(theta)             # stream 0 of the family
    eng.noise_generate(7, N, D, seed=13, stream=0, kind=_lib.NOISE_STUDENT_T, df=df)
    noise = eng.noise_get_host(7, N, D)
    ov, og = oobj.exclusive_kl(ofam.MFStudentT(D, df), omod.Funnel(D, 5), theta, noise)
    assert abs(value - ov) <= 1e-12 * abs(ov)
    np.testing.assert_allclose(grad, og, rtol=0, atol=1e-11 * np.max(np.abs(og)))
    assert approx.sample(theta, 10).shape == (10, D)


def test_philox_many_rows(vb):
    """More rows than one launch's gridDim.y covers (vi_diagnostics draws 1e5 samples; 6e5 rows here)."""
    from viabel_amd import _lib
    eng = _lib.default_engine()
    N, D = 600000, 2
    eng.noise_generate(8, N, D, seed=1, stream=0)
    x = eng.noise_get_host(8, N, D)
    assert abs(x.mean()) < 0.01 and abs(x.var() - 1) < 0.01
    eng.noise_generate(9, 1000, D, seed=1, stream=0, row_offset=599000)
    np.testing.assert_array_equal(eng.noise_get_host(9, 1000, D), x[599000:])
