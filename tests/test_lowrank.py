"""LRGaussian (viabel/approximations.py:610-731): oracle and host-side family against the reference's own forward
code (tests/golden/lowrank_*.npz), and the HIP ExclusiveKL path against the reference closure / the oracle.

Tolerances: forward values 1e-11 relative; device value 1e-12, gradient 1e-11 relative to max|grad|; 2e-7 against
the reference's finite-difference gradients.
"""
import numpy as np
import pytest

import _golden as G
from oracle import families as ofam
from oracle import models as omod
from oracle import objectives as oobj

FIXTURES = G.fixtures('lowrank_')


def _models(fx, tag, vb=None):
    if str(fx[tag + 'model_kind']) == 'gauss_diag':
        o = omod.GaussDiag(fx[tag + 'model_mean'], fx[tag + 'model_stdev'])
        p = vb.GaussianModel(fx[tag + 'model_mean'], fx[tag + 'model_stdev']) if vb else None
    else:
        D = int(fx['dim'])
        o = omod.Funnel(D, int(fx[tag + 'model_scale_index']), float(fx[tag + 'model_log_sigma_stdev']))
        p = vb.FunnelModel(D, int(fx[tag + 'model_scale_index']), float(fx[tag + 'model_log_sigma_stdev'])) if vb else None
    return o, p


@pytest.mark.parametrize('path', FIXTURES, ids=G.ids(FIXTURES))
def test_oracle_and_host_family_match_reference(path):
    import viabel_amd as vb
    fx = G.load(path)
    D, k, seed, N = int(fx['dim']), int(fx['k']), int(fx['seed']), int(fx['n'])
    th0, th1, x = fx['theta0'], fx['theta1'], fx['samples']
    orc = ofam.LRGaussian(D, k)
    fam = vb.LRGaussian(D, seed=seed, k=k)
    np.testing.assert_allclose(fam.init_param(), fx['init_param'], rtol=1e-15)   # same D k draws
    np.testing.assert_allclose(fam.sample(th0, N), x, rtol=1e-13, atol=1e-14)    # next: z then eps
    for f in (orc, fam):
        np.testing.assert_allclose(f.log_density(th1, x), fx['log_density'], rtol=1e-11)
        np.testing.assert_allclose(f.entropy(th0), float(fx['entropy']), rtol=1e-12)
        np.testing.assert_allclose(f.kl(th0, th1), float(fx['kl']), rtol=1e-10)
    np.testing.assert_allclose(orc.sample_from_noise(th0, (fx['noise_z'], fx['noise_eps'])), x, rtol=1e-13, atol=1e-14)
    mean, cov = fam.mean_and_cov(th0)
    np.testing.assert_allclose(mean, fx['mean'], rtol=0, atol=0)
    np.testing.assert_allclose(cov, fx['cov'], rtol=1e-13)
    np.testing.assert_allclose(fam.pth_moment(th0, 2), float(fx['pth2']), rtol=1e-12)
    np.testing.assert_allclose(fam.pth_moment(th0, 4), float(fx['pth4']), rtol=1e-12)
    assert fam.var_param_dim == 2 * D + D * k and fam.supports_kl and fam.supports_entropy
    assert fam.log_density(th1, x[0]).shape == (1,)
    for m in (0, 1):
        tag = 'm%d_' % m
        omodel, _ = _models(fx, tag)
        ov, og = oobj.exclusive_kl(orc, omodel, th0, (fx[tag + 'noise_z'], fx[tag + 'noise_eps']))
        assert abs(ov - float(fx[tag + 'value'])) <= 1e-12 * abs(float(fx[tag + 'value']))
        np.testing.assert_allclose(og, fx[tag + 'grad'], rtol=0, atol=1e-13 * np.max(np.abs(og)))
        np.testing.assert_allclose(og, fx[tag + 'grad_fd'], rtol=0, atol=2e-7 * np.max(np.abs(og)))
        pv, pg = oobj.exclusive_kl(orc, omodel, th0, (fx[tag + 'noise_z'], fx[tag + 'noise_eps']), True)
        assert abs(pv - float(fx[tag + 'pd_value'])) <= 1e-12 * abs(pv)
        np.testing.assert_allclose(pg, fx[tag + 'pd_grad_fd'], rtol=0, atol=2e-7 * np.max(np.abs(pg)))


@pytest.mark.gpu
@pytest.mark.parametrize('path', FIXTURES, ids=G.ids(FIXTURES))
def test_device_exclusive_kl_matches_reference(path):
    import viabel_amd as vb
    fx = G.load(path)
    D, k, seed, N = int(fx['dim']), int(fx['k']), int(fx['seed']), int(fx['n'])
    for m in (0, 1):
        tag = 'm%d_' % m
        _, model = _models(fx, tag, vb)
        objective = vb.ExclusiveKL(vb.LRGaussian(D, seed=seed, k=k), model, N)
        value, grad = objective(fx['theta0'])
        ref_v, ref_g = float(fx[tag + 'value']), fx[tag + 'grad']
        assert abs(value - ref_v) <= 1e-12 * abs(ref_v), (value, ref_v)
        np.testing.assert_allclose(grad, ref_g, rtol=0, atol=1e-11 * np.max(np.abs(ref_g)))
        np.testing.assert_allclose(grad, fx[tag + 'grad_fd'], rtol=0, atol=2e-7 * np.max(np.abs(ref_g)))
        # the reference closure with use_path_deriv=True (objectives.py:156-159)
        objective = vb.ExclusiveKL(vb.LRGaussian(D, seed=seed, k=k), model, N, use_path_deriv=True)
        value, grad = objective(fx['theta0'])
        ref_v, ref_g = float(fx[tag + 'pd_value']), fx[tag + 'pd_grad']
        assert abs(value - ref_v) <= 1e-11 * abs(ref_v), (value, ref_v)
        np.testing.assert_allclose(grad, ref_g, rtol=0, atol=1e-10 * np.max(np.abs(ref_g)))
        np.testing.assert_allclose(grad, fx[tag + 'pd_grad_fd'], rtol=0, atol=2e-7 * np.max(np.abs(ref_g)))


@pytest.mark.gpu
@pytest.mark.parametrize('target', ['gauss_diag', 'funnel'])
@pytest.mark.parametrize('D,k,N', [(1024, 8, 4096), (1000, 16, 777), (130, 3, 1), (257, 5, 1030), (64, 1, 300)])
def test_device_exclusive_kl_matches_oracle(target, D, k, N):
    """BASELINE-sized and ragged shapes (odd D, N not a multiple of the row tile, N = 1, k = 1 .. 16)."""
    import viabel_amd as vb
    rng = np.random.RandomState(D + k)
    if target == 'gauss_diag':
        mean, sd = rng.randn(D), np.exp(0.3 * rng.randn(D))
        model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    else:
        model, omodel = vb.FunnelModel(D, D // 3), omod.Funnel(D, D // 3)
    fam = vb.LRGaussian(D, seed=4, k=k)
    theta = fam.pack(0.2 * rng.randn(D), -1.0 + 0.1 * rng.randn(D), 0.1 * rng.randn(D, k))
    value, grad = vb.ExclusiveKL(fam, model, N)(theta)
    noise = ofam.LRGaussian(D, k).draw_noise(np.random.RandomState(4), N)
    ov, og = oobj.exclusive_kl(ofam.LRGaussian(D, k), omodel, theta, noise)
    assert abs(value - ov) <= 1e-12 * abs(ov), (value, ov)
    np.testing.assert_allclose(grad, og, rtol=0, atol=1e-11 * np.max(np.abs(og)))
    value, grad = vb.ExclusiveKL(vb.LRGaussian(D, seed=4, k=k), model, N, use_path_deriv=True)(theta)
    ov, og = oobj.exclusive_kl(ofam.LRGaussian(D, k), omodel, theta, noise, True)
    assert abs(value - ov) <= 1e-11 * abs(ov), (value, ov)
    np.testing.assert_allclose(grad, og, rtol=0, atol=1e-10 * np.max(np.abs(og)))


@pytest.mark.gpu
@pytest.mark.parametrize('target', ['gauss_diag', 'funnel'])
@pytest.mark.parametrize('D,k,N', [(1024, 32, 4096), (1024, 64, 4096), (300, 17, 777), (130, 40, 33), (64, 100, 256)])
def test_device_exclusive_kl_any_rank_matches_oracle(target, D, k, N):
    """Ranks beyond the streaming kernel's 16 (the reference's LRGaussian has no limit, approximations.py:610-644):
    samples, G' Z and the column sums from MFMA GEMMs (vb_elbo_sums_lowrank), entropy terms on the host."""
    import viabel_amd as vb
    rng = np.random.RandomState(D + k)
    if target == 'gauss_diag':
        mean, sd = rng.randn(D), np.exp(0.3 * rng.randn(D))
        model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    else:
        model, omodel = vb.FunnelModel(D, D // 3), omod.Funnel(D, D // 3)
    fam = vb.LRGaussian(D, seed=4, k=k)
    theta = fam.pack(0.2 * rng.randn(D), -1.0 + 0.1 * rng.randn(D), 0.05 * rng.randn(D, k))
    value, grad = vb.ExclusiveKL(fam, model, N)(theta)
    noise = ofam.LRGaussian(D, k).draw_noise(np.random.RandomState(4), N)
    ov, og = oobj.exclusive_kl(ofam.LRGaussian(D, k), omodel, theta, noise)
    assert abs(value - ov) <= 1e-12 * abs(ov), (value, ov)
    np.testing.assert_allclose(grad, og, rtol=0, atol=1e-11 * np.max(np.abs(og)))
    value, grad = vb.ExclusiveKL(vb.LRGaussian(D, seed=4, k=k), model, N, use_path_deriv=True)(theta)
    ov, og = oobj.exclusive_kl(ofam.LRGaussian(D, k), omodel, theta, noise, True)
    assert abs(value - ov) <= 1e-11 * abs(ov), (value, ov)
    np.testing.assert_allclose(grad, og, rtol=0, atol=1e-10 * np.max(np.abs(og)))


@pytest.mark.gpu
def test_lowrank_rejects_unsupported():
    import viabel_amd as vb
    model = vb.GaussianModel(np.zeros(4), np.ones(4))
    with pytest.raises(NotImplementedError):
        vb.ExclusiveKL(vb.LRGaussian(4, k=2), model, 10, hessian_approx_method='full')
    with pytest.raises(NotImplementedError):
        vb.DISInclusiveKL(vb.LRGaussian(4, k=65), model, 10, ess_target=5, temper_prior=vb.MFGaussian(4),
                          temper_prior_params=np.zeros(8))(np.zeros(4 * 2 + 4 * 65))
    with pytest.raises(ValueError):
        vb.ExclusiveKL(vb.LRGaussian(4, k=2), model, 10)(np.zeros(3))


@pytest.mark.gpu
def test_lowrank_philox_mode_matches_oracle_on_the_same_noise():
    """rng='philox': both noise blocks are generated on the device (streams 2 c / 2 c + 1 of call c); read back and
    fed to the oracle they reproduce the device objective, and the host sample() sees the same convention."""
    import viabel_amd as vb
    from viabel_amd import _lib
    D, k, N = 96, 4, 500
    rng = np.random.RandomState(2)
    mean, sd = rng.randn(D), np.exp(0.3 * rng.randn(D))
    fam = vb.LRGaussian(D, seed=11, k=k, rng='philox')
    theta = fam.pack(0.2 * rng.randn(D), -1.0 + 0.1 * rng.randn(D), 0.1 * rng.randn(D, k))
    obj = vb.ExclusiveKL(fam, vb.GaussianModel(mean, sd), N)
    assert obj.supports_device_fit()
    for call in range(2):
        value, grad = obj(theta)
        eng = _lib.default_engine()
        eps = eng.noise_get_host(0, N, D)
        z = eng.noise_get_host(3, N, k)
        ov, og = oobj.exclusive_kl(ofam.LRGaussian(D, k), omod.GaussDiag(mean, sd), theta, (z, eps))
        assert abs(value - ov) <= 1e-12 * abs(ov)
        np.testing.assert_allclose(grad, og, rtol=0, atol=1e-11 * np.max(np.abs(og)))
        # the same call index through the generator directly
        eng.noise_generate(40, N, D, 11, 2 * call)
        eng.noise_generate(41, N, k, 11, 2 * call + 1)
        np.testing.assert_array_equal(eng.noise_get_host(40, N, D), eps)
        np.testing.assert_array_equal(eng.noise_get_host(41, N, k), z)
    x = fam.sample(theta, 2000)
    assert x.shape == (2000, D) and abs(x.mean() - theta[:D].mean()) < 0.1


@pytest.mark.gpu
def test_lowrank_path_terms_plain_and_sharded_context():
    """vb_lowrank_path_terms against numpy second moments, and through a context with a (one-rank) communicator."""
    import viabel_amd  # noqa: F401
    from viabel_amd import _lib
    D, k, N = 70, 5, 513
    rng = np.random.RandomState(3)
    sw = rng.randn(D, k)
    plain = _lib.default_engine()
    comm = _lib.Engine(plain.device)
    comm.comm_init(_lib.Engine.comm_unique_id(), 1, 0)
    try:
        outs = []
        for eng in (plain, comm):
            eng.noise_generate(50, N, D, seed=2, stream=0)
            eng.noise_generate(51, N, k, seed=2, stream=1)
            outs.append(eng.lowrank_path_terms(50, 51, N, D, k, sw))
        E, Z = plain.noise_get_host(50, N, D), plain.noise_get_host(51, N, k)
        T = np.concatenate([Z, E @ sw], axis=1)
        want = (E.T @ T, T.T @ T, E.sum(0), (E * E).sum(0), T.sum(0))
        for got, ref in zip(outs[0], want):
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12 * max(1.0, np.max(np.abs(ref))))
        for a, b in zip(outs[0], outs[1]):
            np.testing.assert_array_equal(b, a)
    finally:
        comm.comm_destroy()
        comm.close()
