"""GPU: small results and parameters through mapped memory + a polled completion word (fetch_blocking / push_small,
vb_api.hip) against the plain copies + stream synchronisation they replace (VB_FETCH_FLAGSYNC=0): the same numbers through
another door -- bit-identical -- on every family of blocking entry points that uses them."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import viabel_amd as vb
    from viabel_amd import _lib
    return vb, _lib.default_engine(), _lib


def _both(call):
    old = os.environ.get('VB_FETCH_FLAGSYNC')
    try:
        os.environ['VB_FETCH_FLAGSYNC'] = '0'
        plain = call()
        os.environ['VB_FETCH_FLAGSYNC'] = '1'
        flagged = call()
    finally:
        if old is None:
            os.environ.pop('VB_FETCH_FLAGSYNC', None)
        else:
            os.environ['VB_FETCH_FLAGSYNC'] = old
    return plain, flagged


def _same(a, b):
    if isinstance(a, (tuple, list)):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            _same(x, y)
    elif isinstance(a, np.ndarray):
        np.testing.assert_array_equal(a, b)
    else:
        assert a == b or (a != a and b != b)


@pytest.mark.parametrize('d,n', [(3, 17), (64, 1000), (300, 513)])
def test_fullrank_blocking_call(env, d, n):
    vb, eng, _lib = env
    rng = np.random.RandomState(d)
    A = rng.randn(d, d)
    model = vb.CorrelatedGaussianModel(rng.randn(d), covariance=A @ A.T / d + np.eye(d))
    fam = vb.FullRankGaussian(d)
    theta = fam.pack(0.1 * rng.randn(d), np.exp(-1.0) * np.eye(d) + 0.01 * np.tril(rng.randn(d, d)))

    def call():
        eng.set_model(model.device_spec())
        eng.noise_generate(5, n, d, seed=3, stream=1)
        return eng.elbo_grad_fullrank(5, n, d, theta)
    _same(*_both(call))


@pytest.mark.parametrize('k', [4, 32])
def test_lowrank_blocking_calls(env, k):
    vb, eng, _lib = env
    D, N = 200, 1000
    rng = np.random.RandomState(k)
    fam = vb.LRGaussian(D, k=k)
    theta = fam.pack(0.1 * rng.randn(D), -np.ones(D), 0.05 * rng.randn(D, k))

    def call():
        eng.set_model(vb.FunnelModel(D).device_spec())
        eng.noise_generate(0, N, D, seed=1, stream=0)
        eng.noise_generate(1, N, k, seed=2, stream=0)
        return (eng.elbo_sums_lowrank(0, 1, N, D, k, theta) if k > 16 else eng.elbo_grad_lowrank(0, 1, N, D, k, theta))
    _same(*_both(call))


def test_psis_and_dis_and_t_family(env):
    vb, eng, _lib = env
    from viabel_amd._psis import psislw
    rng = np.random.RandomState(1)
    lw = 2.0 * rng.standard_t(3.0, 5000)
    _same(*_both(lambda: psislw(lw)))
    D, N = 40, 2048

    def dis():
        r = np.random.RandomState(7)
        model = vb.GaussianModel(0.3 + 0.3 * r.randn(D), np.exp(0.2 * r.randn(D)))
        prior = np.zeros(2 * D)
        obj = vb.DISInclusiveKL(vb.MFGaussian(D, seed=11, rng='philox'), model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                                temper_prior_params=prior, use_resampling=False)
        v, g = obj(prior + 0.02 * r.randn(2 * D))
        return v, g, obj._eps, obj._ess
    _same(*_both(dis))

    def mvt():
        r = np.random.RandomState(9)
        approx = vb.MultivariateT(D, 30.0, seed=4, rng='philox')
        model = vb.GaussianModel(0.2 * r.randn(D), np.exp(0.1 * r.randn(D)))
        theta = approx.init_param() * 0.3
        v1, g1 = vb.ExclusiveKL(approx, model, N)(theta)
        np.random.seed(3)
        v2, g2 = vb.AlphaDivergence(vb.MultivariateT(D, 30.0, seed=4, rng='philox'), model, N, 0.5)(theta)
        return v1, g1, v2, g2
    _same(*_both(mvt))
