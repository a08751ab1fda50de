"""GPU: look-ahead generation of Philox noise (vb_api.hip, noise_prefetch).  After two requests that walked the stream
index in equal steps a blocking call generates the NEXT request's values behind its last kernel, and the matching
vb_noise_generate / vb_chisq_generate adopts that buffer.  Counter-based streams: the values -- and everything computed
from them -- must be the ones a plain generation gives (VB_NOISE_AHEAD=0), bit for bit, whatever the caller does next:
the predicted request, another stream, another shape, a host-set matrix, numpy's streams."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import viabel_amd as vb
    from viabel_amd import _lib
    return vb, _lib.default_engine(), _lib


def _with(value, call):
    old = os.environ.get('VB_NOISE_AHEAD')
    os.environ['VB_NOISE_AHEAD'] = value
    try:
        return call()
    finally:
        if old is None:
            del os.environ['VB_NOISE_AHEAD']
        else:
            os.environ['VB_NOISE_AHEAD'] = old


def _blocking_call(vb, eng, _lib, slot, n, d):
    """Any blocking entry point that ends in fetch_blocking: a small dense-family evaluation on the slot's noise."""
    model = vb.GaussianModel(np.zeros(d), np.ones(d))
    eng.set_model(model.device_spec())
    fam = vb.FullRankGaussian(d)
    theta = fam.pack(np.zeros(d), np.eye(d))
    return eng.elbo_grad_fullrank(slot, n, d, theta)


@pytest.mark.parametrize('n,d', [(64, 8), (1000, 130), (4096, 256)])
def test_adopted_noise_equals_generated_noise(env, n, d):
    vb, eng, _lib = env
    slot = 11

    def run(streams):
        out = []
        for st in streams:
            eng.noise_generate(slot, n, d, seed=7, stream=st)
            eng.chisq_generate(9.0, n, seed=7, stream=st)
            v, g = _blocking_call(vb, eng, _lib, slot, n, d)
            out.append((eng.noise_get_host(slot, n, d).copy(), eng.chisq_get_host(n).copy(), v, g.copy()))
        return out
    # equal steps (the look-ahead engages from the third request), a jump, another step size, a repeat
    streams = [3, 4, 5, 6, 7, 20, 22, 24, 26, 26, 27]
    plain = _with('0', lambda: run(streams))
    ahead = _with('1', lambda: run(streams))
    for (e0, c0, v0, g0), (e1, c1, v1, g1) in zip(plain, ahead):
        np.testing.assert_array_equal(e0, e1)
        np.testing.assert_array_equal(c0, c1)
        assert v0 == v1
        np.testing.assert_array_equal(g0, g1)


def test_other_writers_end_the_history(env):
    vb, eng, _lib = env
    slot, n, d = 12, 512, 64
    rng = np.random.RandomState(0)

    def run():
        out = []
        for st in range(4):
            eng.noise_generate(slot, n, d, seed=3, stream=st)
            _blocking_call(vb, eng, _lib, slot, n, d)
        host = rng.randn(n, d)
        eng.noise_set_host(slot, host)                       # a host matrix in between
        out.append(eng.noise_get_host(slot, n, d).copy())
        _blocking_call(vb, eng, _lib, slot, n, d)
        eng.noise_generate(slot, n, d, seed=3, stream=4)     # what the look-ahead had predicted before the host write
        out.append(eng.noise_get_host(slot, n, d).copy())
        eng.noise_generate(slot, 2 * n, d, seed=3, stream=5)      # another shape
        out.append(eng.noise_get_host(slot, 2 * n, d).copy())
        return out
    rng = np.random.RandomState(0)
    plain = _with('0', run)
    rng = np.random.RandomState(0)
    ahead = _with('1', run)
    for a, b in zip(plain, ahead):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('resample', [False, True])
def test_objective_sequence_is_unchanged(env, resample):
    """The C3-shaped objective (chi-square + normal request per call, one more stream per call with resampling) over a few
    calls: value, gradient, eps, ESS with the look-ahead equal the ones without."""
    vb, eng, _lib = env
    D, N = 64, 4096

    def run():
        rng = np.random.RandomState(5)
        approx = vb.MultivariateT(D, 30.0, seed=4, rng='philox')
        model = vb.GaussianModel(0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D)))
        prior = np.concatenate([np.zeros(D), 0.3 * np.ones(D)])
        obj = vb.DISInclusiveKL(approx, model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D), temper_prior_params=prior,
                                use_resampling=resample)
        theta = approx.init_param() * 0.3
        out = []
        for k in range(6):
            v, g = obj(theta)
            out.append((v, g.copy(), obj._eps, obj._ess))
            theta = theta + 0.002 * np.cos(np.arange(theta.size) + k)
        return out
    plain = _with('0', run)
    ahead = _with('1', run)
    for (v0, g0, e0, s0), (v1, g1, e1, s1) in zip(plain, ahead):
        assert (v0, e0, s0) == (v1, e1, s1)
        np.testing.assert_array_equal(g0, g1)
