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


def test_hinted_seed_is_adopted_bit_for_bit_and_a_wrong_hint_is_harmless(env):
    """vb_noise_hint_seed: the host names the next request's seed (AlphaDivergence draws it from numpy's global generator
    every call, objectives.py:455); right hints, wrong hints and missing hints all give the values of a plain generation."""
    vb, eng, _lib = env
    slot, n, d = 13, 777, 96
    seeds = [11, 4000000000, 12, 12, 99, 5, 6]
    hints = [4000000000, 12, 12, 1234, None, 6, 7]      # right, right, right (a repeat), wrong, none, right, unused

    def run():
        out = []
        for seed, hint in zip(seeds, hints):
            eng.noise_generate(slot, n, d, seed=seed, stream=0)
            eng.chisq_generate(9.0, n, seed=seed, stream=0)
            if hint is not None:
                eng.noise_hint_seed(1 << slot, hint, with_chi=True)
            v, g = _blocking_call(vb, eng, _lib, slot, n, d)
            out.append((eng.noise_get_host(slot, n, d).copy(), eng.chisq_get_host(n).copy(), v, g.copy()))
        return out
    plain = _with('0', run)
    ahead = _with('1', run)
    for (e0, c0, v0, g0), (e1, c1, v1, g1) in zip(plain, ahead):
        np.testing.assert_array_equal(e0, e1)
        np.testing.assert_array_equal(c0, c1)
        assert v0 == v1
        np.testing.assert_array_equal(g0, g1)


@pytest.mark.parametrize('family', ['mf_gaussian', 'mf_student_t', 'fullrank', 'lowrank', 'multivariate_t'])
def test_alpha_divergence_with_the_next_seed_hinted(env, family):
    """AlphaDivergence in throughput mode hints the next call's seed (read off numpy's generator state without drawing):
    the calls return what they return without the look-ahead, and numpy's global stream is consumed as before."""
    vb, eng, _lib = env
    D, N = 24, 600
    rng = np.random.RandomState(5)
    model = vb.GaussianModel(0.3 * rng.randn(D), np.exp(0.2 * rng.randn(D)))
    make = {'mf_gaussian': lambda: vb.MFGaussian(D, rng='philox'), 'mf_student_t': lambda: vb.MFStudentT(D, 7, rng='philox'),
            'fullrank': lambda: vb.FullRankGaussian(D, rng='philox'), 'lowrank': lambda: vb.LRGaussian(D, k=3, rng='philox'),
            'multivariate_t': lambda: vb.MultivariateT(D, 9, rng='philox')}[family]

    def run():
        fam = make()
        obj = vb.AlphaDivergence(fam, model, N, 0.5)
        theta = fam.init_param()
        np.random.seed(17)
        out = []
        for i in range(6):
            if i == 3:
                np.random.randn(5)       # the caller draws in between: the hinted seed is still the generator's next word
            if i == 4:
                np.random.seed(3)        # ... and after a reseed the stale shadow is not adopted
            v, g = obj(theta)
            out.append((v, g.copy()))
        return out, np.random.randint(2 ** 32)
    (plain, tail0), (ahead, tail1) = _with('0', run), _with('1', run)
    assert tail0 == tail1
    for (v0, g0), (v1, g1) in zip(plain, ahead):
        assert v0 == v1
        np.testing.assert_array_equal(g0, g1)


@pytest.mark.parametrize('family', ['mf_gaussian', 'lowrank', 'multivariate_t'])
def test_hinted_seeds_are_actually_adopted(env, family):
    """The point of the hint, observed: in a plain loop of AlphaDivergence calls every call but the first adopts the
    look-ahead buffers (one noise matrix; two for the low-rank family; a noise matrix and the chi-square draws for the t
    family) -- if numpy's generator layout ever stops matching `_peek_next_randint`, results stay right and THIS fails."""
    vb, eng, _lib = env
    D, N, calls = 20, 500, 12
    rng = np.random.RandomState(6)
    model = vb.GaussianModel(0.3 * rng.randn(D), np.exp(0.2 * rng.randn(D)))
    fam, per_call = {'mf_gaussian': (lambda: vb.MFGaussian(D, rng='philox'), 1), 'lowrank': (lambda: vb.LRGaussian(D, k=3, rng='philox'), 2),
                     'multivariate_t': (lambda: vb.MultivariateT(D, 9, rng='philox'), 2)}[family]
    approx = fam()
    obj = vb.AlphaDivergence(approx, model, N, 0.5)
    theta = approx.init_param()
    np.random.seed(2)
    obj(theta)
    g0, a0 = eng.noise_ahead_stats()
    for _ in range(calls):
        obj(theta)
    g1, a1 = eng.noise_ahead_stats()
    assert a1 - a0 == per_call * calls, (g1 - g0, a1 - a0)
    assert g1 - g0 == per_call * calls


def test_stream_walks_are_adopted(env):
    """ExclusiveKL / DISInclusiveKL count their Philox streams 0, 1, 2, ...: from the third call on every request adopts."""
    vb, eng, _lib = env
    D, N, calls = 16, 400, 10
    approx = vb.FullRankGaussian(D, rng='philox')
    obj = vb.ExclusiveKL(approx, vb.GaussianModel(np.zeros(D), np.ones(D)), N)
    theta = approx.init_param()
    for _ in range(3):
        obj(theta)
    g0, a0 = eng.noise_ahead_stats()
    for _ in range(calls):
        obj(theta)
    g1, a1 = eng.noise_ahead_stats()
    assert a1 - a0 == calls and g1 - g0 == calls
