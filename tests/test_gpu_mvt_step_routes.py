"""GPU: the routes of the device-resident MultivariateT DIS step (vb_mvt.hip, round 5) against one another.

  VB_MVT_DIRECT=0        chain rule through U = E' L^-1 (an N x D x D product) instead of L^-T M on the Gram matrix of the
                         residuals -- a different grouping of the same sums: equal to rounding;
  VB_MVT_SIDE_INVERSE=0  the triangular inverse on the main stream instead of beside the sampling product: same kernels,
                         same inputs -- bit-identical;
  VB_MVT_FLAGSYNC=0      gradient by a device-to-host copy + stream synchronisation instead of mapped memory and a
                         polled completion word: the same numbers through another door -- bit-identical;
  VB_MVT_FUSED_ROWS=0    (round 6) log p / log prior and maha / log q / c_n by two row kernels instead of one pass over samples
                         and noise: every sum is formed in the same order -- bit-identical;
  VB_MVT_CHAIN=0         (round 6) the chain rule's D x D x D product as an MFMA launch + a pack kernel instead of one launch
                         over the lower 32 x 32 tiles: another order of the k sum -- equal to rounding;
  VB_MVT_CHAIN_FETCH=0   (round 6) the gradient gathered into mapped memory by a launch of its own behind the chain-rule kernel
                         instead of that kernel's own second stores: the same numbers -- bit-identical;
  VB_MVT_UNPACK=0        (round 6) the parameter read across the bus as the 32 x 32 tiles it is transposed in, and a prep launch on the
                         main stream for the row scales, instead of one coalesced pass that takes them along: the same values by
                         the same expressions -- bit-identical;
  VB_GRAM_XCD=0          (round 6) the Gram product's split workgroups in plain dispatch order instead of one split per XCD:
                         the same tiles and slabs, placed elsewhere -- bit-identical.
Parity with the oracle is tests/test_gpu_objectives.py / test_gpu_full_size.py (all switches at their defaults)."""
import os

import numpy as np
import pytest

import _golden as G
from oracle import families as ofam

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import viabel_amd
    from viabel_amd import _lib
    _lib.default_engine()
    return viabel_amd


def _step(vb, D, N, df, resample, batches, steps, seed):
    rng = np.random.RandomState(seed)
    approx = vb.MultivariateT(D, df, seed=8, rng='philox')
    model = vb.GaussianModel(0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D)))
    prior = np.concatenate([np.zeros(D), 0.3 * np.ones(D)])
    obj = vb.DISInclusiveKL(approx, model, N, ess_target=max(8, N // 8), temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, use_resampling=resample, num_resampling_batches=batches)
    A = rng.randn(D, D)
    theta = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(A @ A.T / D + 0.7 * np.eye(D))])
    out = []
    for k in range(steps):
        v, g = obj(theta)
        out.append((v, g.copy(), obj._eps, obj._ess))
        theta = theta + 0.003 * np.cos(np.arange(theta.size) + k)      # (a later step differentiates at a parameter that
                                                                       # is not the state's; the walk is route-independent)
    return out


def _with(env, call):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return call()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


SHAPES = [(5, 64), (50, 1000), (130, 513), (256, 4096), (300, 2048)]


@pytest.mark.parametrize('D,N', SHAPES)
@pytest.mark.parametrize('df', [7.0, 100.0])
@pytest.mark.parametrize('resample,batches', [(False, 1), (True, 1), (True, 3)])
def test_step_routes_agree(vb, D, N, df, resample, batches):
    call = lambda: _step(vb, D, N, df, resample, batches, steps=3, seed=D + N)
    base = call()
    for env, exact in (({'VB_MVT_SIDE_INVERSE': '0'}, True), ({'VB_MVT_FLAGSYNC': '0'}, True),
                       ({'VB_MVT_FUSED_ROWS': '0'}, True), ({'VB_MVT_CHAIN': '0'}, False),
                       ({'VB_MVT_CHAIN_FETCH': '0'}, True), ({'VB_GRAM_XCD': '0'}, True), ({'VB_MVT_UNPACK': '0'}, True),
                       ({'VB_MVT_DIRECT': '0'}, False),
                       ({'VB_MVT_DIRECT': '0', 'VB_MVT_SIDE_INVERSE': '0', 'VB_MVT_FLAGSYNC': '0'}, False)):
        other = _with(env, call)
        for (v0, g0, e0, s0), (v1, g1, e1, s1) in zip(base, other):
            assert e0 == e1 and s0 == s1, env
            if exact:
                assert v0 == v1, env
                np.testing.assert_array_equal(g0, g1, err_msg=str(env))
            else:
                assert abs(v0 - v1) <= 1e-13 * abs(v0), env
                assert G.rel_err(g0, g1) < 1e-11, (env, G.rel_err(g0, g1))


def test_gaussian_member_routes_agree(vb):
    """df = 0 of the same kernels (the full-covariance Gaussian member, no chi-square scales)."""
    D, N = 64, 1024
    rng = np.random.RandomState(3)

    def call():
        approx = vb.FullRankGaussian(D, seed=5, rng='philox')
        model = vb.GaussianModel(0.2 * rng.randn(D) * 0 + 0.1, np.ones(D))
        prior = np.concatenate([np.zeros(D), 0.3 * np.ones(D)])
        obj = vb.DISInclusiveKL(approx, model, N, ess_target=128, temper_prior=vb.MFGaussian(D), temper_prior_params=prior,
                                use_resampling=False)
        theta = approx.init_param() * 0.1
        return obj(theta)
    v0, g0 = call()
    v1, g1 = _with({'VB_MVT_DIRECT': '0', 'VB_MVT_SIDE_INVERSE': '0', 'VB_MVT_FLAGSYNC': '0'}, call)
    assert abs(v0 - v1) <= 1e-13 * abs(v0)
    assert G.rel_err(g0, g1) < 1e-11
