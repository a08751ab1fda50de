"""GPU: the routes of the device-resident MultivariateT DIS step (vb_mvt.hip, round 5) against one another.

  VB_MVT_DIRECT=0        chain rule through U = E' L^-1 (an N x D x D product) instead of L^-T M on the Gram matrix of the
                         residuals -- a different grouping of the same sums: equal to rounding;
  VB_MVT_SIDE_INVERSE=0  the triangular inverse on the main stream instead of beside the sampling product: same kernels,
                         same inputs -- bit-identical;
  VB_MVT_FLAGSYNC=0      gradient by a device-to-host copy + stream synchronisation instead of mapped memory and a
                         polled completion word: the same numbers through another door -- bit-identical;
  VB_MVT_FUSED_ROWS=0    (round 6) log p / log prior and maha / log q / c_n by two row kernels instead of one pass over samples
                         and noise: every sum is formed in the same order -- bit-identical (compared with VB_MVT_EPI_ROWS=0 on
                         both sides: where the epilogue route applies it replaces both);
  VB_MVT_CHAIN=0         (round 6) the chain rule's D x D x D product as an MFMA launch + a pack kernel instead of one launch
                         over the lower 32 x 32 tiles: another order of the k sum -- equal to rounding;
  VB_MVT_CHAIN_FETCH=0   (round 6) the gradient gathered into mapped memory by a launch of its own behind the chain-rule kernel
                         instead of that kernel's own second stores: the same numbers -- bit-identical;
  VB_MVT_UNPACK=0        (round 6) the parameter read across the bus as the 32 x 32 tiles it is transposed in, and a prep launch on the
                         main stream for the row scales, instead of one coalesced pass that takes them along: the same values by
                         the same expressions -- bit-identical;
  VB_MVT_EPI_ROWS=0      (round 6) log p / log prior / maha by a pass over samples and noise instead of the sampling product's
                         epilogue + the noise rows' norms: other groupings of the same sums -- equal to rounding, the tempering
                         walk's outcome included (eps, ESS: compared to rounding as well);
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
    # the two row-pass routes against each other (bit for bit), the epilogue route switched off in both
    pass_fused = _with({'VB_MVT_EPI_ROWS': '0'}, call)
    pass_split = _with({'VB_MVT_EPI_ROWS': '0', 'VB_MVT_FUSED_ROWS': '0'}, call)
    for (v0, g0, e0, s0), (v1, g1, e1, s1) in zip(pass_fused, pass_split):
        assert e0 == e1 and s0 == s1 and v0 == v1
        np.testing.assert_array_equal(g0, g1)
    for env, exact in (({'VB_MVT_SIDE_INVERSE': '0'}, True), ({'VB_MVT_FLAGSYNC': '0'}, True),
                       ({'VB_MVT_FUSED_ROWS': '0'}, None), ({'VB_MVT_CHAIN': '0'}, False),
                       ({'VB_MVT_CHAIN_FETCH': '0'}, True), ({'VB_GRAM_XCD': '0'}, True), ({'VB_MVT_UNPACK': '0'}, True),
                       ({'VB_MVT_EPI_ROWS': '0'}, None),
                       ({'VB_MVT_DIRECT': '0'}, False),
                       ({'VB_MVT_DIRECT': '0', 'VB_MVT_SIDE_INVERSE': '0', 'VB_MVT_FLAGSYNC': '0'}, False)):
        other = _with(env, call)
        for (v0, g0, e0, s0), (v1, g1, e1, s1) in zip(base, other):
            if exact is None:      # (log q itself differs in the last bits: so may the walk's last levels)
                assert abs(e0 - e1) <= 1e-10 * max(abs(e0), 1e-300) and abs(s0 - s1) <= 1e-8 * abs(s0), (env, e0, e1, s0, s1)
                assert abs(v0 - v1) <= 1e-10 * abs(v0), env
                assert G.rel_err(g0, g1) < 1e-9, (env, G.rel_err(g0, g1))
                continue
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


def test_epilogue_rows_route_does_not_depend_on_history(vb):
    """Round 6: the C3-type call takes log p / log prior out of the sampling product's epilogue and the Mahalanobis terms from
    the noise rows' norms (VB_MVT_EPI_ROWS).  The norms come with the Philox normals once a reader has asked for them, and
    from a pass over the noise before that -- the same bits either way: the first call of a fresh engine (nobody has asked) and
    a later one (norms generated with the noise, look-ahead included) return identical results for identical inputs."""
    from viabel_amd import _lib
    D, N, df = 48, 2048, 9.0
    rng = np.random.RandomState(5)
    model_args = (0.3 * rng.randn(D), np.exp(0.2 * rng.randn(D)))
    prior = np.concatenate([0.1 * rng.randn(D), 0.4 + 0.1 * rng.rand(D)])
    A = rng.randn(D, D)
    theta = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(A @ A.T / D + 0.5 * np.eye(D))])

    def calls(k):
        approx = vb.MultivariateT(D, df, seed=3, rng='philox')
        obj = vb.DISInclusiveKL(approx, vb.GaussianModel(*model_args), N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                                temper_prior_params=prior, use_resampling=False)
        return [obj(theta + 0.01 * i) for i in range(k)]
    old = _lib.default_engine()
    fresh = _lib.Engine(0)
    _lib.set_default_engine(fresh)
    try:
        first = calls(5)          # call 0: norms by a pass; later ones: generated with the (look-ahead) noise
        again = calls(5)          # the same requests on an engine whose slot has been asked already
        for (v0, g0), (v1, g1) in zip(first, again):
            assert v0 == v1
            np.testing.assert_array_equal(g0, g1)
        off = _with({'VB_MVT_EPI_ROWS': '0'}, lambda: calls(2))
        for (v0, g0), (v1, g1) in zip(first, off):
            assert abs(v0 - v1) <= 1e-10 * abs(v0) and G.rel_err(g0, g1) < 1e-9
    finally:
        _lib.set_default_engine(old)
        fresh.close()


@pytest.mark.parametrize('N,exact', [(4096, True), (16384, True), (40000, False)])
def test_psis_weights_in_weights_out(vb, N, exact):
    """Round 6: the smoothing kernel reads the weights and writes them back itself (PsisWeightsIo) instead of a prep and an apply
    launch around it (VB_PSIS_FUSED_IO=0).  The same values; the same bits while a thread owns one weight (N <= 16 384: the
    prep kernel's slices and summation order), the total of the weights in another grouping beyond."""
    D, df = 32, 9.0
    rng = np.random.RandomState(N % 97)
    model = vb.GaussianModel(0.3 * rng.randn(D), np.exp(0.2 * rng.randn(D)))
    prior = np.concatenate([np.zeros(D), 0.3 * np.ones(D)])
    A = rng.randn(D, D)
    theta = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(A @ A.T / D + 0.6 * np.eye(D))])

    def call():
        obj = vb.DISInclusiveKL(vb.MultivariateT(D, df, seed=4, rng='philox'), model, N, ess_target=N // 8,
                                temper_prior=vb.MFGaussian(D), temper_prior_params=prior, use_resampling=False, psis_smooth=True)
        out = [obj(theta + 0.01 * i) for i in range(3)]
        return out, obj._khat
    (a, ka), (b, kb) = call(), _with({'VB_PSIS_FUSED_IO': '0'}, call)
    assert np.isfinite(ka) and ka == kb
    for (v0, g0), (v1, g1) in zip(a, b):
        if exact:
            assert v0 == v1
            np.testing.assert_array_equal(g0, g1)
        else:
            assert abs(v0 - v1) <= 1e-13 * abs(v0) and G.rel_err(g0, g1) < 1e-12


@pytest.mark.parametrize('D,N', [(48, 77), (80, 1000), (144, 4100), (256, 1024), (512, 2048)])
@pytest.mark.parametrize('df', [0.0, 6.0])
def test_epilogue_rows_route_over_tile_shapes(vb, D, N, df):
    """The sampling product's row-summing epilogue (EpiRowSums) where the column blocks are ragged (D not a multiple of 64), the
    row tiles are (N not a multiple of 64), one block and eight blocks wide -- and for the Gaussian member (df = 0: no row
    scales): against the row-pass route, to rounding."""
    rng = np.random.RandomState(D + N)
    model = vb.GaussianModel(0.3 * rng.randn(D), np.exp(0.2 * rng.randn(D)))
    prior = np.concatenate([0.1 * rng.randn(D), 0.4 + 0.1 * rng.rand(D)])
    A = rng.randn(D, D)
    theta = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(A @ A.T / D + 0.5 * np.eye(D))])

    def call():
        approx = vb.MultivariateT(D, df, seed=6, rng='philox') if df > 0 else vb.FullRankGaussian(D, seed=6, rng='philox')
        obj = vb.DISInclusiveKL(approx, model, N, ess_target=max(8, N // 8), temper_prior=vb.MFGaussian(D),
                                temper_prior_params=prior, use_resampling=False)
        return [obj(theta + 0.01 * i) + (obj._eps, obj._ess) for i in range(2)]
    from viabel_amd import _lib
    eng = _lib.default_engine()
    before = eng.mvt_route_stats()
    on = call()
    mid = eng.mvt_route_stats()
    off = _with({'VB_MVT_EPI_ROWS': '0'}, call)
    after = eng.mvt_route_stats()
    assert mid[0] == before[0] + 2 and after[0] == mid[0]      # both refreshes took the epilogue route; none with the switch off
    small = (D + D * (D + 1) // 2) * 8 + 128 <= (1 << 20)       # (the mapped staging buffer takes results up to 1 MB)
    assert mid[1] == before[1] + (2 if small else 0)           # ... and the chain-rule kernel brought the gradient home
    for (v0, g0, e0, s0), (v1, g1, e1, s1) in zip(on, off):
        assert abs(e0 - e1) <= 1e-10 * max(abs(e0), 1e-300) and abs(s0 - s1) <= 1e-8 * abs(s0)
        assert abs(v0 - v1) <= 1e-10 * abs(v0) and G.rel_err(g0, g1) < 1e-9


def test_newton_schulz_hinted_step_count_is_bit_identical(vb):
    """Round 6: the symmetric root's Newton-Schulz iteration (reference-identical mode, approximations.py:348) launches as many
    steps at once as the previous root of that size needed and replays the step-by-step control on their residuals afterwards
    (one synchronisation instead of three); the states are the same states, so results are bit-identical to the step-by-step
    control (VB_NS_HINT=0) -- over parameters whose roots need more steps than the hint, fewer by one, fewer by many (the
    restart) and the same number."""
    D, N, df = 64, 4096, 7.0
    rng = np.random.RandomState(17)
    model = vb.GaussianModel(0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D)))
    A = rng.randn(D, D)
    mu = 0.1 * rng.randn(D)
    covs = [A @ A.T / D + 0.5 * np.eye(D),                 # moderate
            A @ A.T / D + 0.5 * np.eye(D) + 1e-3,          # nearly the same: the hint holds
            A @ A.T / D + 1e-4 * np.eye(D),                # ill conditioned: more steps
            np.eye(D) * 1.7,                               # a multiple of the identity: far fewer steps (restart)
            A @ A.T / D + 0.5 * np.eye(D),
            np.diag(np.linspace(0.5, 2.0, D))]
    thetas = [np.concatenate([mu, ofam.psd_to_free(c)]) for c in covs]

    def call():
        out = []
        ekl = vb.ExclusiveKL(vb.MultivariateT(D, df, seed=2), model, N)
        dis = vb.DISInclusiveKL(vb.MultivariateT(D, df, seed=2), model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                                temper_prior_params=np.concatenate([np.zeros(D), 0.3 * np.ones(D)]), use_resampling=False)
        for th in thetas:
            np.random.seed(5)
            out.append(ekl(th))
            out.append(dis(th))
        return out
    hinted, plain = call(), _with({'VB_NS_HINT': '0'}, call)
    for (v0, g0), (v1, g1) in zip(hinted, plain):
        assert v0 == v1
        np.testing.assert_array_equal(g0, g1)
