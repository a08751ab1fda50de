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


# ---- round 6: the blocking full-rank call with its parameter upload pipelined against the sampling product -----------
def _pipe_both(call):
    old = os.environ.get('VB_FR_UPLOAD_PIPE')
    try:
        os.environ['VB_FR_UPLOAD_PIPE'] = '0'
        plain = call()
        os.environ['VB_FR_UPLOAD_PIPE'] = '1'
        piped = call()
    finally:
        if old is None:
            os.environ.pop('VB_FR_UPLOAD_PIPE', None)
        else:
            os.environ['VB_FR_UPLOAD_PIPE'] = old
    return plain, piped


@pytest.mark.parametrize('d,n', [(1024, 4096), (512, 4096), (1008, 1000), (576, 257), (2048, 512)])
@pytest.mark.parametrize('target', ['gauss_full', 'funnel', 'gauss_diag'])
def test_fullrank_pipelined_upload_is_the_same_evaluation(env, d, n, target):
    """vb_elbo_grad_fullrank above 1 MB of parameter with VB_FR_UPLOAD_PIPE=1 (built in round 6, measured slower, off by
    default): three row chunks of L, heaviest first, the sampling product of each chunk's column blocks behind its copy.  Every element of Z is the same k loop
    in the same order whichever launch computes it, so value and gradient are bit-identical -- back to back with changing
    parameters, targets whose first product has a reducing epilogue (gauss_diag: not chunked, but behind the upload) and
    shards too short for the plain product (k-split) included."""
    vb, eng, _lib = env
    rng = np.random.RandomState(d + n)
    if target == 'gauss_full':
        A = rng.randn(d, d)
        model = vb.CorrelatedGaussianModel(rng.randn(d), covariance=A @ A.T / d + np.eye(d))
    elif target == 'funnel':
        model = vb.FunnelModel(d, d // 3)
    else:
        model = vb.GaussianModel(rng.randn(d), np.exp(0.1 * rng.randn(d)))
    fam = vb.FullRankGaussian(d)
    thetas = [fam.pack(0.1 * rng.randn(d), np.exp(-1.0) * np.eye(d) + 0.01 * np.tril(rng.randn(d, d))) for _ in range(3)]

    def call():
        eng.set_model(model.device_spec())
        out = []
        for k, theta in enumerate(thetas):
            eng.noise_generate(5, n, d, seed=3, stream=k)
            out.append(eng.elbo_grad_fullrank(5, n, d, theta))
        return out
    before = eng.fullrank_upload_stats()
    plain, piped = _pipe_both(call)
    assert eng.fullrank_upload_stats() == before + len(thetas)      # the second pass really took the pipelined route
    _same(plain, piped)


def test_fullrank_pipelined_upload_after_asynchronous_evaluations(env):
    """The upload must stay behind evaluations still reading the PREVIOUS parameter (vb_elbo_grad_fullrank_enqueue leaves work
    in flight) and in front of everything that reads the new one."""
    vb, eng, _lib = env
    d, n = 1024, 4096
    rng = np.random.RandomState(9)
    A = rng.randn(d, d)
    model = vb.CorrelatedGaussianModel(rng.randn(d), covariance=A @ A.T / d + np.eye(d))
    fam = vb.FullRankGaussian(d)
    th = [fam.pack(0.1 * rng.randn(d), np.exp(-1.0) * np.eye(d) + 0.01 * np.tril(rng.randn(d, d))) for _ in range(2)]
    eng.set_model(model.device_spec())
    for s in range(4):
        eng.noise_generate(20 + s, n, d, seed=4, stream=s)
    want = [eng.elbo_grad_fullrank(20 + s, n, d, th[s & 1]) for s in range(4)]
    eng.fullrank_set_theta(th[0], d)
    for rep in range(3):
        for s in (0, 2):
            eng.elbo_grad_fullrank_enqueue(20 + s, n, d)              # in flight on the old parameter ...
        got1 = eng.elbo_grad_fullrank(21, n, d, th[1])                  # ... when the new one starts to arrive
        _same(want[1], got1)
        eng.elbo_grad_fullrank_enqueue(23, n, d)                        # the resident parameter is the uploaded one
        _same(want[3], eng.fullrank_get(d))
        eng.fullrank_set_theta(th[0], d)
        eng.elbo_grad_fullrank_enqueue(22, n, d)
        _same(want[2], eng.fullrank_get(d))


def test_pinned_result_arrays_are_ordinary_arrays_and_recycled(env):
    vb, eng, _lib = env
    a = _lib.pinned_array(200000)
    assert isinstance(a, np.ndarray) and a.flags.writeable and a.dtype == np.float64 and a.shape == (200000,)
    a[:] = 1.5
    view = a[10:20]
    addr = a.ctypes.data
    del a
    assert view[0] == 1.5                       # a view keeps the block
    b = _lib.pinned_array(200000)
    assert b.ctypes.data != addr
    del view
    c = _lib.pinned_array(200000)               # ... and the block comes back once the last view is gone
    assert c.ctypes.data == addr
    assert type(_lib.pinned_array(10)) is np.ndarray


def test_pinned_pool_stops_pinning_for_callers_that_keep_everything(env):
    """A caller that keeps every gradient (FASO's history) must not pin the host's memory: beyond MAX_OUTSTANDING the result
    arrays are ordinary pageable ones again -- and pinned ones come back once the kept arrays are dropped."""
    vb, eng, _lib = env
    pool = _lib.PinnedPool(_lib.load(), keep=2)
    pool.MAX_OUTSTANDING = 3 * 8 * 200000
    kept = [pool.array(200000) for _ in range(5)]
    for a in kept:
        a[:] = 2.0
    assert pool._out == 3 * 8 * 200000             # three page-locked blocks out, the other two arrays pageable
    del kept, a
    assert pool._out == 0 and pool.idle_blocks() == 2      # (keep = 2: the third block was freed)
    b = pool.array(200000)
    assert pool._out == 8 * 200000 and pool.idle_blocks() == 1
