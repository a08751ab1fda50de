"""GPU: the sharded code path (reduce-only finalize -> RCCL all-reduce -> epilogue kernel) with a
one-rank communicator must reproduce the fused single-GPU path; with n_total > n it must scale like a
shard of a larger job.  (N > 1 ranks need N GPUs: covered by the driver's multi-GPU bench; the
sharding arithmetic itself is covered on CPU by tests/test_distributed_cpu.py.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def engines():
    import viabel_amd  # noqa: F401
    from viabel_amd import _lib
    plain = _lib.default_engine()
    comm = _lib.Engine(plain.device)
    comm.comm_init(_lib.Engine.comm_unique_id(), 1, 0)
    yield plain, comm
    comm.comm_destroy()
    comm.close()


def _theta(D, seed):
    rng = np.random.RandomState(seed)
    return np.concatenate([0.3 * rng.randn(D), -1.0 + 0.2 * rng.randn(D)])


@pytest.mark.parametrize('flags,cv', [(0, 0), (1, 0), (0, 1), (1, 3)])
def test_meanfield_comm_path_matches_fused(engines, flags, cv):
    import viabel_amd as vb
    from viabel_amd import _lib
    plain, comm = engines
    D, N = 1024, 4096
    spec = vb.FunnelModel(D).device_spec()
    theta = _theta(D, 1)
    out = []
    for eng in (plain, comm):
        eng.set_model(spec)
        eng.noise_generate(3, N, D, seed=5, stream=1)
        out.append(eng.elbo_grad_meanfield(3, N, D, theta, _lib.FAMILY_MF_GAUSSIAN, flags=flags, cv_mode=cv))
    assert abs(out[0][0] - out[1][0]) < 1e-13 * abs(out[0][0])
    np.testing.assert_allclose(out[1][1], out[0][1], rtol=0, atol=1e-13 * np.max(np.abs(out[0][1])))


def test_meanfield_batch_through_comm(engines):
    import viabel_amd as vb
    from viabel_amd import _lib
    plain, comm = engines
    D, N, B = 256, 1024, 5
    spec = vb.GaussianModel(np.zeros(D), np.ones(D)).device_spec()
    thetas = np.stack([_theta(D, s) for s in range(B)])
    res = []
    for eng in (plain, comm):
        eng.set_model(spec)
        for s in range(B):
            eng.noise_generate(10 + s, N, D, seed=2, stream=s)
        eng.elbo_grad_meanfield_batch_async(list(range(10, 10 + B)), N, D, thetas, _lib.FAMILY_MF_GAUSSIAN,
                                            list(range(B)))
        eng.sync()
        res.append([eng.result_get(b, 2 * D) for b in range(B)])
    for b in range(B):
        assert abs(res[0][b][0] - res[1][b][0]) < 1e-13 * abs(res[0][b][0])
        np.testing.assert_allclose(res[1][b][1], res[0][b][1], rtol=0, atol=1e-13)


def test_shard_of_a_larger_job(engines):
    """Two half-size shards evaluated with n_total = N: their (value + H, grad + dH) contributions add up."""
    import viabel_amd as vb
    from viabel_amd import _lib
    plain, comm = engines
    D, N = 128, 512
    spec = vb.FunnelModel(D).device_spec()
    theta = _theta(D, 3)
    plain.set_model(spec)
    plain.noise_generate(4, N, D, seed=8, stream=0)
    full = plain.noise_get_host(4, N, D)
    v, g = plain.elbo_grad_meanfield(4, N, D, theta, _lib.FAMILY_MF_GAUSSIAN)
    comm.set_model(spec)
    parts = []
    for h in range(2):
        comm.noise_set_host(5, full[h * N // 2:(h + 1) * N // 2])
        parts.append(comm.elbo_grad_meanfield(5, N // 2, D, theta, _lib.FAMILY_MF_GAUSSIAN, n_total=N))
    H = 0.5 * D * (1 + np.log(2 * np.pi)) + theta[D:].sum()
    # value_h = -(F_h / N + H): summing the two shards double counts the entropy terms
    assert abs((parts[0][0] + parts[1][0] + H) - v) < 1e-12 * abs(v)
    gsum = parts[0][1] + parts[1][1]
    gsum[D:] += 1.0
    np.testing.assert_allclose(gsum, g, rtol=0, atol=1e-12 * np.max(np.abs(g)))


def test_fullrank_comm_path(engines):
    import viabel_amd as vb
    plain, comm = engines
    D, N = 130, 400
    rng = np.random.RandomState(2)
    spec = vb.GaussianModel(rng.randn(D), np.exp(0.2 * rng.randn(D))).device_spec()
    fr = vb.FullRankGaussian(D)
    L = np.tril(0.05 * rng.randn(D, D), -1) + np.diag(np.exp(-1 + 0.1 * rng.randn(D)))
    theta = fr.pack(0.2 * rng.randn(D), L)
    out = []
    for eng in (plain, comm):
        eng.set_model(spec)
        eng.noise_generate(6, N, D, seed=4, stream=0)
        out.append(eng.elbo_grad_fullrank(6, N, D, theta))
    assert abs(out[0][0] - out[1][0]) < 1e-13 * abs(out[0][0])
    np.testing.assert_allclose(out[1][1], out[0][1], rtol=0, atol=1e-13 * np.max(np.abs(out[0][1])))


def test_fullrank_path_derivative_comm_path(engines):
    """Path derivative through the sharded code path: tr(M2) and the L^-T M2 / L^-T e corrections ride in the
    all-reduced sum vector."""
    import viabel_amd as vb
    from viabel_amd import _lib
    plain, comm = engines
    D, N = 70, 300
    rng = np.random.RandomState(8)
    spec = vb.FunnelModel(D).device_spec()
    fr = vb.FullRankGaussian(D)
    L = np.tril(0.05 * rng.randn(D, D), -1) + np.diag(np.exp(-1 + 0.1 * rng.randn(D)))
    theta = fr.pack(0.2 * rng.randn(D), L)
    out = []
    for eng in (plain, comm):
        eng.set_model(spec)
        eng.noise_generate(6, N, D, seed=4, stream=0)
        out.append(eng.elbo_grad_fullrank(6, N, D, theta, flags=_lib.FLAG_PATH_DERIV))
    assert out[0][0] == out[1][0]
    np.testing.assert_array_equal(out[1][1], out[0][1])
    ent = plain.elbo_grad_fullrank(6, N, D, theta)
    assert ent[0] != out[0][0]


def test_fullrank_overlapped_enqueues_through_comm(engines):
    """Back-to-back sharded full-rank evaluations: the all-reduce + epilogue of one runs on the communication
    stream while the next one's GEMMs run (two sum sets).  Every evaluation, interleaved with parameter changes
    and a mean-field call on the same context, must equal the single-GPU fused result bit for bit."""
    import viabel_amd as vb
    from viabel_amd import _lib
    plain, comm = engines
    D, N = 200, 1024
    rng = np.random.RandomState(12)
    A = rng.randn(D, D)
    spec = vb.CorrelatedGaussianModel(rng.randn(D), covariance=A @ A.T / D + np.eye(D)).device_spec()
    fr = vb.FullRankGaussian(D)
    thetas = []
    for k in range(3):
        L = np.tril(0.05 * rng.randn(D, D), -1) + np.diag(np.exp(-1 + 0.1 * rng.randn(D)))
        thetas.append(fr.pack(0.2 * rng.randn(D), L))
    for eng in (plain, comm):
        eng.set_model(spec)
        for s in range(4):
            eng.noise_generate(20 + s, N, D, seed=9, stream=s)
    want = [[plain.elbo_grad_fullrank(20 + s, N, D, th) for s in range(4)] for th in thetas]
    for k, th in enumerate(thetas):
        comm.fullrank_set_theta(th, D)
        for rep in range(3):                       # 12 evaluations in flight, the last four are checked
            for s in range(4):
                comm.elbo_grad_fullrank_enqueue(20 + s, N, D)
        v, g = comm.fullrank_get(D)
        assert v == want[k][3][0]
        np.testing.assert_array_equal(g, want[k][3][1])
        # a blocking evaluation of an earlier slot right behind the overlapped ones
        v, g = comm.elbo_grad_fullrank(20 + k, N, D, th)
        assert v == want[k][k][0]
        np.testing.assert_array_equal(g, want[k][k][1])
    # the main-stream paths are ordered after the communication stream
    # (binding another model is a main-stream write behind the evaluation in flight, which still sees the old one)
    mf_theta = _theta(D, 4)
    diag_spec = vb.GaussianModel(np.zeros(D), np.ones(D)).device_spec()
    comm.elbo_grad_fullrank_enqueue(21, N, D)
    comm.set_model(diag_spec)
    plain.set_model(diag_spec)
    a = comm.elbo_grad_meanfield(21, N, D, mf_theta, _lib.FAMILY_MF_GAUSSIAN)
    b = plain.elbo_grad_meanfield(21, N, D, mf_theta, _lib.FAMILY_MF_GAUSSIAN)
    assert abs(a[0] - b[0]) < 1e-13 * abs(b[0])
    v, g = comm.fullrank_get(D)
    assert v == want[2][1][0]
    np.testing.assert_array_equal(g, want[2][1][1])


def test_fullrank_shard_of_a_larger_job(engines):
    """Two half shards with n_total = N: summed sums reproduce the one-GPU evaluation (entropy counted once)."""
    import viabel_amd as vb
    plain, comm = engines
    D, N = 96, 512
    rng = np.random.RandomState(5)
    spec = vb.FunnelModel(D).device_spec()
    fr = vb.FullRankGaussian(D)
    L = np.tril(0.05 * rng.randn(D, D), -1) + np.diag(np.exp(-1 + 0.1 * rng.randn(D)))
    theta = fr.pack(0.1 * rng.randn(D), L)
    plain.set_model(spec)
    plain.noise_generate(7, N, D, seed=3, stream=0)
    full = plain.noise_get_host(7, N, D)
    v, g = plain.elbo_grad_fullrank(7, N, D, theta)
    comm.set_model(spec)
    parts = []
    for h in range(2):
        comm.noise_set_host(8, full[h * N // 2:(h + 1) * N // 2])
        parts.append(comm.elbo_grad_fullrank(8, N // 2, D, theta, n_total=N))
    # value_h = -(F_h / N + c0 + H): the entropy and the model's additive constant (added once per evaluation,
    # after the all-reduce) are both counted twice in the sum of the two simulated shards
    H = 0.5 * D * (1 + np.log(2 * np.pi)) + np.log(np.diag(L)).sum()
    c0 = -0.5 * D * np.log(2 * np.pi)
    assert abs((parts[0][0] + parts[1][0] + H + c0) - v) < 1e-12 * abs(v)
    gsum = parts[0][1] + parts[1][1]
    diag = D + np.array([i * (i + 1) // 2 + i for i in range(D)])
    gsum[diag] += 1.0
    np.testing.assert_allclose(gsum, g, rtol=0, atol=1e-12 * np.max(np.abs(g)))


def test_alpha_and_dis_through_comm(engines):
    """AlphaDivergence / DISInclusiveKL with a one-rank communicator: the all-reduce(max/sum) and
    all-gather steps of the sharded path must not change the result."""
    import viabel_amd as vb
    from viabel_amd import _lib
    plain, comm = engines
    D, N = 96, 512
    rng = np.random.RandomState(1)
    model = vb.GaussianModel(rng.randn(D), np.exp(0.1 * rng.randn(D)))
    theta = _theta(D, 7)
    prior = np.concatenate([np.zeros(D), 0.3 * np.ones(D)])
    res = {}
    for name, eng in (('plain', plain), ('comm', comm)):
        _lib.set_default_engine(eng)
        np.random.seed(3)
        a = vb.AlphaDivergence(vb.MFGaussian(D), model, N, 2.0)(theta)
        np.random.seed(3)
        d = vb.DISInclusiveKL(vb.MFStudentT(D, 9, seed=2), model, N, ess_target=100, temper_prior=vb.MFGaussian(D),
                              temper_prior_params=prior)(theta)
        mvt = vb.MultivariateT(D, 30, seed=2)
        np.random.seed(3)
        m = vb.DISInclusiveKL(mvt, model, N, ess_target=100, temper_prior=vb.MFGaussian(D),
                              temper_prior_params=prior)(mvt.init_param())
        res[name] = (a, d, m)
    _lib.set_default_engine(plain)
    for x, y in zip(res['plain'], res['comm']):
        assert abs(x[0] - y[0]) < 1e-12 * abs(x[0])
        np.testing.assert_allclose(y[1], x[1], rtol=0, atol=1e-12 * np.max(np.abs(x[1])))


def test_lowrank_through_comm(engines):
    """LRGaussian ExclusiveKL: sum vector all-reduced on a one-rank communicator == plain path; a shard of a
    larger job (n_total > n) scales the data term only."""
    import viabel_amd as vb
    plain, comm = engines
    D, N, k = 200, 600, 5
    rng = np.random.RandomState(4)
    spec = vb.FunnelModel(D, 7).device_spec()
    fam = vb.LRGaussian(D, k=k)
    theta = fam.pack(0.1 * rng.randn(D), -1.0 + 0.1 * rng.randn(D), 0.05 * rng.randn(D, k))
    out = []
    for eng in (plain, comm):
        eng.set_model(spec)
        eng.noise_generate(3, N, D, seed=5, stream=1)
        eng.noise_generate(4, N, k, seed=6, stream=1)
        out.append(eng.elbo_grad_lowrank(3, 4, N, D, k, theta))
    assert abs(out[0][0] - out[1][0]) < 1e-13 * abs(out[0][0])
    np.testing.assert_allclose(out[1][1], out[0][1], rtol=0, atol=1e-13 * np.max(np.abs(out[0][1])))


def test_meanfield_overlapped_batches_through_comm(engines):
    """Many asynchronous batches in flight with the all-reduce / epilogue on the communication stream: every
    result equals the blocking single-call result for the same (noise, theta)."""
    import viabel_amd as vb
    from viabel_amd import _lib
    plain, comm = engines
    D, N, B, rounds = 384, 2048, 8, 7
    spec = vb.FunnelModel(D).device_spec()
    for eng in (plain, comm):
        eng.set_model(spec)
        for s in range(B):
            eng.noise_generate(20 + s, N, D, seed=3, stream=s)
    thetas = np.stack([_theta(D, 100 + i) for i in range(B * rounds)])
    ref = [plain.elbo_grad_meanfield(20 + i % B, N, D, thetas[i], _lib.FAMILY_MF_GAUSSIAN) for i in range(B * rounds)]
    got = {}
    for r in range(rounds):            # keep 3 batches in flight before collecting the oldest
        idx = list(range(r * B, (r + 1) * B))
        comm.elbo_grad_meanfield_batch_async([20 + i % B for i in idx], N, D, thetas[idx], _lib.FAMILY_MF_GAUSSIAN,
                                             rslots=[(r % 3) * B + j for j in range(B)])
        if r >= 2:
            old = r - 2
            for j in range(B):
                got[old * B + j] = comm.result_get((old % 3) * B + j, 2 * D)
    for old in (rounds - 2, rounds - 1):
        for j in range(B):
            got[old * B + j] = comm.result_get((old % 3) * B + j, 2 * D)
    for i in range(B * rounds):
        assert abs(got[i][0] - ref[i][0]) < 1e-13 * abs(ref[i][0])
        np.testing.assert_allclose(got[i][1], ref[i][1], rtol=0, atol=1e-13 * np.max(np.abs(ref[i][1])))


def test_multivariate_t_elbo_sums_through_comm(engines):
    """vb_elbo_sums_mvt: the sum vector [F | sum g | sum g (z / s)'] all-reduced on a one-rank communicator."""
    import viabel_amd as vb
    plain, comm = engines
    D, N = 96, 700
    rng = np.random.RandomState(9)
    spec = vb.FunnelModel(D, 11).device_spec()
    A = rng.randn(D, D)
    root = A @ A.T / D + np.eye(D)
    mu, inv_s = 0.1 * rng.randn(D), 1.0 / np.sqrt(rng.chisquare(8.0, N) / 8.0)
    out = []
    for eng in (plain, comm):
        eng.set_model(spec)
        eng.noise_generate(3, N, D, seed=5, stream=2)
        out.append(eng.elbo_sums_mvt(3, N, D, mu, root, inv_s))
    assert abs(out[0][0] - out[1][0]) < 1e-13 * abs(out[0][0])
    for a, b in zip(out[0][1:], out[1][1:]):
        np.testing.assert_allclose(b, a, rtol=0, atol=1e-13 * np.max(np.abs(a)))


@pytest.mark.parametrize('family', ['meanfield', 'fullrank'])
def test_device_fit_through_comm(engines, family):
    """vb_fit on a context with a communicator: every iteration's sums go through the all-reduce (and, for the
    full-rank family, through the communication stream) before the optimiser step; with one rank the trajectory
    equals the single-GPU one."""
    import viabel_amd as vb
    from viabel_amd import _lib
    plain, comm = engines
    D, N, iters = 40, 128, 25
    rng = np.random.RandomState(21)
    if family == 'meanfield':
        spec = vb.FunnelModel(D).device_spec()
        fam, theta = _lib.FAMILY_MF_GAUSSIAN, _theta(D, 6)
    else:
        A = rng.randn(D, D)
        spec = vb.CorrelatedGaussianModel(rng.randn(D), covariance=A @ A.T / D + np.eye(D)).device_spec()
        fam = _lib.FAMILY_FULLRANK_GAUSSIAN
        theta = vb.FullRankGaussian(D).pack(np.zeros(D), np.exp(-1.0) * np.eye(D))
    out = []
    for eng in (plain, comm):
        eng.set_model(spec)
        out.append(eng.fit(9, N, D, fam, theta, iters, _lib.OPT_RMSPROP, [0.01, 0.9, 0.0, 1e-8], seed=4,
                           first_stream=3, hist_len=iters, log_gradients=True))
    for a, b in zip(out[0], out[1]):
        if a is not None:
            if family == 'meanfield':   # fused finalize vs reduce-only + epilogue: same sums, different last ulp
                np.testing.assert_allclose(b, a, rtol=1e-9, atol=1e-12)
            else:
                np.testing.assert_array_equal(b, a)


def test_multivariate_t_path_terms_through_comm(engines):
    """The noise-only sums of the t family's path-derivative estimator are all-reduced like the other sum vectors."""
    plain, comm = engines
    D, N, df = 90, 700, 12.0
    rng = np.random.RandomState(31)
    inv_s = 1.0 / np.sqrt(rng.chisquare(df, N) / df)
    out = []
    for eng in (plain, comm):
        eng.noise_generate(12, N, D, seed=5, stream=2)
        out.append(eng.mvt_path_terms(12, N, D, df, inv_s))
    z = plain.noise_get_host(12, N, D)
    maha = np.sum(z * z, axis=1) * inv_s ** 2
    c = (df + D) / (df + maha)
    want_m = (z * (c * inv_s ** 2)[:, None]).T @ z
    np.testing.assert_allclose(out[0][0], want_m, rtol=0, atol=1e-12 * np.max(np.abs(want_m)))
    np.testing.assert_allclose(out[0][1], (c * inv_s) @ z, rtol=0, atol=1e-12 * N)
    assert abs(out[0][2] - np.sum(np.log1p(maha / df))) < 1e-12 * N
    for a, b in zip(out[0], out[1]):
        np.testing.assert_array_equal(np.asarray(b), np.asarray(a))


def test_alpha_fullrank_through_comm(engines):
    """AlphaDivergence for the dense Gaussian through the sharded code path (all-reduced max / weight sum, packed
    sums all-reduced, weighted epilogue kernel) equals the fused single-GPU path."""
    import viabel_amd as vb
    plain, comm = engines
    D, N = 60, 400
    rng = np.random.RandomState(17)
    spec = vb.FunnelModel(D, 7).device_spec()
    L = np.tril(0.05 * rng.randn(D, D), -1) + np.diag(np.exp(-1 + 0.1 * rng.randn(D)))
    theta = vb.FullRankGaussian(D).pack(0.2 * rng.randn(D), L)
    out = []
    for eng in (plain, comm):
        eng.set_model(spec)
        eng.noise_generate(14, N, D, seed=6, stream=1)
        out.append(eng.alpha_grad_fullrank(14, N, D, theta, 2.0))
    assert out[0][0] == out[1][0]
    np.testing.assert_array_equal(out[1][1], out[0][1])


def test_alpha_multivariate_t_through_comm(engines):
    """AlphaDivergence sums of the multivariate t through the sharded code path (all-reduced max / weight sum /
    raw sums) equal the single-GPU path."""
    import viabel_amd as vb
    plain, comm = engines
    D, N, df = 40, 300, 9.0
    rng = np.random.RandomState(23)
    spec = vb.FunnelModel(D, 5).device_spec()
    A = 0.1 * rng.randn(D, D)
    root = A @ A.T + 0.4 * np.eye(D)
    mu = 0.2 * rng.randn(D)
    inv_s = 1.0 / np.sqrt(rng.chisquare(df, N) / df)
    out = []
    for eng in (plain, comm):
        eng.set_model(spec)
        eng.noise_generate(14, N, D, seed=6, stream=2)
        out.append(eng.alpha_sums_mvt(14, N, D, df, 0.5, mu, root, inv_s, -3.0))
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    np.testing.assert_array_equal(out[1][2], out[0][2])
    np.testing.assert_array_equal(out[1][3], out[0][3])


def test_lowrank_alpha_and_dis_through_comm(engines):
    """LRGaussian under AlphaDivergence / DISInclusiveKL: the all-reduced (alpha) and all-gathered (DIS) code paths
    with a one-rank communicator reproduce the plain ones."""
    import viabel_amd as vb
    plain, comm = engines
    D, k, N = 96, 5, 640
    rng = np.random.RandomState(4)
    spec = vb.GaussianModel(0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D))).device_spec()
    mu, ls, B = 0.1 * rng.randn(D), -0.3 + 0.1 * rng.randn(D), 0.3 * rng.randn(D, k)
    Bs = B / np.exp(ls)[:, None]
    M = np.eye(k) + Bs.T @ Bs
    Minv = np.linalg.inv(M)
    cq = -0.5 * (D * np.log(2 * np.pi) + 2 * np.sum(ls) + np.linalg.slogdet(M)[1])
    prior = np.concatenate([np.zeros(D), 0.1 * np.ones(D)])
    out = []
    for eng in (plain, comm):
        eng.set_model(spec)
        eng.noise_generate(5, N, D, seed=9, stream=0)
        eng.noise_generate(6, N, k, seed=9, stream=1)
        a = eng.alpha_sums_lowrank(5, 6, N, D, k, 2.0, mu, ls, B, Minv, cq)
        r = eng.dis_refresh_lowrank(5, 6, N, D, k, mu, ls, B, Minv, cq, prior, 1.0, N / 4)
        g = eng.dis_grad_lowrank(N, D, k, mu + 0.01, ls, B, Minv, cq, r[2])
        out.append((a, r, g))
    for x, y in zip(out[0], out[1]):
        for u, v in zip(x, y):
            np.testing.assert_allclose(np.asarray(v), np.asarray(u), rtol=1e-13, atol=1e-13 * np.max(np.abs(u)))


def test_host_staged_transport_one_rank_and_failing_collective():
    """vb_comm_init_host: with one rank and an identity collective the sharded path reproduces the plain one; the
    collective sees the sum vector and the max request; a collective that raises surfaces as EngineError (VB_ERR_COMM),
    not as an exception unwinding through the C frames."""
    import viabel_amd as vb
    from viabel_amd import _lib
    plain = _lib.default_engine()
    eng = _lib.Engine(plain.device)
    calls = []

    def collective(array, op):
        calls.append((array.size, op))
        if len(calls) > 1000:
            raise RuntimeError('stop')

    eng.comm_init_host(collective, 1, 0)
    assert eng.comm_info() == (1, 0)
    D, N = 40, 512
    model = vb.GaussianModel(np.linspace(-1, 1, D), np.linspace(0.5, 1.5, D))
    theta = _theta(D, 3)
    out = []
    for e in (plain, eng):
        _lib.set_default_engine(e)
        try:
            np.random.seed(5)
            out.append((vb.ExclusiveKL(vb.MFGaussian(D, seed=2, rng='philox'), model, N)(theta),
                        vb.AlphaDivergence(vb.MFGaussian(D, seed=2, rng='philox'), model, N, 0.5)(theta)))
        finally:
            _lib.set_default_engine(plain)
    for (v0, g0), (v1, g1) in zip(out[0], out[1]):
        assert abs(v0 - v1) <= 1e-13 * abs(v0) and np.max(np.abs(g0 - g1)) <= 1e-13 * np.max(np.abs(g0))
    assert {op for _, op in calls} == {0, 1}            # sums and the alpha divergence's max over log weights
    calls.extend([(0, 0)] * 1001)                       # the next collective raises
    _lib.set_default_engine(eng)
    try:
        with pytest.raises(_lib.EngineError):
            vb.ExclusiveKL(vb.MFGaussian(D, seed=2, rng='philox'), model, N)(theta)
    finally:
        _lib.set_default_engine(plain)
    eng.comm_destroy()
    assert eng.comm_info() == (1, 0)
    eng.close()


def test_resident_dense_routes_and_c3_through_a_one_rank_rccl_communicator(engines):
    """Round 6: the device-resident DIS step and the t family's reference-identical ExclusiveKL / AlphaDivergence under a
    communicator -- here RCCL itself with one rank (RCCL refuses two ranks on one device; the two-rank runs of
    tests/test_gpu_two_ranks.py use the other transports): the fused three-vector gather (one in-place ncclAllReduce over
    [log q | log p | log prior]), the sum all-reduce in front of the chain rule, the all-reduced sample sums in front of the
    Frechet derivative are the calls an 8-GPU job issues, and the results are the communicator-free ones."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import viabel_amd as vb
    from viabel_amd import _lib
    import _two_rank_scenarios as S
    plain, comm = engines
    want = {**S.run_resident_dense(vb), **S.run_c3(vb)}
    _lib.set_default_engine(comm)
    try:
        assert comm.comm_info() == (1, 0) and comm._lib.vb_comm_check(comm._ctx) == 0
        got = {**S.run_resident_dense(vb), **S.run_c3(vb)}
    finally:
        _lib.set_default_engine(plain)
    assert set(got) == set(want) and len(got) >= 20
    for name in want:
        v0, g0 = np.asarray(want[name][0], dtype=float), np.asarray(want[name][1], dtype=float)
        v1, g1 = np.asarray(got[name][0], dtype=float), np.asarray(got[name][1], dtype=float)
        assert np.max(np.abs(v1 - v0)) <= 1e-12 * np.max(np.abs(v0)), name
        assert np.max(np.abs(g1 - g0)) <= 1e-12 * np.max(np.abs(g0)), name


def test_bench_sharded_legs_under_a_one_rank_rccl_communicator(engines):
    """bench.py's N > 1 legs (sharded C3 / C4 / C1, the transport legs) run through RCCL itself with a one-rank communicator:
    every collective the 8-GPU line issues is a real ncclAllReduce here, and the legs' results are what the one-GPU legs
    report."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import viabel_amd as vb
    from viabel_amd import _lib, distributed
    plain, comm = engines
    solo = distributed.SocketGroup(0, 1)
    _lib.set_default_engine(comm)
    try:
        c3 = bench.sharded_c3_leg(comm, vb, solo, calls=4)
        c1 = bench.sharded_c1_leg(comm, vb, solo, iters=40)
        c4 = bench.sharded_c4_leg(comm, vb, solo, steps=2)
        t_big = comm.comm_allreduce_time(16 + 1024 + 1024 * 1025 // 2, warm=1, reps=3)
    finally:
        _lib.set_default_engine(plain)
    for scaling in ('strong', 'weak'):
        leg = c3[scaling]
        assert leg['n_mc_global'] == 16384 and leg['allgather_doubles_per_call'] == 3 * 16384
        for mode in ('weighted', 'resampling'):
            # (eps only ever moves down from call to call -- objectives.py:347-368 bisects on [0, eps_prev] -- so after the
            # warm-up calls the effective sample size sits at or above what the first refresh hit)
            assert 0.0 < leg[mode]['eps'] < 1.0 and 1000 < leg[mode]['ess'] < 16384 and np.isfinite(leg[mode]['value'])
            assert 0.05 < leg[mode]['ms_per_call'] < 5.0
    assert 1000 < c3['strong']['parity_mode_weighted']['ess'] < 16384
    assert c1['strong']['us_per_iteration'] > 5.0 and c4['ms_per_eval'] > 1.0 and np.isfinite(c4['value']) and t_big > 0.0
