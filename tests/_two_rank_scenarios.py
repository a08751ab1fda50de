"""Objective evaluations whose result must not depend on how many ranks share the Monte-Carlo axis.

`run_all(vb)` is executed once in a single process (no communicator) and once by each rank of a two-rank job on the
same GPU (tests/test_gpu_two_ranks.py; host-staged transport, because RCCL refuses two ranks on one device).  Every
family draws its noise from Philox counters indexed by the GLOBAL sample row, so the two runs see the same noise and
the results agree up to the order of the partial sums.  Sample counts are odd on purpose: the shards are ragged.
"""
import numpy as np


def _theta_mf(D, rng):
    return np.concatenate([0.1 * rng.randn(D), -0.5 + 0.1 * rng.randn(D)])


def _theta_chol(D, rng, scale=0.7):
    from oracle import families as ofam
    A = rng.randn(D, D)
    return np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(scale * (A @ A.T / D + np.eye(D)))])


def run_all(vb):
    out = {}
    rng = np.random.RandomState(5)
    D = 48
    mean, sd = 0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D))
    gauss = vb.GaussianModel(mean, sd)
    A = rng.randn(D, D)
    corr = vb.CorrelatedGaussianModel(0.2 * rng.randn(D), covariance=A @ A.T / D + np.eye(D))
    funnel = vb.FunnelModel(D, D // 2)

    # ---- ExclusiveKL: every family, entropy and path-derivative forms, control variates ----------------------
    th = _theta_mf(D, rng)
    for name, kw in (('plain', {}), ('pd', {'use_path_deriv': True}), ('cv_mean', {'hessian_approx_method': 'mean_only'}),
                     ('cv_full', {'hessian_approx_method': 'full'})):
        obj = vb.ExclusiveKL(vb.MFGaussian(D, seed=3, rng='philox'), gauss if 'cv' in name else funnel, 1001, **kw)
        out['ekl_mf_' + name] = obj(th)
    out['ekl_mf_t'] = vb.ExclusiveKL(vb.MFStudentT(D, 7.0, seed=3, rng='philox'), funnel, 1001)(th)
    # the reference-parity noise source: every rank draws the whole numpy matrix and uploads its own rows
    out['ekl_mf_numpy_rng'] = vb.ExclusiveKL(vb.MFGaussian(D, seed=3), funnel, 1001)(th)

    thc = _theta_chol(D, rng)
    fr = vb.FullRankGaussian(D, seed=4, rng='philox')
    L = np.exp(-1.0) * np.eye(D) + 0.01 * np.tril(rng.randn(D, D))
    thf = fr.pack(0.1 * rng.randn(D), L)
    out['ekl_fr_corr'] = vb.ExclusiveKL(fr, corr, 1003)(thf)
    out['ekl_fr_funnel'] = vb.ExclusiveKL(vb.FullRankGaussian(D, seed=4, rng='philox'), funnel, 1003)(thf)
    out['ekl_fr_pd'] = vb.ExclusiveKL(vb.FullRankGaussian(D, seed=4, rng='philox'), corr, 1003, use_path_deriv=True)(thf)
    out['ekl_mvt'] = vb.ExclusiveKL(vb.MultivariateT(D, 9.0, seed=5, rng='philox'), corr, 1005)(thc)
    lr = vb.LRGaussian(D, seed=6, k=5, rng='philox')
    thl = lr.pack(0.1 * rng.randn(D), -0.5 + 0.1 * rng.randn(D), 0.1 * rng.randn(D, 5))
    out['ekl_lr'] = vb.ExclusiveKL(lr, gauss, 1001)(thl)
    lr24 = vb.LRGaussian(D, seed=6, k=24, rng='philox')
    thl24 = lr24.pack(0.1 * rng.randn(D), -0.5 + 0.1 * rng.randn(D), 0.1 * rng.randn(D, 24))
    out['ekl_lr24'] = vb.ExclusiveKL(lr24, gauss, 1001)(thl24)

    # ---- AlphaDivergence (the seed of its noise is a host draw: rank 0's travels over the control group) -----
    for fam_name, fam, theta in (('mf', vb.MFGaussian(D, seed=7, rng='philox'), th),
                                 ('fr', vb.FullRankGaussian(D, seed=7, rng='philox'), thf),
                                 ('mvt', vb.MultivariateT(D, 9.0, seed=7, rng='philox'), thc),
                                 ('lr', vb.LRGaussian(D, seed=7, k=5, rng='philox'), thl)):
        np.random.seed(11)
        out['alpha_' + fam_name] = vb.AlphaDivergence(fam, gauss, 1001, 0.5)(theta)

    # ---- DISInclusiveKL: three calls each (state refresh, tempering, weights; ragged gathers of the weights) --
    prior = np.concatenate([np.zeros(D), 0.3 * np.ones(D)])
    for fam_name, make, theta in (('mf', lambda: vb.MFGaussian(D, seed=8, rng='philox'), th),
                                  ('mvt', lambda: vb.MultivariateT(D, 9.0, seed=8, rng='philox'), thc),
                                  ('fr', lambda: vb.FullRankGaussian(D, seed=8, rng='philox'), thf),
                                  ('lr', lambda: vb.LRGaussian(D, seed=8, k=5, rng='philox'), thl)):
        obj = vb.DISInclusiveKL(make(), gauss, 1001, ess_target=150, temper_prior=vb.MFGaussian(D),
                                temper_prior_params=prior, use_resampling=False, num_resampling_batches=2)
        np.random.seed(12)
        res = [obj(theta) for _ in range(3)]
        out['dis_' + fam_name] = (np.array([r[0] for r in res] + [obj._eps]), np.concatenate([r[1] for r in res]))
    # resampling with host draws (rng='numpy': np.random.choice on rank 0 for everybody)
    obj = vb.DISInclusiveKL(vb.MFGaussian(D, seed=8), gauss, 1001, ess_target=150, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, use_resampling=True, num_resampling_batches=2)
    np.random.seed(13)
    res = [obj(th) for _ in range(3)]
    out['dis_mf_resampling'] = (np.array([r[0] for r in res] + [obj._eps]), np.concatenate([r[1] for r in res]))
    obj = vb.DISInclusiveKL(vb.MultivariateT(D, 9.0, seed=8), gauss, 1001, ess_target=150, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, use_resampling=True, num_resampling_batches=2, psis_smooth=True)
    np.random.seed(14)
    res = [obj(thc) for _ in range(2)]
    out['dis_mvt_resampling_psis'] = (np.array([r[0] for r in res] + [obj._eps]), np.concatenate([r[1] for r in res]))

    # ---- the device-resident optimiser loop: 30 RMSProp iterations, one all-reduce per iteration -------------
    for fam_name, fam, theta, model in (('mf', vb.MFGaussian(D, seed=9, rng='philox'), th, gauss),
                                        ('fr', vb.FullRankGaussian(D, seed=9, rng='philox'), thf, corr)):
        obj = vb.ExclusiveKL(fam, model, 257)
        res = vb.RMSProp(0.01, diagnostics=True).optimize(30, obj, theta)
        out['fit_' + fam_name] = (np.asarray(res['value_history'], dtype=float), np.asarray(res['opt_param'], dtype=float))
    out.update(run_resident_dense(vb))
    out.update(run_c3(vb))
    return {k: (np.atleast_1d(np.asarray(v[0], dtype=float)), np.asarray(v[1], dtype=float)) for k, v in out.items()}


def _dis_calls(obj, theta, n_calls, step=0.002):
    """`n_calls` objective calls along a short descent path (kept weights meet a moved parameter when
    num_resampling_batches > 1); returns ([values..., eps, ess-or-0, khat-or-0], gradients concatenated)."""
    vals, grads = [], []
    for _ in range(n_calls):
        v, g = obj(theta)
        vals.append(v)
        grads.append(g)
        theta = theta - step * g / (1.0 + np.abs(g))
    tail = [obj._eps, float(getattr(obj, '_ess', 0.0) or 0.0), float(getattr(obj, '_khat', 0.0) or 0.0)]
    return np.array(vals + tail), np.concatenate(grads)


def run_resident_dense(vb):
    """The device-resident routes of the dense families at ragged sizes (round 6: they used to refuse more than one rank):
    the throughput-mode DIS step with the device's multinomial draw, smoothing and clipping on the resident weights, and
    every reference-identical (rng='numpy') objective of the t family -- chi-square draws and normals of numpy's streams on
    the device, symmetric root, Frechet derivative -- with the sample sums all-reduced and the D^3 algebra redundant."""
    out = {}
    rng = np.random.RandomState(15)
    D = 48
    mean, sd = 0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D))
    gauss = vb.GaussianModel(mean, sd)
    funnel = vb.FunnelModel(D, D // 2)
    thc = _theta_chol(D, rng)
    prior = np.concatenate([np.zeros(D), 0.3 * np.ones(D)])
    kw = dict(ess_target=150, temper_prior=vb.MFGaussian(D), temper_prior_params=prior)
    np.random.seed(21)
    # device multinomial draw over the gathered weights (Philox: the same counts on every rank)
    obj = vb.DISInclusiveKL(vb.MultivariateT(D, 9.0, seed=8, rng='philox'), gauss, 1003, use_resampling=True,
                            num_resampling_batches=2, **kw)
    out['res_dis_mvt_philox_resampling'] = _dis_calls(obj, thc, 3)
    obj = vb.DISInclusiveKL(vb.MultivariateT(D, 9.0, seed=8, rng='philox'), gauss, 1003, use_resampling=False,
                            psis_smooth=True, w_clip_threshold=0.02, **kw)
    out['res_dis_mvt_philox_psis_clip'] = _dis_calls(obj, thc, 2)
    obj = vb.DISInclusiveKL(vb.FullRankGaussian(D, seed=8, rng='philox'), funnel, 1003, use_resampling=True, **kw)
    out['res_dis_fr_philox_resampling'] = _dis_calls(obj, thc, 2)
    # reference-identical mode: numpy's streams on the device, host resampling draw on rank 0 for everybody
    obj = vb.DISInclusiveKL(vb.MultivariateT(D, 9.0, seed=8), gauss, 1003, use_resampling=False, **kw)
    out['res_dis_mvt_numpy_weighted'] = _dis_calls(obj, thc, 2)
    obj = vb.DISInclusiveKL(vb.MultivariateT(D, 9.0, seed=8), gauss, 1003, use_resampling=True, num_resampling_batches=2,
                            w_clip_threshold=0.02, **kw)
    out['res_dis_mvt_numpy_resampling_clip'] = _dis_calls(obj, thc, 3)
    obj = vb.DISInclusiveKL(vb.FullRankGaussian(D, seed=8), gauss, 1003, use_resampling=True, num_resampling_batches=2, **kw)
    out['res_dis_fr_numpy_resampling'] = _dis_calls(obj, thc, 3)
    A = rng.randn(D, D)
    corr = vb.CorrelatedGaussianModel(0.2 * rng.randn(D), covariance=A @ A.T / D + np.eye(D))
    out['res_ekl_mvt_numpy'] = vb.ExclusiveKL(vb.MultivariateT(D, 9.0, seed=5), corr, 1005)(thc)
    out['res_ekl_mvt_numpy_pd'] = vb.ExclusiveKL(vb.MultivariateT(D, 9.0, seed=5), funnel, 1005, use_path_deriv=True)(thc)
    np.random.seed(22)
    out['res_alpha_mvt_numpy'] = vb.AlphaDivergence(vb.MultivariateT(D, 9.0, seed=7), gauss, 1005, 0.5)(thc)
    return out


def c3_problem(rng, D):
    """BASELINE configs[3]'s problem as tests/test_gpu_full_size.py poses it: q on the tempering prior up to a small
    correlated perturbation, the target shifted away -- the tempering bisection has to find an interior eps."""
    from oracle import families as ofam
    mean = 0.3 * rng.randn(D)
    sd = np.exp(0.5 + 0.02 * rng.randn(D))
    prior = np.concatenate([np.zeros(D), 0.5 * np.ones(D)])
    A = rng.randn(D, D)
    Sigma = np.e * np.eye(D) + 0.04 * (A @ A.T / D - np.eye(D))
    theta = np.concatenate([0.02 * rng.randn(D), ofam.psd_to_free(Sigma)])
    return mean, sd, prior, theta


def run_c3(vb):
    """BASELINE configs[3] at FULL size -- MultivariateT(256, df = 100) + DISInclusiveKL, N_mc = 16 384, the MC axis
    sharded -- weighted / resampling / PSIS, throughput mode and reference-identical mode, plus the t family's
    reference-identical ExclusiveKL (both forms) and AlphaDivergence at the same shape."""
    out = {}
    D, N, df, ess_target = 256, 16384, 100, 2048
    rng = np.random.RandomState(33)
    mean, sd, prior, theta = c3_problem(rng, D)
    model = vb.GaussianModel(mean, sd)
    kw = dict(ess_target=ess_target, temper_prior=vb.MFGaussian(D), temper_prior_params=prior)
    np.random.seed(31)
    for mode in ('philox', 'numpy'):
        def family():
            return vb.MultivariateT(D, df, seed=6, rng=mode)
        out['c3_%s_weighted' % mode] = _dis_calls(vb.DISInclusiveKL(family(), model, N, use_resampling=False, **kw), theta, 2)
        out['c3_%s_resampling' % mode] = _dis_calls(
            vb.DISInclusiveKL(family(), model, N, use_resampling=True, num_resampling_batches=2, **kw), theta, 3)
        out['c3_%s_weighted_psis' % mode] = _dis_calls(
            vb.DISInclusiveKL(family(), model, N, use_resampling=False, psis_smooth=True, **kw), theta, 2)
        out['c3_%s_resampling_psis' % mode] = _dis_calls(
            vb.DISInclusiveKL(family(), model, N, use_resampling=True, psis_smooth=True, **kw), theta, 2)
    out['c3_ekl_numpy'] = vb.ExclusiveKL(vb.MultivariateT(D, df, seed=6), model, N)(theta)
    out['c3_ekl_numpy_pd'] = vb.ExclusiveKL(vb.MultivariateT(D, df, seed=6), model, N, use_path_deriv=True)(theta)
    out['c3_alpha_numpy'] = vb.AlphaDivergence(vb.MultivariateT(D, df, seed=6), model, N, 0.5)(theta)
    return out



def run_three_ranks(vb):
    """A reduced set for a THREE-rank job (tests/test_gpu_two_ranks.py::test_three_ranks...): the middle rank's shard begins
    inside the vectors and ends inside them -- offsets neither of the two-rank job's ranks has -- with ragged shards
    (1003 = 335 + 334 + 334; 16 384 = 5462 + 5461 + 5461)."""
    out = dict(run_resident_dense(vb))
    D, N, df, ess_target = 256, 16384, 100, 2048
    rng = np.random.RandomState(33)
    mean, sd, prior, theta = c3_problem(rng, D)
    model = vb.GaussianModel(mean, sd)
    kw = dict(ess_target=ess_target, temper_prior=vb.MFGaussian(D), temper_prior_params=prior)
    np.random.seed(41)
    for mode in ('philox', 'numpy'):
        out['c3x3_%s_resampling_psis' % mode] = _dis_calls(
            vb.DISInclusiveKL(vb.MultivariateT(D, df, seed=6, rng=mode), model, N, use_resampling=True, psis_smooth=True,
                              num_resampling_batches=2, **kw), theta, 3)
    out['c3x3_ekl_numpy_pd'] = vb.ExclusiveKL(vb.MultivariateT(D, df, seed=6), model, N, use_path_deriv=True)(theta)
    # the mean-field and low-rank refreshes gather their three vectors through the same collective
    D2 = 48
    g2 = vb.GaussianModel(np.zeros(D2), np.ones(D2))
    pr2 = np.concatenate([np.zeros(D2), 0.3 * np.ones(D2)])
    th2 = np.concatenate([0.1 * np.ones(D2), -0.5 * np.ones(D2)])
    np.random.seed(42)
    out['x3_dis_mf'] = _dis_calls(vb.DISInclusiveKL(vb.MFGaussian(D2, seed=8, rng='philox'), g2, 1003, ess_target=150,
                                                    temper_prior=vb.MFGaussian(D2), temper_prior_params=pr2,
                                                    use_resampling=False), th2, 2)
    lr = vb.LRGaussian(D2, seed=8, k=5, rng='philox')
    out['x3_dis_lr'] = _dis_calls(vb.DISInclusiveKL(lr, g2, 1003, ess_target=150, temper_prior=vb.MFGaussian(D2),
                                                    temper_prior_params=pr2, use_resampling=False),
                                  lr.pack(np.zeros(D2), -0.5 * np.ones(D2), 0.1 * np.ones((D2, 5))), 2)
    return {k: (np.atleast_1d(np.asarray(v[0], dtype=float)), np.asarray(v[1], dtype=float)) for k, v in out.items()}
