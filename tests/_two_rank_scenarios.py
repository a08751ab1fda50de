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
    return {k: (np.atleast_1d(np.asarray(v[0], dtype=float)), np.asarray(v[1], dtype=float)) for k, v in out.items()}
