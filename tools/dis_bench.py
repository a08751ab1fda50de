"""Dev tool: DISInclusiveKL call times (state refresh every call, rng=philox) for the four families at one shape."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb

D, N = 256, 16384
rng = np.random.RandomState(3)
model = vb.GaussianModel(0.3 * rng.randn(D), np.exp(0.5 + 0.02 * rng.randn(D)))
prior = np.concatenate([np.zeros(D), 0.5 * np.ones(D)])
for name, approx in (('MFGaussian', vb.MFGaussian(D, rng='philox')), ('MFStudentT(100)', vb.MFStudentT(D, 100, rng='philox')),
                     ('FullRankGaussian', vb.FullRankGaussian(D, rng='philox')),
                     ('MultivariateT(100)', vb.MultivariateT(D, 100, rng='philox')),
                     ('LRGaussian(k=8)', vb.LRGaussian(D, k=8, rng='philox'))):
    theta = approx.init_param()
    if isinstance(approx, (vb.MFGaussian, vb.MFStudentT)):
        theta[D:] = 0.5
    elif isinstance(approx, vb.FullRankGaussian):
        theta = approx.pack(np.zeros(D), np.exp(0.5) * np.eye(D))
    elif isinstance(approx, vb.MultivariateT):
        L = np.exp(0.5) * np.eye(D)
        Lf = L.copy()
        Lf[np.diag_indices(D)] = np.log(np.diag(L))
        theta = np.concatenate([np.zeros(D), Lf[np.tril_indices(D)]])
    else:
        theta = approx.pack(np.zeros(D), 0.5 * np.ones(D), 0.01 * rng.randn(D, 8))
    for resample in (False, True):
        obj = vb.DISInclusiveKL(approx, model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                                temper_prior_params=prior, use_resampling=resample)
        np.random.seed(2)
        for _ in range(10):
            obj(theta)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(20):
                v, g = obj(theta)
            ts.append((time.perf_counter() - t0) / 20)
        print('%-20s resampling=%-5s %8.1f us per call  eps %.3f' % (name, resample, 1e6 * sorted(ts)[1], obj._eps))
