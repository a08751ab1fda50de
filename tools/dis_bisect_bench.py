"""Dev tool: the DIS state refresh (objectives.py:393-401) on fresh problems -- eps_prev = 1 every time, so the
tempering bisection starts from [0, 1] (tools/c3_bench.py re-uses eps of the previous call, where the walk hugs the
upper end).  Mean-field family, N = 16384; VB_DIS_TRACE=1 prints what every round planned."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb

D, N = 64, 16384
rng = np.random.RandomState(7)
model = vb.GaussianModel(0.3 + 0.3 * rng.randn(D), np.exp(0.2 * rng.randn(D)))
approx = vb.MFGaussian(D, seed=11, rng='philox')
prior = np.zeros(2 * D)
theta = prior + 0.02 * rng.randn(2 * D)
for target in (N // 50, N // 8, N // 2):
    obj = vb.DISInclusiveKL(approx, model, N, ess_target=target, temper_prior=vb.MFGaussian(D), temper_prior_params=prior,
                            use_resampling=False)
    ts = []
    for i in range(30):
        obj._eps = 1.0
        t0 = time.perf_counter()
        obj(theta)
        ts.append(time.perf_counter() - t0)
    print('target %6d: %.1f us per call (median of 30), eps %.6f ess %.1f' % (target, 1e6 * np.median(ts[5:]), obj._eps, obj._ess))
