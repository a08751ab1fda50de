#!/usr/bin/env python3
"""Dev tool: BASELINE configs[3] shape -- MultivariateT(256, df=100) + DISInclusiveKL, N_mc=16384 on one GPU
(the 8-GPU job gives each GPU 2048 rows), then PSIS of the 16384 log weights."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb
from viabel_amd._psis import psislw

if '--blas-threads' in sys.argv:          # see viabel_amd.set_host_blas_threads
    i = sys.argv.index('--blas-threads')
    vb.set_host_blas_threads(int(sys.argv[i + 1]))
    del sys.argv[i:i + 2]

D, N = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 16384
np.random.seed(5)
rng_kind = sys.argv[3] if len(sys.argv) > 3 else 'philox'
approx = vb.MultivariateT(D, 100, seed=1, rng=rng_kind)
# the problem of tests/test_gpu_full_size.py: q on the tempering prior, target shifted away -> interior eps, ESS on target
# (init_param's Sigma = 10 I collapses the weights to ESS = 1 in 256 dimensions)
sys.path.insert(0, 'tests')
from test_gpu_full_size import c3_problem
mean, sd, prior, theta = c3_problem(np.random.RandomState(33), D)
model = vb.FunnelModel(D) if (len(sys.argv) > 2 and sys.argv[2] == 'funnel') else vb.GaussianModel(mean, sd)
for resample in (False, True):
    obj = vb.DISInclusiveKL(approx, model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, use_resampling=resample, num_resampling_batches=1)
    times = []
    for i in range(6):
        t0 = time.perf_counter()
        v, g = obj(theta)
        times.append(time.perf_counter() - t0)
    print('C3 shape (rng=' + rng_kind + ', N=%d, resampling=%s): %.2f ms per objective call (min %.2f); value %.6g |grad| %.4g; eps %.3g ess %.0f'
          % (N, resample, 1e3 * np.median(times[1:]), 1e3 * min(times), v, np.linalg.norm(g), obj._eps, obj._ess))
lw = obj._state_log_p_unnormalized - obj._state_log_q
t0 = time.perf_counter()
for i in range(20):
    sm, k = psislw(lw)
print('PSIS of %d log weights: %.3f ms per call, khat %.3f' % (N, 1e3 * (time.perf_counter() - t0) / 20, k))
