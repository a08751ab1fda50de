#!/usr/bin/env python3
"""Dev tool: where the host time of a parity-mode (rng='numpy') C3 call goes (cProfile over steady-state calls)."""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import viabel_amd as vb
from test_gpu_full_size import c3_problem

D, N = 256, 16384
resample = len(sys.argv) > 1 and sys.argv[1] == 'resample'
np.random.seed(5)
approx = vb.MultivariateT(D, 100, seed=1)
mean, sd, prior, theta = c3_problem(np.random.RandomState(33), D)
obj = vb.DISInclusiveKL(approx, vb.GaussianModel(mean, sd), N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                        temper_prior_params=prior, use_resampling=resample, num_resampling_batches=1)
for _ in range(5):
    obj(theta)
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    obj(theta)
    ts.append(time.perf_counter() - t0)
print('parity-mode C3 call: median %.3f ms' % (1e3 * np.median(ts)))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    obj(theta)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
