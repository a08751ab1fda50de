#!/usr/bin/env python3
"""Dev tool: host-side profile of one C3-shape objective call (where the 8 ms go)."""
import cProfile
import pstats
import sys

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb

D, N = 256, 16384
np.random.seed(5)
sys.path.insert(0, 'tests')
from test_gpu_full_size import c3_problem
mean, sd, prior, theta = c3_problem(np.random.RandomState(33), D)
model = vb.GaussianModel(mean, sd)
approx = vb.MultivariateT(D, 100, seed=1, rng='philox')
obj = vb.DISInclusiveKL(approx, model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                        temper_prior_params=prior, use_resampling=True, num_resampling_batches=1)
for i in range(3):
    obj(theta)
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    obj(theta)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
