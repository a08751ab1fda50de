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
model = vb.GaussianModel(np.zeros(D), 3 * np.ones(D))
approx = vb.MultivariateT(D, 100, seed=1, rng='philox')
obj = vb.DISInclusiveKL(approx, model, N, ess_target=N // 2, temper_prior=vb.MFGaussian(D),
                        temper_prior_params=np.zeros(2 * D), use_resampling=True, num_resampling_batches=1)
theta = approx.init_param() + 0.01 * np.random.randn(approx.var_param_dim)
for i in range(3):
    obj(theta)
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    obj(theta)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
