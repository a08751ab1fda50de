#!/usr/bin/env python3
"""Dev tool: where the host time of a parity-mode (rng='numpy') MultivariateT(256) + ExclusiveKL call goes."""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb

D, N, df = 256, 16384, 100
rng = np.random.RandomState(7)
mean, sd = 0.3 * rng.randn(D), np.exp(0.1 * rng.randn(D))
A = rng.randn(D, D)
Sigma = np.e * np.eye(D) + 0.04 * (A @ A.T / D - np.eye(D))
approx = vb.MultivariateT(D, df, seed=1)
L = np.linalg.cholesky(Sigma)
Lf = L.copy()
Lf[np.diag_indices(D)] = np.log(np.diag(L))
theta = np.concatenate([0.02 * rng.randn(D), Lf[np.tril_indices(D)]])
obj = vb.ExclusiveKL(approx, vb.GaussianModel(mean, sd), N)
for _ in range(4):
    obj(theta)
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    obj(theta)
    ts.append(time.perf_counter() - t0)
print('parity-mode MultivariateT ExclusiveKL call: median %.3f ms' % (1e3 * np.median(ts)))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    obj(theta)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
