#!/usr/bin/env python3
"""Dev tool: BASELINE configs[4] shape -- MFGaussian + ExclusiveKL on logistic regression, D=2000,
n_data=8192, N_mc=8192 (one GPU's share)."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb
from viabel_amd import _lib

D, n_data, N = 2000, 8192, 8192
rng = np.random.RandomState(4)
X = rng.randn(n_data, D) / np.sqrt(D)
beta = rng.randn(D)
y = (rng.rand(n_data) < 1 / (1 + np.exp(-X @ beta))).astype(float)
model = vb.LogisticRegressionModel(X, y, 10.0)
eng = _lib.default_engine()
t0 = time.perf_counter()
eng.set_model(model.device_spec())
print('model upload %.2f s' % (time.perf_counter() - t0))
approx = vb.MFGaussian(D, rng='philox')
obj = vb.ExclusiveKL(approx, model, N)
theta = np.concatenate([np.zeros(D), -2 * np.ones(D)])
for i in range(3):
    v, g = obj(theta)
eng.sync()
t0 = time.perf_counter()
steps = 10
for i in range(steps):
    v, g = obj(theta)
dt = (time.perf_counter() - t0) / steps
flops = 2 * 2.0 * N * n_data * D
print('C4 shape: %.2f ms/eval, %.1f evals/s, %.1f TFLOP/s; value %.6g |grad| %.4g'
      % (dt * 1e3, 1 / dt, flops / dt / 1e12, v, np.linalg.norm(g)))
