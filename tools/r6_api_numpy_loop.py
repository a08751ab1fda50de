"""Round 6: blocking FullRankGaussian(1024) + ExclusiveKL calls in the default rng='numpy' mode (look-ahead generation of
numpy's randn stream beside the evaluation), for rocprofv3 / timing."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import viabel_amd as vb
from viabel_amd import _lib

d, n = 1024, 4096
rng = np.random.RandomState(2)
A = rng.randn(d, d)
model = vb.CorrelatedGaussianModel(rng.randn(d), covariance=A @ A.T / d + np.eye(d))
fam = vb.FullRankGaussian(d)
obj = vb.ExclusiveKL(fam, model, n)
theta = fam.init_param()
for _ in range(10):
    obj(theta)
t0 = time.perf_counter()
for _ in range(30):
    obj(theta)
print('rng=numpy blocking call: %.1f us; look-ahead (launched, adopted, discarded) = %s'
      % (1e6 * (time.perf_counter() - t0) / 30, _lib.default_engine().legacy_ahead_stats()))
