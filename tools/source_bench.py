#!/usr/bin/env python3
"""Dev tool: timing of a SourceModel (log density as HIP source, vb_usermodel.hip) under ExclusiveKL for the three
Gaussian families, next to the numpy oracle of the same density on the host.  Robust (Student-t) regression of the
reference's docs; D coefficients, n_data observations, N Monte-Carlo samples.
usage: python tools/source_bench.py [D n_data N calls]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import viabel_amd as vb
from test_gpu_source_model import ROBUST_REGRESSION_SRC, ROBUST_REGRESSION_PARTS_SRC, RobustRegressionOracle

D, n_data, N, calls = (int(a) for a in (sys.argv[1:5] + ['64', '512', '4096', '200'][len(sys.argv) - 1:]))
rng = np.random.RandomState(3)
X = rng.randn(n_data, D) / np.sqrt(D)
y = X @ rng.randn(D) + 0.3 * rng.standard_t(3.0, size=n_data)
params = np.concatenate([[n_data, 4.0, 0.5, 3.0], X.ravel(), y])
oracle = RobustRegressionOracle(X, y, 4.0, 0.5, 3.0)
print('robust regression as HIP source: D=%d n_data=%d N=%d' % (D, n_data, N))
variants = [('one thread per sample', ROBUST_REGRESSION_SRC)] + [
    ('%d threads per sample' % k, ROBUST_REGRESSION_PARTS_SRC % k) for k in (8, 32)]
for vname, src in variants:
  model = vb.SourceModel(D, src, params)
  print('-- %s; gradient check %.1e' % (vname, model.check_gradient(rng.randn(4, D))))
  fams = [('MFGaussian', vb.MFGaussian(D, rng='philox')), ('LRGaussian k=4', vb.LRGaussian(D, k=4, rng='philox')),
          ('FullRankGaussian', vb.FullRankGaussian(D, rng='philox'))]
  for name, fam in fams:
      obj = vb.ExclusiveKL(fam, model, N)
      theta = fam.init_param()
      if name.startswith('MF') or name.startswith('LR'):
          theta[D:2 * D] = -1.0
      for _ in range(5):
          obj(theta)
      t0 = time.perf_counter()
      for _ in range(calls):
          v, g = obj(theta)
      dt = (time.perf_counter() - t0) / calls
      print('%-18s %8.1f us per objective call (value %.6g, |grad| %.4g)' % (name, 1e6 * dt, v, np.linalg.norm(g)))
z = rng.randn(N, D)
t0 = time.perf_counter()
for _ in range(3):
    oracle.logp(z), oracle.grad(z)
print('numpy oracle, f and grad of the same %d samples on the host: %.1f ms' % (N, 1e3 * (time.perf_counter() - t0) / 3))
