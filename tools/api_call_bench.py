#!/usr/bin/env python3
"""Dev tool: the API-level blocking call objective(theta) -> (value, grad) at the headline shape
(FullRankGaussian D = 1024, N = 4096, correlated-Gaussian target), rng='philox' and rng='numpy'."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb
from viabel_amd import _lib

D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rng = np.random.RandomState(2)
A = rng.randn(D, D)
model = vb.CorrelatedGaussianModel(rng.randn(D), covariance=A @ A.T / D + np.eye(D))
eng = _lib.default_engine()
for kind in ('philox', 'numpy'):
    fr = vb.FullRankGaussian(D, seed=1, rng=kind)
    obj = vb.ExclusiveKL(fr, model, N)
    theta = fr.pack(np.zeros(D), np.exp(-1.0) * np.eye(D) + 0.01 * np.tril(np.random.RandomState(3).randn(D, D)))
    n = reps if kind == 'philox' else max(5, reps // 20)
    for _ in range(3):
        obj(theta)
    t0 = time.perf_counter()
    for _ in range(n):
        v, g = obj(theta)
    dt = (time.perf_counter() - t0) / n
    print('rng=%-6s objective(theta): %.1f us per blocking call (value %.6g)' % (kind, dt * 1e6, v))
# pieces (philox): upload, enqueue, fetch
theta = fr.pack(np.zeros(D), np.exp(-1.0) * np.eye(D))
eng.set_model(model.device_spec())
eng.noise_generate(0, N, D, seed=1, stream=0)
eng.sync()
for name, fn in (('set_theta (H2D + sync)', lambda: eng.fullrank_set_theta(theta, D)),
                 ('noise_generate + sync', lambda: (eng.noise_generate(0, N, D, seed=1, stream=1), eng.sync())),
                 ('enqueue + sync (4 kernels)', lambda: (eng.elbo_grad_fullrank_enqueue(0, N, D), eng.sync())),
                 ('fullrank_get (D2H + sync)', lambda: eng.fullrank_get(D))):
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    print('  %-28s %.1f us' % (name, (time.perf_counter() - t0) / reps * 1e6))
