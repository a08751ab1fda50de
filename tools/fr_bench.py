#!/usr/bin/env python3
"""Dev tool: time the full-rank ExclusiveKL pipeline with the parameter resident on the device."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb
from viabel_amd import _lib

D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
model_kind = sys.argv[3] if len(sys.argv) > 3 else 'gauss_full'
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
flags = _lib.FLAG_PATH_DERIV if (len(sys.argv) > 5 and sys.argv[5] == 'path_deriv') else 0
eng = _lib.default_engine()
rng = np.random.RandomState(2)
if model_kind == 'gauss_full':
    A = rng.randn(D, D)
    S = A @ A.T / D + np.eye(D)
    model = vb.CorrelatedGaussianModel(rng.randn(D), covariance=S)
elif model_kind == 'gauss_diag':
    model = vb.GaussianModel(rng.randn(D), np.exp(0.3 * rng.randn(D)))
else:
    model = vb.FunnelModel(D)
eng.set_model(model.device_spec())
fr = vb.FullRankGaussian(D)
L = np.exp(-1.0) * np.eye(D) + 0.01 * np.tril(np.random.RandomState(3).randn(D, D))
theta = fr.pack(np.zeros(D), L)
ring = 8
for s in range(ring):
    eng.noise_generate(s, N, D, seed=1, stream=s)
eng.fullrank_set_theta(theta, D)
for i in range(5):
    eng.elbo_grad_fullrank_enqueue(i % ring, N, D, flags=flags)
eng.sync()
t0 = time.perf_counter()
for i in range(steps):
    eng.elbo_grad_fullrank_enqueue(i % ring, N, D, flags=flags)
t_enq = (time.perf_counter() - t0) / steps
eng.sync()
dt = (time.perf_counter() - t0) / steps
print('host enqueue %.1f us/eval' % (t_enq * 1e6))
flops = 4.0 * N * D * D + (2.0 * N * D * D if model_kind == 'gauss_full' else 0.0)
print('D=%d N=%d model=%s%s: %.1f us/eval, %.0f evals/s, %.2f TFLOP/s (dense convention %.2f GFLOP/eval)'
      % (D, N, model_kind, ' path_deriv' if flags else '', dt * 1e6, 1 / dt, flops / dt / 1e12, flops / 1e9))
v, g = eng.fullrank_get(D)
print('value %.10g |grad| %.6g' % (v, np.linalg.norm(g)))
