#!/usr/bin/env python3
"""Dev tool: time the low-rank Gaussian ExclusiveKL path (blocking calls; noise resident on the device)."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb
from viabel_amd import _lib

D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
k = int(sys.argv[3]) if len(sys.argv) > 3 else 8
kind = sys.argv[4] if len(sys.argv) > 4 else 'funnel'
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 300
eng = _lib.default_engine()
model = vb.FunnelModel(D) if kind == 'funnel' else vb.GaussianModel(np.zeros(D), np.ones(D))
eng.set_model(model.device_spec())
fam = vb.LRGaussian(D, k=k)
rng = np.random.RandomState(0)
theta = fam.pack(np.zeros(D), -np.ones(D), 0.05 * rng.randn(D, k))
ring = 8
for s in range(ring):
    eng.noise_generate(s, N, D, seed=1, stream=s)
    eng.noise_generate(ring + s, N, k, seed=2, stream=s)
if k > 16:      # ranks beyond the streaming kernel: the GEMM-assembled sums (vb_elbo_sums_lowrank)
    def call(i):
        f, g, ge, gz = eng.elbo_sums_lowrank(i % ring, ring + i % ring, N, D, k, theta)
        return f, gz
else:
    def call(i):
        return eng.elbo_grad_lowrank(i % ring, ring + i % ring, N, D, k, theta)
for i in range(20):
    call(i)
t0 = time.perf_counter()
for i in range(steps):
    v, g = call(i)
dt = (time.perf_counter() - t0) / steps
print('LRGaussian D=%d N=%d k=%d %s: %.1f us per blocking call (%.0f evals/s); value %.8g |grad| %.6g'
      % (D, N, k, kind, dt * 1e6, 1 / dt, v, np.linalg.norm(g)))
