#!/usr/bin/env python3
"""Dev tool: RMSProp iterations of the dense Gaussian family at the headline shape (D = 1024, N = 4096, correlated-
Gaussian target, Philox noise) through the device-resident loop (vb_fit) and through the host loop."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import viabel_amd as vb
from viabel_amd.optimization import RMSProp
D, N = 1024, 4096
rng = np.random.RandomState(2)
A = rng.randn(D, D)
model = vb.CorrelatedGaussianModel(rng.randn(D), covariance=A @ A.T / D + np.eye(D))
fam = vb.FullRankGaussian(D, rng='philox')
obj = vb.ExclusiveKL(fam, model, N)
theta = fam.init_param() if hasattr(fam, 'init_param') else None
opt = RMSProp(0.001)
import contextlib, io
with contextlib.redirect_stderr(io.StringIO()):
    opt.optimize(100, obj, theta, on_device=True)
    for iters in (300,):
        t0 = time.perf_counter()
        r = opt.optimize(iters, obj, theta, on_device=True)
        dt = (time.perf_counter() - t0) / iters
        print('device loop, FullRankGaussian D=%d N=%d: %.1f us per iteration' % (D, N, dt * 1e6))
        t0 = time.perf_counter()
        r = opt.optimize(100, obj, theta, on_device=False)
        dt = (time.perf_counter() - t0) / 100
        print('host loop: %.1f us per iteration' % (dt * 1e6))
