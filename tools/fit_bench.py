#!/usr/bin/env python3
"""Optimiser-loop rate: RMSProp iterations per second through the host loop (one blocking objective call +
numpy update per iteration) and through the device-resident loop (vb_fit), Philox noise regenerated on the
device every iteration.
usage: tools/fit_bench.py [D N iters]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viabel_amd as vb  # noqa: E402
from viabel_amd.optimization import RMSProp  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2000

approx = vb.MFGaussian(D, rng='philox')
obj = vb.ExclusiveKL(approx, vb.FunnelModel(D), N)
theta0 = np.concatenate([np.zeros(D), -np.ones(D)])
opt = RMSProp(0.01, diagnostics=False)
opt.optimize(200, obj, theta0, on_device=False)     # warm-up
t0 = time.perf_counter()
res = opt.optimize(iters, obj, theta0, on_device=False)
dt = time.perf_counter() - t0
print('host loop  D=%d N=%d: %.1f us/iteration (%.0f it/s), final value %.6f' % (
    D, N, 1e6 * dt / iters, iters / dt, res['value_history'][-1]))
if True:
    approx2 = vb.MFGaussian(D, rng='philox')
    obj2 = vb.ExclusiveKL(approx2, vb.FunnelModel(D), N)
    opt2 = RMSProp(0.01)
    opt2.optimize(200, obj2, theta0, on_device=True)
    t0 = time.perf_counter()
    res2 = opt2.optimize(iters, obj2, theta0, on_device=True)
    dt = time.perf_counter() - t0
    same = (np.array_equal(res['value_history'], res2['value_history'])
            and np.array_equal(res['opt_param'], res2['opt_param']))
    print('trajectories identical:', same)
    print('device loop D=%d N=%d: %.1f us/iteration (%.0f it/s), final value %.6f' % (
        D, N, 1e6 * dt / iters, iters / dt, res2['value_history'][-1]))
