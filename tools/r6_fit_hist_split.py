"""Round 6: RMSProp.optimize on the device at the headline shape: time inside the engine's fit call against the whole call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import viabel_amd as vb
from viabel_amd import _lib
from viabel_amd.optimization import RMSProp

d, n = 1024, 4096
rng = np.random.RandomState(2)
A = rng.randn(d, d)
model = vb.CorrelatedGaussianModel(rng.randn(d), covariance=A @ A.T / d + np.eye(d))
theta = vb.FullRankGaussian(d).init_param()
eng = _lib.default_engine()
inside = {}
for name in dir(eng):
    if name.startswith('_') or not callable(getattr(eng, name)):
        continue

    def make(real, name):
        def wrapped(*a, **k):
            t0 = time.perf_counter()
            try:
                return real(*a, **k)
            finally:
                inside[name] = inside.get(name, 0.0) + time.perf_counter() - t0
        return wrapped
    setattr(eng, name, make(getattr(eng, name), name))
for iters in (300, 300):
    obj = vb.ExclusiveKL(vb.FullRankGaussian(d, rng='philox'), model, n)
    inside.clear()
    t0 = time.perf_counter()
    res = RMSProp(0.001).optimize(iters, obj, theta, on_device=True)
    dt = time.perf_counter() - t0
    print('%d iterations: %.1f us each; inside engine calls: %s' % (iters, 1e6 * dt / iters,
          {k: round(1e6 * v / iters, 1) for k, v in sorted(inside.items(), key=lambda kv: -kv[1])[:4]}), flush=True)
    del res
    obj = vb.ExclusiveKL(vb.FullRankGaussian(d, rng='philox'), model, n)
    ropt = RMSProp(0.001)
    t0 = time.perf_counter()
    obj.device_fit(iters, theta, ropt._device_kind, ropt._device_hyper())
    print('device_fit without history: %.1f us each' % (1e6 * (time.perf_counter() - t0) / iters), flush=True)
