"""Round 6: where the host's share of the C3 call goes -- time of the ctypes calls inside one objective call against the whole
call, and what a 265-KB memcpy into pinned memory costs on this host."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import viabel_amd as vb
from viabel_amd import _lib

D, N, df = 256, 16384, 100
model, prior, theta = bench._c3_problem(vb, D)
obj = vb.DISInclusiveKL(vb.MultivariateT(D, df, seed=1, rng='philox'), model, N, ess_target=N // 8,
                        temper_prior=vb.MFGaussian(D), temper_prior_params=prior, use_resampling=False)
for _ in range(30):
    obj(theta)
eng = _lib.default_engine()
inside = [0.0]
counts = {}
for name in dir(eng):
    if name.startswith('_'):
        continue
    real = getattr(eng, name)
    if not callable(real):
        continue

    def make(real, name):
        def wrapped(*a, **k):
            t0 = time.perf_counter()
            try:
                return real(*a, **k)
            finally:
                inside[0] += time.perf_counter() - t0
                counts[name] = counts.get(name, 0) + 1
        return wrapped
    try:
        setattr(eng, name, make(real, name))
    except Exception:
        pass
calls = 200
t0 = time.perf_counter()
for _ in range(calls):
    obj(theta)
total = (time.perf_counter() - t0) / calls
print('per call %.1f us, inside engine methods %.1f us (%s)' % (1e6 * total, 1e6 * inside[0] / calls,
                                                                 {k: v // calls for k, v in counts.items() if v >= calls}))
dst = _lib.pinned_array(theta.size) if hasattr(_lib, 'pinned_array') else np.empty(theta.size)
plain = np.empty(theta.size)
for label, d in (('pinned', dst), ('pageable', plain)):
    ts = []
    for _ in range(200):
        t0 = time.perf_counter()
        np.copyto(d, theta)
        ts.append(time.perf_counter() - t0)
    print('%d doubles -> %s: median %.2f us' % (theta.size, label, 1e6 * statistics.median(ts)))
