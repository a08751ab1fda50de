"""Dev tool: blocking objective calls whose cost is mostly launch and host overhead (mean-field C1 shape, small shapes)."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb

rng = np.random.RandomState(1)
for D, N, kind in ((1024, 4096, 'funnel'), (100, 10, 'gauss'), (256, 1024, 'gauss')):
    model = vb.FunnelModel(D) if kind == 'funnel' else vb.GaussianModel(0.1 * rng.randn(D), np.exp(0.1 * rng.randn(D)))
    obj = vb.ExclusiveKL(vb.MFGaussian(D, seed=3, rng='philox'), model, N)
    theta = np.concatenate([0.1 * rng.randn(D), -1.0 + 0.1 * rng.randn(D)])
    for _ in range(50):
        obj(theta)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(500):
            v, g = obj(theta)
        ts.append((time.perf_counter() - t0) / 500)
    print('MFGaussian D=%4d N=%5d %-6s: %.1f us per blocking call (median of 5 blocks; min %.1f)' % (D, N, kind, 1e6 * sorted(ts)[2], 1e6 * min(ts)))
