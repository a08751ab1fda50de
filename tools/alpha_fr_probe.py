"""Dev tool: AlphaDivergence over the dense Gaussian family at the headline shape, 40 blocking calls (for a kernel trace)."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb

D, N = 1024, 4096
approx = vb.FullRankGaussian(D, rng='philox')
model = vb.FunnelModel(D)
theta = approx.pack(np.zeros(D), np.exp(-1.0) * np.eye(D))
obj = vb.AlphaDivergence(approx, model, N, 0.5)
np.random.seed(1)
for _ in range(10):
    obj(theta)
t0 = time.perf_counter()
for _ in range(40):
    obj(theta)
print('%.1f us per call' % (1e6 * (time.perf_counter() - t0) / 40))
