import sys, time
import numpy as np
sys.path.insert(0, '.')
import viabel_amd as vb
D, N = 50, 100
rng = np.random.RandomState(1)
model = vb.GaussianModel(0.1 * rng.randn(D), np.exp(0.1 * rng.randn(D)))
fam = vb.FullRankGaussian(D, seed=3, rng='philox')
obj = vb.ExclusiveKL(fam, model, N)
theta = fam.init_param()
for _ in range(20):
    obj(theta)
t0 = time.perf_counter()
for _ in range(200):
    obj(theta)
print('FullRankGaussian D=%d N=%d: %.1f us per call' % (D, N, 1e6 * (time.perf_counter() - t0) / 200))
