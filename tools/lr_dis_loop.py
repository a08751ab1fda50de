"""Dev tool: blocking DISInclusiveKL calls over LRGaussian(D = 100, k = 4), N = 1000, rng='philox' -- the loop behind
profiles/r05_lr_dis_timeline.txt (tools/timeline.sh lrdis 600 50 tools/lr_dis_loop.py)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import viabel_amd as vb
D, N = 100, 1000
rng = np.random.RandomState(D)
model = vb.GaussianModel(0.3 * rng.randn(D), np.exp(0.2 * rng.randn(D)))
prior = np.concatenate([np.zeros(D), 0.5 * np.ones(D)])
fam = vb.LRGaussian(D, k=4, rng='philox')
theta = fam.init_param()
theta[D:2 * D] = -0.5
obj = vb.DISInclusiveKL(fam, model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D), temper_prior_params=prior)
np.random.seed(1)
for _ in range(40):
    obj(theta)
t0 = time.perf_counter()
for _ in range(100):
    obj(theta)
print('LR DIS philox: %.3f ms per call, eps %.3g ess %.4g' % (10 * (time.perf_counter() - t0), obj._eps, obj._ess))
