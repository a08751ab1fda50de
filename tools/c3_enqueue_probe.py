"""Dev tool: how long does the HOST take to enqueue one C3-shape call (chi-square draws, normals, deferred refresh) before
the step call that synchronises?  If this is close to the whole call, the call is host-bound and GPU-side savings do
not show."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import viabel_amd as vb
from viabel_amd import _lib
from test_gpu_full_size import c3_problem

D, N = 256, 16384
mean, sd, prior, theta = c3_problem(np.random.RandomState(33), D)
model = vb.GaussianModel(mean, sd)
approx = vb.MultivariateT(D, 100, seed=1, rng='philox')
obj = vb.DISInclusiveKL(approx, model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                        temper_prior_params=prior, use_resampling=False, num_resampling_batches=1)
eng = _lib.default_engine()
for _ in range(20):
    obj(theta)
# wrap the engine's methods with timers
acc = {}
def wrap(name):
    f = getattr(eng, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    setattr(eng, name, g)
for name in ('chisq_generate', 'noise_generate', 'dis_refresh_mvt_deferred', 'dis_step_mvt_packed', 'set_model'):
    if hasattr(eng, name):
        wrap(name)
K = 200
eng.sync()
t0 = time.perf_counter()
for _ in range(K):
    obj(theta)
tot = time.perf_counter() - t0
print('call %.1f us' % (1e6 * tot / K))
for k, v in acc.items():
    print('  %-28s %.1f us' % (k, 1e6 * v / K))
print('  python outside the engine    %.1f us' % (1e6 * (tot - sum(acc.values())) / K))
