"""Dev tool: C3 shape with psis_smooth=True (BASELINE configs[3]: "DISInclusiveKL with PSIS reweighting"), per-call time
and the engine calls it is made of."""
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
for resample in (False, True):
    approx = vb.MultivariateT(D, 100, seed=1, rng='philox')
    obj = vb.DISInclusiveKL(approx, model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, use_resampling=resample, num_resampling_batches=1, psis_smooth=True)
    eng = _lib.default_engine()
    np.random.seed(3)
    for _ in range(10):
        obj(theta)
    acc = {}

    def wrap(name):
        f = getattr(eng, name)

        def g(*a, **k):
            t0 = time.perf_counter()
            r = f(*a, **k)
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
            return r
        setattr(eng, name, g)
    for name in [n for n in dir(eng) if not n.startswith('_') and callable(getattr(eng, n))]:
        wrap(name)
    K = 50
    t0 = time.perf_counter()
    for _ in range(K):
        v, g = obj(theta)
    tot = time.perf_counter() - t0
    print('psis_smooth=True, resampling=%s: %.1f us per call, khat %.3f, value %.6g' % (resample, 1e6 * tot / K, obj._khat, v))
    for k, t in sorted(acc.items(), key=lambda kv: -kv[1])[:8]:
        print('  %-28s %.1f us' % (k, 1e6 * t / K))
    print('  python outside the engine    %.1f us' % (1e6 * (tot - sum(acc.values())) / K))
    for name in list(acc):
        delattr(eng, name)
