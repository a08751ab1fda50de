#!/usr/bin/env python3
"""Dev tool: a few thousand blocking calls of the paths reworked late in round 5 (resident DIS step with look-ahead noise,
PSIS, the reference-identical step, mean-field DIS, bbvi) with device memory watched: no growth, no stall, finite results."""
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import viabel_amd as vb
from test_gpu_full_size import c3_problem


def used_mb():
    out = subprocess.run(['rocm-smi', '--showmemuse', '--json'], capture_output=True, text=True).stdout
    import json
    try:
        d = json.loads(out)
        k = sorted(d)[0]
        return d[k]
    except Exception:
        return out.strip()[:200]


D, N = 256, 16384
np.random.seed(5)
mean, sd, prior, theta = c3_problem(np.random.RandomState(33), D)
model = vb.GaussianModel(mean, sd)
t0 = time.time()
for kind, kw, calls in (('philox', dict(use_resampling=False), 1500), ('philox', dict(use_resampling=True, num_resampling_batches=3), 1500),
                        ('philox', dict(use_resampling=False, psis_smooth=True), 800), ('numpy', dict(use_resampling=False), 300)):
    obj = vb.DISInclusiveKL(vb.MultivariateT(D, 100, seed=1, rng=kind), model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, **kw)
    th = theta.copy()
    for i in range(calls):
        v, g = obj(th)
        if not (np.isfinite(v) and np.all(np.isfinite(g))):
            raise SystemExit('non-finite result at call %d of %s %s' % (i, kind, kw))
        th = th - 1e-4 * g
    print('%-7s %-60s %5d calls, %.1f s so far, value %.6g; memory: %s' % (kind, kw, calls, time.time() - t0, v, used_mb()))
res = vb.bbvi(2, n_iters=4000, num_mc_samples=10, objective=vb.ExclusiveKL(vb.MFGaussian(2), vb.FunnelModel(2), 10), learning_rate=0.5)
print('bbvi done; %.1f s; memory: %s' % (time.time() - t0, used_mb()))
