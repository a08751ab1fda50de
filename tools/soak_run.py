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

# the paths reworked last: AlphaDivergence with the hinted seed (all five families), the low-rank DIS / alpha calls with
# the staged parameter image and the pack kernel, optimize() with streamed rows
Ds, Ns = 96, 2000
rng = np.random.RandomState(4)
model_s = vb.GaussianModel(0.3 * rng.randn(Ds), np.exp(0.2 * rng.randn(Ds)))
prior_s = np.concatenate([np.zeros(Ds), 0.5 * np.ones(Ds)])
for name, fam in (('MFGaussian', vb.MFGaussian(Ds, rng='philox')), ('MFStudentT', vb.MFStudentT(Ds, 7, rng='philox')),
                  ('FullRankGaussian', vb.FullRankGaussian(Ds, rng='philox')), ('LRGaussian', vb.LRGaussian(Ds, k=5, rng='philox')),
                  ('MultivariateT', vb.MultivariateT(Ds, 9, rng='philox'))):
    th = fam.init_param()
    objs = [vb.AlphaDivergence(fam, model_s, Ns, 0.5)]
    if name == 'LRGaussian':
        objs.append(vb.DISInclusiveKL(fam, model_s, Ns, ess_target=Ns // 8, temper_prior=vb.MFGaussian(Ds), temper_prior_params=prior_s))
    for obj in objs:
        for i in range(600):
            if i % 97 == 0:
                np.random.randn(3)            # the caller's own draws in between: hints go stale and must not be adopted
            v, g = obj(th)
            if not (np.isfinite(v) and np.all(np.isfinite(g))):
                raise SystemExit('non-finite result at call %d of %s %s' % (i, name, type(obj).__name__))
            th = th - 1e-4 * g
        print('%-18s %-16s 600 calls, %.1f s so far, value %.6g; memory: %s' % (name, type(obj).__name__, time.time() - t0, v, used_mb()))
from viabel_amd.optimization import RMSProp
import contextlib, io
fam = vb.FullRankGaussian(300, rng='philox')          # rows of 45 450 doubles = 355 KB: streamed
A = rng.randn(300, 300)
obj = vb.ExclusiveKL(fam, vb.CorrelatedGaussianModel(rng.randn(300), covariance=A @ A.T / 300 + np.eye(300)), 1024)
with contextlib.redirect_stderr(io.StringIO()):
    for rep in range(6):
        r = RMSProp(0.001, diagnostics=True).optimize(150, obj, fam.init_param(), on_device=True)
        assert np.all(np.isfinite(r['variational_param_history'])) and np.all(np.isfinite(r['value_history']))
print('optimize with streamed rows: 6 x 150 iterations, %.1f s; memory: %s' % (time.time() - t0, used_mb()))
