#!/usr/bin/env python3
"""Dev tool: one of the short blocking-call workloads of bench.py's secondary legs in a plain loop, for tools/timeline.sh
(where does the GPU wait for the host?).  usage: small_legs_run.py lr8 | lr32 | dis_mf | alpha_mf | bbvi | mf_call"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb
from viabel_amd import _lib

what = sys.argv[1]
eng = _lib.default_engine()
rng = np.random.RandomState(0)
D, N = 1024, 4096
if what in ('lr8', 'lr32'):
    k = int(what[2:])
    eng.set_model(vb.FunnelModel(D).device_spec())
    fam = vb.LRGaussian(D, k=k)
    theta = fam.pack(np.zeros(D), -np.ones(D), 0.05 * rng.randn(D, k))
    eng.noise_generate(0, N, D, seed=1, stream=0)
    eng.noise_generate(1, N, k, seed=2, stream=0)
    call = ((lambda: eng.elbo_sums_lowrank(0, 1, N, D, k, theta)) if k > 16 else (lambda: eng.elbo_grad_lowrank(0, 1, N, D, k, theta)))
elif what == 'dis_mf':
    Dm, Nm = 64, 16384
    mrng = np.random.RandomState(7)
    model = vb.GaussianModel(0.3 + 0.3 * mrng.randn(Dm), np.exp(0.2 * mrng.randn(Dm)))
    prior = np.zeros(2 * Dm)
    theta = prior + 0.02 * mrng.randn(2 * Dm)
    obj = vb.DISInclusiveKL(vb.MFGaussian(Dm, seed=11, rng='philox'), model, Nm, ess_target=Nm // 8,
                            temper_prior=vb.MFGaussian(Dm), temper_prior_params=prior, use_resampling=False)

    def call():
        obj._eps = 1.0
        obj(theta)
elif what == 'alpha_mf':
    obj = vb.AlphaDivergence(vb.MFGaussian(D, rng='philox'), vb.FunnelModel(D), N, alpha=0.5)
    theta = np.concatenate([np.zeros(D), -np.ones(D)])
    call = lambda: obj(theta)
elif what == 'mf_call':
    obj = vb.ExclusiveKL(vb.MFGaussian(D, rng='philox'), vb.FunnelModel(D), N)
    theta = np.concatenate([np.zeros(D), -np.ones(D)])
    call = lambda: obj(theta)
elif what == 'bbvi':
    t0 = time.perf_counter()
    res = vb.bbvi(2, n_iters=3000, num_mc_samples=10, approx=None, log_density=None, objective=vb.ExclusiveKL(
        vb.MFGaussian(2, rng='philox'), vb.FunnelModel(2), 10), learning_rate=0.5) if False else None
    import bench
    print(bench.bbvi_quickstart_leg(vb, n_iters=2000)['device_funnel_model'])
    sys.exit(0)
for _ in range(10):
    call()
ts = []
for _ in range(60):
    t0 = time.perf_counter()
    call()
    ts.append(time.perf_counter() - t0)
print('%s: %.1f us per call (median of 60)' % (what, 1e6 * float(np.median(ts))))
