#!/usr/bin/env python3
"""Dev tool: blocking objective(theta) calls in the DEFAULT mode (rng='numpy': the reference's own noise streams) over
families x objectives at a few mid-size shapes -- a table to look for outliers (a stale host/device gate, a per-value copy)."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb


def med_ms(obj, theta, reps=12):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        obj(theta)
        ts.append(time.perf_counter() - t0)
    ts = ts[3:]
    return 1e3 * float(np.median(ts)), 1e3 * float(max(ts))


for D, N in ((10, 100), (50, 500), (100, 1000), (100, 8000), (300, 2000)):
    rng = np.random.RandomState(D)
    model = vb.GaussianModel(0.3 * rng.randn(D), np.exp(0.2 * rng.randn(D)))
    prior = np.concatenate([np.zeros(D), 0.5 * np.ones(D)])
    kw = {'rng': sys.argv[1]} if len(sys.argv) > 1 else {}
    fams = {'MFGaussian': vb.MFGaussian(D, **kw), 'MFStudentT': vb.MFStudentT(D, 7.0, **kw), 'FullRank': vb.FullRankGaussian(D, **kw),
            'MultivariateT': vb.MultivariateT(D, 9.0, **kw), 'LRGaussian(k=4)': vb.LRGaussian(D, k=4, **kw)}
    print('D = %d, N = %d   (median / max ms per call)' % (D, N))
    for name, fam in fams.items():
        theta = fam.init_param()
        if name.startswith('MF'):
            theta[D:] = -0.5
        row = []
        for oname, make in (('ExclusiveKL', lambda: vb.ExclusiveKL(fam, model, N)),
                            ('DIS', lambda: vb.DISInclusiveKL(fam, model, N, ess_target=max(2, N // 8), temper_prior=vb.MFGaussian(D),
                                                              temper_prior_params=prior)),
                            ('Alpha', lambda: vb.AlphaDivergence(fam, model, N, 0.5))):
            try:
                np.random.seed(1)
                m, mx = med_ms(make(), theta)
                row.append('%s %.2f / %.2f' % (oname, m, mx))
            except Exception as exc:      # noqa: BLE001
                row.append('%s -- %s' % (oname, type(exc).__name__))
        print('  %-16s %s' % (name, '   '.join(row)))
