"""Dev tool: the speculative tempering bisection against the look-ahead rounds on random problems -- dimensions, sample
counts, targets, priors that make ESS(eps) non-monotone, eps_prev, max_bisection_its.  Prints the worst differences."""
import os
import sys

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst_eps, worst_ess, bad = 0.0, 0.0, []
interior = 0
for c in range(cases):
    D = int(rng.choice([2, 5, 16, 40, 64]))
    N = int(rng.choice([64, 333, 1000, 4096, 20000]))
    target = float(rng.uniform(1.0, N))
    eps_prev = float(rng.choice([1.0, rng.uniform(0.05, 1.0)]))
    its = int(rng.choice([0, 3, 6, 7, 20, 50, 50, 50, 70]))
    mean = rng.uniform(-1.0, 1.0) + 0.5 * rng.randn(D)
    sd = np.exp(rng.uniform(-1.0, 1.0) + 0.3 * rng.randn(D))
    prior = np.concatenate([rng.uniform(-0.5, 0.5) + 0.3 * rng.randn(D), rng.uniform(-0.7, 0.7) + 0.2 * rng.randn(D)])
    theta = np.concatenate([0.3 * rng.randn(D), rng.uniform(-0.7, 0.3) + 0.2 * rng.randn(D)])
    seed = int(rng.randint(1, 1 << 30))
    out = []
    for env in ({}, {'VB_DIS_BISECT': '0'}):
        for k in ('VB_DIS_BISECT',):
            os.environ.pop(k, None)
        os.environ.update(env)
        obj = vb.DISInclusiveKL(vb.MFGaussian(D, seed=seed, rng='philox'), vb.GaussianModel(mean, sd), N, ess_target=target,
                                temper_prior=vb.MFGaussian(D), temper_prior_params=prior, use_resampling=False)
        obj._max_bisection_its = its
        obj._eps = eps_prev
        try:
            obj(theta)
            out.append((obj._eps, obj._ess))
        except ValueError as exc:
            out.append(('error', str(exc)))
    os.environ.pop('VB_DIS_BISECT', None)
    (e1, s1), (e0, s0) = out
    if e1 == 'error' or e0 == 'error':
        if e1 != e0:
            bad.append((c, out))
        continue
    de = abs(e1 - e0)
    ds = abs(s1 - s0) / max(abs(s0), 1e-300) if np.isfinite(s0) else (0.0 if (np.isnan(s1) == np.isnan(s0)) else 1.0)
    worst_eps, worst_ess = max(worst_eps, de), max(worst_ess, ds)
    interior += 0.0 < e1 < eps_prev
    if de > 1e-12 or ds > 1e-7:
        bad.append((c, D, N, target, eps_prev, its, out))
print('%d cases (%d with an interior eps): worst |eps - eps_lookahead| %.3g, worst relative ESS difference %.3g, mismatches %d'
      % (cases, interior, worst_eps, worst_ess, len(bad)))
for b in bad[:10]:
    print('  ', b)
