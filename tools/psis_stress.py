"""Dev tool: the multi-workgroup PSIS kernel against the single-workgroup one on random weight vectors (sizes across the
workgroup-count and registers-per-thread boundaries, heavy / light / discrete / clustered weights, Reff)."""
import os
import sys

import numpy as np

sys.path.insert(0, '.')
from viabel_amd import _lib
from viabel_amd._psis import psislw

_lib.default_engine()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst, kbad, bad = 0.0, 0, []
for c in range(cases):
    n = int(rng.choice([1025, 1100, 2047, 2048, 2049, 5000, 16383, 16384, 16385, 30000, 65535, 65536, 65537, 131073, 200000,
                        262144, int(rng.randint(1025, 262145))]))
    kind = rng.randint(6)
    if kind == 0:
        lw = rng.uniform(0.5, 3.0) * rng.standard_t(rng.uniform(1.5, 8.0), n)
    elif kind == 1:
        lw = rng.uniform(-100, 100) + rng.uniform(1e-6, 1.0) * rng.randn(n)
    elif kind == 2:
        lw = np.round(rng.uniform(0.5, 2.0) * rng.standard_t(3.0, n), int(rng.randint(0, 3)))
    elif kind == 3:
        lw = -rng.uniform(0.1, 2.0) * rng.randn(n) ** 2
    elif kind == 4:
        lw = rng.uniform(0.5, 2.0) * rng.randn(n)
        lw[rng.rand(n) < 0.05] = -np.inf
    else:
        lw = np.log(rng.pareto(rng.uniform(0.5, 3.0), n) + 1e-300)
    reff = float(rng.choice([1.0, 1.0, rng.uniform(0.2, 3.0)]))
    res = []
    for grid in ('1', '0'):
        os.environ['VB_PSIS_GRID'] = grid
        res.append(psislw(lw, Reff=reff))
    (gs, gk), (ss, sk) = res
    same_k = gk == sk or (np.isnan(gk) and np.isnan(sk))
    fin = np.isfinite(ss)
    d = float(np.max(np.abs(gs[fin] - ss[fin]))) if fin.any() else 0.0
    ok_inf = np.array_equal(np.isfinite(gs), fin)
    worst = max(worst, d)
    if not same_k:
        kbad += 1
    if not same_k or d > 1e-11 or not ok_inf:
        bad.append((c, n, kind, reff, gk, sk, d, ok_inf))
os.environ.pop('VB_PSIS_GRID', None)
print('%d cases: k-hat differs in %d, worst |smoothed - smoothed_single| %.3g, mismatches %d' % (cases, kbad, worst, len(bad)))
for b in bad[:10]:
    print('  ', b)
