#!/usr/bin/env python3
"""Dev tool: where does a default bbvi() run (RAABBVI over RMSProp) spend its time?  usage: tools/bbvi_prof.py [D N iters]"""
import cProfile
import contextlib
import io
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viabel_amd as vb  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 100
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
rng = np.random.RandomState(0)
mean, sd = rng.randn(D), np.exp(0.5 * rng.randn(D))
for kind in ('philox', 'numpy'):
    approx = vb.MFGaussian(D, rng=kind)
    objective = vb.ExclusiveKL(approx, vb.GaussianModel(mean, sd), N)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    with contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):
        pr.enable()
        res = vb.bbvi(D, objective=objective, n_iters=iters, learning_rate=0.1)
        pr.disable()
    dt = time.perf_counter() - t0
    n = len(res['value_history'])
    m, c = approx.mean_and_cov(res['opt_param'])
    print('rng=%s: %d iterations in %.2f s (%.0f us / iteration); max |mean err| / sd %.3f' % (
        kind, n, dt, 1e6 * dt / n, np.max(np.abs(m - mean) / sd)))
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats('tottime').print_stats(8)
    print('\n'.join(l for l in out.getvalue().splitlines() if '/' in l or 'ncalls' in l or '{' in l)[:1800])
