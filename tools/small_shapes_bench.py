"""Dev tool: blocking objective calls at the shapes the reference's users run (tens of dimensions, tens to a thousand
samples), every family, rng='philox' and the default rng='numpy'."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb

rng = np.random.RandomState(1)
for D, N in ((10, 10), (50, 100), (100, 1000)):
    model = vb.GaussianModel(0.1 * rng.randn(D), np.exp(0.1 * rng.randn(D)))
    for kind in ('philox', 'numpy'):
        fams = (('MFGaussian', vb.MFGaussian(D, seed=3, rng=kind)), ('MFStudentT', vb.MFStudentT(D, 40, seed=3, rng=kind)),
                ('FullRankGaussian', vb.FullRankGaussian(D, seed=3, rng=kind)), ('MultivariateT', vb.MultivariateT(D, 40, seed=3, rng=kind)),
                ('LRGaussian k=2', vb.LRGaussian(D, seed=3, k=2, rng=kind)))
        row = []
        for name, fam in fams:
            obj = vb.ExclusiveKL(fam, model, N)
            theta = fam.init_param()
            for _ in range(20):
                obj(theta)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(100):
                    obj(theta)
                ts.append((time.perf_counter() - t0) / 100)
            row.append('%s %.0f' % (name, 1e6 * sorted(ts)[1]))
        print('D=%3d N=%4d rng=%-6s us per call: %s' % (D, N, kind, ' | '.join(row)))
