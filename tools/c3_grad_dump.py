#!/usr/bin/env python3
"""Dev tool: the C3-shaped DIS gradient (weighted, throughput mode) dumped to gpurun_out/c3_grad_<tag>.npy -- run under
different VB_MVT_* settings and compare with tools/c3_grad_dump.py --compare a b."""
import sys

import numpy as np

sys.path.insert(0, '.')
if sys.argv[1] == '--compare':
    a, b = (np.load('gpurun_out/c3_grad_%s.npy' % t) for t in sys.argv[2:4])
    D = 256
    for name, sl in (('value', slice(0, 1)), ('mu', slice(1, 1 + D)), ('L', slice(1 + D, None))):
        x, y = a[sl], b[sl]
        print('%-5s max |a - b| %.3e   |a| %.6g |b| %.6g' % (name, np.abs(x - y).max(), np.linalg.norm(x), np.linalg.norm(y)))
    sys.exit(0)
import viabel_amd as vb
sys.path.insert(0, 'tests')
from test_gpu_full_size import c3_problem
D, N = 256, 16384
np.random.seed(5)
approx = vb.MultivariateT(D, 100, seed=1, rng='philox')
mean, sd, prior, theta = c3_problem(np.random.RandomState(33), D)
model = vb.GaussianModel(mean, sd)
obj = vb.DISInclusiveKL(approx, model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D), temper_prior_params=prior,
                        use_resampling=False, num_resampling_batches=1)
v, g = obj(theta)
np.save('gpurun_out/c3_grad_%s.npy' % sys.argv[1], np.concatenate([[v], g]))
print(sys.argv[1], v, np.linalg.norm(g))
