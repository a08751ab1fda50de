"""Dev tool: AlphaDivergence (objectives.py:443-463) timings -- blocking objective(theta) calls with fresh Philox noise:
mean-field at the C1 shape (D = 1024 funnel, N = 4096) and the dense family at the headline shape."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb

N = 4096
rng = np.random.RandomState(2)
for name, D, fam, model in (
        ('MFGaussian, funnel D=1024', 1024, lambda D: vb.MFGaussian(D, rng='philox'), lambda D: vb.FunnelModel(D)),
        ('FullRankGaussian, funnel D=1024', 1024, lambda D: vb.FullRankGaussian(D, rng='philox'), lambda D: vb.FunnelModel(D)),
        ('FullRankGaussian, correlated Gaussian D=1024', 1024, lambda D: vb.FullRankGaussian(D, rng='philox'), None),
        ('MultivariateT(df=100), diagonal Gaussian D=256, N=16384', 256, lambda D: vb.MultivariateT(D, 100, rng='philox'), 'diag')):
    if model is None:
        A = rng.randn(D, D)
        m = vb.CorrelatedGaussianModel(rng.randn(D), covariance=A @ A.T / D + np.eye(D))
    elif model == 'diag':
        m = vb.GaussianModel(0.1 * rng.randn(D), np.exp(0.1 * rng.randn(D)))
    else:
        m = model(D)
    approx = fam(D)
    n = 16384 if D == 256 else N
    obj = vb.AlphaDivergence(approx, m, n, 0.5)
    theta = approx.init_param()
    if isinstance(approx, vb.MFGaussian):
        theta[D:] = -1.0
    elif isinstance(approx, vb.FullRankGaussian):
        theta = approx.pack(np.zeros(D), np.exp(-1.0) * np.eye(D))
    else:
        theta = approx.init_param()
    np.random.seed(1)
    for _ in range(10):
        obj(theta)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(30):
            v, g = obj(theta)
        ts.append((time.perf_counter() - t0) / 30)
    print('%-58s %8.1f us per call   value %.6g' % (name, 1e6 * sorted(ts)[1], v))
