#!/usr/bin/env python3
"""Bayesian regression with a dense-covariance Gaussian, fitted entirely on the device.

    python examples/regression.py [logistic|poisson|linear] [D] [n_data]

Target: a GLM with a N(0, prior_sd) prior on the coefficients (`LogisticRegressionModel`,
`PoissonRegressionModel`, `LinearRegressionModel`: the likelihood runs as two fp64 MFMA GEMMs per objective call).
Approximation: `FullRankGaussian` with Philox noise, ExclusiveKL with the path-derivative estimator
(`use_path_deriv=True`, viabel/objectives.py:156-159).  Optimiser: the reference's Adam through the device-resident
loop (`vb_fit`): noise generation, objective and parameter update are chained on one HIP stream, the host only sees
the result.  For the linear model the exact posterior is printed next to the fit.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viabel_amd as vb   # noqa: E402
from viabel_amd.optimization import Adam   # noqa: E402


def main(kind='logistic', D=64, n_data=2000, num_mc_samples=256):
    rng = np.random.RandomState(0)
    X = rng.randn(n_data, D) / np.sqrt(D)
    beta = rng.randn(D)
    if kind == 'logistic':
        y = (rng.rand(n_data) < 1.0 / (1.0 + np.exp(-X @ beta))).astype(float)
        model = vb.LogisticRegressionModel(X, y, prior_sd=10.0)
    elif kind == 'poisson':
        y = rng.poisson(np.exp(X @ beta)).astype(float)
        model = vb.PoissonRegressionModel(X, y, prior_sd=10.0)
    else:
        y = X @ beta + 0.5 * rng.randn(n_data)
        model = vb.LinearRegressionModel(X, y, prior_sd=10.0, noise_sd=0.5)
    approx = vb.FullRankGaussian(D, rng='philox')
    objective = vb.ExclusiveKL(approx, model, num_mc_samples, use_path_deriv=True)
    theta = approx.pack(np.zeros(D), np.eye(D))
    opt = Adam(0.05, iterate_avg_prop=None)
    t0 = time.perf_counter()
    n_total = 0
    for lr, iters in ((0.05, 3000), (0.01, 3000), (0.002, 2000)):
        opt._learning_rate = lr
        res = opt.optimize(iters, objective, theta)          # device-resident: objective.supports_device_fit()
        theta = res['opt_param']
        n_total += iters
        print('lr %.3f: -ELBO estimate %.4f' % (lr, np.mean(res['value_history'][-200:])))
    dt = time.perf_counter() - t0
    mean, cov = approx.mean_and_cov(theta)
    print('%d iterations in %.2f s (%.0f us per iteration, %d-sample gradients, %d parameters)'
          % (n_total, dt, 1e6 * dt / n_total, num_mc_samples, theta.size))
    print('coefficient error |mean - beta| / |beta| = %.3f, mean posterior sd %.3f'
          % (np.linalg.norm(mean - beta) / np.linalg.norm(beta), np.sqrt(np.diag(cov)).mean()))
    if kind == 'linear':
        exact_cov = np.linalg.inv(X.T @ X / 0.25 + np.eye(D) / 100.0)
        exact_mean = exact_cov @ X.T @ y / 0.25
        print('exact posterior: max |mean error| / sd = %.2e, max |cov error| / max var = %.2e'
              % (np.max(np.abs(mean - exact_mean) / np.sqrt(np.diag(exact_cov))),
                 np.max(np.abs(cov - exact_cov)) / np.max(np.diag(exact_cov))))
    return theta


if __name__ == '__main__':
    a = sys.argv[1:]
    main(a[0] if a else 'logistic', int(a[1]) if len(a) > 1 else 64, int(a[2]) if len(a) > 2 else 2000)
