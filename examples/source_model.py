#!/usr/bin/env python3
"""A target outside the built-in set: the log density (and its gradient) written as a HIP device function.

The reference's primary entry is ``bbvi(dimension, log_density=<Python callable>)`` with autograd supplying the
gradient (viabel/convenience.py:75).  On the GPU the callable is device code: ``vb_log_density`` below is compiled for
the MI355X with hiprtc when the model is first used and runs as a row kernel inside the ELBO-gradient pipeline.  The
model is the robust (Student-t) regression of the reference's documentation (docs/source/robust-regression.ipynb).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import viabel_amd as vb  # noqa: E402

SRC = r"""
// params = [n, nu, s, tau | X (n x d, row-major) | y (n)]:  y_i ~ StudentT(nu, x_i' z, s),  z ~ N(0, tau^2 I)
__device__ double vb_log_density(const double* z, int d, const double* p, double* g) {
  const int n = (int)p[0];
  const double nu = p[1], s = p[2], tau = p[3];
  const double* X = p + 4;
  const double* y = X + (long long)n * d;
  double f = 0.0;
  for (int j = 0; j < d; ++j) {
    f -= 0.5 * z[j] * z[j] / (tau * tau);
    if (g) g[j] = -z[j] / (tau * tau);
  }
  for (int i = 0; i < n; ++i) {
    double eta = 0.0;
    for (int j = 0; j < d; ++j) eta += X[(long long)i * d + j] * z[j];
    const double r = y[i] - eta, q = 1.0 + r * r / (nu * s * s);
    f -= 0.5 * (nu + 1.0) * log(q);
    if (g) {
      const double c = (nu + 1.0) * r / (nu * s * s * q);
      for (int j = 0; j < d; ++j) g[j] += c * X[(long long)i * d + j];
    }
  }
  return f;
}
"""

# The same target as the density ALONE -- what the reference takes (autograd derives the gradient there, forward-mode dual
# numbers on the device here).  `vb::dot` is the linear predictor as one operation of that arithmetic: the threads of a
# sample share the product and the derived gradient runs as fast as the hand-written one.
AUTO_SRC = r"""
template <class T>
__device__ T vb_log_density(vb::vec<T> z, int d, const double* p) {
  const int n = (int)p[0];
  const double nu = p[1], s = p[2], tau = p[3];
  const double* X = p + 4;
  const double* y = X + (long long)n * d;
  T f = 0.0;
  for (int j = 0; j < d; ++j) f -= 0.5 * z[j] * z[j] / (tau * tau);
  for (int i = 0; i < n; ++i) {
    const T r = y[i] - vb::dot(X + (long long)i * d, z, d);
    f -= 0.5 * (nu + 1.0) * log(1.0 + r * r / (nu * s * s));
  }
  return f;
}
"""

# (with more data: `#define VB_LOG_DENSITY_PARTS 8` + vb_log_density_part(z, d, p, g, part, n_parts) summing the
#  observations part, part + 8, ... puts eight threads on every sample -- see SourceModel's docstring)
rng = np.random.RandomState(0)
D, n = 5, 300
X = rng.randn(n, D)
beta = np.array([2.0, -1.0, 0.5, 0.0, 3.0])
y = X @ beta + 0.3 * rng.standard_t(3.0, size=n)
y[:10] += 15.0                                   # outliers the t likelihood shrugs off
model = vb.SourceModel(D, SRC, np.concatenate([[n, 4.0, 0.3, 10.0], X.ravel(), y]))
# the gradient above is hand-written: compare it with differences of the density before trusting a fit to it
print('gradient check (max relative deviation from central differences): %.1e' % model.check_gradient(rng.randn(8, D)))

auto = vb.SourceModel(D, AUTO_SRC, model.params, grad='auto')
pts = rng.randn(8, D)
print('density-only model: gradient differs from the hand-written one by %.1e'
      % (np.max(np.abs(auto.grad(pts) - model.grad(pts))) / np.max(np.abs(model.grad(pts)))))

res = vb.bbvi(D, log_density=model, approx=vb.MFGaussian(D, rng='philox'), n_iters=4000, num_mc_samples=64,
              learning_rate=0.05)
mean, sd = res['opt_param'][:D], np.exp(res['opt_param'][D:])
print('true coefficients ', beta)
print('posterior mean    ', np.round(mean, 3))
print('posterior stdev   ', np.round(sd, 3))
out = vb.vi_diagnostics(res['opt_param'], objective=res['objective'], n_samples=20000)
