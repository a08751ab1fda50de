#!/usr/bin/env python3
"""Dev tool: how far inside their tolerances (1e-11 value, 1e-9 gradient) the resident reference-identical forms of the t
family sit at D = 256 / N = 16 384 -- ExclusiveKL entropy / path-derivative form, AlphaDivergence -- against the oracle."""
import sys

import numpy as np

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import viabel_amd as vb
from oracle import families as ofam
from oracle import models as omod
from oracle import objectives as oobj

D, N, df = 256, 16384, 9.0
rng = np.random.RandomState(41)
mean, sd = 0.3 * rng.randn(D), np.exp(0.2 * rng.randn(D))
A = rng.randn(D, D)
S = A @ A.T / D + np.eye(D)
m2 = rng.randn(D)
targets = {'gauss_diag': (vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)),
           'funnel': (vb.FunnelModel(D, D // 2), omod.Funnel(D, D // 2)),
           'gauss_full': (vb.CorrelatedGaussianModel(m2, covariance=S), omod.GaussFull(m2, np.linalg.inv(S)))}
B = rng.randn(D, D)
theta = np.concatenate([0.2 * rng.randn(D), ofam.psd_to_free(0.05 * (B @ B.T / D + 0.5 * np.eye(D)))])
omvt = ofam.MultivariateT(D, df)
for name, (model, omodel) in targets.items():
    for pd in (False, True):
        v, g = vb.ExclusiveKL(vb.MultivariateT(D, df, seed=6), model, N, use_path_deriv=pd)(theta)
        ov, og = oobj.exclusive_kl(omvt, omodel, theta, omvt.draw_noise(np.random.RandomState(6), N), pd)
        print('%-10s ekl path=%-5s value %.1e grad %.1e' % (name, pd, abs(v - ov) / abs(ov), np.max(np.abs(g - og)) / np.max(np.abs(og))))
    for alpha in (2.0, 0.5):
        np.random.seed(17)
        v, g = vb.AlphaDivergence(vb.MultivariateT(D, df), model, N, alpha)(theta)
        np.random.seed(17)
        noise = omvt.draw_noise(np.random.RandomState(np.random.randint(2 ** 32)), N)
        ov, og = oobj.alpha_divergence(omvt, omodel, theta, noise, alpha)
        print('%-10s alpha=%.1f        value %.1e grad %.1e' % (name, alpha, abs(v - ov) / abs(ov), np.max(np.abs(g - og)) / np.max(np.abs(og))))
