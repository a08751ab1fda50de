#!/usr/bin/env python3
"""The reference's quickstart (docs/source/quickstart.ipynb) on the HIP engine: Neal's funnel in 2 dimensions,
black-box VI with the default settings (ELBO, mean-field Gaussian, RMSProp with RAABBVI step-size adaptation),
then the Pareto k-hat / 2-divergence diagnostics and PSIS importance resampling.

    python examples/quickstart.py [n_iters]

Differences from the notebook: the target is a device model (`FunnelModel(2, scale_index=1)` is exactly the
notebook's `log_density`: x[:, 1] ~ N(0, 1) is the log scale, x[:, 0] ~ N(0, exp(x[:, 1]))) instead of an
autograd callable, and nothing is plotted.
"""
import sys
import warnings

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viabel_amd as vb   # noqa: E402


def main(n_iters=30000):
    D = 2
    model = vb.FunnelModel(D, scale_index=1, log_sigma_stdev=1.0)
    np.random.seed(0)
    results = vb.bbvi(D, log_density=model, learning_rate=0.5, n_iters=n_iters)
    opt = results['opt_param']
    print('variational mean      ', opt[:D])
    print('variational log-stdev ', opt[D:])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        diagnostics = vb.vi_diagnostics(opt, objective=results['objective'], n_samples=100000)
    weights = np.exp(diagnostics['smoothed_log_weights'])
    samples = diagnostics['samples']
    subset = np.random.choice(samples.shape[1], size=1000, p=weights)
    resampled = samples[:, subset]
    print('importance-resampled mean %s, stdev %s (funnel: mean 0; log-scale stdev 1)'
          % (resampled.mean(axis=1), resampled.std(axis=1)))
    return results, diagnostics


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 30000)
