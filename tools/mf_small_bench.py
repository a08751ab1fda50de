"""Dev tool: mean-field calls on narrow shapes with the noise in memory (rng='numpy' ELBO, DIS score pass), where the
number of row blocks decides between the streaming pass and the finalize kernel's chain of partial sums.
VB_MF_MAX_ROW_BLOCKS=512 restores the round-3 split."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb
from viabel_amd import _lib

eng = _lib.default_engine()
rng = np.random.RandomState(1)
for D, N in ((64, 16384), (128, 8192), (256, 4096), (64, 4096)):
    model = vb.GaussianModel(0.1 * rng.randn(D), np.exp(0.1 * rng.randn(D)))
    eng.set_model(model.device_spec())
    eng.noise_generate(0, N, D, seed=1, stream=0)
    theta = np.concatenate([0.1 * rng.randn(D), -0.5 + 0.1 * rng.randn(D)])
    for _ in range(20):
        eng.elbo_grad_meanfield(0, N, D, theta, _lib.FAMILY_MF_GAUSSIAN)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(300):
            v, g = eng.elbo_grad_meanfield(0, N, D, theta, _lib.FAMILY_MF_GAUSSIAN)
        ts.append((time.perf_counter() - t0) / 300)
    print('ELBO, noise in memory, D=%4d N=%6d: %.1f us per blocking call (min %.1f), value %.10g' % (D, N, 1e6 * sorted(ts)[2], 1e6 * min(ts), v))
