"""Dev tool: MultivariateT + ExclusiveKL in parity mode (rng='numpy'): host eigh against the device iterations for the
symmetric root and the Sylvester solve (VIABEL_AMD_HOST_ROOT_MAX_DIM)."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb

rng = np.random.RandomState(1)
for D in (int(a) for a in sys.argv[1:]):
    model = vb.GaussianModel(0.1 * rng.randn(D), np.exp(0.1 * rng.randn(D)))
    fam = vb.MultivariateT(D, 40, seed=3)
    obj = vb.ExclusiveKL(fam, model, 1000)
    theta = fam.init_param()
    for _ in range(5):
        obj(theta)
    t0 = time.perf_counter()
    for _ in range(30):
        obj(theta)
    print('D=%4d: %.0f us per call' % (D, 1e6 * (time.perf_counter() - t0) / 30))
