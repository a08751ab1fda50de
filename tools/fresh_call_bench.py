#!/usr/bin/env python3
"""Dev tool: blocking ExclusiveKL calls on resident noise vs on noise generated inside the streaming kernel."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viabel_amd as vb  # noqa: E402
from viabel_amd import _lib  # noqa: E402

D, N, reps = 1024, 4096, 400
eng = _lib.default_engine()
eng.set_model(vb.FunnelModel(D).device_spec())
theta = np.concatenate([np.zeros(D), -np.ones(D)])
eng.noise_generate(0, N, D, seed=1, stream=0)
for name, call in (('resident', lambda i: eng.elbo_grad_meanfield(0, N, D, theta, _lib.FAMILY_MF_GAUSSIAN)),
                   ('in-register', lambda i: eng.elbo_grad_meanfield_philox(0, N, D, theta, _lib.FAMILY_MF_GAUSSIAN, 1, i))):
    for i in range(50):
        call(i)
    t0 = time.perf_counter()
    for i in range(reps):
        call(i)
    print('%-12s %.1f us per blocking call' % (name, 1e6 * (time.perf_counter() - t0) / reps))
