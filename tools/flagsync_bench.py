#!/usr/bin/env python3
"""Dev tool: the blocking mean-field call waiting on the finalize kernel's completion words (default) against
hipStreamSynchronize (VB_MF_FLAGSYNC=0; read once per process: run twice)."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, '.')
import viabel_amd as vb
from viabel_amd import _lib

eng = _lib.default_engine()
def med_us(call, reps=500, blocks=5):
    for _ in range(100):
        call()
    ts = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        ts.append((time.perf_counter() - t0) / reps)
    return 1e6 * sorted(ts)[len(ts) // 2]
for d, n in ((1024, 4096), (10, 100), (2, 10)):
    eng.set_model(vb.FunnelModel(d).device_spec())
    theta = np.concatenate([np.zeros(d), -np.ones(d)])
    eng.noise_generate(0, n, d, seed=1, stream=0)
    a = med_us(lambda: eng.elbo_grad_meanfield(0, n, d, theta, _lib.FAMILY_MF_GAUSSIAN))
    b = med_us(lambda: eng.elbo_grad_meanfield_philox(0, n, d, theta, _lib.FAMILY_MF_GAUSSIAN, 1, 5))
    print('VB_MF_FLAGSYNC=%s D=%d N=%d: blocking call %.1f us, fresh-noise call %.1f us' % (os.environ.get('VB_MF_FLAGSYNC', '1'), d, n, a, b))
