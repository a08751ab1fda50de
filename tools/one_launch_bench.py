#!/usr/bin/env python3
"""Dev tool: the C1 / C0 mean-field evaluation as one launch (default) against the prep -> accumulate -> finalize chain
(VB_MF_ONE=0): blocking call with the noise in memory, with the noise generated in registers, and the device fit loop."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb
from viabel_amd import _lib
from viabel_amd.optimization import RMSProp

eng = _lib.default_engine()


def med_us(call, reps=400, blocks=5):
    for _ in range(50):
        call()
    ts = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        ts.append((time.perf_counter() - t0) / reps)
    return 1e6 * sorted(ts)[len(ts) // 2]


for d, n in ((1024, 4096), (10, 100)):
    model = vb.FunnelModel(d)
    eng.set_model(model.device_spec())
    theta = np.concatenate([np.zeros(d), -np.ones(d)])
    eng.noise_generate(0, n, d, seed=1, stream=0)
    for one in ('1', '0'):
        os.environ['VB_MF_ONE'] = one
        a = med_us(lambda: eng.elbo_grad_meanfield(0, n, d, theta, _lib.FAMILY_MF_GAUSSIAN))
        b = med_us(lambda: eng.elbo_grad_meanfield_philox(0, n, d, theta, _lib.FAMILY_MF_GAUSSIAN, 1, 5))
        obj = vb.ExclusiveKL(vb.MFGaussian(d, rng='philox'), model, n)
        opt = RMSProp(0.01)
        obj.device_fit(200, theta, opt._device_kind, opt._device_hyper())
        t0 = time.perf_counter()
        obj.device_fit(2000, theta, opt._device_kind, opt._device_hyper())
        c = 1e6 * (time.perf_counter() - t0) / 2000
        print('D=%d N=%d VB_MF_ONE=%s: blocking call %.1f us, fresh-noise call %.1f us, device-loop iteration %.2f us' % (d, n, one, a, b, c))
