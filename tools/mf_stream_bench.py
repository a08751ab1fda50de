#!/usr/bin/env python3
"""Dev tool: the mean-field streaming kernel by itself at BASELINE configs[1] (D = 1024 funnel, N = 4096), 32 evaluations
per launch over a ring of 32 resident noise matrices -- the launch shape of bench.py's c1_meanfield leg; run under
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for the HBM-side bytes of mf_accum_kernel."""
import sys
import numpy as np
sys.path.insert(0, '.')
import viabel_amd as vb
from viabel_amd import _lib
eng = _lib.default_engine()
D, N, batch = 1024, 4096, 32
eng.set_model(vb.FunnelModel(D).device_spec())
theta = np.concatenate([np.zeros(D), -np.ones(D)])
for s in range(batch):
    eng.noise_generate(s, N, D, seed=1, stream=s)
thetas = np.tile(theta, (batch, 1))
for it in range(40):
    eng.elbo_grad_meanfield_batch_async(list(range(batch)), N, D, thetas, _lib.FAMILY_MF_GAUSSIAN, list(range(batch)))
eng.sync()
print('mean-field streaming kernel: 40 launches of 32 evaluations (D=1024 funnel, N=4096), 1 074.8 MB algorithmic per launch')
