#!/usr/bin/env python3
"""Dev tool: where the gate between the host-root route and the resident device route of the t family's reference-identical
mode (objectives._RESIDENT_GATE) should sit: both routes timed at several D (N = 16 384, DIS weighted and ExclusiveKL)."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import viabel_amd as vb
from viabel_amd import objectives as vobj
from oracle import families as ofam

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for D in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (32, 64, 100, 128, 160, 200)):
    rng = np.random.RandomState(D)
    mean, sd = 0.3 * rng.randn(D), np.exp(0.2 * rng.randn(D))
    model = vb.GaussianModel(mean, sd)
    B = rng.randn(D, D)
    theta = np.concatenate([0.2 * rng.randn(D), ofam.psd_to_free(np.e * np.eye(D) + 0.04 * (B @ B.T / D - np.eye(D)))])
    prior = np.concatenate([np.zeros(D), 0.5 * np.ones(D)])
    row = []
    for gate in (10 ** 6, 0):
        keep = vobj._RESIDENT_GATE
        vobj._RESIDENT_GATE = gate      # 10**6: the host-root route (its own root policy: LAPACK up to D = 160, device iteration above)
        try:
            for make in (lambda: vb.DISInclusiveKL(vb.MultivariateT(D, 100, seed=1), model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                                                   temper_prior_params=prior, use_resampling=False),
                         lambda: vb.ExclusiveKL(vb.MultivariateT(D, 100, seed=1), model, N)):
                obj = make()
                for _ in range(3):
                    obj(theta)
                t0 = time.perf_counter()
                for _ in range(8):
                    obj(theta)
                row.append(1e3 * (time.perf_counter() - t0) / 8)
        finally:
            vobj._RESIDENT_GATE = keep
    print('D = %3d  N = %d: DIS host-root %.2f ms, resident %.2f ms;  ExclusiveKL host-root %.2f ms, resident %.2f ms'
          % (D, N, row[0], row[2], row[1], row[3]))
