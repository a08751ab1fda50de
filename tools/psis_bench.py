"""Dev tool: PSIS smoothing of device-resident log weights, multi-workgroup kernel against the single-workgroup one
(VB_PSIS_GRID=0), blocking calls through psislw (upload + kernel + download)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from viabel_amd import _lib
from viabel_amd._psis import psislw

_lib.default_engine()
rng = np.random.RandomState(1)
for n in (16384, 100000):
    lw = 2.0 * rng.standard_t(3.0, n)
    for grid in ('1', '0'):
        os.environ['VB_PSIS_GRID'] = grid
        for _ in range(5):
            psislw(lw)
        t0 = time.perf_counter()
        for _ in range(50):
            sm, k = psislw(lw)
        print('n = %6d grid = %s: %.1f us per call, khat %.6f' % (n, grid, 1e6 * (time.perf_counter() - t0) / 50, k))
