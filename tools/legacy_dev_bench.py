#!/usr/bin/env python3
"""Dev tool: RandomState.randn(N, D) into a noise slot -- host draw + upload against the device stream."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from viabel_amd import _lib
from viabel_amd._legacy_rng import LegacyRandomState

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
eng = _lib.default_engine()
rs = LegacyRandomState(1)
for _ in range(2):
    eng.noise_legacy_randn(3, rs._h, N, D)
eng.sync()
t0 = time.perf_counter()
for _ in range(reps):
    eng.noise_legacy_randn(3, rs._h, N, D)
eng.sync()
print('device stream: %.1f us per randn(%d, %d)' % ((time.perf_counter() - t0) / reps * 1e6, N, D))
t0 = time.perf_counter()
for _ in range(max(2, reps // 4)):
    eng.noise_set_host(3, rs.randn(N, D))
eng.sync()
print('host draw + upload: %.1f us' % ((time.perf_counter() - t0) / max(2, reps // 4) * 1e6))
t0 = time.perf_counter()
for _ in range(max(2, reps // 4)):
    np.random.RandomState(1).randn(N, D)
print('numpy randn alone: %.1f us' % ((time.perf_counter() - t0) / max(2, reps // 4) * 1e6))
