#!/usr/bin/env python3
"""Dev tool: RandomState.standard_t(df, (N, D)) into a noise slot and chisquare(df, N) on the device against the host loop."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from viabel_amd import _lib
from viabel_amd._legacy_rng import LegacyRandomState

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
df = float(sys.argv[4]) if len(sys.argv) > 4 else 7.0
eng = _lib.default_engine()
rs = LegacyRandomState(1)
for _ in range(2):
    eng.noise_legacy_standard_t(3, rs._h, df, N, D)
eng.sync()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    eng.noise_legacy_standard_t(3, rs._h, df, N, D)
    ts.append(time.perf_counter() - t0)
ts.sort()
print('device standard_t(%g, (%d, %d)): median %.1f us, min %.1f, max %.1f' % (df, N, D, 1e6 * ts[len(ts) // 2], 1e6 * ts[0], 1e6 * ts[-1]))
n_chi = 16384
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    eng.chisq_legacy(rs._h, 100.0, n_chi)
    ts.append(time.perf_counter() - t0)
ts.sort()
print('device chisquare(100, %d): median %.1f us' % (n_chi, 1e6 * ts[len(ts) // 2]))
t0 = time.perf_counter()
rs.standard_t(df, (N // 4, D))
print('host loop standard_t: %.1f us (scaled to the full draw)' % ((time.perf_counter() - t0) * 4e6))
t0 = time.perf_counter()
rs.chisquare(100.0, n_chi)
print('host loop chisquare(100, %d): %.1f us' % (n_chi, (time.perf_counter() - t0) * 1e6))
