#!/usr/bin/env python3
"""Dev tool: Philox noise generation rate.  usage: tools/rng_bench.py [N D reps kind df]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viabel_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
kind = int(sys.argv[4]) if len(sys.argv) > 4 else 0
df = float(sys.argv[5]) if len(sys.argv) > 5 else 8.0
eng = _lib.default_engine()
for i in range(200):
    eng.noise_generate(i % 4, N, D, seed=1, stream=i, kind=kind, df=df)
eng.sync()
t0 = time.perf_counter()
for i in range(reps):
    eng.noise_generate(i % 4, N, D, seed=1, stream=i, kind=kind, df=df)
eng.sync()
dt = (time.perf_counter() - t0) / reps
x = eng.noise_get_host(0, N, D)
print('%d x %d kind %d: %.2f us per matrix, %.1f G values/s; mean %.5f var %.5f kurt %.4f  checksum %.17g' % (
    N, D, kind, 1e6 * dt, N * D / dt / 1e9, x.mean(), x.var(), (x ** 4).mean() / x.var() ** 2, float(x.sum())))
