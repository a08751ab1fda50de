import sys, time
import numpy as np
sys.path.insert(0, '.')
import viabel_amd as vb
from viabel_amd import _lib
from viabel_amd._legacy_rng import LegacyRandomState
eng = _lib.default_engine()
D = 64
for n in (128, 192, 256, 384, 512, 768, 1024, 1536, 2048, 4096, 8192):
    rs = LegacyRandomState(1)
    ts = []
    for _ in range(15):
        t0 = time.perf_counter()
        ok = eng.noise_legacy_standard_t(7, rs._h, 7.0, n, D, 0, n)
        eng.sync()
        ts.append(1e6 * (time.perf_counter() - t0))
    ts = ts[2:]
    print('%7d values: standard_t device min %7.1f median %7.1f max %7.1f  ok=%s' % (n * D, min(ts), np.median(ts), max(ts), ok))
