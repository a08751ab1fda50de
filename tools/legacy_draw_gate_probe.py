#!/usr/bin/env python3
"""Dev tool: where the gates between the host draw + upload and the device draw of numpy's streams should sit
(approximations._DEVICE_DRAW_FROM / _DEVICE_CHI_FROM): both routes timed for randn, standard_t and chisquare over sizes."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb
from viabel_amd import _lib
from viabel_amd._legacy_rng import LegacyRandomState

eng = _lib.default_engine()
D = 64


def med(f, reps=15):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts[2:]))


for n in (16, 64, 128, 256, 512, 1024, 2048, 4096):
    vals = n * D
    rs = LegacyRandomState(1)

    def host_randn():
        eng.noise_set_host(7, rs.randn(n, D))
        eng.sync()

    def dev_randn():
        if eng.noise_legacy_randn(7, rs._h, n, D, 0, n) is False:
            raise SystemExit('declined')
        eng.sync()

    def host_t():
        eng.noise_set_host(7, rs.standard_t(7.0, (n, D)))
        eng.sync()

    def dev_t():
        if eng.noise_legacy_standard_t(7, rs._h, 7.0, n, D, 0, n) is False:
            raise SystemExit('declined')
        eng.sync()

    def host_chi():
        return rs.chisquare(9.0, vals)

    def dev_chi():
        return eng.chisq_legacy(rs._h, 9.0, vals, to_host=False)
    print('%7d values: randn host %7.1f us device %7.1f | standard_t host %7.1f device %7.1f | chisquare host %7.1f device %7.1f'
          % (vals, med(host_randn), med(dev_randn), med(host_t), med(dev_t), med(host_chi), med(dev_chi)))
