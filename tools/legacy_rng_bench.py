"""Host time of the parity-mode noise: `numpy.random.RandomState.randn(4096, 1024)` against the library's restatement of
the same stream (vb_legacy_rng.cpp) at several thread counts; values and state compared.  Runs anywhere (no GPU)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viabel_amd._legacy_rng import LegacyRandomState      # noqa: E402

N, D = 4096, 1024


def best(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return 1e3 * min(ts)


if __name__ == '__main__':
    print('host cpus: %d' % os.cpu_count())
    b = np.random.RandomState(7)
    print('numpy RandomState.randn(%d, %d): %.1f ms' % (N, D, best(lambda: b.randn(N, D))))
    print('numpy RandomState.standard_t(7, (%d, %d)): %.1f ms' % (N, D, best(lambda: b.standard_t(7.0, size=(N, D)), 2)))
    for th in (1, 2, 4, 8, 16, 32):
        if th > 2 * os.cpu_count():
            break
        os.environ['VIABEL_AMD_RNG_THREADS'] = str(th)
        a = LegacyRandomState(7)
        ms = best(lambda: a.randn(N, D))
        ref = np.random.RandomState(7)
        a = LegacyRandomState(7)
        same = np.array_equal(a.randn(N, D), ref.randn(N, D)) and np.array_equal(a.get_state()[1], ref.get_state()[1])
        print('library randn, %2d threads: %.1f ms   bit-identical to numpy: %s' % (th, ms, same))
    a = LegacyRandomState(7)
    print('library standard_t(7, (%d, %d)) (sequential): %.1f ms' % (N, D, best(lambda: a.standard_t(7.0, size=(N, D)), 2)))
