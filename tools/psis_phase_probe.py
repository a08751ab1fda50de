"""Dev tool: phase times inside psis_kernel (library built with -DVB_PSIS_CLOCK: tools/build_variant.sh psisclk
"-DVB_PSIS_CLOCK", run with VIABEL_AMD_LIB=tools/libviabel_hip_psisclk.so)."""
import sys

import numpy as np

sys.path.insert(0, '.')
from viabel_amd import _lib

eng = _lib.default_engine()
rng = np.random.RandomState(1)
for name, lw in (('student-t tails', 2.0 * rng.standard_t(3.0, 16384)), ('clustered (C3-like)', -50.0 + 0.3 * rng.randn(16384)),
                 ('n = 100000', 1.5 * rng.standard_t(4.0, 100000))):
    print(name)
    for _ in range(3):
        eng.psis_smooth(lw.size, lw)
