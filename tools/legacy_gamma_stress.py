#!/usr/bin/env python3
"""Dev tool: random standard_t / chisquare / randn requests on one generator, interleaved, against numpy (values + state).
python tools/legacy_gamma_stress.py [cases] [max values]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from viabel_amd import _lib
from viabel_amd._legacy_rng import LegacyRandomState

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 3000000
eng = _lib.default_engine()
gen = np.random.RandomState(99)
bad = 0
ours, ref = LegacyRandomState(7), np.random.RandomState(7)
for case in range(cases):
    kind = gen.randint(3)
    df = float(gen.choice([2.01, 2.5, 4.0, 7.0, 19.5, 100.0, 5e4]))
    total = int(np.exp(gen.uniform(0, np.log(cap))))
    d = int(min(total, gen.randint(1, 2100)))
    n = max(1, total // d)
    if kind == 0:
        ok = eng.noise_legacy_standard_t(6, ours._h, df, n, d)
        want = ref.standard_t(df, (n, d))
        got = eng.noise_get_host(6, n, d) if ok else None
    elif kind == 1:
        got = eng.chisq_legacy(ours._h, df, n * d)
        want = ref.chisquare(df, n * d)
        ok = got is not None
    else:
        ok = eng.noise_legacy_randn(6, ours._h, n, d)
        want = ref.randn(n, d)
        got = eng.noise_get_host(6, n, d) if ok else None
    if not ok:
        # declined (small budgets can fall short by design): the host generator must then produce the same values
        got = ours.standard_t(df, (n, d)) if kind == 0 else (ours.chisquare(df, n * d) if kind == 1 else ours.randn(n, d))
    a, b = ours.get_state(), ref.get_state()
    good = np.array_equal(got, want) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
    if not good or not ok:
        print('case', case, ['t', 'chi', 'n'][kind], df, n, d, 'device' if ok else 'DECLINED', 'OK' if good else 'MISMATCH')
    bad += not good
print('cases', cases, 'failures', bad)
