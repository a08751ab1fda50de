"""numpy's legacy chisquare / standard_t on the device against numpy itself (values and state), with timings.
python tools/legacy_gamma_check.py [quick]"""
import sys
import time
import numpy as np
import viabel_amd as vb
from viabel_amd import _lib
from viabel_amd._legacy_rng import LegacyRandomState

eng = _lib.default_engine()
print('log proven:', _lib.load().vb_legacy_rng_log_proven())
quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'


def same_state(ours, ref):
    a, b = ours.get_state(), ref.get_state()
    return np.array_equal(a[1], b[1]) and a[2:] == b[2:]


bad = 0
for seed in (1, 851):
    for df in (2.5, 7.0, 100.0):
        for (n, d) in [(1, 1), (3, 5), (100, 7), (1001, 77), (4096, 1024)] if not quick else [(3, 5), (1001, 77)]:
            ours, ref = LegacyRandomState(seed), np.random.RandomState(seed)
            ok = eng.noise_legacy_standard_t(7, ours._h, df, n, d)
            want = ref.standard_t(df, (n, d))
            if not ok:
                print('standard_t', seed, df, n, d, 'UNSUPPORTED')
                bad += 1
                continue
            got = eng.noise_get_host(7, n, d)
            eq, st = np.array_equal(got, want), same_state(ours, ref)
            nxt = np.array_equal(ours.randn(5), ref.randn(5))
            print('standard_t', seed, df, n, d, 'values', eq, 'state', st, 'next', nxt,
                  '' if eq else 'mismatches %d first %s' % (np.sum(got != want), np.argwhere(got != want)[:3].tolist()))
            bad += not (eq and st and nxt)
        for n in [1, 2, 17, 1000, 16384, 262144] if not quick else [17, 16384]:
            for pre in (0, 3):        # pre: an odd host draw first leaves a cached normal
                ours, ref = LegacyRandomState(seed), np.random.RandomState(seed)
                if pre:
                    ours.randn(pre), ref.randn(pre)
                got = eng.chisq_legacy(ours._h, df, n)
                want = ref.chisquare(df, n)
                if got is None:
                    print('chisquare', seed, df, n, pre, 'UNSUPPORTED')
                    bad += 1
                    continue
                eq, st = np.array_equal(got, want), same_state(ours, ref)
                nxt = np.array_equal(ours.randn(4), ref.randn(4))
                print('chisquare', seed, df, n, pre, 'values', eq, 'state', st, 'next', nxt)
                bad += not (eq and st and nxt)
print('FAILURES:', bad)

for (n, d, df) in [(4096, 1024, 7.0), (16384, 1, 40.0)]:
    ours = LegacyRandomState(3)
    ts = []
    for rep in range(6):
        t0 = time.perf_counter()
        if d > 1:
            eng.noise_legacy_standard_t(7, ours._h, df, n, d)
        else:
            eng.chisq_legacy(ours._h, df, n)
        ts.append(time.perf_counter() - t0)
    print('timing', n, d, df, ['%.3f ms' % (1e3 * t) for t in ts])
ours = LegacyRandomState(3)
ts = []
for rep in range(6):
    t0 = time.perf_counter()
    eng.noise_legacy_randn(7, ours._h, 4096, 1024)
    ts.append(time.perf_counter() - t0)
print('timing randn 4096x1024', ['%.3f ms' % (1e3 * t) for t in ts])
