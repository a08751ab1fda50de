"""Dev tool: distribution check of the device normal generator on 16 x 4096 x 1024 = 67 M draws: moments against their
sampling errors, tail counts against the normal law, a Kolmogorov-Smirnov distance on a subsample, correlation of the
two rows / two columns that share a Philox call.  GPU box."""
import sys

import numpy as np
from scipy import special, stats

sys.path.insert(0, '.')
from viabel_amd import _lib

eng = _lib.default_engine()
N, D, S = 4096, 1024, 16
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 12345
acc = np.zeros(7)
tails = {3: 0, 4: 0, 5: 0, 6: 0}
mx = 0.0
corr_rows = corr_cols = 0.0
sub = []
for s in range(S):
    eng.noise_generate(5, N, D, SEED, s)
    z = eng.noise_get_host(5, N, D)
    for k in range(1, 7):
        acc[k] += np.sum(z ** k)
    for t in tails:
        tails[t] += int(np.sum(np.abs(z) > t))
    mx = max(mx, float(np.abs(z).max()))
    zz = z.reshape(N // 8, 8, D)
    corr_rows += float(np.sum(zz[:, :4, :] * zz[:, 4:, :]))          # rows g and g + 4 share a call
    corr_cols += float(np.sum(z[:, 0::2] * z[:, 1::2]))              # the two columns of a pair
    sub.append(z[::64, ::16].ravel())
n = S * N * D
m = acc / n
print('n = %d' % n)
for k, expect, var in ((1, 0, 1), (2, 1, 2), (3, 0, 15), (4, 3, 96), (5, 0, 945), (6, 15, 10170)):
    print('  E z^%d = %+.6f (expected %g, %.1f sigma)' % (k, m[k], expect, (m[k] - expect) / np.sqrt(var / n)))
for t, c in tails.items():
    p = special.erfc(t / np.sqrt(2.0))
    print('  |z| > %d: %d (expected %.1f, %.1f sigma)' % (t, c, n * p, (c - n * p) / np.sqrt(n * p)))
print('  max |z| = %.3f' % mx)
print('  sum z_g z_{g+4} / sqrt(n/2) = %.2f sigma, sum z_2j z_2j+1 / sqrt(n/2) = %.2f sigma'
      % (corr_rows / np.sqrt(n / 2), corr_cols / np.sqrt(n / 2)))
ks = stats.kstest(np.concatenate(sub), 'norm')
print('  KS on %d draws: D = %.2e, p = %.3f' % (np.concatenate(sub).size, ks.statistic, ks.pvalue))
