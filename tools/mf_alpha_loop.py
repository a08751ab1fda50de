"""Dev tool: blocking AlphaDivergence and ExclusiveKL calls over MFGaussian at the C1 shape (funnel D = 1024, N = 4096),
rng='philox' -- the loop behind profiles/r05_mf_alpha_timeline.txt (tools/timeline.sh mfalpha 120 24 tools/mf_alpha_loop.py)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import viabel_amd as vb
D, N = 1024, 4096
fam = vb.MFGaussian(D, rng='philox')
m = vb.FunnelModel(D)
theta = fam.init_param(); theta[D:] = -1.0
for name, obj in (('alpha', vb.AlphaDivergence(fam, m, N, 0.5)), ('ekl', vb.ExclusiveKL(fam, m, N))):
    np.random.seed(1)
    for _ in range(40): obj(theta)
    t0 = time.perf_counter()
    for _ in range(200): obj(theta)
    print('%s: %.1f us per call' % (name, 5e3 * (time.perf_counter() - t0)))
