"""Dev tool: how many speculative rounds the tempering walk needs (VB_DIS_TRACE parsed): for random problems with
max_bisection_its = 50 and an interior root, the level at which each round starts and whether the last (spare) round still
had candidates to evaluate."""
import os
import re
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == '--child':
    import numpy as np
    sys.path.insert(0, '.')
    import viabel_amd as vb
    rng = np.random.RandomState(int(sys.argv[2]))
    for c in range(int(sys.argv[3])):
        D = int(rng.choice([2, 5, 16, 40, 64, 256]))
        N = int(rng.choice([333, 1000, 4096, 16384]))
        target = float(rng.uniform(0.02, 0.6) * N)
        mean = rng.uniform(-1.0, 1.0) + 0.5 * rng.randn(D)
        sd = np.exp(rng.uniform(-1.0, 1.0) + 0.3 * rng.randn(D))
        prior = np.concatenate([rng.uniform(-0.5, 0.5) + 0.3 * rng.randn(D), rng.uniform(-0.7, 0.7) + 0.2 * rng.randn(D)])
        theta = np.concatenate([0.3 * rng.randn(D), rng.uniform(-0.7, 0.3) + 0.2 * rng.randn(D)])
        obj = vb.DISInclusiveKL(vb.MFGaussian(D, seed=int(rng.randint(1, 1 << 30)), rng='philox'), vb.GaussianModel(mean, sd), N,
                                ess_target=target, temper_prior=vb.MFGaussian(D), temper_prior_params=prior, use_resampling=False)
        obj._eps = float(rng.choice([1.0, rng.uniform(0.05, 1.0)]))
        sys.stderr.write('[case %d]\n' % c)
        try:
            obj(theta)
            sys.stderr.write('[eps %.6g of %.6g]\n' % (obj._eps, 1.0))
        except ValueError:
            sys.stderr.write('[eps error]\n')
    sys.exit(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
env = dict(os.environ, VB_DIS_TRACE='1')
p = subprocess.run([sys.executable, __file__, '--child', '0', str(cases)], env=env, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
blocks = p.stderr.split('[case ')[1:]
hist = {}
spare_busy = interior = 0
for b in blocks:
    m = re.search(r'\[eps ([^ ]+) of', b)
    if not m or m.group(1) in ('error', '0', '1'):
        continue
    interior += 1
    rounds = re.findall(r'round (\d+): level (\d+) mode (\d+) candidates (\d+)', b)
    busy = sum(1 for r in rounds if int(r[3]) > 0)
    hist[busy] = hist.get(busy, 0) + 1
    if rounds and int(rounds[-1][3]) > 0:
        spare_busy += 1
        if spare_busy <= 2 and '--show' in sys.argv:
            print(b[:1200])
print('%d cases with an interior root; evaluating rounds needed: %s; the last launched round still had candidates in %d'
      % (interior, dict(sorted(hist.items())), spare_busy))
