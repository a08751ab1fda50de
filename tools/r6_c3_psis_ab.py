"""Round 6: the C3 call with Pareto smoothing, fused weights-in / weights-out (default) against prep + apply launches
(VB_PSIS_FUSED_IO=0), alternating on one box."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import viabel_amd as vb

D, N, df = 256, 16384, 100
model, prior, theta = bench._c3_problem(vb, D)


def run():
    obj = vb.DISInclusiveKL(vb.MultivariateT(D, df, seed=1, rng='philox'), model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                            temper_prior_params=prior, use_resampling=False, psis_smooth=True)
    for _ in range(30):
        obj(theta)
    bl = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(60):
            obj(theta)
        bl.append(1e6 * (time.perf_counter() - t0) / 60)
    return statistics.median(bl), min(bl)


for rep in range(3):
    for flag in ('1', '0'):
        os.environ['VB_PSIS_FUSED_IO'] = flag
        print('fused io %s: %.1f us (min %.1f)' % ((flag,) + run()), flush=True)
