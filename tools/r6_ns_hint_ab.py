"""Round 6: the reference-identical (rng='numpy') calls of the t family at the C3 shape, Newton-Schulz steps launched from the
previous root's count (default) against the step-by-step control (VB_NS_HINT=0), alternating on one box."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import viabel_amd as vb

D, N, df = 256, 16384, 100
model, prior, theta = bench._c3_problem(vb, D)


def med(obj, calls=30):
    for _ in range(8):
        obj(theta)
    bl = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(calls):
            obj(theta)
        bl.append(1e6 * (time.perf_counter() - t0) / calls)
    return statistics.median(bl)


for rep in range(3):
    for flag in ('1', '0'):
        os.environ['VB_NS_HINT'] = flag
        c3 = vb.DISInclusiveKL(vb.MultivariateT(D, df, seed=1), model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                               temper_prior_params=prior, use_resampling=False)
        ekl = vb.ExclusiveKL(vb.MultivariateT(D, df, seed=1), model, N)
        alpha = vb.AlphaDivergence(vb.MultivariateT(D, df, seed=1), model, N, 0.5)
        print('hint %s: C3 %.1f us, ExclusiveKL %.1f us, AlphaDivergence %.1f us' % (flag, med(c3), med(ekl), med(alpha)), flush=True)
