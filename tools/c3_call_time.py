"""Round 6: per-call time of the C3 blocking call (weighted and resampling), medians of 5 blocks of 60 calls."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import viabel_amd as vb

D, N, df = 256, 16384, 100
model, prior, theta = bench._c3_problem(vb, D)


def run(resample, calls=60):
    obj = vb.DISInclusiveKL(vb.MultivariateT(D, df, seed=1, rng='philox'), model, N, ess_target=N // 8,
                            temper_prior=vb.MFGaussian(D), temper_prior_params=prior, use_resampling=resample)
    for _ in range(20):
        obj(theta)
    blocks = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(calls):
            obj(theta)
        blocks.append((time.perf_counter() - t0) / calls)
    return 1e6 * statistics.median(blocks), 1e6 * min(blocks)


for rep in range(2):
    print('weighted %.1f us (min %.1f), resampling %.1f us (min %.1f)' % (run(False) + run(True)), flush=True)
