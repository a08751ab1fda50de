"""Round 6: C3 one-rank call, A/B of the round-6 routes on ONE box (VB_MVT_CHAIN, VB_MVT_FUSED_ROWS)."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(calls):
            obj(theta)
        blocks.append((time.perf_counter() - t0) / calls)
    return 1e6 * statistics.median(blocks)


for rep in range(2):
    for chain, rows in (('1', '1'), ('0', '1'), ('1', '0'), ('0', '0')):
        os.environ['VB_MVT_CHAIN'], os.environ['VB_MVT_FUSED_ROWS'] = chain, rows
        print('chain %s fused_rows %s: weighted %.1f us, resampling %.1f us' % (chain, rows, run(False), run(True)), flush=True)
