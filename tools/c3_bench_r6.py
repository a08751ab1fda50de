"""Round 6: the C3 blocking call in a plain loop (for rocprofv3 timelines / kernel stats).
usage: c3_bench_r6.py [philox|numpy] [psis] [resample]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import viabel_amd as vb

D, N, df = 256, 16384, 100
model, prior, theta = bench._c3_problem(vb, D)
mode = sys.argv[1] if len(sys.argv) > 1 else 'philox'
obj = vb.DISInclusiveKL(vb.MultivariateT(D, df, seed=1, rng=mode), model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                        temper_prior_params=prior, use_resampling='resample' in sys.argv[2:], psis_smooth='psis' in sys.argv[2:])
for _ in range(40):
    obj(theta)
