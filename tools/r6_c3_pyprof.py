"""Round 6: cProfile of the C3 blocking call's Python side (weighted form, philox): which host functions the ~30 us between two
calls' device work go to."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import viabel_amd as vb

D, N, df = 256, 16384, 100
model, prior, theta = bench._c3_problem(vb, D)
obj = vb.DISInclusiveKL(vb.MultivariateT(D, df, seed=1, rng='philox'), model, N, ess_target=N // 8,
                        temper_prior=vb.MFGaussian(D), temper_prior_params=prior, use_resampling=False)
for _ in range(30):
    obj(theta)
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    obj(theta)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
