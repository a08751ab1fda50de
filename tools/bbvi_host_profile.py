#!/usr/bin/env python3
"""Dev tool: where the host's time goes in a bbvi() iteration at the quickstart's shape (2-D funnel, 10 samples):
cProfile over the same call bench.py's bbvi_quickstart leg times."""
import cProfile
import pstats
import sys

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb

D = 2
model = vb.FunnelModel(D, scale_index=1) if 'scale_index' in vb.FunnelModel.__init__.__code__.co_varnames else vb.FunnelModel(D)
vb.bbvi(D, n_iters=300, num_mc_samples=10, objective=vb.ExclusiveKL(vb.MFGaussian(D), model, 10), learning_rate=0.5)
pr = cProfile.Profile()
pr.enable()
res = vb.bbvi(D, n_iters=3000, num_mc_samples=10, objective=vb.ExclusiveKL(vb.MFGaussian(D), model, 10), learning_rate=0.5)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
