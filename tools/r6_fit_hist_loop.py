"""Round 6: RMSProp.optimize on the device at the headline shape, 80 iterations (for rocprofv3 timelines)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import viabel_amd as vb
from viabel_amd.optimization import RMSProp

d, n = 1024, 4096
rng = np.random.RandomState(2)
A = rng.randn(d, d)
model = vb.CorrelatedGaussianModel(rng.randn(d), covariance=A @ A.T / d + np.eye(d))
theta = vb.FullRankGaussian(d).init_param()
obj = vb.ExclusiveKL(vb.FullRankGaussian(d, rng='philox'), model, n)
RMSProp(0.001).optimize(80, obj, theta, on_device=True)
