"""Round 6: blocking calls in the default mode (rng='numpy') with the look-ahead job's start deferred to the caller's first wait
when the last job had slack (default) against always at the round's end (VB_LEGACY_DEFER=0): the dense Gaussian family at the
headline shape, the two mean-field families at the C1 shape, the low-rank family."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import viabel_amd as vb

d, n = 1024, 4096
rng = np.random.RandomState(2)
A = rng.randn(d, d)
dense = vb.CorrelatedGaussianModel(rng.randn(d), covariance=A @ A.T / d + np.eye(d))
funnel = vb.FunnelModel(d)


def med(obj, theta, calls):
    for _ in range(8):
        obj(theta)
    bl = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(calls):
            obj(theta)
        bl.append(1e6 * (time.perf_counter() - t0) / calls)
    return statistics.median(bl)


for rep in range(2):
    for flag in ('1', '0'):
        os.environ['VB_LEGACY_DEFER'] = flag
        fr = vb.FullRankGaussian(d)
        mfg, mft = vb.MFGaussian(d), vb.MFStudentT(d, 7)
        lr = vb.LRGaussian(d, k=8)
        th_mf = np.concatenate([np.zeros(d), -np.ones(d)])
        print('defer %s: full rank %.1f us, MFGaussian %.1f us, MFStudentT %.1f us, LRGaussian(k=8) %.1f us' % (
            flag, med(vb.ExclusiveKL(fr, dense, n), fr.init_param(), 40), med(vb.ExclusiveKL(mfg, funnel, n), th_mf, 40),
            med(vb.ExclusiveKL(mft, funnel, n), th_mf, 20), med(vb.ExclusiveKL(lr, funnel, n), lr.init_param(), 40)), flush=True)
