"""Round 6: the blocking full-rank call, pipelined upload on / off, pinned gradient array on / off."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import viabel_amd as vb
from viabel_amd import _lib

eng = _lib.default_engine()
d, n = int(os.environ.get('D', 1024)), 4096
rng = np.random.RandomState(2)
A = rng.randn(d, d)
model = vb.CorrelatedGaussianModel(rng.randn(d), covariance=A @ A.T / d + np.eye(d))
fam = vb.FullRankGaussian(d, rng='philox')
obj = vb.ExclusiveKL(fam, model, n)
theta = fam.init_param()


def run(calls=200):
    for _ in range(30):
        obj(theta)
    best = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(calls):
            v, g = obj(theta)
        best.append(1e6 * (time.perf_counter() - t0) / calls)
    return sorted(best)[1]


real_pinned = _lib.pinned_array
theta_pageable = theta
theta_pinned = real_pinned(theta.size)
theta_pinned[:] = theta
for src in ('pageable', 'pinned'):
    theta = theta_pinned if src == 'pinned' else theta_pageable
    for pipe in ('0', '1'):
        for pinned in (False, True):
            os.environ['VB_FR_UPLOAD_PIPE'] = pipe
            _lib.pinned_array = real_pinned if pinned else (lambda k: np.empty(k, dtype=np.float64))
            for chunks in (('3',) if pipe == '0' else ('1', '2', '3', '4')):
                os.environ['VB_FR_UPLOAD_CHUNKS'] = chunks
                print('theta %-8s pipe %s chunks %s pinned grad %-5s: %.1f us per blocking call'
                      % (src, pipe, chunks, pinned, run()), flush=True)
