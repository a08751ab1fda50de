"""Dev tool: what would a pinned result buffer save?  fullrank_get into a pageable numpy array against the same call
into hipHostMalloc'ed memory (and theta uploaded from pinned memory), headline shape."""
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import viabel_amd as vb
from viabel_amd import _lib

D, N, reps = 1024, 4096, 200
rng = np.random.RandomState(2)
A = rng.randn(D, D)
model = vb.CorrelatedGaussianModel(rng.randn(D), covariance=A @ A.T / D + np.eye(D))
eng = _lib.default_engine()
fr = vb.FullRankGaussian(D, seed=1, rng='philox')
theta = fr.pack(np.zeros(D), np.exp(-1.0) * np.eye(D))
p = theta.size
eng.set_model(model.device_spec())
eng.noise_generate(0, N, D, seed=1, stream=0)
eng.fullrank_set_theta(theta, D)
eng.elbo_grad_fullrank_enqueue(0, N, D)
eng.sync()
hip = ctypes.CDLL('libamdhip64.so')
ptr = ctypes.c_void_p()
assert hip.hipHostMalloc(ctypes.byref(ptr), ctypes.c_size_t(8 * (p + 16)), 0) == 0
pinned = np.frombuffer((ctypes.c_double * (p + 16)).from_address(ptr.value), dtype=np.float64)
lib = eng._lib
value = ctypes.c_double(0.0)
pageable = np.empty(p)
for name, arr in (('pageable grad', pageable), ('pinned grad', pinned[:p])):
    dp = arr.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    for _ in range(5):
        lib.vb_fullrank_get(eng._ctx, ctypes.byref(value), dp, p)
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.vb_fullrank_get(eng._ctx, ctypes.byref(value), dp, p)
    print('%-16s fullrank_get: %.1f us' % (name, 1e6 * (time.perf_counter() - t0) / reps))
ptr2 = ctypes.c_void_p()
assert hip.hipHostMalloc(ctypes.byref(ptr2), ctypes.c_size_t(8 * p), 0) == 0
pth = np.frombuffer((ctypes.c_double * p).from_address(ptr2.value), dtype=np.float64)
pth[:] = theta
for name, arr in (('pageable theta', theta), ('pinned theta', pth)):
    dp = arr.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    for _ in range(5):
        lib.vb_fullrank_set_theta(eng._ctx, dp, D)
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.vb_fullrank_set_theta(eng._ctx, dp, D)
    print('%-16s set_theta: %.1f us' % (name, 1e6 * (time.perf_counter() - t0) / reps))
