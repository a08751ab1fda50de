import sys, time
import numpy as np
sys.path.insert(0, '.')
import viabel_amd as vb
from viabel_amd import _lib
D, N = 1024, 4096
eng = _lib.default_engine()
eng.set_model(vb.FunnelModel(D).device_spec())
for s in range(8):
    eng.noise_generate(s, N, D, seed=1, stream=s)
theta = np.concatenate([np.zeros(D), -np.ones(D)])
for i in range(200):
    eng.elbo_grad_meanfield(i % 8, N, D, theta, _lib.FAMILY_MF_GAUSSIAN)
K = 2000
t0 = time.perf_counter()
for i in range(K):
    eng.elbo_grad_meanfield(i % 8, N, D, theta, _lib.FAMILY_MF_GAUSSIAN)
t_block = (time.perf_counter() - t0) / K
te = ts = 0.0
for i in range(K):
    a = time.perf_counter()
    eng.elbo_grad_meanfield_async(i % 8, N, D, theta, _lib.FAMILY_MF_GAUSSIAN, rslot=0)
    b = time.perf_counter()
    eng.result_get(0, 2 * D)
    c = time.perf_counter()
    te += b - a
    ts += c - b
print('blocking call %.1f us; async enqueue %.1f us + result_get (event wait + copy) %.1f us' % (t_block * 1e6, te / K * 1e6, ts / K * 1e6))
