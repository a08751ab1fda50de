"""Do the three GEMMs of the full-rank evaluation fill each other's tails when row slabs of one evaluation run as
independent chains on separate HIP streams?  (Every kernel of the evaluation is independent across sample rows up to
the split reduction.)  Probe: K engines (one stream each) evaluate slabs of n_k rows concurrently, against one engine
with all 4096 rows.  GPU box.  usage: fr_pipe_probe.py [rows,rows,...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viabel_amd as vb                                     # noqa: E402
from viabel_amd import _lib                                 # noqa: E402

D = 1024
STEPS = 200


def setup(eng, n_rows, ring=4, slot0=0):
    rng = np.random.RandomState(2)
    A = rng.randn(D, D)
    model = vb.CorrelatedGaussianModel(rng.randn(D), covariance=A @ A.T / D + np.eye(D))
    eng.set_model(model.device_spec())
    fr = vb.FullRankGaussian(D)
    L = np.exp(-1.0) * np.eye(D) + 0.01 * np.tril(np.random.RandomState(3).randn(D, D))
    eng.fullrank_set_theta(fr.pack(np.zeros(D), L), D)
    for s in range(ring):
        eng.noise_generate(slot0 + s, n_rows, D, seed=2, stream=s)
    return ring


def measure(slabs):
    engs = [_lib.Engine() for _ in slabs]
    for e, n in zip(engs, slabs):
        setup(e, n)

    def run(k):
        for i in range(k):
            for e, n in zip(engs, slabs):
                e.elbo_grad_fullrank_enqueue(i % 4, n, D, n_total=sum(slabs))

    def sync():
        for e in engs:
            e.sync()
    run(600)
    sync()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        run(STEPS)
        sync()
        best = min(best, (time.perf_counter() - t0) / STEPS)
    for e in engs:
        e.close() if hasattr(e, 'close') else None
    return best * 1e6


if __name__ == '__main__':
    cases = [[4096], [2048, 2048], [2432, 1664], [2816, 1280], [1408, 1408, 1280], [1792, 1280, 1024], [1024] * 4,
             [1536, 1152, 896, 512]]
    if len(sys.argv) > 1:
        cases = [[int(x) for x in a.split(',')] for a in sys.argv[1:]]
    for c in cases:
        print('slabs %-28s %8.1f us per %d rows' % (c, measure(c), sum(c)), flush=True)
