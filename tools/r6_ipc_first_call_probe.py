"""Round 6 probe: how long do the first blocking C3 calls of a two-rank IPC job on ONE GPU take (the first call allocates,
loads code objects, creates streams while the peer's collective kernels already spin)?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if 'WORLD_SIZE' in os.environ:
    import numpy as np
    import bench
    from viabel_amd import _lib, distributed
    import viabel_amd as vb
    eng = _lib.Engine(0)
    _lib.set_default_engine(eng)
    group = distributed.SocketGroup.from_env()
    distributed.attach(eng, group, transport=sys.argv[1])
    model, prior, theta = bench._c3_problem(vb, 256)
    N = 16384
    obj = vb.DISInclusiveKL(vb.MultivariateT(256, 100, seed=1, rng='philox'), model, N, ess_target=N // 8,
                            temper_prior=vb.MFGaussian(256), temper_prior_params=prior, use_resampling=False)
    if len(sys.argv) > 2 and sys.argv[2] == 'stagger' and group.rank == 1:
        time.sleep(1.0)
    for k in range(4):
        t0 = time.time()
        try:
            obj(theta)
            msg = 'ok'
        except Exception as e:
            msg = str(e)[-120:]
        sys.stderr.write('rank %d call %d: %.3f s %s\n' % (group.rank, k, time.time() - t0, msg))
    group.barrier()
    group.close()
else:
    import bench
    rc, lines = bench.spawn_ranks(2, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], timeout_s=600,
                                  extra_env={'VB_IPC_TIMEOUT_S': os.environ.get('VB_IPC_TIMEOUT_S', '120')})
    print('rc', rc)
