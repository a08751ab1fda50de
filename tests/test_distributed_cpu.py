"""Multi-process (gloo, world_size 2, CPU) checks of the Monte-Carlo-axis sharding logic.

The device path all-reduces a vector of partial sums over RCCL and applies an O(P) epilogue
(vb_meanfield.hip / vb_fullrank.hip).  Here two CPU processes each take their `shard_rows` block,
form the same partial sums with the oracle's model derivatives (test infrastructure), all-reduce
them with gloo and apply the epilogue; the result must equal the unsharded oracle.  Also covers the
unique-id hand-off used to bootstrap RCCL.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from viabel_amd import distributed
        from viabel_amd.objectives import shard_rows
        from oracle import families as ofam, models as omod, objectives as oobj

        # 1. RCCL bootstrap hand-off: rank 0's id reaches everybody
        uid = distributed.broadcast_unique_id(rank, lambda: b'id-from-rank-0' + bytes(114))
        assert uid == b'id-from-rank-0' + bytes(114)

        # 2. sharded mean-field ELBO gradient == unsharded
        D, N = 12, 101                       # N not divisible by the world size
        rng = np.random.RandomState(0)
        theta = np.concatenate([0.3 * rng.randn(D), -0.5 + 0.2 * rng.randn(D)])
        noise = np.random.RandomState(1).randn(N, D)          # every rank draws the same stream
        fam, model = ofam.MFGaussian(D), omod.Funnel(D, 4)
        b, e = shard_rows(N, world, rank)
        mu, ls = theta[:D], theta[D:]
        sig = np.exp(ls)
        z = mu + sig * noise[b:e]
        g = model.grad(z)
        partial = np.concatenate([[model.logp(z).sum()], g.sum(0), (g * noise[b:e]).sum(0)])
        t = torch.from_numpy(partial.copy())
        dist.all_reduce(t)                                   # the exchange the device does over RCCL
        s = t.numpy()
        value = -(s[0] / N + fam.entropy(theta))
        grad = -np.concatenate([s[1:1 + D] / N, s[1 + D:] * sig / N + 1.0])
        ov, og = oobj.exclusive_kl(fam, model, theta, noise)
        assert abs(value - ov) < 1e-12 * abs(ov)
        np.testing.assert_allclose(grad, og, rtol=0, atol=1e-12 * np.max(np.abs(og)))

        # 3. combine_partial_sums is the all-reduce's reference semantics
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(partial.copy()))
        np.testing.assert_allclose(distributed.combine_partial_sums([x.numpy() for x in gathered]), s,
                                   rtol=1e-15)

        # 4. AlphaDivergence exchange (vb_rowstats.hip alpha_enqueue): all-reduce(max) of the log
        #    weights, then all-reduce(sum) of [sum s, sum s g, sum s (g e sigma + 1)]
        N2 = 64
        noise2 = np.random.RandomState(2).randn(N2, D)
        alpha = 2.0
        b, e = shard_rows(N2, world, rank)
        z = mu + sig * noise2[b:e]
        lw = model.logp(z) - fam.log_density(theta, z)
        mx = torch.tensor([lw.max()])
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sv = np.exp(lw - mx.item()) ** alpha
        g = model.grad(z)
        part = np.concatenate([[sv.sum()], (sv[:, None] * g).sum(0),
                               (sv[:, None] * (g * noise2[b:e] * sig + 1.0)).sum(0)])
        t = torch.from_numpy(part)
        dist.all_reduce(t)
        s = t.numpy()
        value = np.log(s[0] / N2) / alpha + mx.item()
        grad = alpha * s[1:] / N2
        ov, og = oobj.alpha_divergence(fam, model, theta, noise2, alpha)
        assert abs(value - ov) < 1e-12 * abs(ov)
        np.testing.assert_allclose(grad, og, rtol=0, atol=1e-12 * np.max(np.abs(og)))

        # 5. DISInclusiveKL exchange (dis_refresh_enqueue): all-gather of the per-sample log p /
        #    log q / log prior, every rank runs the same bisection over all N, then the weighted
        #    score sums of the local block are all-reduced
        tfam = ofam.MFGaussian(D)
        prior = np.concatenate([np.zeros(D), 0.5 * np.ones(D)])
        ref = oobj.DISInclusiveKL(fam, model, N2, 20, tfam, prior, use_resampling=False)
        ov, og = ref(theta, noise2)
        mine = oobj.DISInclusiveKL(fam, model, N2, 20, tfam, prior, use_resampling=False)
        vecs = np.stack([model.logp(z), fam.log_density(theta, z), tfam.log_density(prior, z)])
        gathered = [torch.zeros(3, e - b, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(vecs))
        allv = np.concatenate([x.numpy() for x in gathered], axis=1)
        _, _, w = mine._eps_and_weights(mine._eps, allv[2], allv[0], allv[1])
        wc = mine._clip(w)
        t = torch.from_numpy(np.concatenate([
            fam.log_density_grad_weighted(theta, z, wc[b:e]), [np.dot(wc[b:e], vecs[1])]]))
        dist.all_reduce(t)
        s = t.numpy()
        np.testing.assert_allclose(-s[:-1] / N2, og, rtol=0, atol=1e-12 * np.max(np.abs(og)))
        assert abs(-s[-1] / N2 - ov) < 1e-12 * max(1.0, abs(ov))
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    world = 2
    port = 29500 + os.getpid() % 2000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: 1, 1: 1}


@pytest.mark.parametrize('n,g', [(4096, 8), (101, 2), (7, 8), (16384, 3)])
def test_shard_rows_partitions(n, g):
    sys.path.insert(0, ROOT)
    from viabel_amd.objectives import shard_rows
    blocks = [shard_rows(n, g, r) for r in range(g)]
    assert blocks[0][0] == 0 and blocks[-1][1] == n
    for (b0, e0), (b1, e1) in zip(blocks, blocks[1:]):
        assert e0 == b1 and e0 >= b0
    sizes = [e - b for b, e in blocks]
    assert max(sizes) - min(sizes) <= 1
