"""Monte-Carlo-axis data parallelism: one process per GPU, one RCCL all-reduce per call.

The reference is single-process; every estimator on the hot path is a mean (or a
weighted sum) over the ``num_mc_samples`` rows, so rank r evaluates the contiguous row block
``shard_rows(N, G, r)`` and the fp64 partial-sum vector ``[scalars | column sums]`` is
sum-all-reduced over RCCL/xGMI on the device before the O(P) epilogue
(``vb_comm_init`` in ``include/viabel_hip.h``).  The variational parameter is replicated.

``torch.distributed`` (any backend, gloo is enough) is used only to hand the 128-byte RCCL
unique id from rank 0 to the other ranks; no tensor ever goes through torch.
"""
import os

import numpy as np

from . import _lib
from .objectives import shard_rows  # noqa: F401  (re-exported)


def combine_partial_sums(partials):
    """Reference semantics of the device all-reduce: elementwise sum of the ranks' vectors."""
    return np.sum(np.stack([np.asarray(p, dtype=np.float64) for p in partials]), axis=0)


def broadcast_unique_id(rank, make_id):
    """Rank 0 creates the RCCL unique id; everybody receives it via torch.distributed."""
    import torch.distributed as dist
    if not dist.is_initialized():
        raise RuntimeError('torch.distributed is not initialised; call init_process_group first')
    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def attach(engine=None):
    """Attach an RCCL communicator spanning the torch.distributed world to ``engine``."""
    import torch.distributed as dist
    engine = engine or _lib.default_engine()
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        return engine
    uid = broadcast_unique_id(rank, _lib.Engine.comm_unique_id)
    engine.comm_init(uid, world, rank)
    return engine


def init_from_env(backend='gloo'):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun) and attach RCCL."""
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend=backend)
    engine = _lib.default_engine()
    if world > 1:
        attach(engine)
    return engine
