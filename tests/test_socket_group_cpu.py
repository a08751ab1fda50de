"""The torch-free control plane of a sharded job (viabel_amd.distributed.SocketGroup): three CPU processes
rendezvous over TCP on 127.0.0.1, hand rank 0's RCCL unique id to everybody, run barriers and scalar reductions
(what bench.py brackets its timed region with)."""
import multiprocessing as mp
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q, occupy_first):
    sys.path.insert(0, ROOT)
    try:
        from viabel_amd import distributed
        g = distributed.SocketGroup(rank, world, '127.0.0.1', port, timeout=60.0)
        uid = distributed.broadcast_unique_id(rank, lambda: b'uid-from-rank-0' + bytes(113), g)
        assert uid == b'uid-from-rank-0' + bytes(113) and len(uid) == 128
        g.barrier()
        assert g.allreduce_max(float(rank) + 0.25) == world - 1 + 0.25
        assert g.allreduce_sum(float(rank + 1)) == world * (world + 1) / 2
        for i in range(50):                  # many back-to-back collectives stay paired
            assert g.allreduce_max(i * 10.0 + rank) == i * 10.0 + world - 1
        g.barrier()
        g.close()
        q.put((rank, 'ok'))
    except Exception as exc:                 # pragma: no cover
        q.put((rank, repr(exc)))


def _run(world, occupy_first=False):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    blocker = None
    if occupy_first:                         # the nominal port is taken: rank 0 walks upwards, the others follow
        blocker = socket.socket()
        blocker.bind(('127.0.0.1', port))
        blocker.listen(1)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, occupy_first)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
    if blocker is not None:
        blocker.close()
    assert results == {r: 'ok' for r in range(world)}, results


def test_socket_group_three_ranks():
    _run(3)


def test_socket_group_port_taken():
    _run(2, occupy_first=True)


def test_single_rank_group_is_trivial():
    sys.path.insert(0, ROOT)
    from viabel_amd import distributed
    g = distributed.SocketGroup(0, 1)
    g.barrier()
    assert g.allreduce_max(3.5) == 3.5
    assert g.broadcast_bytes(b'x') == b'x'
    g.close()
