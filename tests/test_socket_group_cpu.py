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
        # every rank's 64-byte IPC window handle on every rank, in rank order (attach(..., transport='ipc'))
        got = g.allgather_bytes(bytes([rank]) * 64)
        assert got == [bytes([r]) * 64 for r in range(world)]
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
    assert g.allgather_bytes(b'abc') == [b'abc']
    g.close()


def _token_worker(rank, world, port, token, q):
    sys.path.insert(0, ROOT)
    try:
        from viabel_amd import distributed
        g = distributed.SocketGroup(rank, world, '127.0.0.1', port, timeout=60.0, token=token)
        got = g.broadcast_bytes(token if rank == 0 else None)
        g.barrier()
        g.close()
        q.put((token, rank, got))
    except Exception as exc:                 # pragma: no cover
        q.put((token, rank, repr(exc)))


def test_two_jobs_with_the_same_port_do_not_capture_each_other():
    """two jobs of the same world size whose control ports collide (job B's rank 0 walks to the next
    port) -- every rank must end up in its own job: the handshake carries a job token that both sides check."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_token_worker, args=(r, 2, port, tok, q)) for tok in (b'job-A', b'job-B') for r in (0, 1)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(4)]
    for p in procs:
        p.join(30)
    for tok, rank, got in results:
        assert got == tok, results


def test_job_token_from_env():
    sys.path.insert(0, ROOT)
    from viabel_amd.distributed import SocketGroup
    a = SocketGroup.job_token({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29500', 'TORCHELASTIC_RUN_ID': 'x'})
    b = SocketGroup.job_token({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29501', 'TORCHELASTIC_RUN_ID': 'x'})
    c = SocketGroup.job_token({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29500', 'VIABEL_AMD_JOB_ID': 'y'})
    assert len({a, b, c}) == 3


def test_rendezvous_timeout_names_the_missing_ranks_and_ignores_silent_clients():
    sys.path.insert(0, ROOT)
    import threading
    import time
    from viabel_amd import distributed
    port = _free_port()
    stray = []

    def silent_client():                     # connects, says nothing: must not stall the accept loop for the deadline
        time.sleep(0.3)
        s = socket.socket()
        try:
            s.connect(('127.0.0.1', port))
            stray.append(s)
        except OSError:
            pass
    t = threading.Thread(target=silent_client)
    t.start()
    t0 = time.time()
    try:
        distributed.SocketGroup(0, 3, '127.0.0.1', port, timeout=4.0)
        raise AssertionError('rendezvous should have timed out')
    except RuntimeError as exc:
        assert 'ranks [1, 2] never arrived' in str(exc), exc
    assert time.time() - t0 < 10
    t.join()
    for s in stray:
        s.close()


def _array_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    try:
        import numpy as np
        from viabel_amd import distributed
        g = distributed.SocketGroup(rank, world, '127.0.0.1', port, timeout=60.0)
        a = np.arange(1000, dtype=np.float64) * (rank + 1)
        g.allreduce_array(a)
        m = np.full((4, 5), float(rank))
        g.allreduce_array(m, op=1)
        big = np.full(600_000, 0.5 + rank)           # larger than one TCP segment / socket buffer
        g.allreduce_array(big)
        ok = (np.array_equal(a, np.arange(1000.0) * (world * (world + 1) / 2)) and np.all(m == world - 1)
              and np.all(big == sum(0.5 + r for r in range(world))))
        g.barrier()
        g.close()
        q.put((rank, 'ok' if ok else 'wrong result'))
    except Exception as exc:                 # pragma: no cover
        q.put((rank, repr(exc)))


def test_allreduce_array_sum_and_max():
    """The vector collective of the host-staged transport (`attach(..., transport='host')`)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_array_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(30)
    assert results == {0: 'ok', 1: 'ok', 2: 'ok'}, results
