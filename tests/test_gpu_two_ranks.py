"""A real two-rank job on ONE GPU: both ranks open device 0, the Monte-Carlo axis is sharded (ragged: odd sample
counts), and every device collective of the sharded path runs -- through the host-staged transport
(`vb_comm_init_host`, `distributed.attach(..., transport='host')`), because RCCL refuses two ranks on one device -- and through the xGMI-native transport (`vb_comm_init_ipc`,
transport='ipc': the ranks map each other's windows through IPC handles and reduce on the device, device-side flags),
which between two processes on one GPU runs the same kernels and flag protocol an 8-GPU node would -- but both ranks
share one L2 here, so nothing of the cross-device visibility the transport depends on (fine-grained windows, system-scope
flags over xGMI) is exercised: that needs two GPUs and is unmeasured; RCCL stays the default transport.
What this covers that the one-rank communicator tests (test_gpu_comm.py) cannot: shard offsets of the second rank,
ragged gathers of per-sample vectors, rank 0's host random draws reaching rank 1, the collective sequence of every
objective staying paired across ranks (a mismatch deadlocks or trips the size check), the device fit loop with one
all-reduce per iteration.  Results are compared with the same evaluations in a single process: the noise is indexed by
global sample row, so only the order of the partial sums differs.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS = os.path.join(ROOT, 'tests')

WORKER = '''
import os, sys
# Two processes on ONE device can hold each other up for tens of seconds (see the comment at the prewarm below: 42 s seen;
# about one run in three of this test sees such a stall somewhere, with or without the prewarm): the transport's wall-time
# bound -- 20 s by default, meant for a peer that died -- is raised so that a stall resolves itself instead of poisoning the
# communicator.  With a GPU per rank none of this applies; the give-up itself is tested below with a one-second bound.
os.environ.setdefault('VB_IPC_TIMEOUT_S', '60')
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np
from viabel_amd import _lib, distributed
import viabel_amd as vb
eng = _lib.Engine(0)                                  # both ranks on the one GPU of the box
_lib.set_default_engine(eng)
group = distributed.SocketGroup.from_env(timeout=120.0)
import _two_rank_scenarios as S
# Two processes on ONE device: a process that sets up queues / streams / copy engines for the first time while the other
# process's collective kernel spins on the same GPU can stall for tens of seconds (round 6, tools/r6_ipc_first_call_probe.py:
# 42 s, or until the spinning kernel gives up and poisons the communicator).  Every code path once without a communicator
# first -- the scenarios reseed everything they draw from, so the sharded pass below computes what it would have anyway.
S.run_all(vb)
group.barrier()
distributed.attach(eng, group, transport=%(transport)r)
assert eng.comm_info() == (2, group.rank)
# round 6: every dense-family objective stays on its device-resident route under a communicator -- the host-root / host-weight
# entry points the sharded jobs used to fall back to must not be called at all
fallbacks = {}
def spy(name):
    real = getattr(eng, name)
    def wrapped(*a, **k):
        fallbacks[name] = fallbacks.get(name, 0) + 1
        return real(*a, **k)
    setattr(eng, name, wrapped)
for name in ('dis_refresh_mvt', 'dis_grad_mvt', 'elbo_sums_mvt', 'alpha_sums_mvt', 'sym_sqrt'):
    spy(name)
try:
    res = S.run_all(vb)
except Exception as exc:      # (the parent decides what a failure means: it reads this file)
    open(os.path.join(%(out)r, 'rank%%d.err' %% group.rank), 'w').write('%%s: %%s' %% (type(exc).__name__, exc))
    raise
assert not fallbacks, fallbacks
np.savez(os.path.join(%(out)r, 'rank%%d.npz' %% group.rank),
         **{k + '__v': v[0] for k, v in res.items()}, **{k + '__g': v[1] for k, v in res.items()})
group.barrier()
group.close()
print('{"rank": %%d, "done": true}' %% group.rank)
'''


def _rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


@pytest.mark.parametrize('transport', ['host', 'ipc'])
def test_two_ranks_on_one_gpu_match_one_rank(tmp_path, transport):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, TESTS)
    import bench
    import viabel_amd as vb
    from viabel_amd import _lib
    import _two_rank_scenarios as S

    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'root': ROOT, 'tests': TESTS, 'out': str(tmp_path), 'transport': transport})
    for attempt in range(3):
        rc, lines = bench.spawn_ranks(2, [sys.executable, str(script)], timeout_s=1500)
        errs = [(tmp_path / ('rank%d.err' % r)).read_text() for r in (0, 1) if (tmp_path / ('rank%d.err' % r)).exists()]
        stalled = transport == 'ipc' and rc != 0 and any('did not reach collective phase' in e for e in errs)
        if not stalled:
            break
        # Two PROCESSES on one GPU: a device-side wait of the IPC transport ran into its wall-time bound because the two
        # processes held each other up on the shared device (measured in round 6, with a GPU per rank it cannot
        # happen; every collective had completed when the give-up was reported).  Not a result of the code under test:
        # run the job again, and if the box keeps doing it say so instead of failing.
        for r in (0, 1):
            for suffix in ('err', 'npz'):
                f = tmp_path / ('rank%d.%s' % (r, suffix))
                if f.exists():
                    f.unlink()
    else:
        pytest.skip('two processes on one GPU stalled each other past the IPC transport\'s wall-time bound three times in a '
                    'row (a property of sharing the device, DESIGN 6); the same scenarios pass over the host-staged transport')
    assert rc == 0, (lines[-5:], errs)

    eng = _lib.default_engine()
    assert eng.comm_info() == (1, 0)
    single = S.run_all(vb)
    ranks = [np.load(tmp_path / ('rank%d.npz' % r)) for r in (0, 1)]
    worst = {}
    for name, (v, g) in single.items():
        for r in (0, 1):
            rv, rg = ranks[r][name + '__v'], ranks[r][name + '__g']
            assert rv.shape == v.shape and rg.shape == g.shape, name
            worst[name] = max(worst.get(name, 0.0), _rel(rv, v), _rel(rg, g))
        # the two ranks hold the same replicated result, bit for bit (they ran the same epilogue on the same sums)
        assert np.array_equal(ranks[0][name + '__v'], ranks[1][name + '__v']), name
        assert np.array_equal(ranks[0][name + '__g'], ranks[1][name + '__g']), name
    bad = {k: e for k, e in worst.items() if not e < (1e-9 if k.startswith('fit_') else 1e-11)}
    print('two ranks vs one: %d scenarios, worst relative differences: %s'
          % (len(worst), ', '.join('%s %.1e' % kv for kv in sorted(worst.items(), key=lambda kv: -kv[1])[:6])))
    assert len(worst) >= 44 and not bad, (bad, worst)
    assert sum(k.startswith('c3_') for k in worst) >= 11


GIVE_UP_WORKER = '''
import os, sys, time
os.environ['VB_IPC_TIMEOUT_S'] = '1'
sys.path.insert(0, %(root)r)
from viabel_amd import _lib, distributed
eng = _lib.Engine(0)
group = distributed.SocketGroup.from_env(timeout=120.0)
distributed.attach(eng, group, transport='ipc')
group.barrier()
if group.rank == 0:
    t0 = time.time()
    try:
        eng.comm_allreduce_time(1024, warm=0, reps=1)      # the peer never joins this collective
        raise SystemExit('the abandoned collective returned without an error')
    except _lib.EngineError as e:
        assert 'did not reach collective phase' in str(e) and 'VB_IPC_TIMEOUT_S' in str(e), str(e)
    waited = time.time() - t0
    assert 0.8 < waited < 10.0, waited      # seconds of wall time, not minutes of polls
else:
    time.sleep(3.0)
group.barrier()
print('{"rank": %%d, "done": true}' %% group.rank, flush=True)
os._exit(0)       # (the communicator is poisoned by design: no orderly teardown of a collective that never completed)
'''


def test_ipc_wait_gives_up_after_wall_time_not_poll_counts(tmp_path):
    """a peer that never arrives used to keep the spinning kernels resident for minutes (2^27 polls); the bound is
    wall time now (VB_IPC_TIMEOUT_S), the give-up is reported as VB_ERR_COMM by the call that ran into it."""
    sys.path.insert(0, ROOT)
    import bench
    script = tmp_path / 'give_up.py'
    script.write_text(GIVE_UP_WORKER % {'root': ROOT})
    rc, lines = bench.spawn_ranks(2, [sys.executable, str(script)], timeout_s=300)
    assert rc == 0, lines[-8:]


def test_three_ranks_on_one_gpu_match_one_rank(tmp_path):
    """Three ranks (host-staged transport) on one GPU: the MIDDLE rank's shard starts and ends inside the gathered vectors --
    shard offsets, ragged block lengths and the one-collective gather of [log q | log p | log prior] as an 8-GPU job's inner
    ranks have them -- on the device-resident routes of the dense families and at the C3 size."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, TESTS)
    import bench
    import viabel_amd as vb
    import _two_rank_scenarios as S
    script = tmp_path / 'worker3.py'
    script.write_text((WORKER % {'root': ROOT, 'tests': TESTS, 'out': str(tmp_path), 'transport': 'host'})
                      .replace('S.run_all(vb)', 'S.run_three_ranks(vb)').replace('== (2, group.rank)', '== (3, group.rank)'))
    rc, lines = bench.spawn_ranks(3, [sys.executable, str(script)], timeout_s=1500)
    assert rc == 0, lines[-5:]
    single = S.run_three_ranks(vb)
    ranks = [np.load(tmp_path / ('rank%d.npz' % r)) for r in (0, 1, 2)]
    worst = {}
    for name, (v, g) in single.items():
        for r in (0, 1, 2):
            worst[name] = max(worst.get(name, 0.0), _rel(ranks[r][name + '__v'], v), _rel(ranks[r][name + '__g'], g))
            assert np.array_equal(ranks[r][name + '__v'], ranks[0][name + '__v']), name      # every rank: the same bits
            assert np.array_equal(ranks[r][name + '__g'], ranks[0][name + '__g']), name
    bad = {k: e for k, e in worst.items() if not e < 1e-11}
    assert len(worst) >= 14 and not bad, (bad, worst)
