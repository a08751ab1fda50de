"""bench.py's own rank launcher (`python bench.py --gpus N` with no torchrun around it), exercised on the CPU with stub
children: environment hand-out, rank 0's line relayed, a failing rank stops the job, the timeout kills whole process
groups, and a result line whose n_gpus / rccl_ranks is not what was asked for is refused.  No GPU, no viabel_amd import
in the parent (that is the point of the launcher: it must not have touched HIP before it starts the ranks)."""
import json
import os
import subprocess
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _stub(tmp_path, body):
    path = tmp_path / 'stub_rank.py'
    path.write_text(textwrap.dedent(body))
    return [sys.executable, str(path)]


def test_spawn_ranks_env_and_relay(tmp_path):
    import bench
    argv = _stub(tmp_path, '''
        import json, os, sys
        r = int(os.environ['RANK'])
        assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'
        assert os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and len(os.environ['VIABEL_AMD_JOB_ID']) == 32
        print('banner from rank %d' % r)
        print(json.dumps({'metric': 'm', 'rank': r, 'world': int(os.environ['WORLD_SIZE']),
                          'port': int(os.environ['MASTER_PORT'])}))
    ''')
    rc, lines = bench.spawn_ranks(3, argv, timeout_s=60)
    assert rc == 0
    assert lines[0] == 'banner from rank 0'                 # only rank 0's stdout is collected
    out = json.loads(lines[-1])
    assert out['rank'] == 0 and out['world'] == 3 and out['port'] > 0 and len(lines) == 2


def test_spawn_ranks_real_socket_group(tmp_path):
    """The children of the launcher rendezvous through SocketGroup.from_env with the job token it hands out."""
    import bench
    argv = _stub(tmp_path, '''
        import json, os, sys
        sys.path.insert(0, %r)
        from viabel_amd import distributed
        g = distributed.SocketGroup.from_env(timeout=60.0)
        s = g.allreduce_sum(g.rank + 1.0)
        g.barrier(); g.close()
        if g.rank == 0:
            print(json.dumps({'metric': 'm', 'sum': s}))
    ''' % ROOT)
    rc, lines = bench.spawn_ranks(3, argv, timeout_s=120)
    assert rc == 0, lines
    assert json.loads(lines[-1])['sum'] == 6.0


def test_failing_rank_stops_the_job(tmp_path):
    import bench
    argv = _stub(tmp_path, '''
        import os, sys, time
        if os.environ['RANK'] == '1':
            sys.exit(7)
        time.sleep(600)
    ''')
    t0 = time.time()
    rc, lines = bench.spawn_ranks(3, argv, timeout_s=120)
    assert rc == 7 and time.time() - t0 < 30


def test_timeout_kills_process_groups(tmp_path):
    import bench
    marker = tmp_path / 'grandchild.pid'
    argv = _stub(tmp_path, '''
        import os, subprocess, sys, time
        if os.environ['RANK'] == '0':                       # a rank with a child of its own: the group must die
            p = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)'])
            open(%r, 'w').write(str(p.pid))
        time.sleep(600)
    ''' % str(marker))
    t0 = time.time()
    rc, _ = bench.spawn_ranks(2, argv, timeout_s=3.0)
    assert rc == 124 and time.time() - t0 < 40
    pid = int(marker.read_text())
    for _ in range(100):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            break
        # a zombie re-parented to init still answers kill(0) until reaped: check its state
        try:
            state = open('/proc/%d/stat' % pid).read().split(') ')[-1][0]
            if state == 'Z':
                break
        except FileNotFoundError:
            break
        time.sleep(0.1)
    else:
        pytest.fail('grandchild %d survived the launcher timeout' % pid)


def _run_bench_with_stub(tmp_path, line, gpus=2, env_extra=None):
    """bench.launch_main with the rank command replaced by a stub that prints `line` on rank 0."""
    stub = _stub(tmp_path, '''
        import os
        if os.environ['RANK'] == '0':
            print(%r)
    ''' % line)
    code = textwrap.dedent('''
        import sys, types
        sys.path.insert(0, %r)
        import bench
        assert 'viabel_amd' not in sys.modules               # the parent never imports the package
        real = bench.spawn_ranks
        bench.spawn_ranks = lambda world, argv, timeout: real(world, %r, timeout)
        sys.argv = ['bench.py', '--gpus', '%d']
        bench.main()
        assert 'viabel_amd' not in sys.modules
    ''') % (ROOT, stub, gpus)
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=120)


def test_launch_main_relays_a_valid_line(tmp_path):
    line = json.dumps({'metric': 'ELBO-gradient evals/sec (D=1024, N_mc=4096)', 'value': 1.0, 'n_gpus': 2, 'rccl_ranks': 2})
    res = _run_bench_with_stub(tmp_path, line)
    assert res.returncode == 0, res.stderr
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out['n_gpus'] == 2 and 'self-launch' in out['launcher']


@pytest.mark.parametrize('n_gpus,rccl', [(1, 1), (2, 1)])
def test_launch_main_refuses_a_wrong_rank_count(tmp_path, n_gpus, rccl):
    line = json.dumps({'metric': 'm', 'value': 1.0, 'n_gpus': n_gpus, 'rccl_ranks': rccl})
    res = _run_bench_with_stub(tmp_path, line)
    assert res.returncode != 0 and 'refusing' in res.stderr and res.stdout.strip() == ''


def test_launch_main_dry_run_needs_no_communicator(tmp_path):
    line = json.dumps({'metric': 'm', 'value': 1.0, 'n_gpus': 2, 'rccl_ranks': 1})
    res = _run_bench_with_stub(tmp_path, line, env_extra={'VB_BENCH_NO_RCCL': '1'})
    assert res.returncode == 0, res.stderr


def test_rank_refuses_world_size_mismatch():
    """A rank process (WORLD_SIZE set) whose world differs from --gpus exits before importing the engine."""
    env = dict(os.environ, WORLD_SIZE='1', RANK='0')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], capture_output=True, text=True,
                         env=env, timeout=120)
    assert res.returncode != 0 and 'WORLD_SIZE=1' in res.stderr and res.stdout.strip() == ''


# ---- round 6: the second transport of an N > 1 line is measured by CHILD processes of the ranks -------------------------
def _probe_job(tmp_path, child_body, timeout_s=60.0):
    """Two rank processes (bench.spawn_ranks) that each call bench.second_transport_probe with a stub child; rank 0 prints
    what it got."""
    import bench
    child = tmp_path / 'probe_child.py'
    child.write_text(textwrap.dedent(child_body))
    rank = tmp_path / 'probe_rank.py'
    rank.write_text(textwrap.dedent('''
        import json, os, sys
        sys.path.insert(0, %r)
        import bench
        from viabel_amd import distributed
        g = distributed.SocketGroup.from_env(timeout=60.0)
        res = bench.second_transport_probe(g, 'ipc', timeout_s=%r, child_argv=[sys.executable, %r])
        g.barrier(); g.close()
        if g.rank == 0:
            print(json.dumps({'metric': 'm', 'probe': res}))
    ''') % (ROOT, timeout_s, str(child)))
    rc, lines = bench.spawn_ranks(2, [sys.executable, str(rank)], timeout_s=180)
    assert rc == 0, lines
    return json.loads(lines[-1])['probe']


def test_second_transport_probe_relays_the_childs_line(tmp_path):
    res = _probe_job(tmp_path, '''
        import json, os
        # the children get a control port and a job token of their own, and the transport's name
        assert os.environ['VB_BENCH_TRANSPORT'] == 'ipc' and os.environ['VIABEL_AMD_JOB_ID'].endswith('-probe-ipc')
        assert int(os.environ['VIABEL_AMD_CONTROL_PORT']) == int(os.environ['MASTER_PORT']) + 23 + 41
        print('a banner')
        if os.environ['RANK'] == '0':
            print(json.dumps({'dependent_chain': {'weak': {'us_per_iteration': 1.5}}, 'allreduce_us': {'us_per_allreduce': 2.5}}))
    ''')
    assert res['dependent_chain']['weak']['us_per_iteration'] == 1.5 and res['allreduce_us']['us_per_allreduce'] == 2.5


def test_second_transport_probe_reports_a_failing_child_and_goes_on(tmp_path):
    """A child that dies (on ANY rank) costs this transport's numbers, not the line: rank 0 gets the error text."""
    res = _probe_job(tmp_path, '''
        import json, os, sys
        if os.environ['RANK'] == '1':
            sys.exit(9)
        sys.stderr.write('what the rank-0 child said on stderr\\n')
        print(json.dumps({'dependent_chain': {}, 'allreduce_us': {}}))
    ''')
    assert 'what the rank-0 child said' in res['rank0_child_stderr_tail']
    assert set(res) == {'error', 'rank0_child_stderr_tail'} and 'exit code 9' in res['error'] and 'ipc' in res['error']


def test_second_transport_probe_kills_children_at_the_deadline(tmp_path):
    t0 = time.time()
    res = _probe_job(tmp_path, '''
        import time
        time.sleep(600)
    ''', timeout_s=2.0)
    assert set(res) == {'error', 'rank0_child_stderr_tail'} and 'exit code 124' in res['error'] and time.time() - t0 < 60
