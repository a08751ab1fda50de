"""Round 6 probe: which of bench.py's collective legs trips the IPC transport between two ranks of one GPU.
usage (rank processes are started by bench.spawn_ranks):  python tools/r6_ipc_legs_probe.py <transport> <leg>[,<leg>...]
legs: chain, c1, c3, c4, allreduce"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_main(transport, legs):
    import bench
    from viabel_amd import _lib, distributed
    import viabel_amd as vb
    eng = _lib.Engine(0)
    _lib.set_default_engine(eng)
    group = distributed.SocketGroup.from_env()
    distributed.attach(eng, group, transport=transport)
    out = {}
    for leg in legs:
        sys.stderr.write('rank %d: leg %s\n' % (group.rank, leg))
        if leg == 'chain':
            out[leg] = bench.dependent_chain_leg(eng, vb, group)
        elif leg == 'c1':
            out[leg] = bench.sharded_c1_leg(eng, vb, group)
        elif leg == 'c3':
            out[leg] = bench.sharded_c3_leg(eng, vb, group)
        elif leg == 'c4':
            out[leg] = bench.sharded_c4_leg(eng, vb, group)
        elif leg == 'allreduce':
            out[leg] = eng.comm_allreduce_time(525840, warm=5, reps=30)
        eng.sync()
        group.barrier()
    if group.rank == 0:
        print(json.dumps(out))
    group.barrier()
    group.close()


if __name__ == '__main__':
    if 'WORLD_SIZE' in os.environ:
        rank_main(sys.argv[1], sys.argv[2].split(','))
    else:
        import bench
        rc, lines = bench.spawn_ranks(2, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], timeout_s=600,
                                      extra_env={'VB_IPC_TIMEOUT_S': '4'})
        print('rc', rc)
        for line in lines[-3:]:
            print(line[:1500])
