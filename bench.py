#!/usr/bin/env python3
"""Headline benchmark: ELBO-gradient evaluations per second (BASELINE.json `metric`).

Workload = BASELINE.json configs[1]: MFGaussian + ExclusiveKL on the D=1024 funnel with
N_mc=4096 Monte-Carlo samples per GPU, fp64.  One "step" = one objective evaluation
(value + gradient) over one N x D noise matrix resident in HBM.  A ring of noise matrices
larger than the 256-MiB Infinity Cache is cycled so every step streams its noise from HBM,
as an optimisation loop (fresh noise per iteration) does.

N > 1 GPUs (launched by torch.distributed.run, one rank per GPU): the Monte-Carlo axis is
sharded -- every rank holds 4096 rows (weak scaling, N_mc global = 4096 x GPUs) and the
partial sums are all-reduced over RCCL inside every evaluation.  `value` counts
4096-sample evaluation units: GPUs x evaluations / second.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

D, N_MC = 1024, 4096
ALGO_BYTES = N_MC * D * 8 + 4 * D * 8 + 8        # noise read + theta read + grad write + value
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: 8.0 TB/s spec
# HBM bytes per evaluation of the accumulate kernel from the PMC passes committed under
# profiles/r01_meanfield_c1_pmc_hbm.txt: (2 x FETCH_SIZE [gfx950 correction] + WRITE_SIZE) KiB per
# 32-evaluation launch
PMC_TRAFFIC_BYTES_PER_EVAL = (2 * 530474.5 + 8352.7) * 1024 / 32
FP64_MFMA_PEAK_TFLOPS = 78.6                     # datasheet; tools/fp64_peak.hip measures 63-73 (4x4x4 form)


def fullrank_leg(eng, vb, steps=150, warmup=60, world=1, rank=0, barrier=None):
    """Secondary measurement: the dense (full-rank) Gaussian family named by north_star, D=1024, N=4096 rows
    per GPU, correlated-Gaussian target, parameter resident on the device.  fp64 MFMA-bound.  With world > 1
    the Monte-Carlo axis is sharded like the headline leg: every evaluation all-reduces its
    1 + D + D(D+1)/2 partial sums over RCCL (4.2 MB), so all ranks have to call this together."""
    d, n = 1024, N_MC
    n_total = n * world
    barrier = barrier or (lambda: None)
    rng = np.random.RandomState(2)
    A = rng.randn(d, d)
    model = vb.CorrelatedGaussianModel(rng.randn(d), covariance=A @ A.T / d + np.eye(d))
    eng.set_model(model.device_spec())
    fr = vb.FullRankGaussian(d)
    L = np.exp(-1.0) * np.eye(d) + 0.01 * np.tril(np.random.RandomState(3).randn(d, d))
    eng.fullrank_set_theta(fr.pack(np.zeros(d), L), d)
    ring = 8
    for s in range(ring):
        eng.noise_generate(40 + s, n, d, seed=2, stream=s, row_offset=rank * n)
    t_ramp = time.perf_counter()
    for _ in range(4 if world > 1 else 1000):          # untimed clock ramp (see main); fixed count when the
        for i in range(warmup):                        # calls are collective so that every rank issues the same
            eng.elbo_grad_fullrank_enqueue(40 + i % ring, n, d, n_total=n_total)
        eng.sync()
        if world == 1 and time.perf_counter() - t_ramp > 0.2:
            break
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.elbo_grad_fullrank_enqueue(40 + i % ring, n, d, n_total=n_total)
    eng.sync()
    dt = (time.perf_counter() - t0) / steps
    barrier()
    value, grad = eng.fullrank_get(d)
    flops = 4.0 * n * d * d + 2.0 * n * d * d          # Z = E L^T, G^T E (dense convention) + target's (Z - m) P
    return {
        'workload': 'FullRankGaussian + ExclusiveKL, D=1024, N_mc=4096 per GPU, correlated-Gaussian target, fp64',
        'evals_per_s': world / dt, 'us_per_eval': 1e6 * dt, 'steps': steps, 'n_gpus': world,
        'counting': '4096-sample evaluation units over all GPUs per second (rank 0 clock)',
        'roofline': {'bound': 'mfma', 'achieved': flops / dt / 1e12, 'peak': FP64_MFMA_PEAK_TFLOPS,
                     'unit': 'TFLOP/s', 'frac': flops / dt / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                     'flops_per_eval_dense_convention': flops,
                     'note': 'whole evaluation (3 MFMA GEMMs + O(ND) kernels); kernel split in profiles/'},
        'check': {'value': value, 'grad_norm': float(np.linalg.norm(grad))},
    }


def fit_leg(vb, theta, iters=1500):
    """Secondary measurement: a whole RMSProp fit at the C1 shape (fresh Philox noise every iteration) through
    the host loop (one blocking objective call + numpy step per iteration, optimization.py:91-112) and through
    the device-resident loop (vb_fit); the two trajectories are the same bit for bit."""
    from viabel_amd.optimization import RMSProp
    out = {'workload': 'RMSProp(0.01), MFGaussian(rng=philox) + ExclusiveKL, D=1024 funnel, N_mc=4096, %d iterations'
                       % iters}
    hist = {}
    for mode, on_device in (('host_loop', False), ('device_loop', True)):
        obj = vb.ExclusiveKL(vb.MFGaussian(D, rng='philox'), vb.FunnelModel(D), N_MC)
        opt = RMSProp(0.01)
        opt.optimize(200, obj, theta, on_device=on_device)
        t0 = time.perf_counter()
        hist[mode] = opt.optimize(iters, obj, theta, on_device=on_device)['value_history']
        out[mode + '_us_per_iteration'] = 1e6 * (time.perf_counter() - t0) / iters
    out['trajectories_identical'] = bool(np.array_equal(hist['host_loop'], hist['device_loop']))
    return out


def cpu_baseline(theta, budget_s=12.0):
    """Oracle (numpy fp64 restatement of objectives.py:154-168) on the host cores, bounded."""
    from oracle import families as ofam, models as omod, objectives as oobj
    fam, model = ofam.MFGaussian(D), omod.Funnel(D)
    t0 = time.perf_counter()
    noise = np.random.RandomState(1).randn(N_MC, D)
    t_rng = time.perf_counter() - t0
    oobj.exclusive_kl(fam, model, theta, noise)          # warm-up
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        oobj.exclusive_kl(fam, model, theta, noise)
        n += 1
    dt = time.perf_counter() - t0
    try:
        import threadpoolctl
        blas_threads = max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        blas_threads = 1
    return {
        'value': n / dt, 'unit': 'evals/s', 'cores': 1, 'kind': 'port',
        'sample': '%d evaluations of the numpy oracle at the full C1 shape (D=1024, N_mc=4096), noise '
                  'pre-generated; RandomState.randn for one matrix took %.3f s on top; host has %d cpus, '
                  'elementwise numpy is single-threaded (BLAS threads %d unused: no GEMM on this path)'
                  % (n, t_rng, os.cpu_count(), blas_threads),
    }


def parity_check(eng, theta, fam_id):
    """BASELINE's second metric: relative ELBO error of the HIP path against the CPU restatement on the same
    noise (one of the resident Philox matrices read back from the device).  north_star asks <= 1e-5."""
    from oracle import families as ofam, models as omod, objectives as oobj
    noise = eng.noise_get_host(0, N_MC, D)
    dv, dg = eng.elbo_grad_meanfield(0, N_MC, D, theta, fam_id)
    ov, og = oobj.exclusive_kl(ofam.MFGaussian(D), omod.Funnel(D), theta, noise)
    return {'rel_elbo_err': abs(dv - ov) / abs(ov), 'rel_grad_err': float(np.max(np.abs(dg - og)) / np.max(np.abs(og))),
            'tolerance': 1e-5, 'against': 'numpy oracle (oracle/objectives.py) on the same 4096 x 1024 noise matrix'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--ring', type=int, default=16, help='noise matrices cycled (16 x 33.5 MB > L3)')
    ap.add_argument('--engines', type=int, default=1,
                    help='independent HIP contexts (streams) the evaluations are spread over, so the '
                         'small prep / finalize kernels of one evaluation overlap the streaming kernel of another')
    ap.add_argument('--batch', type=int, default=32,
                    help='independent evaluations per API call (share one launch of each kernel)')
    ap.add_argument('--force-comm', action='store_true',
                    help='attach an RCCL communicator even with one rank (exercises the sharded code path)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fit', action='store_true', help='skip the secondary optimiser-loop measurement')
    ap.add_argument('--no-fullrank', action='store_true', help='skip the secondary full-rank measurement')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))

    from viabel_amd import _lib, distributed
    import viabel_amd as vb

    dist = None
    use_comm = world > 1 or args.force_comm
    if use_comm:
        import torch.distributed as dist
        if 'RANK' not in os.environ:      # plain `python bench.py --force-comm`
            os.environ.update(RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29531')
        dist.init_process_group(backend='gloo')
    eng = _lib.default_engine()
    # VB_BENCH_NO_RCCL=1: control-flow dry run of the N > 1 branches on a box with fewer GPUs than ranks (the ranks
    # then share a device, RCCL refuses that, and no communicator is attached: the numbers mean nothing)
    no_rccl = world > 1 and os.environ.get('VB_BENCH_NO_RCCL') == '1'
    if use_comm and not no_rccl:
        if world > 1:
            distributed.attach(eng)
        else:
            eng.comm_init(_lib.Engine.comm_unique_id(), 1, 0)

    model = vb.FunnelModel(D)
    theta = np.concatenate([np.zeros(D), -np.ones(D)])       # mu = 0, log sigma = -1
    n_total = N_MC * world
    batch = max(1, min(args.batch, 32))
    engines = [eng] + [_lib.Engine(eng.device) for _ in range(max(1, args.engines) - 1)]
    if world > 1 and not no_rccl:
        for e in engines[1:]:
            distributed.attach(e)
    elif use_comm:
        for e in engines[1:]:
            e.comm_init(_lib.Engine.comm_unique_id(), 1, 0)
    n_eng = len(engines)
    # noise matrices per engine: at least one batch, and > 256 MiB L3 over all engines
    ring = min(max(batch, (args.ring + n_eng - 1) // n_eng), _lib.MAX_SLOTS - 8)
    for k, e in enumerate(engines):                              # synthetic noise, resident in HBM
        e.set_model(model.device_spec())
        for s in range(ring):
            e.noise_generate(s, N_MC, D, seed=1, stream=k * ring + s, row_offset=rank * N_MC)
    fam = _lib.FAMILY_MF_GAUSSIAN
    thetas = np.tile(theta, (batch, 1))
    n_rsets = _lib.MAX_SLOTS // batch

    def sync_all():
        for e in engines:
            e.sync()

    def barrier():
        sync_all()
        if dist is not None:
            dist.barrier()

    def run(steps):
        """Enqueue exactly `steps` evaluations, `batch` per call, round-robin over the engines."""
        done, call = 0, 0
        last = None
        while done < steps:
            b = min(batch, steps - done)
            e = engines[call % n_eng]
            j = call // n_eng
            slots = [(j * batch + i) % ring for i in range(b)]
            rslots = [(j % n_rsets) * batch + i for i in range(b)]
            e.elbo_grad_meanfield_batch_async(slots, N_MC, D, thetas[:b], fam, rslots, n_total=n_total)
            last = (e, rslots[-1])
            done += b
            call += 1
        return last

    # untimed clock ramp: the GPU's power state follows load with tens of milliseconds of lag, and the timed
    # region below is only ~13 ms long at the default K, so bring the device to its sustained clocks first
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.3:
        run(4 * batch)
        sync_all()
    run(args.warmup)
    barrier()
    for e in engines:
        e.profile_enable(True)
        e.profile_read(reset=True)
    barrier()
    t0 = time.perf_counter()
    last = run(args.steps)
    sync_all()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    barrier()
    launches, evals_timed, kernel_ms = 0, 0, 0.0
    for e in engines:
        n_l, n_e, ms = e.profile_read(reset=True)
        launches += n_l
        evals_timed += n_e
        kernel_ms += ms
        e.profile_enable(False)
    last_value, last_grad = last[0].result_get(last[1], 2 * D)

    # blocking-call rate (what a host-side optimiser loop sees: one evaluation per call, wait for it)
    n_sync = min(args.steps, 500)
    t2 = time.perf_counter()
    for i in range(n_sync):
        eng.elbo_grad_meanfield(i % ring, N_MC, D, theta, fam, n_total=n_total)
    sync_rate = n_sync / (time.perf_counter() - t2)
    # the same blocking call on FRESH noise generated inside the streaming kernel (Philox + Box-Muller in
    # registers: what an optimiser loop in rng='philox' mode issues every iteration)
    t3 = time.perf_counter()
    for i in range(n_sync):
        eng.elbo_grad_meanfield_philox(0, N_MC, D, theta, fam, 1, 1000 + i, n_total=n_total, row_offset=rank * N_MC)
    fresh_rate = n_sync / (time.perf_counter() - t3)

    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        kernel_us = 1e3 * kernel_ms / max(1, launches)
        bytes_per_launch = ALGO_BYTES * evals_timed / max(1, launches)
        achieved = bytes_per_launch / (kernel_us * 1e-6) / 1e9
        out = {
            'metric': 'ELBO-gradient evals/sec (D=1024, N_mc=4096)',
            'value': world * args.steps / elapsed,
            'unit': 'evals/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {
                'workload': 'BASELINE configs[1]: MFGaussian + ExclusiveKL, D=1024 funnel, N_mc=4096 per GPU',
                'family': 'MFGaussian', 'objective': 'ExclusiveKL (entropy form)', 'model': 'funnel',
                'dim': D, 'n_mc_per_gpu': N_MC, 'n_mc_global': n_total,
                'noise': 'Philox4x32-10 normals resident in HBM, %d matrices cycled (%.0f MB > 256 MiB L3)' % (
                    ring * n_eng, ring * n_eng * N_MC * D * 8 / 1e6),
                'parallelism': 'mc-axis dp%d, one RCCL all-reduce of %d doubles per evaluation' % (
                    world, 8 + 2 * D) if world > 1 else 'single GPU',
                'pipelining': '%d independent evaluations (own noise matrix, own theta, own result) per API '
                              'call share one launch of each kernel; calls go round-robin to %d HIP streams; '
                              'results are written to pinned host memory by the finalize kernel; the '
                              'one-evaluation blocking-call rate is sync_call_evals_per_s' % (batch, n_eng),
            },
            'sync_call_evals_per_s': world * sync_rate,
            'fresh_noise_sync_call_evals_per_s': world * fresh_rate,
            'check': {'value': last_value, 'grad_norm': float(np.linalg.norm(last_grad))},
            'roofline': {
                'bound': 'hbm', 'kernel': 'mf_accum_kernel', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                'traffic': PMC_TRAFFIC_BYTES_PER_EVAL * evals_timed / max(1, launches),
                'traffic_source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/r01_meanfield_c1_pmc_hbm.txt',
                'algorithmic_bytes_per_launch': bytes_per_launch, 'evals_per_launch': evals_timed / max(1, launches),
                'avg_kernel_us': kernel_us, 'launches_timed': launches,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            # the cpu_baseline leg, after the timed region: the only place this file touches oracle/ -- timed as
            # the CPU baseline and, on the same inputs, used as the checker of the HIP result (never measured as ours)
            out['cpu_baseline'] = cpu_baseline(theta)
            out['parity'] = parity_check(eng, theta, fam)
    if out is not None and world == 1 and not args.no_fit:
        import contextlib
        import io
        with contextlib.redirect_stderr(io.StringIO()):    # tqdm progress bars of the host loop
            out['fit_loop'] = fit_leg(vb, theta)
    if not args.no_fullrank:                               # collective when world > 1: every rank runs it
        fr_out = fullrank_leg(eng, vb, world=world, rank=rank, barrier=barrier)
        if out is not None:
            out['fullrank'] = fr_out
    if rank == 0:
        # RCCL prints a version banner through C stdio; push it out first so the JSON is the last line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
