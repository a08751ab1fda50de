#!/usr/bin/env python3
"""Headline benchmark: ELBO-gradient evaluations per second (BASELINE.json `metric`).

Workload = the north_star target: FullRankGaussian + ExclusiveKL, D=1024, N_mc=4096 Monte-Carlo samples,
correlated-Gaussian target (dense precision matrix), fp64.  One "step" = one objective evaluation (value +
gradient of all 525 824 parameters) over one N x D noise matrix resident in HBM: sampling GEMM Z = E L' + mu
(triangular), model GEMM G = -(Z - m) P (dense), gradient GEMM C = G' E (lower triangle), reductions and the
O(P) epilogue -- every evaluation is enqueued on ONE HIP stream behind the previous one, as an optimiser loop
issues them.  A ring of 8 noise matrices (268 MB > the 256-MiB Infinity Cache) is cycled.

N > 1 GPUs: one rank per GPU.  Under torch.distributed.run (WORLD_SIZE in the environment) this process IS a rank;
a bare `python bench.py --gpus N` starts its own N rank processes (spawn_ranks: subprocesses, never exec) and relays
rank 0's line.  The ranks talk over RCCL, the control path is a plain TCP socket group -- no torch import anywhere:
  --scaling weak   (default)  every rank holds 4096 rows, N_mc global = 4096 x GPUs, `value` counts
                              4096-sample evaluation units: GPUs x evaluations / second;
  --scaling strong            N_mc = 4096 global, rank r holds rows shard_rows(4096, G, r), `value` =
                              evaluations / second of the whole job.
Either way every evaluation all-reduces its 16 + D + D(D+1)/2 partial sums (4.2 MB) over RCCL.

Secondary legs (rank 0 of a 1-GPU run only, after the timed region): BASELINE configs[1] (mean field, HBM-bound),
configs[2] (D=512 full rank), configs[3] (MultivariateT + DIS), configs[4] (logistic regression), the optimiser loop,
the CPU baseline and the parity check.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FR_D, N_MC = 1024, 4096
D1 = 1024                                        # configs[1] dimension
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_MFMA_PEAK_TFLOPS = 78.6                     # datasheet fp64 matrix peak = 256 CUs x 4 SIMDs x 512 flop / 16 clk x 2.4 GHz;
#                                                  tools/mfma_barrier_probe.hip measures 77.3 with constant operands
# HBM-side bytes per launch of the headline's dense model GEMM at D=1024, N=4096 (see roofline.traffic_source)
MODEL_GEMM_HBM_BYTES = int((2 * 64267.8 + 34348.7) * 1024)      # profiles/r06_fullrank_gemm_pmc.txt (FETCH_SIZE x 2 + WRITE_SIZE, KiB)
MF_ACCUM_HBM_BYTES = int((2 * 530477.5 + 8352.6) * 1024)           # profiles/r06_meanfield_c1_pmc_hbm.txt, per 32-evaluation launch
MIN_TIMED_S = 0.05                               # the timed blocks are repeated until they cover at least this


def fr_flops(n, d):
    """Executed-work model of one full-rank evaluation (exact triangles: the kernels skip the zero halves)."""
    tri = float(n) * d * (d + 1)                 # sum_j 2 n (j + 1): Z = E L' with lower-triangular L; likewise tril(G' E)
    dense = 2.0 * n * d * d
    return {'sample_gemm': tri, 'model_gemm': dense, 'grad_gemm': tri, 'total': 2 * tri + dense,
            'dense_convention': 3 * dense}


def fr_setup(eng, vb, d, n_rows, rank_row_offset, ring, slot0, seed=2):
    rng = np.random.RandomState(seed)
    A = rng.randn(d, d)
    model = vb.CorrelatedGaussianModel(rng.randn(d), covariance=A @ A.T / d + np.eye(d))
    eng.set_model(model.device_spec())
    fr = vb.FullRankGaussian(d)
    L = np.exp(-1.0) * np.eye(d) + 0.01 * np.tril(np.random.RandomState(3).randn(d, d))
    theta = fr.pack(np.zeros(d), L)
    eng.fullrank_set_theta(theta, d)
    for s in range(ring):
        eng.noise_generate(slot0 + s, n_rows, d, seed=seed, stream=s, row_offset=rank_row_offset)
    return model, theta


def timed_blocks(run_block, sync, group, steps, min_total_s=MIN_TIMED_S, max_blocks=200):
    """Time blocks of exactly `steps` steps, each bracketed by barrier + device sync on both sides and reduced
    with MAX over ranks; blocks are repeated until the timed total reaches `min_total_s` (the number of blocks is
    decided from the first block's all-reduced time, so every rank runs the same number).  Returns the list of
    per-block seconds."""
    times = []
    n_blocks = 1
    b = 0
    while b < n_blocks:
        sync()
        group.barrier()
        t0 = time.perf_counter()
        run_block(steps)
        sync()
        dt = time.perf_counter() - t0
        group.barrier()
        dt = group.allreduce_max(dt)
        times.append(dt)
        if b == 0:
            n_blocks = int(min(max_blocks, max(3, np.ceil(min_total_s / max(dt, 1e-9)))))
        b += 1
    return times


def fullrank_leg(eng, vb, _lib, group, d, steps, warmup, scaling='weak', slot0=40, ring=8, profile=True):
    """FullRankGaussian + ExclusiveKL on the correlated-Gaussian target with the parameter resident on the device."""
    world, rank = group.world, group.rank
    if scaling == 'strong':
        from viabel_amd.objectives import shard_rows
        lo, hi = shard_rows(N_MC, world, rank)
        n_rows, n_total, row_off = hi - lo, N_MC, lo
    else:
        n_rows, n_total, row_off = N_MC, N_MC * world, rank * N_MC
    model, theta = fr_setup(eng, vb, d, n_rows, row_off, ring, slot0)

    def run(k):
        for i in range(k):
            eng.elbo_grad_fullrank_enqueue(slot0 + i % ring, n_rows, d, n_total=n_total)

    # untimed clock ramp: the GPU's power state follows load with tens of milliseconds of lag; a FIXED number of
    # evaluations, because with a communicator every evaluation is a collective and every rank must issue the same
    # number of them (a wall-clock ramp can deadlock the ranks)
    run(600 if d >= 1024 else 1500)
    eng.sync()
    run(warmup)
    eng.sync()
    times = timed_blocks(run, eng.sync, group, steps)
    kern = {}
    if profile:
        # per-kernel durations: the same K steps once more with hipExtLaunchKernel start / stop events on the three
        # GEMMs.  Kept out of the timed region: a launch that carries events costs ~4.7 us more on this stack
        # (338.8 -> 353.0 us per evaluation with all three instrumented), the kernels themselves run the same.
        eng.profile_enable(True)
        for k in (_lib.PROF_FR_SAMPLE_GEMM, _lib.PROF_FR_MODEL_GEMM, _lib.PROF_FR_GRAD_GEMM):
            eng.profile_read(reset=True, kernel=k)
        group.barrier()
        for _ in range(max(1, min(len(times), 5))):
            run(steps)
        eng.sync()
        group.barrier()
        for name, k in (('sample_gemm', _lib.PROF_FR_SAMPLE_GEMM), ('model_gemm', _lib.PROF_FR_MODEL_GEMM),
                        ('grad_gemm', _lib.PROF_FR_GRAD_GEMM)):
            n_l, _, ms = eng.profile_read(reset=True, kernel=k)
            kern[name] = (n_l, 1e3 * ms / max(1, n_l))
        eng.profile_enable(False)
    value, grad = eng.fullrank_get(d)
    sec_per_step = statistics.median(times) / steps
    fl = fr_flops(n_rows, d)
    per_kernel = {}
    for name, (n_l, us) in kern.items():
        tf = fl[name] / (us * 1e-6) / 1e12 if us > 0 else 0.0
        per_kernel[name] = {'avg_kernel_us': us, 'launches_timed': n_l, 'flops_per_launch': fl[name],
                            'achieved': tf, 'frac': tf / FP64_MFMA_PEAK_TFLOPS}
    whole_tf = fl['total'] / sec_per_step / 1e12
    return {
        'sec_per_step': sec_per_step, 'block_seconds': times, 'n_rows': n_rows, 'n_total': n_total,
        '_model': model, '_theta': theta,
        'value': value, 'grad_norm': float(np.linalg.norm(grad)), 'per_kernel': per_kernel,
        'whole_evaluation': {'flops_executed': fl['total'], 'us_per_eval': 1e6 * sec_per_step, 'achieved': whole_tf,
                             'frac': whole_tf / FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                             'flops_dense_convention': fl['dense_convention'],
                             'note': 'executed flops count the triangles exactly (n d (d+1) for each triangular GEMM); '
                                     'the dense convention would count 6 n d^2'},
    }


def dependent_chain_leg(eng, vb, group, iters=60):
    """N > 1: what an OPTIMISER sees under the communicator.  The headline enqueues independent evaluations of one resident
    theta, so evaluation k's all-reduce runs beside evaluation k + 1's GEMMs -- an overlap no optimiser has
    (optimization.py:95-97: theta_{k+1} needs grad_k).  Here: RMSProp iterations of the device-resident loop (vb_fit: fresh
    Philox noise -> sharded evaluation -> all-reduce of the 4.2 MB sum vector -> step -> unpack), every iteration behind
    the previous one's collective; weak (N_mc per GPU fixed) and strong (N_mc global fixed) sharding.  Collective: every
    rank runs it; max over ranks."""
    from viabel_amd.optimization import RMSProp
    d, world = FR_D, group.world
    rng = np.random.RandomState(2)
    A = rng.randn(d, d)
    model = vb.CorrelatedGaussianModel(rng.randn(d), covariance=A @ A.T / d + np.eye(d))
    out = {'workload': 'RMSProp iterations of vb_fit under the communicator: FullRankGaussian(%d, rng=philox) + ExclusiveKL, '
                       'fresh noise and one all-reduce of %d doubles per iteration' % (d, 16 + d + d * (d + 1) // 2)}
    for scaling, n_global in (('weak', N_MC * world), ('strong', N_MC)):
        obj = vb.ExclusiveKL(vb.FullRankGaussian(d, rng='philox'), model, n_global)
        theta = obj.approx.init_param()
        ropt = RMSProp(0.001)
        obj.device_fit(10, theta, ropt._device_kind, ropt._device_hyper())

        def run(k):
            obj.device_fit(k, theta, ropt._device_kind, ropt._device_hyper())
        times = timed_blocks(run, eng.sync, group, iters, min_total_s=0.5, max_blocks=5)
        out[scaling] = {'us_per_iteration': 1e6 * statistics.median(times) / iters, 'n_mc_global': n_global,
                        'iterations_per_s': iters / statistics.median(times), 'blocks': len(times)}
    return out


# --------------------------------------------------------------------------------------------------------------
# sharded legs (N > 1: collective -- every rank runs them; times are max over ranks)
# --------------------------------------------------------------------------------------------------------------
def _c3_problem(vb, D=256):
    """The problem of c3_leg / tests/test_gpu_full_size.py: interior tempering eps, ESS on target."""
    rng = np.random.RandomState(33)
    mean = 0.3 * rng.randn(D)
    sd = np.exp(0.5 + 0.02 * rng.randn(D))
    prior = np.concatenate([np.zeros(D), 0.5 * np.ones(D)])
    A = rng.randn(D, D)
    Sigma = np.e * np.eye(D) + 0.04 * (A @ A.T / D - np.eye(D))
    L = np.linalg.cholesky(Sigma)
    Lf = L.copy()
    Lf[np.diag_indices(D)] = np.log(np.diag(L))
    theta = np.concatenate([0.02 * rng.randn(D), Lf[np.tril_indices(D)]])
    return vb.GaussianModel(mean, sd), prior, theta


def sharded_c3_leg(eng, vb, group, calls=20):
    """BASELINE configs[3] AS IT IS STATED: MultivariateT(256, df=100) + DISInclusiveKL, N_mc = 16 384, the MC axis sharded
    over the ranks -- on the device-resident route (round 6: it used to refuse more than one rank).  Per blocking call:
    every rank samples and scores its rows, ONE device all-gather of [log p | log q | log prior] (3 N doubles), bisection /
    multinomial draw redundantly on the whole vectors, per-rank weighted Gram product, ONE all-reduce of
    [16 + D + D x D] doubles, the D^3 chain rule redundantly.  strong: N_mc = 16 384 global; weak: 16 384 per GPU."""
    D, df, world = 256, 100, group.world
    model, prior, theta = _c3_problem(vb, D)
    ld = (D + 15) // 16 * 16
    out = {'workload': 'BASELINE configs[3] sharded: MultivariateT(256, df=100) + DISInclusiveKL, state refresh every call, '
                       'rng=philox, MC axis over %d ranks; max over ranks of blocks of %d blocking calls' % (world, calls),
           'allreduce_doubles_per_call': 16 + ld + D * ld}
    np.random.seed(5)
    for scaling, N in (('strong', 16384), ('weak', 16384 * world)):
        leg = {'n_mc_global': N, 'n_mc_per_gpu': N // world, 'allgather_doubles_per_call': 3 * N}
        for mode, resample, n_warm in (('weighted', False, 10), ('resampling', True, 10), ('parity_mode_weighted', False, 3)):
            if mode.startswith('parity') and scaling == 'weak':
                continue
            approx = vb.MultivariateT(D, df, seed=1, rng='numpy' if mode.startswith('parity') else 'philox')
            obj = vb.DISInclusiveKL(approx, model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                                    temper_prior_params=prior, use_resampling=resample)
            last = {}
            if os.environ.get('VB_BENCH_TRACE'):
                sys.stderr.write('rank %d: c3 %s %s\n' % (group.rank, scaling, mode))

            def run(k):
                for _ in range(k):
                    last['vg'] = obj(theta)
            run(n_warm)
            times = timed_blocks(run, eng.sync, group, calls if n_warm > 3 else max(3, calls // 4), min_total_s=0.1, max_blocks=5)
            per = statistics.median(times) / (calls if n_warm > 3 else max(3, calls // 4))
            v, g = last['vg']
            leg[mode] = {'ms_per_call': 1e3 * per, 'calls_per_s': 1.0 / per, 'blocks': len(times), 'eps': float(obj._eps),
                         'ess': float(obj._ess), 'value': float(v), 'grad_norm': float(np.linalg.norm(g))}
        out[scaling] = leg
    return out


def sharded_c4_leg(eng, vb, group, steps=10):
    """BASELINE configs[4] sharded: MFGaussian + ExclusiveKL on Bayesian logistic regression, D = 2000, n_data = 8192,
    N_mc = 8192 GLOBAL (strong scaling: 8192 / world rows per GPU), fresh Philox noise per evaluation; one all-reduce of
    the mean-field sum vector per evaluation.  Blocking objective calls, and RMSProp iterations of the device-resident
    loop (vb_fit) that FASO / RAABBVI run between convergence checks."""
    from viabel_amd.optimization import RMSProp
    D, n_data, N = 2000, 8192, 8192
    rng = np.random.RandomState(4)
    X = rng.randn(n_data, D) / np.sqrt(D)
    beta = rng.randn(D)
    y = (rng.rand(n_data) < 1 / (1 + np.exp(-X @ beta))).astype(float)
    model = vb.LogisticRegressionModel(X, y, 10.0)
    obj = vb.ExclusiveKL(vb.MFGaussian(D, rng='philox'), model, N)
    theta = np.concatenate([np.zeros(D), -2 * np.ones(D)])
    last = {}

    def run(k):
        for _ in range(k):
            last['vg'] = obj(theta)
    run(3)
    times = timed_blocks(run, eng.sync, group, steps, min_total_s=0.1, max_blocks=5)
    dt = statistics.median(times) / steps
    ropt = RMSProp(0.02)
    obj.device_fit(5, theta, ropt._device_kind, ropt._device_hyper())

    def loop(k):
        obj.device_fit(k, theta, ropt._device_kind, ropt._device_hyper())
    ltimes = timed_blocks(loop, eng.sync, group, steps, min_total_s=0.1, max_blocks=5)
    v, g = last['vg']
    flops = 2 * 2.0 * N * n_data * D
    return {'workload': 'BASELINE configs[4] sharded: MFGaussian + ExclusiveKL, logistic regression D=2000, n_data=8192, '
                        'N_mc=8192 global over %d ranks (strong), rng=philox' % group.world,
            'n_mc_global': N, 'n_mc_per_gpu': N // group.world, 'allreduce_doubles_per_evaluation': 2 * D + 12,
            'ms_per_eval': 1e3 * dt, 'evals_per_s': 1.0 / dt,
            'device_loop_ms_per_iteration': 1e3 * statistics.median(ltimes) / steps,
            'value': float(v), 'grad_norm': float(np.linalg.norm(g)),
            'whole_job_tflops': flops / dt / 1e12}


def sharded_c1_leg(eng, vb, group, iters=400):
    """BASELINE configs[1] as an optimiser sees it under the communicator: RMSProp iterations of vb_fit with
    MFGaussian(1024, rng=philox) + ExclusiveKL on the funnel -- noise in registers, one streaming pass, ONE all-reduce of
    2 060 doubles (16 KB: the latency-bound collective SURVEY 5 warns about), finalize + step -- every iteration behind
    the previous one's collective.  strong: N_mc = 4096 global; weak: 4096 per GPU."""
    from viabel_amd.optimization import RMSProp
    d, world = D1, group.world
    model = vb.FunnelModel(d)
    theta = np.concatenate([np.zeros(d), -np.ones(d)])
    out = {'workload': 'RMSProp iterations of vb_fit under the communicator: MFGaussian(%d, rng=philox) + ExclusiveKL, funnel, '
                       'one all-reduce of %d doubles per iteration' % (d, 2 * d + 12)}
    for scaling, n_global in (('strong', N_MC), ('weak', N_MC * world)):
        obj = vb.ExclusiveKL(vb.MFGaussian(d, rng='philox'), model, n_global)
        ropt = RMSProp(0.01)
        obj.device_fit(50, theta, ropt._device_kind, ropt._device_hyper())

        def run(k):
            obj.device_fit(k, theta, ropt._device_kind, ropt._device_hyper())
        times = timed_blocks(run, eng.sync, group, iters, min_total_s=0.2, max_blocks=5)
        out[scaling] = {'us_per_iteration': 1e6 * statistics.median(times) / iters, 'n_mc_global': n_global,
                        'iterations_per_s': iters / statistics.median(times), 'blocks': len(times)}
    return out


def transport_legs(eng, vb, group):
    """The legs that depend on the TRANSPORT alone: the dependent full-rank chain, the 4.2-MB collective by itself and the
    mean-field chain with its 16-KB collective.  Run under the job's own transport by the ranks themselves and, for the
    other transport, by the children of `second_transport_probe`."""
    n_sum = 16 + FR_D + FR_D * (FR_D + 1) // 2
    chain = dependent_chain_leg(eng, vb, group)
    group.barrier()
    big = group.allreduce_max(eng.comm_allreduce_time(n_sum, warm=5, reps=30))
    group.barrier()
    small = group.allreduce_max(eng.comm_allreduce_time(2 * D1 + 12, warm=10, reps=200))
    return {'dependent_chain': chain,
            'allreduce_us': {'us_per_allreduce': big, 'doubles': n_sum,
                             'note': 'sum all-reduce of the full-rank sum vector alone, back to back on one stream between HIP '
                                     'events (vb_comm_allreduce_time), max over ranks'},
            'allreduce_small_us': {'us_per_allreduce': small, 'doubles': 2 * D1 + 12,
                                   'note': 'the mean-field sum vector (16 KB): latency-bound'},
            'c1_meanfield_chain': sharded_c1_leg(eng, vb, group)}


def prewarm_shared_device(eng, vb, distributed, which=('transport', 'c3', 'c4')):
    """Functional N > 1 runs on ONE GPU (VB_BENCH_TRANSPORT=host|ipc on a box with fewer GPUs than ranks) only: every leg
    once WITHOUT a communicator before it is attached.  Measured in round 6 (tools/r6_ipc_first_call_probe.py): when a
    process sets up device queues / streams / copy engines for the first time (the first call of a new code path) while
    the OTHER process's collective kernel is spinning on the same GPU, that call stalls for tens of seconds -- 42 s seen --
    or until the spinning kernel gives up (VB_IPC_TIMEOUT_S), and the give-up poisons the communicator: two processes
    cannot always make progress against each other on one device.  With its own GPU per rank (the case the transport
    exists for) nothing of the kind can happen; here the first-time set-up is simply done before anything can spin."""
    solo = distributed.SocketGroup(0, 1)
    if 'transport' in which:
        dependent_chain_leg(eng, vb, solo, iters=3)
        sharded_c1_leg(eng, vb, solo, iters=20)
        eng.comm_allreduce_time(16 + FR_D + FR_D * (FR_D + 1) // 2, warm=1, reps=1)
    if 'c3' in which:
        sharded_c3_leg(eng, vb, solo, calls=2)
    if 'c4' in which:
        sharded_c4_leg(eng, vb, solo, steps=1)
    eng.sync()


def run_probe_child(argv, env, timeout_s, want_stdout):
    """One child process of a rank (its own session; never an exec of this process): returns (exit code or 124 on
    timeout, its stdout lines, the tail of its stderr).  A child that is still running at the deadline has its whole
    process group killed."""
    import signal
    import subprocess
    import tempfile
    with tempfile.TemporaryFile(mode='w+') as err:
        p = subprocess.Popen(argv, env=env, start_new_session=True, text=True,
                             stdout=subprocess.PIPE if want_stdout else subprocess.DEVNULL, stderr=err)
        try:
            out, _ = p.communicate(timeout=timeout_s)
            rc, lines = p.returncode, (out or '').splitlines()
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
            try:
                p.communicate(timeout=10)
            except Exception:
                pass
            rc, lines = 124, []
        err.seek(0)
        tail = [ln for ln in err.read().splitlines() if ln.strip() and 'RCCL version' not in ln and 'version  :' not in ln]
        return rc, lines, ' | '.join(tail[-3:])[-400:]


def second_transport_probe(group, transport, timeout_s=240.0, child_argv=None):
    """`transports[<the other transport>]` of an N > 1 line WITHOUT putting the line at risk: every rank starts ONE child
    process (rank / world / device as its own, a control port and a job token of their own) that attaches the other
    transport, runs `transport_legs` and -- rank 0's child -- prints them as one JSON line.  The xGMI-native IPC
    transport has never crossed a device boundary on this pool: whatever it does on a real node -- a refused
    handle, a give-up of its device-side waits, a fault -- ends a CHILD; the parent ranks report the error text instead
    of the numbers and the headline line is printed regardless.  Collective: every rank calls this; rank 0 gets the dict."""
    env = dict(os.environ)
    base_port = int(env.get('VIABEL_AMD_CONTROL_PORT', int(env.get('MASTER_PORT', '29500')) + 23))
    env['VIABEL_AMD_CONTROL_PORT'] = str(base_port + 41)
    env['VIABEL_AMD_JOB_ID'] = env.get('VIABEL_AMD_JOB_ID', '') + '-probe-' + transport
    env['VB_BENCH_TRANSPORT'] = transport
    argv = child_argv or [sys.executable, os.path.abspath(__file__), '--gpus', str(group.world), '--probe-transport', transport]
    group.barrier()
    rc, lines, err_tail = run_probe_child(argv, env, timeout_s, want_stdout=group.rank == 0)
    worst = group.allreduce_max(float(rc if rc >= 0 else 128 - rc))
    if group.rank != 0:
        return None
    result = None
    for line in lines:
        try:
            obj = json.loads(line)
            if isinstance(obj, dict) and 'dependent_chain' in obj:
                result = obj
        except ValueError:
            pass
    if worst != 0 or result is None:
        return {'error': 'probe children of the %s transport: worst exit code %d over the ranks%s -- the numbers of this '
                         'transport are missing, nothing else is affected' % (transport, int(worst), '' if result is not None
                                                                              else ', no result line'),
                'rank0_child_stderr_tail': err_tail}
    return result


def probe_main(args):
    """A child of `second_transport_probe`: this rank's share of `transport_legs` under the named transport."""
    from viabel_amd import _lib, distributed
    import viabel_amd as vb
    world = int(os.environ['WORLD_SIZE'])
    group = distributed.SocketGroup.from_env()
    shared = _lib.device_count() < world
    eng = _lib.Engine(0) if shared else _lib.default_engine()
    _lib.set_default_engine(eng)
    if shared:
        os.environ.setdefault('VB_IPC_TIMEOUT_S', '150')      # (a stall between two processes on one GPU resolves itself; see prewarm_shared_device)
        prewarm_shared_device(eng, vb, distributed, which=('transport',))
        group.barrier()
    distributed.attach(eng, group, transport=args.probe_transport)
    legs = transport_legs(eng, vb, group)
    if group.rank == 0:
        legs['transport'] = args.probe_transport + (' (all ranks on device 0: a functional run)' if shared else '')
        print(json.dumps(legs), flush=True)
    group.barrier()
    group.close()


# --------------------------------------------------------------------------------------------------------------
# secondary legs (1 GPU, rank 0)
# --------------------------------------------------------------------------------------------------------------
def meanfield_leg(eng, vb, _lib, steps=2000, warmup=200, batch=32, ring_total=16):
    """BASELINE configs[1]: MFGaussian + ExclusiveKL, D=1024 funnel, N_mc=4096; HBM-bound streaming kernel.
    `evals_per_s_batched` shares one launch of each kernel between 32 independent evaluations (own noise matrix,
    own theta, own result); `sync_call_evals_per_s` is the one-evaluation blocking call an optimiser loop issues."""
    d = D1
    algo_bytes = N_MC * d * 8 + 4 * d * 8 + 8
    model = vb.FunnelModel(d)
    eng.set_model(model.device_spec())
    theta = np.concatenate([np.zeros(d), -np.ones(d)])       # mu = 0, log sigma = -1
    ring = min(max(batch, ring_total), _lib.MAX_SLOTS - 24)
    for s in range(ring):
        eng.noise_generate(s, N_MC, d, seed=1, stream=s)
    fam = _lib.FAMILY_MF_GAUSSIAN
    thetas = np.tile(theta, (batch, 1))
    n_rsets = _lib.MAX_SLOTS // batch

    def run(k):
        done, call = 0, 0
        while done < k:
            b = min(batch, k - done)
            slots = [(call * batch + i) % ring for i in range(b)]
            rslots = [(call % n_rsets) * batch + i for i in range(b)]
            eng.elbo_grad_meanfield_batch_async(slots, N_MC, d, thetas[:b], fam, rslots)
            done += b
            call += 1

    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.3:
        run(4 * batch)
        eng.sync()
    run(warmup)
    eng.sync()
    eng.profile_enable(True)
    eng.profile_read(reset=True)
    t0 = time.perf_counter()
    run(steps)
    eng.sync()
    elapsed = time.perf_counter() - t0
    launches, evals_timed, kernel_ms = eng.profile_read(reset=True)
    eng.profile_enable(False)
    n_sync = 500
    t2 = time.perf_counter()
    for i in range(n_sync):
        eng.elbo_grad_meanfield(i % ring, N_MC, d, theta, fam)
    sync_rate = n_sync / (time.perf_counter() - t2)
    t3 = time.perf_counter()
    for i in range(n_sync):
        eng.elbo_grad_meanfield_philox(0, N_MC, d, theta, fam, 1, 1000 + i)
    fresh_rate = n_sync / (time.perf_counter() - t3)
    # the reference-identical mode (default rng='numpy'): every call draws RandomState(seed).randn(N, D) /
    # .standard_t(df, (N, D)) -- approximations.py:216, :273-274 -- on the device, bit for bit (vb_legacy_dev.hip,
    # vb_legacy_gamma.hip), then evaluates
    parity_mode = {}
    for name, approx_np in (('mf_gaussian', vb.MFGaussian(d)), ('mf_student_t_df7', vb.MFStudentT(d, 7))):
        obj = vb.ExclusiveKL(approx_np, model, N_MC)
        for _ in range(3):
            obj(theta)
        blocks = []
        for _ in range(3):
            t4 = time.perf_counter()
            for _ in range(10):
                obj(theta)
            blocks.append((time.perf_counter() - t4) / 10)
        parity_mode[name + '_ms_per_call'] = 1e3 * statistics.median(blocks)
    parity_mode['note'] = ("blocking objective(theta) with the reference's own noise stream drawn per call on the device "
                           '(4.2 M values); round 4: 14.1 ms / ~160 ms (host draw + upload)')
    kernel_us = 1e3 * kernel_ms / max(1, launches)
    bytes_per_launch = algo_bytes * evals_timed / max(1, launches)
    achieved = bytes_per_launch / (kernel_us * 1e-6) / 1e9
    # parity on the same noise
    from oracle import families as ofam, models as omod, objectives as oobj
    noise = eng.noise_get_host(0, N_MC, d)
    dv, dg = eng.elbo_grad_meanfield(0, N_MC, d, theta, fam)
    ov, og = oobj.exclusive_kl(ofam.MFGaussian(d), omod.Funnel(d), theta, noise)
    return {
        'workload': 'BASELINE configs[1]: MFGaussian + ExclusiveKL, D=1024 funnel, N_mc=4096, fp64',
        'evals_per_s_batched': steps / elapsed, 'evals_per_launch': batch,
        'sync_call_evals_per_s': sync_rate, 'fresh_noise_sync_call_evals_per_s': fresh_rate,
        'parity_mode': parity_mode,
        'roofline': {'bound': 'hbm', 'kernel': 'mf_accum_kernel', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                     'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'avg_kernel_us': kernel_us,
                     'launches_timed': launches, 'algorithmic_bytes_per_launch': bytes_per_launch,
                     'traffic': MF_ACCUM_HBM_BYTES if batch == 32 else None,
                     'traffic_source': 'profiles/r06_meanfield_c1_pmc_hbm.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate '
                                       'passes of tools/mf_stream_bench.py, this launch shape, kernel unchanged since): 1 086.4 MB '
                                       'fetched + 8.6 MB written per 32-evaluation launch = 1.019 x the algorithmic 1 074.8 MB'},
        'parity': {'rel_elbo_err': abs(dv - ov) / abs(ov),
                   'rel_grad_err': float(np.max(np.abs(dg - og)) / np.max(np.abs(og)))},
    }, theta


def fit_leg(vb, theta, iters=1500):
    """A whole RMSProp fit at the C1 shape (fresh Philox noise every iteration) through the host loop (one
    blocking objective call + numpy step per iteration, optimization.py:91-112) and through the device-resident
    loop (vb_fit); the two trajectories are the same bit for bit."""
    from viabel_amd.optimization import RMSProp
    out = {'workload': 'RMSProp(0.01), MFGaussian(rng=philox) + ExclusiveKL, D=1024 funnel, N_mc=4096, %d iterations'
                       % iters}
    hist = {}
    for mode, on_device in (('host_loop', False), ('device_loop', True)):
        obj = vb.ExclusiveKL(vb.MFGaussian(D1, rng='philox'), vb.FunnelModel(D1), N_MC)
        opt = RMSProp(0.01)
        opt.optimize(200, obj, theta, on_device=on_device)
        t0 = time.perf_counter()
        hist[mode] = opt.optimize(iters, obj, theta, on_device=on_device)['value_history']
        out[mode + '_us_per_iteration'] = 1e6 * (time.perf_counter() - t0) / iters
    out['trajectories_identical'] = bool(np.array_equal(hist['host_loop'], hist['device_loop']))
    # What bounds an iteration of the device loop: its streaming kernel generates the noise in registers and is
    # VALU-bound (bytes are irrelevant: it fetches 0.5 MB).  The counters are from separate rocprofv3 --pmc passes
    # (profiles/r03_meanfield_gen_pmc_valu.txt) and cannot be collected in this process; the floor below prices the
    # measured instruction count at the VALU issue rate of the chip.
    n_normals = D1 * N_MC
    wave_instr = 6678165                      # SQ_INSTS_VALU per launch of mf_accum<GEN> at this shape (round 2: 8 425 664)
    simds, clock_ghz = 1024, 2.33             # 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE / duration under the counters
    # of the 102 per normal 5 are quarter-rate v_mad_u64_u32 (Philox, four normals per call) and ~2 quarter-rate
    # v_rcp_f64 / v_sqrt_f64 of the Box-Muller transform
    cycles_per_instr = (95 * 4 + 7 * 16) / 102.0
    floor_us = wave_instr / simds * cycles_per_instr / (clock_ghz * 1e3)
    out['valu_roof'] = {
        'bound': 'VALU issue rate of the noise-generating streaming kernel (mf_accum<GEN>)',
        'valu_instructions_per_normal': round(wave_instr * 64.0 / n_normals, 1),
        'kernel_us': 18.4, 'issue_floor_us': round(floor_us, 1), 'kernel_frac_of_floor': round(floor_us / 18.4, 2),
        'iteration_frac_of_floor': round(floor_us / out['device_loop_us_per_iteration'], 2),
        'rest_of_iteration': 'mf_finalize 5.6 us (4.4 us of latency inside the kernel) + ~2 us of dispatch gaps',
        'source': 'profiles/r03_meanfield_gen_pmc_valu.txt, profiles/r04_fit_loop_kernel_stats.txt',
    }
    return out


def c3_leg(vb, calls=30):
    """BASELINE configs[3]: MultivariateT(256, df=100) + DISInclusiveKL, N_mc = 16 384 (one GPU), throughput mode
    (rng='philox': normals and chi-square draws on the device), state refresh on every call, with and without
    resampling.  The problem is the one of tests/test_gpu_full_size.py (interior tempering eps, ESS on target)."""
    D, N, df = 256, 16384, 100
    rng = np.random.RandomState(33)
    mean = 0.3 * rng.randn(D)
    sd = np.exp(0.5 + 0.02 * rng.randn(D))
    prior = np.concatenate([np.zeros(D), 0.5 * np.ones(D)])
    A = rng.randn(D, D)
    Sigma = np.e * np.eye(D) + 0.04 * (A @ A.T / D - np.eye(D))
    approx = vb.MultivariateT(D, df, seed=1, rng='philox')
    L = np.linalg.cholesky(Sigma)
    Lf = L.copy()
    Lf[np.diag_indices(D)] = np.log(np.diag(L))
    theta = np.concatenate([0.02 * rng.randn(D), Lf[np.tril_indices(D)]])
    model = vb.GaussianModel(mean, sd)
    out = {'workload': 'BASELINE configs[3]: MultivariateT(256, df=100) + DISInclusiveKL, N_mc=16384, ess_target=2048, '
                       'state refresh every call, rng=philox, one GPU'}
    np.random.seed(5)
    # "... with PSIS reweighting" (configs[3]): psis_smooth=True Pareto-smooths the tempered weights of every refresh
    # (viabel/_psis.py:113-209, on the device-resident weights: vb_dis_psis_mvt) before they weight the score
    for resample, psis in ((False, False), (True, False), (False, True), (True, True)):
        obj = vb.DISInclusiveKL(approx, model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                                temper_prior_params=prior, use_resampling=resample, psis_smooth=psis)
        for _ in range(20):
            obj(theta)
        blocks = []                      # three blocks of `calls` calls, the median block is reported (a call is ~45
        for _ in range(3):               # dependent launches: a busy host moves a single block by tens of per cent)
            t0 = time.perf_counter()
            for _ in range(calls):
                v, g = obj(theta)
            blocks.append((time.perf_counter() - t0) / calls)
        dt = statistics.median(blocks)
        key = ('resampling' if resample else 'weighted') + ('_psis' if psis else '')
        out[key] = {
            'ms_per_call': 1e3 * dt, 'calls_per_s': 1.0 / dt, 'block_ms': [1e3 * b for b in blocks],
            'eps': float(obj._eps), 'ess': float(obj._ess),
            'value': float(v), 'grad_norm': float(np.linalg.norm(g))}
        if psis:
            out[key]['khat'] = float(obj._khat)
    # the reference-identical mode (default rng='numpy'): RandomState(seed).chisquare then randn (approximations.py:345-347)
    # generated on the device bit for bit, the symmetric root of :348, the global numpy generator's resampling draw (:408)
    approx_np = vb.MultivariateT(D, df, seed=1)
    for resample in (False, True):
        obj = vb.DISInclusiveKL(approx_np, model, N, ess_target=N // 8, temper_prior=vb.MFGaussian(D),
                                temper_prior_params=prior, use_resampling=resample)
        for _ in range(5):
            obj(theta)
        blocks = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                v, g = obj(theta)
            blocks.append((time.perf_counter() - t0) / 10)
        dt = statistics.median(blocks)
        out['parity_mode_' + ('resampling' if resample else 'weighted')] = {
            'ms_per_call': 1e3 * dt, 'block_ms': [1e3 * b for b in blocks], 'eps': float(obj._eps), 'ess': float(obj._ess),
            'value': float(v), 'grad_norm': float(np.linalg.norm(g)),
            'note': "rng='numpy': the reference's own noise stream (16 384 chi-square draws + 4.2 M normals per call, on the "
                    'device) and its symmetric matrix root'}
    # executed work of one refresh + gradient in this mode (late round 5): the sample GEMM through L' (triangular) and the
    # weighted Gram product of the residuals (lower tiles) -- two half products; the U = E' L^-1 product of rounds 2-4 is
    # gone (the chain rule takes L^-T once, in a D x D x D product), and the residuals of freshly drawn samples are the
    # scaled noise (no product)
    flops = 2.0 * N * D * (D + 1) + 2.0 * D * D * D
    out['flops_executed_per_call'] = flops
    out['note'] = ('19 kernels per call (15 on the critical path: the triangular inverse runs on a side stream), everything '
                   'including the O(D^3) factor algebra on the device; five of them are the tempering bisection (50 levels '
                   'walked along two predicted paths: ~38 us), PSIS smoothing runs on 16 workgroups (43 us), the two N x D x D '
                   'half products take 64 us for %.1f GFLOP, the chain rule is one launch over the lower tiles (14 us): the '
                   'call is a chain of short dependent kernels (~187 us) plus ~30 us of host turn-around, not matrix-pipe time; '
                   'timeline: profiles/r06_c3_timeline.txt, per kernel: profiles/r06_c3_kernel_stats.txt; the parity_mode_* '
                   "calls run the NEXT call's numpy-stream draws beside this call's kernels (look-ahead, DESIGN 4.11)" % (flops / 1e9))
    return out


def mvt_ekl_leg(vb, calls=50):
    """MultivariateT(256, df=100) + ExclusiveKL, N_mc = 16 384, rng='philox' (throughput mode): device chi-square draws and
    normals, Cholesky sampling, the gradient in the free-Cholesky layout straight from the dense-family pipeline --
    no matrix square root and no Sylvester solve.  The family / objective pair of the reference's robust-regression
    notebook at the configs[3] shape.  Blocking objective(theta) calls with fresh noise each."""
    D, N, df = 256, 16384, 100.0
    rng = np.random.RandomState(33)
    mean = 0.3 * rng.randn(D)
    sd = np.exp(0.5 + 0.02 * rng.randn(D))
    A = rng.randn(D, D)
    Sigma = np.e * np.eye(D) + 0.04 * (A @ A.T / D - np.eye(D))
    L = np.linalg.cholesky(Sigma)
    Lf = L.copy()
    Lf[np.diag_indices(D)] = np.log(np.diag(L))
    theta = np.concatenate([0.02 * rng.randn(D), Lf[np.tril_indices(D)]])
    out = {'workload': 'MultivariateT(256, df=100) + ExclusiveKL (entropy form), N_mc=16384, rng=philox, diagonal-Gaussian '
                       'target, blocking objective(theta) calls with fresh noise'}
    for mode in ('philox', 'numpy'):
        obj = vb.ExclusiveKL(vb.MultivariateT(D, df, seed=1, rng=mode), vb.GaussianModel(mean, sd), N)
        n_calls = calls if mode == 'philox' else 10
        for _ in range(3):
            obj(theta)
        t0 = time.perf_counter()
        for _ in range(n_calls):
            v, g = obj(theta)
        dt = (time.perf_counter() - t0) / n_calls
        key = 'throughput_mode' if mode == 'philox' else 'parity_mode'
        out[key] = {'ms_per_call': 1e3 * dt, 'value': float(v), 'grad_norm': float(np.linalg.norm(g))}
    # the path-derivative form of the same call (objectives.py:156-159), resident since late round 5; the host-root route it
    # replaces is timed beside it through the route's dimension gate
    from viabel_amd import objectives as _vobj
    pd = {}
    for name, gate in (('resident', None), ('host_root_route', 10 ** 6)):
        keep = _vobj._HOST_ROOT_MAX_DIM, _vobj._RESIDENT_GATE
        try:
            if gate is not None:
                _vobj._HOST_ROOT_MAX_DIM = _vobj._RESIDENT_GATE = gate
            obj = vb.ExclusiveKL(vb.MultivariateT(D, df, seed=1), vb.GaussianModel(mean, sd), N, use_path_deriv=True)
            for _ in range(2):
                obj(theta)
            t0 = time.perf_counter()
            for _ in range(6):
                v, g = obj(theta)
            pd[name + '_ms_per_call'] = 1e3 * (time.perf_counter() - t0) / 6
        finally:
            _vobj._HOST_ROOT_MAX_DIM, _vobj._RESIDENT_GATE = keep
    pd['value'] = float(v)
    out['parity_mode_path_deriv'] = pd
    out['parity_mode']['note'] = ("rng='numpy': the reference's chi-square + normal streams on the device, its symmetric root "
                                  '(approximations.py:348) and the root\'s Frechet derivative by device iterations, chain rule '
                                  'on the device (vb_elbo_grad_mvt_symroot); rounds 3-4: host root, 5.1 ms')
    flops = 2.0 * float(N) * D * (D + 1)          # sampling and gradient GEMMs (triangles exact)
    tf = flops / (out['throughput_mode']['ms_per_call'] * 1e-3) / 1e12
    out['roofline'] = {'bound': 'mfma', 'flops_executed': flops, 'achieved': tf, 'peak': FP64_MFMA_PEAK_TFLOPS,
                       'unit': 'TFLOP/s', 'frac': tf / FP64_MFMA_PEAK_TFLOPS,
                       'note': 'whole blocking call incl. chi-square / normal generation (4.2 M normals) and the 264-KB '
                               'parameter upload; per-kernel times: profiles/r04_mvt_ekl_kernel_stats.txt'}
    return out


SOURCE_LEG_SRC = r"""
#define VB_LOG_DENSITY_PARTS 8
// robust regression: y_i ~ StudentT(nu, x_i' z, s), z ~ N(0, tau^2 I); params = [n, nu, s, tau | X (n x d) | y (n)]
__device__ double vb_log_density_part(const double* z, int d, const double* p, double* g, int part, int n_parts) {
  const int n = (int)p[0];
  const double nu = p[1], s = p[2], tau = p[3];
  const double* X = p + 4;
  const double* y = X + (long long)n * d;
  double f = 0.0;
  if (part == 0)
    for (int j = 0; j < d; ++j) {
      f -= 0.5 * z[j] * z[j] / (tau * tau);
      if (g) g[j] = -z[j] / (tau * tau);
    }
  for (int i = part; i < n; i += n_parts) {
    double eta = 0.0;
    for (int j = 0; j < d; ++j) eta += X[(long long)i * d + j] * z[j];
    const double r = y[i] - eta, q = 1.0 + r * r / (nu * s * s);
    f -= 0.5 * (nu + 1.0) * log(q);
    if (g) {
      const double c = (nu + 1.0) * r / (nu * s * s * q);
      for (int j = 0; j < d; ++j) g[j] += c * X[(long long)i * d + j];
    }
  }
  return f;
}
"""


SOURCE_LEG_AUTO_SRC = r"""
template <class T>
__device__ T vb_log_density(vb::vec<T> z, int d, const double* p) {
  const int n = (int)p[0];
  const double nu = p[1], s = p[2], tau = p[3];
  const double* X = p + 4;
  const double* y = X + (long long)n * d;
  T f = 0.0;
  for (int j = 0; j < d; ++j) f -= 0.5 * z[j] * z[j] / (tau * tau);
  for (int i = 0; i < n; ++i) {
    const T r = y[i] - vb::dot(X + (long long)i * d, z, d);
    f -= 0.5 * (nu + 1.0) * log(1.0 + r * r / (nu * s * s));
  }
  return f;
}
"""


def source_model_leg(vb, calls=100):
    """SURVEY 8(f) N4, the model adaptor: a log density handed over as HIP source (robust Student-t regression of the
    reference's docs, 64 coefficients, 512 observations), compiled with hiprtc and run inside ExclusiveKL; next to it
    a numpy (BLAS-backed) restatement of the same f and grad f of N samples on the host."""
    D, n_data, N = 64, 512, 4096
    rng = np.random.RandomState(3)
    X = rng.randn(n_data, D) / np.sqrt(D)
    y = X @ rng.randn(D) + 0.3 * rng.standard_t(3.0, size=n_data)
    nu, s, tau = 4.0, 0.5, 3.0
    model = vb.SourceModel(D, SOURCE_LEG_SRC, np.concatenate([[n_data, nu, s, tau], X.ravel(), y]))
    out = {'workload': 'SourceModel (HIP source via hiprtc, 8 threads per sample): robust regression D=64, n_data=512, '
                       'N_mc=4096, ExclusiveKL entropy form, blocking objective(theta) calls, rng=philox',
           'gradient_check': model.check_gradient(rng.randn(4, D))}
    for name, fam in (('mf_gaussian', vb.MFGaussian(D, rng='philox')), ('fullrank_gaussian', vb.FullRankGaussian(D, rng='philox'))):
        obj = vb.ExclusiveKL(fam, model, N)
        theta = fam.init_param()
        if name == 'mf_gaussian':
            theta[D:] = -1.0
        for _ in range(5):
            obj(theta)
        t0 = time.perf_counter()
        for _ in range(calls):
            v, g = obj(theta)
        out[name] = {'us_per_call': 1e6 * (time.perf_counter() - t0) / calls, 'value': float(v),
                     'grad_norm': float(np.linalg.norm(g))}
    # the same density with nothing but the density written down (grad='auto': forward-mode dual numbers on the device,
    # ceil(D / 8) threads per sample) -- what it costs next to the hand-written gradient
    # grad='auto': the density alone.  Twice: the linear predictor as vb::dot(row, z, d) (one operation of the dual
    # arithmetic, shared by the sample's threads) and as a plain loop over dual numbers
    loop_src = SOURCE_LEG_AUTO_SRC.replace('const T r = y[i] - vb::dot(X + (long long)i * d, z, d);',
                                           'T eta = 0.0;\n    for (int j = 0; j < d; ++j) eta += X[(long long)i * d + j] * z[j];\n'
                                           '    const T r = y[i] - eta;')
    assert 'vb::dot' not in loop_src
    xs = rng.randn(16, D)
    out['auto_gradient'] = {}
    for key, src in (('', SOURCE_LEG_AUTO_SRC), ('plain_loop_', loop_src)):
        auto = vb.SourceModel(D, src, model.params, grad='auto')
        out['auto_gradient'][key + 'max_rel_diff_vs_hand_written'] = float(
            np.max(np.abs(auto.grad(xs) - model.grad(xs))) / np.max(np.abs(model.grad(xs))))
        fam = vb.MFGaussian(D, rng='philox')
        obj = vb.ExclusiveKL(fam, auto, N)
        theta = fam.init_param()
        theta[D:] = -1.0
        for _ in range(3):
            obj(theta)
        t0 = time.perf_counter()
        for _ in range(20):
            obj(theta)
        out['auto_gradient'][key + 'mf_gaussian_us_per_call'] = 1e6 * (time.perf_counter() - t0) / 20
        out['auto_gradient'][key + 'cost_ratio_vs_hand_written_8_threads'] = (
            out['auto_gradient'][key + 'mf_gaussian_us_per_call'] / out['mf_gaussian']['us_per_call'])
    z = rng.randn(N, D)
    t0 = time.perf_counter()
    r = y[None, :] - z @ X.T
    q = 1.0 + r * r / (nu * s * s)
    f = -0.5 * np.sum(z * z, axis=1) / tau ** 2 - 0.5 * (nu + 1.0) * np.sum(np.log(q), axis=1)
    gmat = -z / tau ** 2 + ((nu + 1.0) * r / (nu * s * s * q)) @ X
    out['host_numpy_f_and_grad_ms'] = 1e3 * (time.perf_counter() - t0)
    out['host_check'] = float(np.max(np.abs(model.grad(z[:64]) - gmat[:64])) / np.max(np.abs(gmat[:64])))
    del f
    return out


def alpha_leg(vb, calls=30):
    """AlphaDivergence (objectives.py:443-463), the third objective of the hot path: blocking objective(theta) calls with
    fresh Philox noise at the C1 shape (mean field), the headline shape (dense Gaussian family; the call carries the
    525 824-entry parameter up and the gradient down) and the configs[3] shape (MultivariateT, throughput mode)."""
    out = {'workload': 'AlphaDivergence(alpha = 0.5), blocking objective(theta) calls, rng=philox, one GPU'}
    rng = np.random.RandomState(2)
    cases = (('mf_gaussian_funnel_d1024_n4096', vb.MFGaussian(1024, rng='philox'), vb.FunnelModel(1024), 4096),
             ('fullrank_funnel_d1024_n4096', vb.FullRankGaussian(1024, rng='philox'), vb.FunnelModel(1024), 4096),
             ('multivariate_t_gauss_diag_d256_n16384', vb.MultivariateT(256, 100, rng='philox'),
              vb.GaussianModel(0.1 * rng.randn(256), np.exp(0.1 * rng.randn(256))), 16384))
    for name, approx, model, n in cases:
        D = approx.dim
        if isinstance(approx, vb.FullRankGaussian):
            theta = approx.pack(np.zeros(D), np.exp(-1.0) * np.eye(D))
        else:
            theta = approx.init_param()
            if isinstance(approx, vb.MFGaussian):
                theta[D:] = -1.0
        res = {}
        # the same blocking call for ExclusiveKL on the same family / target / shape: the like-for-like reference (both
        # carry the parameter up and the gradient down and draw fresh noise)
        for key, obj in (('us_per_call', vb.AlphaDivergence(approx, model, n, 0.5)),
                         ('exclusive_kl_same_call_us', vb.ExclusiveKL(approx, model, n))):
            np.random.seed(1)
            for _ in range(10):
                obj(theta)
            blocks = []
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(calls):
                    v, g = obj(theta)
                blocks.append((time.perf_counter() - t0) / calls)
            res[key] = 1e6 * statistics.median(blocks)
            if key == 'us_per_call':
                res['value'], res['grad_norm'] = float(v), float(np.linalg.norm(g))
        res['ratio_to_exclusive_kl'] = res['us_per_call'] / res['exclusive_kl_same_call_us']
        out[name] = res
    # the t family in the reference-identical mode (rng='numpy', the call's fresh RandomState drawn on the device, the
    # symmetric root): resident since late round 5 (vb_alpha_grad_mvt_symroot); the host-root route through its gate
    from viabel_amd import objectives as _vobj
    mrng = np.random.RandomState(2)
    mean, sd = 0.1 * mrng.randn(256), np.exp(0.1 * mrng.randn(256))
    par = {}
    for name, gate in (('resident', None), ('host_root_route', 10 ** 6)):
        keep = _vobj._HOST_ROOT_MAX_DIM, _vobj._RESIDENT_GATE
        try:
            if gate is not None:
                _vobj._HOST_ROOT_MAX_DIM = _vobj._RESIDENT_GATE = gate
            approx = vb.MultivariateT(256, 100)
            obj = vb.AlphaDivergence(approx, vb.GaussianModel(mean, sd), 16384, 0.5)
            theta = approx.init_param()
            np.random.seed(1)
            for _ in range(2):
                obj(theta)
            t0 = time.perf_counter()
            for _ in range(6):
                v, g = obj(theta)
            par[name + '_ms_per_call'] = 1e3 * (time.perf_counter() - t0) / 6
        finally:
            _vobj._HOST_ROOT_MAX_DIM, _vobj._RESIDENT_GATE = keep
    out['multivariate_t_gauss_diag_d256_n16384']['parity_mode'] = par
    return out


def c4_leg(eng, vb, steps=20):
    """BASELINE configs[4]: MFGaussian + ExclusiveKL on Bayesian logistic regression, D=2000, n_data=8192,
    N_mc=8192 (one GPU), fresh Philox noise per evaluation: blocking objective calls, and RMSProp iterations of the
    device-resident loop that FASO / RAABBVI run between convergence checks.  MFMA-bound: eta = Z X' and
    G = R X are two dense n_mc x n_data x D products."""
    from viabel_amd.optimization import RMSProp
    D, n_data, N = 2000, 8192, 8192
    rng = np.random.RandomState(4)
    X = rng.randn(n_data, D) / np.sqrt(D)
    beta = rng.randn(D)
    y = (rng.rand(n_data) < 1 / (1 + np.exp(-X @ beta))).astype(float)
    model = vb.LogisticRegressionModel(X, y, 10.0)
    eng.set_model(model.device_spec())
    obj = vb.ExclusiveKL(vb.MFGaussian(D, rng='philox'), model, N)
    theta = np.concatenate([np.zeros(D), -2 * np.ones(D)])
    for _ in range(3):
        v, g = obj(theta)
    t0 = time.perf_counter()
    for _ in range(steps):
        v, g = obj(theta)
    dt = (time.perf_counter() - t0) / steps
    import contextlib
    import io
    with contextlib.redirect_stderr(io.StringIO()):
        opt = RMSProp(0.02)
        opt.optimize(5, obj, theta, on_device=True)
        t0 = time.perf_counter()
        opt.optimize(steps, obj, theta, on_device=True)
        dt_loop = (time.perf_counter() - t0) / steps
    flops = 2 * 2.0 * N * n_data * D
    tf = flops / dt / 1e12
    return {'workload': 'BASELINE configs[4]: MFGaussian + ExclusiveKL, logistic regression D=2000, n_data=8192, '
                        'N_mc=8192, rng=philox, one GPU',
            'ms_per_eval': 1e3 * dt, 'evals_per_s': 1.0 / dt, 'device_loop_ms_per_iteration': 1e3 * dt_loop,
            'value': float(v), 'grad_norm': float(np.linalg.norm(g)),
            'roofline': {'bound': 'mfma', 'flops_executed': flops, 'achieved': tf, 'peak': FP64_MFMA_PEAK_TFLOPS,
                         'unit': 'TFLOP/s', 'frac': tf / FP64_MFMA_PEAK_TFLOPS,
                         'note': 'whole blocking call (two dense GEMMs + noise generation + streaming pass)'}}


def fullrank_fit_leg(vb, iters=300):
    """What an optimiser sees at the headline shape: RMSProp iterations of FullRankGaussian(1024) + ExclusiveKL on the
    correlated-Gaussian target with FRESH Philox noise every iteration -- noise generation, the evaluation, the
    optimiser step and the unpack of the stepped parameter all inside the timed region -- through the device-resident
    loop (vb_fit, optimization.py:83-127 restated as one stream of launches) and through the host loop."""
    from viabel_amd.optimization import RMSProp
    d = FR_D
    rng = np.random.RandomState(2)
    A = rng.randn(d, d)
    model = vb.CorrelatedGaussianModel(rng.randn(d), covariance=A @ A.T / d + np.eye(d))
    fam = vb.FullRankGaussian(d, rng='philox')
    obj = vb.ExclusiveKL(fam, model, N_MC)
    theta = fam.init_param()
    out = {'workload': 'RMSProp(0.001), FullRankGaussian(%d, rng=philox) + ExclusiveKL, correlated-Gaussian target, '
                       'N_mc=%d, fresh noise every iteration' % (d, N_MC)}
    hist = {}
    n_host = max(20, iters // 10)
    for mode, on_device, n_it in (('optimize_device', True, iters), ('optimize_host', False, n_host)):
        # (warm-up of the timed size on the device, with an objective and optimiser of its own: the 1.26-GB iterate history on
        # both sides -- device rows, the host array's pages -- is allocated once per size, not per fit)
        RMSProp(0.001).optimize(n_it if on_device else n_host, vb.ExclusiveKL(vb.FullRankGaussian(d, rng='philox'), model, N_MC),
                                theta, on_device=on_device)
        obj = vb.ExclusiveKL(vb.FullRankGaussian(d, rng='philox'), model, N_MC)      # same Philox streams in both modes
        opt = RMSProp(0.001)
        t0 = time.perf_counter()
        res = opt.optimize(n_it, obj, theta, on_device=on_device)
        out[mode + '_us_per_iteration'] = 1e6 * (time.perf_counter() - t0) / n_it
        hist[mode] = np.asarray(res['value_history'])
    out['trajectories_identical'] = bool(np.array_equal(hist['optimize_device'][:n_host], hist['optimize_host'][:n_host]))
    out['note'] = ('optimize_*: RMSProp.optimize (optimization.py:83-127), which returns every iterate -- 4.2 MB per '
                   'iteration brought back to the host; device_loop: the same iterations through device_fit without the '
                   'iterate history (what FASO / RAABBVI run between two convergence checks, minus their window)')
    # the loop itself: vb_fit with no iterate history (noise generation, evaluation, step, unpack of the stepped parameter)
    obj = vb.ExclusiveKL(vb.FullRankGaussian(d, rng='philox'), model, N_MC)
    ropt = RMSProp(0.001)
    obj.device_fit(30, theta, ropt._device_kind, ropt._device_hyper())
    t0 = time.perf_counter()
    obj.device_fit(iters, theta, ropt._device_kind, ropt._device_hyper())
    out['device_loop_us_per_iteration'] = 1e6 * (time.perf_counter() - t0) / iters
    fl = fr_flops(N_MC, d)['total']
    tf = fl / (out['device_loop_us_per_iteration'] * 1e-6) / 1e12
    out['roofline'] = {'bound': 'mfma', 'flops_executed': fl, 'achieved': tf, 'peak': FP64_MFMA_PEAK_TFLOPS,
                       'unit': 'TFLOP/s', 'frac': tf / FP64_MFMA_PEAK_TFLOPS,
                       'note': 'device_loop iteration; per-kernel times: profiles/r04_fullrank_fit_kernel_stats.txt'}
    return out


def api_call_leg(eng, vb, calls=200):
    """The API-level call the headline leaves out: a BLOCKING
    ``ExclusiveKL(FullRankGaussian(1024, rng=...), CorrelatedGaussianModel, 4096)(theta) -> (value, grad)`` from a host
    parameter to a host gradient (objectives.py:32-44 as an optimiser calls it, optimization.py:95), with fresh noise per
    call: rng='philox' (device generator) and rng='numpy' (the reference's exact RandomState stream -- generated on the
    device since round 4, vb_legacy_rng_randn_device).  The pieces are timed separately on the same engine:
    parameter upload (4.2 MB H2D + unpack), noise, the evaluation's kernels, gradient download (4.2 MB D2H)."""
    d = FR_D
    rng = np.random.RandomState(2)
    A = rng.randn(d, d)
    model = vb.CorrelatedGaussianModel(rng.randn(d), covariance=A @ A.T / d + np.eye(d))
    out = {'workload': 'blocking objective(theta) -> (value, grad), FullRankGaussian(%d) + ExclusiveKL, N_mc=%d, '
                       'host parameter in, host gradient out, fresh noise per call' % (d, N_MC)}
    for kind, n_calls in (('philox', calls // 2), ('numpy', max(10, calls // 8))):
        fam = vb.FullRankGaussian(d, seed=1, rng=kind)
        obj = vb.ExclusiveKL(fam, model, N_MC)
        theta = fam.pack(np.zeros(d), np.exp(-1.0) * np.eye(d) + 0.01 * np.tril(np.random.RandomState(3).randn(d, d)))
        for _ in range(3):
            obj(theta)
        blocks = []      # the median of three blocks: a blocking call is host-latency-sensitive (the CPU baseline's BLAS
        for _ in range(3):      # threads may still be spinning down when this leg starts)
            t0 = time.perf_counter()
            for _ in range(n_calls):
                value, grad = obj(theta)
            blocks.append(1e6 * (time.perf_counter() - t0) / n_calls)
        out['%s_us_per_call' % kind] = statistics.median(blocks)
        out['%s_block_us' % kind] = blocks
        if kind == 'numpy':      # look-ahead generation of numpy's stream: jobs launched / requests adopted / discarded
            out['numpy_look_ahead'] = dict(zip(('launched', 'adopted', 'discarded'), eng.legacy_ahead_stats()))
    from viabel_amd._legacy_rng import LegacyRandomState
    eng.set_model(model.device_spec())
    rs = LegacyRandomState(1)
    pieces = (('theta_upload_us', lambda: eng.fullrank_set_theta(theta, d)),
              ('philox_noise_us', lambda: (eng.noise_generate(0, N_MC, d, seed=1, stream=1), eng.sync())),
              ('numpy_stream_noise_on_device_us', lambda: eng.noise_legacy_randn(0, rs._h, N_MC, d)),
              ('evaluation_kernels_us', lambda: (eng.elbo_grad_fullrank_enqueue(0, N_MC, d), eng.sync())),
              ('gradient_download_us', lambda: eng.fullrank_get(d)))
    split = {}
    for name, fn in pieces:
        for _ in range(3):
            fn()
        n_rep = 20 if 'numpy' in name else calls
        t0 = time.perf_counter()
        for _ in range(n_rep):
            fn()
        split[name] = 1e6 * (time.perf_counter() - t0) / n_rep
    out['split'] = split
    t0 = time.perf_counter()
    host = np.random.RandomState(1).randn(N_MC, d)
    out['numpy_randn_on_this_host_us'] = 1e6 * (time.perf_counter() - t0)
    out['note'] = ('the headline value is the theta-resident enqueue rate (no PCIe in the timed region); this leg is the '
                   'same evaluation with 2 x 4.2 MB across PCIe and one synchronisation per call.  Floor without '
                   'overlapping transfers and kernels: upload + kernels + download (the overlap was built in round 6 and '
                   "measured slower: DESIGN 4.4).  rng='numpy': the NEXT call's randn(4096, 1024) is generated on the device "
                   "beside this call's GEMMs and adopted by a pointer swap (values and generator state numpy's: DESIGN 4.11)")
    del host
    return out


def fullrank_funnel_leg(eng, vb, steps=200, ring=8, slot0=40):
    """The headline shape on a target with no closed-form shortcuts: FullRankGaussian(1024) + ExclusiveKL on the
    D-dimensional funnel, N_mc=4096 -- sampling GEMM, the funnel's row kernel (f and G per sample), the column-sum
    pass, the gradient GEMM and the split reduction; parameter resident, one evaluation per call on one stream."""
    d = FR_D
    model = vb.FunnelModel(d)
    eng.set_model(model.device_spec())
    fr = vb.FullRankGaussian(d)
    L = np.exp(-1.0) * np.eye(d) + 0.01 * np.tril(np.random.RandomState(3).randn(d, d))
    theta = fr.pack(np.zeros(d), L)
    eng.fullrank_set_theta(theta, d)
    for s in range(ring):
        eng.noise_generate(slot0 + s, N_MC, d, seed=2, stream=s)

    def run(k):
        for i in range(k):
            eng.elbo_grad_fullrank_enqueue(slot0 + i % ring, N_MC, d)
    run(300)
    eng.sync()
    t0 = time.perf_counter()
    run(steps)
    eng.sync()
    us = 1e6 * (time.perf_counter() - t0) / steps
    value, grad = eng.fullrank_get(d)
    from oracle import families as ofam, models as omod, objectives as oobj
    last = slot0 + (steps - 1) % ring
    ov, og = oobj.exclusive_kl(ofam.FullRankGaussian(d), omod.Funnel(d), theta, eng.noise_get_host(last, N_MC, d))
    fl = 2.0 * float(N_MC) * d * (d + 1)             # the two triangular GEMMs; the row kernel and sums are O(N D)
    tf = fl / (us * 1e-6) / 1e12
    return {'workload': 'FullRankGaussian(%d) + ExclusiveKL on the funnel target, N_mc=%d, fp64 (no target-specific '
                        'identities: row kernel + column-sum pass between the two triangular GEMMs)' % (d, N_MC),
            'us_per_eval': us, 'evals_per_s': 1e6 / us, 'value': float(value), 'grad_norm': float(np.linalg.norm(grad)),
            'parity': {'rel_elbo_err': abs(value - ov) / abs(ov),
                       'rel_grad_err': float(np.max(np.abs(grad - og)) / np.max(np.abs(og)))},
            'roofline': {'bound': 'mfma', 'flops_executed': fl, 'achieved': tf, 'peak': FP64_MFMA_PEAK_TFLOPS,
                         'unit': 'TFLOP/s', 'frac': tf / FP64_MFMA_PEAK_TFLOPS,
                         'note': 'per-kernel times: profiles/r04_fullrank_funnel_kernel_stats.txt'}}


def bbvi_quickstart_leg(vb, n_iters=6000):
    """The one workload the reference publishes a rate for: docs/source/quickstart.ipynb:431,
    ``bbvi(2, log_density=<2-D funnel>, learning_rate=0.5, n_iters=30000)`` -- MFGaussian + ExclusiveKL, num_mc_samples=10,
    RAABBVI over RMSProp -- 623-661 it/s in the notebook's tqdm output (CPU, hardware not stated: context, not a like-for-like
    baseline).  Iterations/s through ``bbvi()`` here with the device funnel and with the notebook's own Python callable.
    At D = 2 / N = 10 an iteration is pure latency: one blocking objective call + the optimiser's O(D) host arithmetic."""
    import warnings

    def log_density(x):                       # the notebook's density (quickstart.ipynb:23-29), numpy instead of autograd
        mu, log_sigma = x[:, 0], x[:, 1]
        return (-0.5 * log_sigma ** 2 - 0.5 * np.log(2 * np.pi)
                - 0.5 * (mu * np.exp(-log_sigma)) ** 2 - log_sigma - 0.5 * np.log(2 * np.pi))

    def grad_log_density(x):
        mu, v = x[:, 0], x[:, 1]
        q = (mu * np.exp(-v)) ** 2
        return np.stack([-mu * np.exp(-2 * v), -v + q - 1.0], axis=1)

    out = {'workload': 'docs/source/quickstart.ipynb:431: bbvi(2, log_density=funnel, learning_rate=0.5), MFGaussian + '
                       'ExclusiveKL, num_mc_samples=10, RAABBVI(RMSProp); %d iterations asked for' % n_iters,
           'reference_notebook_it_per_s': [623.5, 661.0, 642.4, 644.8],
           'reference_hardware': 'unknown CPU (tqdm rates in the notebook cell outputs; BASELINE.md 1)'}
    cases = (('device_funnel_model', dict(log_density=vb.FunnelModel(2))),
             ('python_callable_with_gradient', dict(log_density=log_density, grad_log_density=grad_log_density)),
             ('python_callable_central_differences', dict(log_density=log_density)))
    for name, kw in cases:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            vb.bbvi(2, learning_rate=0.5, n_iters=300, **kw)                  # warm-up (kernels, buffers)
            t0 = time.perf_counter()
            res = vb.bbvi(2, learning_rate=0.5, n_iters=n_iters, **kw)
            dt = time.perf_counter() - t0
        its = int(len(res['value_history']))
        out[name] = {'it_per_s': its / dt, 'iterations_run': its, 'us_per_iteration': 1e6 * dt / max(1, its),
                     'vs_reference_notebook': (its / dt) / 642.4}
    out['note'] = ('vs_reference_notebook divides by the median of the four notebook rates; the notebook differentiates the '
                   'callable with autograd, the callable cases here take an explicit gradient or central differences')
    return out


def other_paths_leg(eng, vb):
    """Blocking calls of the paths next to the headline that round 4 reworked: LRGaussian at ranks 8 / 32 / 64 (the streaming
    kernel up to 16, the GEMM-assembled sums beyond), the full-rank path derivative, PSIS of N log weights, and the DIS
    state refresh of a fresh problem (tempering bisection from [0, 1])."""
    from viabel_amd._psis import psislw

    def median_us(call, reps, blocks=3):      # median of `blocks` timed blocks (host jitter on short calls)
        ts = []
        for _ in range(blocks):
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
            ts.append((time.perf_counter() - t0) / reps)
        return 1e6 * sorted(ts)[len(ts) // 2]

    out = {}
    D, N = 1024, N_MC
    rng = np.random.RandomState(0)
    model = vb.FunnelModel(D)
    eng.set_model(model.device_spec())
    lr = {}
    for k in (8, 32, 64):
        fam = vb.LRGaussian(D, k=k)
        theta = fam.pack(np.zeros(D), -np.ones(D), 0.05 * rng.randn(D, k))
        eng.noise_generate(0, N, D, seed=1, stream=0)
        eng.noise_generate(1, N, k, seed=2, stream=0)
        call = ((lambda: eng.elbo_sums_lowrank(0, 1, N, D, k, theta)) if k > 16
                else (lambda: eng.elbo_grad_lowrank(0, 1, N, D, k, theta)))
        for _ in range(10):
            call()
        lr['k=%d' % k] = median_us(call, 50)
    out['lr_gaussian_us_per_call'] = lr
    out['lr_gaussian_note'] = ('D=1024, N_mc=4096, funnel target; ranks above 16 take the GEMM-assembled sums (526 / 700 us at '
                               'k = 32 / 64 before the parameter pieces, the column passes and the copies were merged)')
    # full-rank path derivative at the headline shape (theta resident, enqueue rate like the headline)
    A = rng.randn(D, D)
    gm = vb.CorrelatedGaussianModel(rng.randn(D), covariance=A @ A.T / D + np.eye(D))
    fr = vb.FullRankGaussian(D, rng='philox')
    theta = fr.pack(np.zeros(D), np.exp(-1.0) * np.eye(D) + 0.01 * np.tril(rng.randn(D, D)))
    obj_pd = vb.ExclusiveKL(fr, gm, N, use_path_deriv=True)
    obj_en = vb.ExclusiveKL(vb.FullRankGaussian(D, rng='philox'), gm, N)
    res = {}
    for name, obj in (('path_derivative', obj_pd), ('entropy_form', obj_en)):
        for _ in range(5):
            obj(theta)
        res[name] = median_us(lambda: obj(theta), 20)
    out['fullrank_path_derivative'] = {'blocking_call_us': res, 'ratio': res['path_derivative'] / res['entropy_form'],
                                       'note': 'same blocking host-to-host call both ways (parameter up, gradient down); '
                                               'the score enters as G~ = G + E L^-1 (one triangular product + a blocked '
                                               'triangular inverse); kernels: profiles/r04_fullrank_path_deriv_kernel_stats.txt'}
    ps = {}
    for n in (16384, 100000):
        lw = 2.0 * rng.standard_t(3.0, n)
        for _ in range(5):
            sm, khat = psislw(lw)
        ps['n=%d' % n] = {'us_per_call': median_us(lambda: psislw(lw), 20), 'khat': float(khat)}
    out['psis'] = ps
    out['psis_note'] = ('blocking psislw(lw): upload, kernel, download; the kernel runs on ceil(n / 1024) <= 64 workgroups '
                        '(43 us at n = 16 384, 70 us at 100 000; the single-workgroup kernel: 80 / 477 us) -- '
                        'profiles/r04_psis_kernel_stats.txt')
    Dm, Nm = 64, 16384
    mrng = np.random.RandomState(7)
    model = vb.GaussianModel(0.3 + 0.3 * mrng.randn(Dm), np.exp(0.2 * mrng.randn(Dm)))
    prior = np.zeros(2 * Dm)
    theta = prior + 0.02 * mrng.randn(2 * Dm)
    obj = vb.DISInclusiveKL(vb.MFGaussian(Dm, seed=11, rng='philox'), model, Nm, ess_target=Nm // 8,
                            temper_prior=vb.MFGaussian(Dm), temper_prior_params=prior, use_resampling=False)
    ts = []
    for _ in range(30):
        obj._eps = 1.0
        t0 = time.perf_counter()
        obj(theta)
        ts.append(time.perf_counter() - t0)
    out['dis_refresh_fresh_problem'] = {'us_per_call': 1e6 * float(np.median(ts[5:])), 'eps': float(obj._eps), 'ess': float(obj._ess),
                                        'note': 'MFGaussian(64), N = 16 384, eps restarts at 1: the 50-level bisection is four '
                                                'speculative launches + a final one (~50 us of kernels; ten launches and ~96 us '
                                                'before) -- profiles/r04_dis_bisect_kernel_stats.txt'}
    return out


def blas_threads_for_baseline():
    """Threads the CPU baseline's GEMMs run on: the GPU boxes of this pool show 256 cores to a container whose
    cgroup quota is far smaller, and an oversubscribed OpenBLAS pool is slower than a modest one."""
    try:
        quota = len(os.sched_getaffinity(0))
    except Exception:
        quota = os.cpu_count() or 1
    try:
        with open('/sys/fs/cgroup/cpu.max') as fh:
            q, p = fh.read().split()
            if q != 'max':
                quota = min(quota, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(quota, 16))


def cpu_baseline_and_parity(eng, d, slot, dev_model, theta, budget_s=12.0):
    """Oracle (numpy fp64 restatement of objectives.py:154-164 for the dense Gaussian family, oracle/objectives.py)
    on the host cores: timed as the CPU baseline and, on the same noise matrix read back from the device, used as
    the checker of the HIP result.  The only place this file touches oracle/."""
    from oracle import families as ofam, models as omod, objectives as oobj
    fam, model = ofam.FullRankGaussian(d), omod.GaussFull(dev_model.mean, dev_model.precision)
    noise = eng.noise_get_host(slot, N_MC, d)
    eng.set_model(dev_model.device_spec())
    eng.fullrank_set_theta(theta, d)
    eng.elbo_grad_fullrank_enqueue(slot, N_MC, d)
    dv, dg = eng.fullrank_get(d)
    threads = blas_threads_for_baseline()
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads, user_api='blas')
    except Exception:
        limiter, threads = None, -1
    try:
        ov, og = oobj.exclusive_kl(fam, model, theta, noise)          # warm-up + parity
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s and n < 200:
            oobj.exclusive_kl(fam, model, theta, noise)
            n += 1
        dt = time.perf_counter() - t0
    finally:
        if limiter is not None:
            limiter.restore_original_limits() if hasattr(limiter, 'restore_original_limits') else None
    t0 = time.perf_counter()
    np.random.RandomState(1).randn(N_MC // 8, d)
    t_rng = 8 * (time.perf_counter() - t0)
    base = {
        'value': n / dt, 'unit': 'evals/s', 'cores': threads, 'kind': 'port',
        'sample': '%d evaluations of the numpy oracle at the full headline shape (FullRankGaussian D=%d, N_mc=%d, '
                  'correlated-Gaussian target: three %d x %d x %d fp64 GEMMs through the host BLAS on %d threads + '
                  'elementwise numpy on one), noise pre-generated; RandomState.randn for one matrix would add ~%.2f s; '
                  'host shows %d cpus' % (n, d, N_MC, N_MC, d, d, threads, t_rng, os.cpu_count()),
    }
    parity = {'rel_elbo_err': abs(dv - ov) / abs(ov), 'rel_grad_err': float(np.max(np.abs(dg - og)) / np.max(np.abs(og))),
              'tolerance': 1e-5, 'against': 'numpy oracle (oracle/objectives.py) on the same %d x %d noise matrix read '
                                            'back from the device; north_star asks 1e-5 on the ELBO' % (N_MC, d)}
    return base, parity


# --------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` with no launcher around it starts its own N ranks
# --------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(world, child_argv, timeout_s=1500.0, extra_env=None, poll_s=0.05):
    """Start `world` fresh child processes (rank r gets RANK / LOCAL_RANK = r, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, a
    free MASTER_PORT and a job id for the control handshake), wait for all of them and return
    (exit code, rank 0's stdout lines).  The caller is a plain parent: it has not imported viabel_amd or touched HIP,
    and nothing here replaces a process image (no os.exec*): children are subprocesses in their own sessions, so on
    the first failing child or on the timeout every rank's whole process group is killed.  Exit code: 0 when every
    rank exited 0, else the first failing rank's code (124 for the timeout)."""
    import signal
    import subprocess
    import threading
    import uuid
    env0 = dict(os.environ)
    env0.update({'WORLD_SIZE': str(world), 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(_free_port()),
                 'VIABEL_AMD_JOB_ID': uuid.uuid4().hex, 'HSA_ENABLE_IPC_MODE_LEGACY': '0'})
    env0.update(extra_env or {})
    procs, lines = [], []
    for r in range(world):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(child_argv, env=env, start_new_session=True,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))

    def pump():
        for line in procs[0].stdout:
            lines.append(line.rstrip('\n'))
    reader = threading.Thread(target=pump, daemon=True)
    reader.start()

    def kill_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)       # the child's own session: pgid == pid
                except (ProcessLookupError, PermissionError):
                    pass
        for p in procs:
            try:
                p.wait(10)
            except Exception:
                pass

    rc, deadline = 0, time.time() + timeout_s
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                sys.stderr.write('bench launcher: rank %d exited with code %s; stopping the other ranks\n' % (r, c))
                rc = c if c > 0 else 128 - c
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                sys.stderr.write('bench launcher: ranks still running after %.0f s; killing them\n' % timeout_s)
                rc = 124
                break
            time.sleep(poll_s)
    finally:
        kill_all()
    reader.join(5)
    return rc, lines


def launch_main(args, argv):
    """Parent of a self-launched N-rank run: relay rank 0's JSON line as the LAST line of stdout after checking that
    it really describes an N-rank job."""
    rc, lines = spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + argv, args.launch_timeout)
    result = None
    for line in lines:
        try:
            obj = json.loads(line)
            if isinstance(obj, dict) and 'metric' in obj:
                result = obj
                continue
        except ValueError:
            pass
        sys.stderr.write(line + '\n')           # banners of RCCL etc.: not on stdout
    if rc != 0:
        raise SystemExit(rc)
    if result is None:
        raise SystemExit('bench launcher: rank 0 printed no result line')
    dry = os.environ.get('VB_BENCH_NO_RCCL') == '1'
    if result.get('n_gpus') != args.gpus or (not dry and result.get('rccl_ranks') != args.gpus):
        raise SystemExit('bench launcher: asked for %d GPUs, the ranks report n_gpus=%r rccl_ranks=%r -- refusing to '
                         'print that line' % (args.gpus, result.get('n_gpus'), result.get('rccl_ranks')))
    result['launcher'] = 'bench.py self-launch: %d child processes (subprocess), rank 0 relayed' % args.gpus
    sys.stderr.flush()
    print(json.dumps(result), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak')
    ap.add_argument('--force-comm', action='store_true',
                    help='attach an RCCL communicator even with one rank (exercises the sharded code path)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-legs', action='store_true', help='headline only (skip the secondary legs)')
    ap.add_argument('--no-profile', action='store_true', help='no per-kernel HIP events in the timed region')
    ap.add_argument('--launch-timeout', type=float, default=1500.0,
                    help='self-launched ranks (--gpus N > 1 without a launcher) are killed after this many seconds')
    ap.add_argument('--probe-transport', choices=['rccl', 'ipc', 'host'], default=None,
                    help='internal: a child of second_transport_probe (runs transport_legs under that transport)')
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit('--gpus must be at least 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher around us: become the parent of N fresh rank processes (nothing has touched HIP or imported
        # viabel_amd in this process, and it never will)
        return launch_main(args, sys.argv[1:])
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world != args.gpus:
        # a line whose n_gpus differs from what was asked for must never be printed
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.probe_transport:
        return probe_main(args)

    from viabel_amd import _lib, distributed
    import viabel_amd as vb

    group = distributed.SocketGroup.from_env() if world > 1 else distributed.SocketGroup(0, 1)
    # VB_BENCH_NO_RCCL=1: control-flow dry run of the N > 1 branches on a box with fewer GPUs than ranks (the ranks
    # share device 0, RCCL refuses that, so no communicator is attached: the numbers mean nothing)
    no_rccl = world > 1 and os.environ.get('VB_BENCH_NO_RCCL') == '1'
    # VB_BENCH_TRANSPORT=host: the same N ranks with the host-staged transport (vb_comm_init_host) on device 0 -- every
    # sharded code path and every collective runs, through pinned host memory and the control sockets instead of
    # RCCL / xGMI (functional check of the N > 1 path on a one-GPU box; the numbers are not a scaling measurement)
    host_transport = world > 1 and os.environ.get('VB_BENCH_TRANSPORT') == 'host'
    # VB_BENCH_TRANSPORT=ipc: the xGMI-native all-reduce (vb_comm_init_ipc) instead of RCCL -- one GPU per rank where the
    # node has them, otherwise all ranks on device 0 (a functional run, as the host-staged one)
    ipc_transport = world > 1 and os.environ.get('VB_BENCH_TRANSPORT') == 'ipc'
    ipc_shared_gpu = ipc_transport and _lib.device_count() < world
    eng = _lib.Engine(0) if (no_rccl or host_transport or ipc_shared_gpu) else _lib.default_engine()
    _lib.set_default_engine(eng)
    if world > 1 and (host_transport or ipc_shared_gpu):
        os.environ.setdefault('VB_IPC_TIMEOUT_S', '150')
        prewarm_shared_device(eng, vb, distributed)      # (one-GPU functional runs only: see there)
        group.barrier()
    if host_transport:
        distributed.attach(eng, group, transport='host')
    elif ipc_transport:
        distributed.attach(eng, group, transport='ipc')
    elif world > 1 and not no_rccl:
        distributed.attach(eng, group)
    elif args.force_comm and world == 1:
        eng.comm_init(_lib.Engine.comm_unique_id(), 1, 0)
    rccl_ranks = eng.comm_info()[0]

    head = fullrank_leg(eng, vb, _lib, group, FR_D, args.steps, args.warmup, scaling=args.scaling,
                        profile=not args.no_profile)
    other = None
    if world > 1:          # the other scaling mode as a nested leg (collective: every rank runs it)
        other_mode = 'strong' if args.scaling == 'weak' else 'weak'
        other = fullrank_leg(eng, vb, _lib, group, FR_D, args.steps, args.warmup, scaling=other_mode, profile=False)
        other['scaling'] = other_mode

    tlegs, sharded, transports = None, None, None
    if world > 1 and not no_rccl:          # collective legs: every rank runs them
        primary = 'host' if host_transport else 'ipc' if ipc_transport else 'rccl'
        tlegs = transport_legs(eng, vb, group)
        # the configs that NAME 8 GPUs (BASELINE configs[3], configs[4]) on their sharded device-resident routes
        sharded = {'c3_mvt_dis': sharded_c3_leg(eng, vb, group), 'c4_logistic': sharded_c4_leg(eng, vb, group)}
        # ... and the transport-dependent legs once more under the OTHER transport, by child processes (see there)
        other_t = os.environ.get('VB_BENCH_SECOND_TRANSPORT', 'host' if primary == 'ipc' else 'ipc')
        transports = {primary: tlegs}
        if other_t not in ('0', 'none', primary):
            eng.sync()
            transports[other_t] = second_transport_probe(group, other_t)

    out = None
    if rank == 0:
        sec = head['sec_per_step']
        units = world if args.scaling == 'weak' else 1
        mg = head['per_kernel'].get('model_gemm')
        roof = {'bound': 'mfma', 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'traffic': None,
                'whole_evaluation': head['whole_evaluation'], 'per_kernel': head['per_kernel']}
        if mg:
            # HBM-side bytes per launch of that kernel: PMC counters cannot be collected from inside this process,
            # so the figure is the one measured with `tools/prof_r2.sh pmc` on this shape and committed under
            # profiles/ (FETCH_SIZE doubled as the gfx950 note in MI355X_MICROARCH.md prescribes, + WRITE_SIZE)
            if head['n_rows'] == N_MC and FR_D == 1024:
                roof['traffic'] = MODEL_GEMM_HBM_BYTES
                roof['traffic_source'] = ('profiles/r06_fullrank_gemm_pmc.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, '
                                          'separate passes, FETCH_SIZE doubled per the gfx950 note): 131.6 MB fetched + '
                                          '35.2 MB written per launch vs 75.5 MB of operands and result (Z 33.6 + P 8.4 '
                                          'read, G 33.6 written) + 33.6 MB for the epilogue reading z - m back for sum f; '
                                          'P is fetched once per XCD; 1.25 TB/s, a sixth of HBM peak: MFMA-bound')
            roof.update({'kernel': 'gemm_f64_dma_kernel<A[m][k], 128x64, EpiNegate>: G = -(Z - m) P, dense %d x %d x %d '
                                   '(the dominant kernel of the evaluation)' % (head['n_rows'], FR_D, FR_D),
                         'achieved': mg['achieved'], 'frac': mg['frac'], 'avg_kernel_us': mg['avg_kernel_us'],
                         'launches_timed': mg['launches_timed'], 'flops_per_launch': mg['flops_per_launch'],
                         'peak_source': 'datasheet fp64 matrix = 128 flop/clk/CU x 256 CUs x 2.4 GHz; a stream of independent '
                                        'v_mfma_f64_4x4x4_4b_f64 measures 77.3 TFLOP/s (tools/mfma_barrier_probe.hip); under this '
                                        'kernel the part sustains 2.37-2.39 GHz at ~1300 W of its 1400 W limit '
                                        '(profiles/r02_power_clock.txt; 1.85-1.95 GHz only in the first milliseconds of a burst)'})
        else:
            we = head['whole_evaluation']
            roof.update({'kernel': 'whole evaluation', 'achieved': we['achieved'], 'frac': we['frac']})
        out = {
            'metric': 'ELBO-gradient evals/sec (D=1024, N_mc=4096)',
            'value': units / sec, 'unit': 'evals/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * sec, 'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {
                'workload': 'north_star headline: FullRankGaussian + ExclusiveKL, D=1024, N_mc=4096%s, '
                            'correlated-Gaussian target (dense precision)' % (' per GPU' if args.scaling == 'weak' and world > 1 else ''),
                'family': 'FullRankGaussian (theta = [mu | free Cholesky], 525 824 parameters)',
                'objective': 'ExclusiveKL (entropy form)', 'model': 'correlated Gaussian', 'dim': FR_D,
                'n_mc_per_gpu': head['n_rows'], 'n_mc_global': head['n_total'],
                'noise': 'Philox4x32-10 normals resident in HBM, 8 matrices cycled (268 MB > 256 MiB L3)',
                'parallelism': ('mc-axis dp%d, one RCCL all-reduce of %d doubles per evaluation'
                                % (world, 16 + FR_D + FR_D * (FR_D + 1) // 2)) if world > 1 else 'single GPU',
                'pipelining': ('none: one evaluation per call, all calls on one HIP stream, each behind the previous one; '
                               'theta resident on the device, results not copied out inside the timed region (the '
                               'blocking host-to-host call is the api_call leg)') if world == 1 else
                              ('the timed evaluations are INDEPENDENT (one resident theta): evaluation k\'s all-reduce runs on '
                               'the post stream beside evaluation k + 1\'s GEMMs (two sum sets, vb_fullrank.hip), so `value` is '
                               'an enqueue rate with the collective hidden -- an overlap no optimiser has (theta_{k+1} needs '
                               'grad_k); the rate of the dependent chain is `dependent_chain`, the collective alone `allreduce_us`'),
            },
            'rccl_ranks': rccl_ranks,
            'transport': ('host-staged (VB_BENCH_TRANSPORT=host): all ranks on device 0, collectives through pinned host '
                          'memory and the control sockets -- a functional run of the N > 1 path, not a scaling figure')
                         if host_transport else
                         ('xGMI-native IPC all-reduce (VB_BENCH_TRANSPORT=ipc)' + (': all ranks on device 0, a functional run'
                                                                                   if ipc_shared_gpu else ''))
                         if ipc_transport else ('rccl' if rccl_ranks > 1 or args.force_comm else 'none'),
            'timing': {'timed_blocks': len(head['block_seconds']), 'steps_per_block': args.steps,
                       'ms_per_step_of': 'median block', 'block_ms': [1e3 * t for t in head['block_seconds']],
                       'timed_total_s': float(sum(head['block_seconds']))},
            'check': {'value': head['value'], 'grad_norm': head['grad_norm']},
            'roofline': roof,
        }
        if tlegs is not None:
            out['dependent_chain'] = tlegs['dependent_chain']
            out['allreduce_us'] = tlegs['allreduce_us']
            out['c1_meanfield'] = {'dependent_chain': tlegs['c1_meanfield_chain'], 'allreduce_us': tlegs['allreduce_small_us']}
            out['c3_mvt_dis'] = sharded['c3_mvt_dis']
            out['c4_logistic'] = sharded['c4_logistic']
            out['transports'] = transports
        if other is not None:
            u2 = world if other['scaling'] == 'weak' else 1
            out['other_scaling'] = {'scaling': other['scaling'], 'value': u2 / other['sec_per_step'], 'unit': 'evals/s',
                                    'ms_per_step': 1e3 * other['sec_per_step'], 'n_mc_per_gpu': other['n_rows'],
                                    'n_mc_global': other['n_total']}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'], out['parity'] = cpu_baseline_and_parity(eng, FR_D, 40, head['_model'], head['_theta'])
        if world == 1 and not args.no_legs:
            import contextlib
            import io
            out['c2_fullrank_d512'] = {k: v for k, v in fullrank_leg(
                eng, vb, _lib, group, 512, max(args.steps, 200), args.warmup, profile=True).items()
                if k in ('whole_evaluation', 'per_kernel', 'value', 'grad_norm')}
            # the same evaluation at D = 256: the sub-30-us GEMMs of C3's shape class
            out['fullrank_d256'] = {k: v for k, v in fullrank_leg(
                eng, vb, _lib, group, 256, max(args.steps, 200), args.warmup, profile=True).items()
                if k in ('whole_evaluation', 'per_kernel', 'value', 'grad_norm')}
            out['fullrank_funnel'] = fullrank_funnel_leg(eng, vb)
            out['api_call'] = api_call_leg(eng, vb)
            with contextlib.redirect_stderr(io.StringIO()):
                out['fullrank_fit_loop'] = fullrank_fit_leg(vb)
            out['c1_meanfield'], theta1 = meanfield_leg(eng, vb, _lib)
            with contextlib.redirect_stderr(io.StringIO()):    # tqdm progress bars of the host loop
                out['fit_loop'] = fit_leg(vb, theta1)
            out['c3_mvt_dis'] = c3_leg(vb)
            out['mvt_ekl'] = mvt_ekl_leg(vb)
            out['alpha_divergence'] = alpha_leg(vb)
            out['c4_logistic'] = c4_leg(eng, vb)
            out['other_paths'] = other_paths_leg(eng, vb)
            # (the optimisers report through print(), as the reference's do: stdout carries the JSON line only)
            with contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):
                out['bbvi_quickstart'] = bbvi_quickstart_leg(vb)
            try:
                out['source_model'] = source_model_leg(vb)
            except Exception as exc:       # (no hiprtc on the box: the adaptor is the one part that needs it)
                out['source_model'] = {'error': '%s: %s' % (type(exc).__name__, exc)}
    if rank == 0:
        # RCCL prints a version banner through C stdio; push it out first so the JSON is the last line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    group.barrier()
    group.close()


if __name__ == '__main__':
    main()
