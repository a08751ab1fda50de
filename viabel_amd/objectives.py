"""Variational objectives evaluated by the HIP engine.

Same classes, constructor signatures, properties and error behaviour as
``viabel/objectives.py``; ``objective(var_param)`` returns ``(value, grad)`` exactly as the
reference's ``VariationalObjective.__call__`` (``:32-44``) does, but the whole Monte-Carlo
estimator -- sampling, model log density and gradient, entropy / log q, control variates,
the mean over samples -- runs in hand-written gfx950 kernels behind the C ABI of
``include/viabel_hip.h``.  There is no CPU path: without the shared library or a GPU the
call raises.

Multi-GPU: when a communicator is attached to the engine (``viabel_amd.distributed``), each
rank evaluates its contiguous block of the ``num_mc_samples`` rows and the partial sums are
all-reduced on the device before the epilogue; every rank returns the same ``(value, grad)``.
"""
from abc import ABC, abstractmethod

import ctypes
import os
import weakref

import numpy as np
from scipy import linalg as _sla
from scipy import special as _special

from . import _lib
from .approximations import MFGaussian, MFStudentT, FullRankGaussian, MultivariateT, LRGaussian, symmetric_eig, symmetric_root
from .models import DeviceModel, SourceModel, as_device_model

__all__ = [
    'VariationalObjective',
    'StochasticVariationalObjective',
    'ExclusiveKL',
    'DISInclusiveKL',
    'AlphaDivergence'
]

_NOISE_SLOT = 0
_DIS_SLOT = 1      # DIS keeps its state samples (as base noise) in a slot of its own
_LR_SLOT = 3       # low-rank family: the n x k block of its noise (slot 2 belongs to the diagnostics)


def shard_rows(n, n_ranks, rank):
    """Contiguous block ``[begin, end)`` of the Monte-Carlo axis owned by ``rank``."""
    base, extra = divmod(n, n_ranks)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def _claim_engine_state(eng, kind, owner):
    """The engine keeps one DIS state per family kind; `owner` (an objective, or None for a call that merely reuses the
    state's buffers: the resident ExclusiveKL / AlphaDivergence of the t family) is about to overwrite it.  The previous
    owner fetches what it left on the device first (``DISInclusiveKL._materialize_state``)."""
    owners = eng.__dict__.setdefault('_dis_state_owner', {})
    ref = owners.get(kind)
    other = ref() if ref is not None else None
    if other is not None and other is not owner:
        other._materialize_state()
        other._park_state(eng, kind)      # (kept weights mid-batch: its state samples leave the context with it)
    owners[kind] = weakref.ref(owner) if owner is not None else None


def _shared_randint(eng):
    """``np.random.randint(2 ** 32)`` from the global numpy RNG (objectives.py:455), the same number on every rank
    of a sharded job: rank 0's draw travels over the engine's control group when it has one
    (``viabel_amd.distributed.attach(engine, group)``); without one the ranks must seed numpy identically."""
    seed = int(np.random.randint(2 ** 32))
    if eng.n_ranks > 1:
        seed = int.from_bytes(_rank0_bytes(eng, seed.to_bytes(8, 'little')), 'little')
    return seed


def _peek_next_randint():
    """What the NEXT ``np.random.randint(2 ** 32)`` of the global numpy generator will return, without drawing it; None
    when it cannot be told.  ``randint(2 ** 32)`` is one tempered 32-bit word of the legacy MT19937 (numpy's bounded-integer
    routine takes the 32-bit path unmasked when the range is exactly 2^32 - 1), so the value is read off the generator's
    state through its ctypes interface: word ``pos`` of the key, or, when the block is used up, the first word of the next
    block by the twist recurrence.  Used only as a HINT for the engine's look-ahead noise (``vb_noise_hint_seed``):
    AlphaDivergence draws its seed this way every call (objectives.py:455), and a wrong hint costs a wasted generation,
    never a wrong result (``tests/test_host_logic.py`` pins the prediction against numpy's draws)."""
    try:
        bit_generator = np.random.mtrand._rand._bit_generator
        if type(bit_generator).__name__ != 'MT19937':      # np.random.set_bit_generator: another state layout
            return None
        key = (ctypes.c_uint32 * 625).from_address(bit_generator.ctypes.state_address)
    except Exception:        # an interface that moved
        return None
    pos = key[624]
    if pos < 624:
        y = key[pos]
    elif pos == 624:
        y1 = key[1]
        y = key[397] ^ (((key[0] & 0x80000000) | (y1 & 0x7fffffff)) >> 1) ^ (0x9908b0df if y1 & 1 else 0)
    else:
        return None
    y ^= y >> 11
    y ^= (y << 7) & 0x9d2c5680
    y ^= (y << 15) & 0xefc60000
    return y ^ (y >> 18)


def _hint_next_seed(eng, slot_mask, with_chi=False):
    """Philox mode of AlphaDivergence, one process: name the next call's seed to the engine before this call blocks."""
    if eng.n_ranks == 1:
        nxt = _peek_next_randint()
        if nxt is not None:
            eng.noise_hint_seed(slot_mask, nxt, with_chi)


def _rank0_bytes(eng, payload):
    """Rank 0's ``payload`` on every rank of a sharded job: over the engine's socket control group, else over an
    initialised ``torch.distributed`` group (the caller attached with one); with neither the ranks would draw
    independently and diverge silently, so that is an error, not a convention."""
    group = getattr(eng, 'control_group', None)
    if group is not None:
        return group.broadcast_bytes(payload)
    try:
        import torch.distributed as dist
        ok = dist.is_available() and dist.is_initialized()
    except ImportError:
        ok = False
    if not ok:
        raise RuntimeError('sharded job without a control group: attach the engine with a SocketGroup or an '
                           'initialised torch.distributed process group, so that rank 0\'s host random draws '
                           '(alpha-divergence seed, DIS resampling indices) reach every rank')
    box = [payload]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def _shared_choice(eng, n, size, p):
    """``np.random.choice(n, size, p=p)`` on the global numpy RNG (objectives.py:408), rank 0's draw on every rank."""
    # np.random.choice(n, size=size, p=p) without its argument checks: the legacy RandomState draws
    # random_sample(size) and inverts the normalised cumulative sum -- the same uniforms from the same global
    # stream, the same indices (tests/test_host_logic.py), 20 % less host time at N = 16 384
    p = np.asarray(p, dtype=np.float64)
    if p.shape != (n,):
        raise ValueError("'a' and 'p' must have same size")
    cdf = p.cumsum()
    # the two checks of np.random.choice that matter for normalised DIS weights (a zero or NaN weight sum)
    if not np.isfinite(cdf[-1]):
        raise ValueError('probabilities contain NaN')
    if cdf[-1] <= 0 or p.min() < 0:
        raise ValueError('probabilities are not non-negative')
    cdf /= cdf[-1]
    indices = cdf.searchsorted(np.random.random_sample(size), side='right')
    if eng.n_ranks > 1:
        raw = _rank0_bytes(eng, np.ascontiguousarray(indices, dtype=np.int64).tobytes())
        indices = np.frombuffer(raw, dtype=np.int64)
    return indices


def _source_model_cv(eng, model, approx, var_param, plain_grad, n_local, n_total, method):
    """RGE control variates (``objectives.py:200-268``) for a model given as device source.

    The four variants reduce algebraically (``oracle.objectives.rge_reduced``, checked against the literal per-sample
    code) to the plain estimator's own sums ``gbar = mean g``, ``ge = mean g * eps`` -- recovered here from the plain
    device gradient ``-[gbar | ge sigma + 1]`` -- plus the noise moments ``ebar = mean eps`` and, for ``full``,
    ``M2 = E'E / N`` (``vb_noise_moments``: one column-sum pass and one MFMA Gram product on the device), and three
    derivatives of the model AT THE MEAN m: ``g_mu = grad f(m)``, ``H (s * ebar)`` and (``full``) ``H`` itself.  The
    reference gets those from autograd Hessian-vector products of the Python callable; a source model carries its
    gradient as device code, so they are fourth-order central differences of that device gradient (``vb_model_grad``
    on 5 points, or 4 D + 1 points for the whole Hessian): relative error ~1e-11 for smooth densities.  Everything of
    order N stays on the device; the O(D^2) combination below is host numpy, as for the low-rank family.

    mean block (all methods)               gbar - H (s * ebar)
    scale block mean_only / loo_direct     ge s + 1
    scale block loo_diag                   ge s + 1 - g_mu s ebar
    scale block full                       ... - s_i sum_j H_ij s_j M2_ij + H_ii s_i^2
    (``s`` = the family's standard deviation, ``eps = (z - m) / s``; for MFStudentT ``s = sigma sqrt(df / (df - 2))``
    and the moments of ``eps`` follow from those of the base noise.)"""
    D = approx.dim
    mu, sigma = var_param[:D], np.exp(var_param[D:])
    c2 = 1.0
    if isinstance(approx, MFStudentT):
        c2 = (approx.df - 2.0) / approx.df                  # (sigma / s)^2
    gbar = -plain_grad[:D]
    scale_block = -plain_grad[D:]                           # ge s + 1 (g (z - m) does not care how z - m is factored)
    colsum, gram = eng.noise_moments(_NOISE_SLOT, n_local, D, want_gram=(method == 'full'))
    v = sigma * colsum / n_total                            # s * mean(eps) = sigma * mean(base noise)
    h = 2e-3 * max(1.0, float(np.max(np.abs(mu))))

    def stencil(directions):
        """grad f at m and the fourth-order central difference of grad f along each row of ``directions``."""
        k = directions.shape[0]
        pts = np.concatenate([mu[None, :], mu + h * directions, mu - h * directions, mu + 2 * h * directions,
                              mu - 2 * h * directions])
        g = eng.model_grad(pts)[1]
        d1 = g[1:1 + k] - g[1 + k:1 + 2 * k]
        d2 = g[1 + 2 * k:1 + 3 * k] - g[1 + 3 * k:]
        return g[0], (8.0 * d1 - d2) / (12.0 * h)

    if method == 'full':
        gmu, Ht = stencil(np.eye(D))                        # row j = d grad f / d z_j
        H = 0.5 * (Ht + Ht.T)
        Hv = H @ v
        M2 = gram / n_total
        scale_block = (scale_block - gmu * v - sigma * np.sum(H * M2 * sigma[None, :], axis=1)
                       + np.diag(H) * sigma * sigma / c2)
    else:
        nv = np.linalg.norm(v)
        if nv > 0.0:
            gmu, Hu = stencil((v / nv)[None, :])
            Hv = Hu[0] * nv
        else:
            gmu, Hv = eng.model_grad(mu[None, :])[1][0], np.zeros(D)
        if method == 'loo_diag_approx':
            scale_block = scale_block - gmu * v
    return -np.concatenate([gbar - Hv, scale_block])


class VariationalObjective(ABC):
    """A variational objective to minimise (``viabel/objectives.py:17-79``)."""

    def __init__(self, approx, model):
        self._approx = approx
        self._model = as_device_model(model, approx.dim) if model is not None else None
        self._objective_and_grad = None
        self._update_objective_and_grad()

    def __call__(self, var_param):
        if self._objective_and_grad is None:
            raise RuntimeError("no objective and gradient available")
        return self._objective_and_grad(var_param)

    @abstractmethod
    def _update_objective_and_grad(self):
        """Rebuild the evaluator after ``approx`` / ``model`` / a setting changed."""

    def update(self, var_param, direction):
        return var_param - direction

    @property
    def approx(self):
        return self._approx

    @approx.setter
    def approx(self, value):
        self._approx = value
        self._update_objective_and_grad()

    @property
    def model(self):
        return self._model

    @model.setter
    def model(self, value):
        self._model = as_device_model(value, self._approx.dim) if value is not None else None
        self._update_objective_and_grad()

    # -- engine plumbing shared by the concrete objectives --------------------------------------
    def _engine(self):
        return _lib.default_engine()

    def _require_device_model(self):
        if not isinstance(self._model, DeviceModel):
            raise TypeError(
                'the HIP engine needs a device-resident model (viabel_amd.models.DeviceModel: '
                'GaussianModel, FunnelModel, CorrelatedGaussianModel, the regression models, a SourceModel '
                'holding the log density as HIP code, or a CallableModel around a host callable); got %r.'
                % type(self._model).__name__)
        if self._model.dim != self._approx.dim:
            raise ValueError('model dimension {} != approximation dimension {}'.format(
                self._model.dim, self._approx.dim))

    def _stage_noise(self, eng, n_samples, slot=_NOISE_SLOT, seed=None):
        """Put this call's base noise into a device slot; returns (n_local, n_total).

        numpy mode consumes exactly the draws the reference's ``approx.sample`` would
        (``approximations.py:212-216``) and uploads this rank's row block; philox mode
        generates the rank's rows on the device (global row index in the counter).
        """
        approx = self._approx
        begin, end = shard_rows(n_samples, eng.n_ranks, eng.rank)
        if approx.rng == 'philox':
            kind, df = approx._philox_kind()
            if seed is None:
                eng.noise_generate(slot, end - begin, approx.dim, approx._seed,
                                   approx._next_philox_stream(), row_offset=begin, kind=kind, df=df)
            else:
                eng.noise_generate(slot, end - begin, approx.dim, seed, 0, row_offset=begin, kind=kind, df=df)
        else:
            approx._stage_base_noise(eng, slot, n_samples, begin, end, seed)
        return end - begin, n_samples


class StochasticVariationalObjective(VariationalObjective):
    """Objective approximated by Monte Carlo (``viabel/objectives.py:82-105``)."""

    def __init__(self, approx, model, num_mc_samples):
        self._num_mc_samples = num_mc_samples
        super().__init__(approx, model)

    @property
    def num_mc_samples(self):
        return self._num_mc_samples

    @num_mc_samples.setter
    def num_mc_samples(self, value):
        self._num_mc_samples = value
        self._update_objective_and_grad()


class ExclusiveKL(StochasticVariationalObjective):
    """Exclusive KL (negative ELBO) with the reparameterisation gradient and, optionally,
    the control variates of Miller et al. (``viabel/objectives.py:108-277``)."""

    def __init__(self, approx, model, num_mc_samples, use_path_deriv=False,
                 hessian_approx_method=None):
        self._use_path_deriv = use_path_deriv
        if hessian_approx_method in [None, 'full', 'mean_only', 'loo_diag_approx',
                                     'loo_direct_approx']:
            self.hessian_approx_method = hessian_approx_method
        else:
            raise ValueError("Name of approximation must be one of 'full', 'mean_only', "
                             "'loo_diag_approx', 'loo_direct_approx' or None object.")
        super().__init__(approx, model, num_mc_samples)

    def _update_objective_and_grad(self):
        approx = self.approx
        self._require_device_model()
        flags = _lib.FLAG_PATH_DERIV if self._use_path_deriv else 0
        cv_mode = _lib.CV_MODES[self.hessian_approx_method]

        if isinstance(approx, (MFGaussian, MFStudentT)):
            def objective_and_grad(var_param):
                var_param = np.asarray(var_param, dtype=np.float64)
                if var_param.shape != (approx.var_param_dim,):
                    raise ValueError('var_param must have shape ({},)'.format(approx.var_param_dim))
                eng = self._engine()
                spec = self.model.device_spec()
                eng.set_model(spec)
                family, df = approx._device_family()
                if approx.rng == 'philox' and spec[0] in (_lib.MODEL_GAUSS_DIAG, _lib.MODEL_FUNNEL):
                    # fresh device noise is consumed where it is generated: it never goes through HBM
                    N = self.num_mc_samples
                    begin, end = shard_rows(N, eng.n_ranks, eng.rank)
                    return eng.elbo_grad_meanfield_philox(
                        _NOISE_SLOT, end - begin, approx.dim, var_param, family, approx._seed,
                        approx._next_philox_stream(), df=df, flags=flags, cv_mode=cv_mode, n_total=N, row_offset=begin)
                n_local, n_total = self._stage_noise(eng, self.num_mc_samples)
                if cv_mode != 0 and spec[0] == _lib.MODEL_SOURCE:
                    # a user model's Hessian is not one of the device epilogue's closed forms: plain sums on the
                    # device, the model's derivatives at the mean from its own device gradient (_source_model_cv)
                    value, grad = eng.elbo_grad_meanfield(_NOISE_SLOT, n_local, approx.dim, var_param, family,
                                                          df=df, flags=flags, cv_mode=0, n_total=n_total)
                    if flags:      # the path-derivative form changes the VALUE only (objectives.py:186-188); RGE's
                        grad = eng.elbo_grad_meanfield(_NOISE_SLOT, n_local, approx.dim, var_param, family, df=df,
                                                       flags=0, cv_mode=0, n_total=n_total)[1]     # gradient does not
                    return value, _source_model_cv(eng, self.model, approx, var_param, grad, n_local, n_total,
                                                   self.hessian_approx_method)
                return eng.elbo_grad_meanfield(_NOISE_SLOT, n_local, approx.dim, var_param, family,
                                               df=df, flags=flags, cv_mode=cv_mode, n_total=n_total)
        elif isinstance(approx, FullRankGaussian):
            if cv_mode != 0:
                raise NotImplementedError(
                    'the RGE control variates treat var_param as [mean | log-scale] '
                    '(objectives.py:196-198) and do not apply to a dense-covariance family')

            def objective_and_grad(var_param):
                var_param = np.asarray(var_param, dtype=np.float64)
                if var_param.shape != (approx.var_param_dim,):
                    raise ValueError('var_param must have shape ({},)'.format(approx.var_param_dim))
                eng = self._engine()
                eng.set_model(self.model.device_spec())
                n_local, n_total = self._stage_noise(eng, self.num_mc_samples)
                return eng.elbo_grad_fullrank(_NOISE_SLOT, n_local, approx.dim, var_param,
                                              flags=flags, n_total=n_total)
        elif isinstance(approx, MultivariateT):
            if cv_mode != 0:
                raise NotImplementedError('the RGE control variates treat var_param as [mean | log-scale] '
                                          '(objectives.py:196-198) and do not apply to MultivariateT')
            objective_and_grad = self._mvt_exclusive_kl(approx)
        elif isinstance(approx, LRGaussian):
            if cv_mode != 0:
                raise NotImplementedError('the RGE control variates treat var_param as [mean | log-scale] '
                                          '(objectives.py:196-198) and do not apply to LRGaussian')
            path_deriv = self._use_path_deriv

            def objective_and_grad(var_param):
                var_param = np.asarray(var_param, dtype=np.float64)
                if var_param.shape != (approx.var_param_dim,):
                    raise ValueError('var_param must have shape ({},)'.format(approx.var_param_dim))
                eng = self._engine()
                eng.set_model(self.model.device_spec())
                N = self.num_mc_samples
                begin, end = shard_rows(N, eng.n_ranks, eng.rank)
                if approx.rng == 'philox':
                    approx._philox_noise(eng, end - begin, None, begin, _NOISE_SLOT, _LR_SLOT)
                else:
                    # low-rank block first (approximations.py:639-640)
                    approx._stage_base_noise(eng, _NOISE_SLOT, N, begin, end, None, slot_aux=_LR_SLOT)
                if approx.k > 16:
                    # beyond the streaming kernel's register budget: the sums from GEMMs (vb_elbo_sums_lowrank), the
                    # entropy and its gradient through the k x k capacitance matrix on the host (approximations.py:559-573)
                    value, grad = _lowrank_any_rank(eng, approx, var_param, end - begin, N)
                    if path_deriv:
                        value, grad = _lowrank_path_correction(eng, approx, var_param, value, grad, end - begin, N)
                    return value, grad
                value, grad = eng.elbo_grad_lowrank(_NOISE_SLOT, _LR_SLOT, end - begin, approx.dim, approx.k, var_param,
                                                    n_total=N)
                if path_deriv:
                    value, grad = _lowrank_path_correction(eng, approx, var_param, value, grad, end - begin, N)
                return value, grad
        else:
            raise NotImplementedError(
                'ExclusiveKL on the HIP engine supports MFGaussian, MFStudentT, FullRankGaussian, MultivariateT '
                'and LRGaussian; got {}'.format(type(approx).__name__))
        self._objective_and_grad = objective_and_grad


    def _hessian_vector_product(self, var_param, x):
        """Hessian-vector product of the stochastic objective on ONE noise draw (``objectives.py:166``, ``:275-277``:
        ``make_hvp(variational_objective)(var_param)[0](x)`` -- the forward pass samples once, the product is taken
        on those samples).

        autograd differentiates the reference's graph twice; here the gradient kernels exist and the noise is
        resident, so the product is the symmetric second difference of the DEVICE gradient along ``x`` on the same
        noise matrix, ``(grad(theta + h u) - grad(theta - h u)) / (2 h) * |x|`` with ``u = x / |x|``: two
        evaluations, O(h^2) truncation (h = 6e-6 (1 + |theta|_inf), relative error ~1e-9 in the tests; exact up to
        rounding wherever the gradient is affine in theta along ``u``).  Plain estimator only
        (``hessian_approx_method=None``), as in the reference; mean-field and dense Gaussian families.

        ``use_path_deriv=True`` (``objectives.py:156-159``): autograd keeps the stopped copy ``theta_s`` of the parameter
        fixed through BOTH differentiations, which a difference of path-derivative gradients would not reproduce.  The
        objective splits as ``-mean f(z(theta)) + mean log q(z(theta); theta_s)``.  The entropy of all three families
        is linear in theta (sum of the log scales), so the first term has the Hessian of the entropy-form objective --
        the device difference above.  The second term depends on the noise only through a handful of moments and its
        Hessian at ``theta = theta_s`` is written out (``_path_logq_hvp``)."""
        approx = self.approx
        if self.hessian_approx_method is not None:
            raise AttributeError("'ExclusiveKL' object has no attribute '_hvp'")      # what the reference raises (:275)
        if not isinstance(approx, (MFGaussian, MFStudentT, FullRankGaussian)):
            raise NotImplementedError('_hessian_vector_product: MFGaussian, MFStudentT and FullRankGaussian')
        var_param = np.asarray(var_param, dtype=np.float64)
        x = np.asarray(x, dtype=np.float64)
        if var_param.shape != (approx.var_param_dim,) or x.shape != var_param.shape:
            raise ValueError('var_param and x must have shape ({},)'.format(approx.var_param_dim))
        norm = np.linalg.norm(x)
        if norm == 0.0:
            return np.zeros_like(x)
        eng = self._engine()
        eng.set_model(self.model.device_spec())
        flags = 0
        n_local, n_total = self._stage_noise(eng, self.num_mc_samples)       # one draw, as the forward pass
        h = 6e-6 * (1.0 + np.max(np.abs(var_param)))
        u = x / norm

        def grad_at(theta):
            if isinstance(approx, FullRankGaussian):
                return eng.elbo_grad_fullrank(_NOISE_SLOT, n_local, approx.dim, theta, flags=flags, n_total=n_total)[1]
            family, df = approx._device_family()
            return eng.elbo_grad_meanfield(_NOISE_SLOT, n_local, approx.dim, theta, family, df=df, flags=flags,
                                           n_total=n_total)[1]

        hv = (grad_at(var_param + h * u) - grad_at(var_param - h * u)) * (norm / (2.0 * h))
        if self._use_path_deriv:
            hv = hv + self._path_logq_hvp(eng, var_param, x, n_local, n_total)
        return hv

    def _path_logq_hvp(self, eng, theta, x, n_local, n_total):
        """Hessian-vector product of ``T(theta) = mean_n log q(z_n(theta); theta_s)`` at ``theta = theta_s`` on the
        staged noise (the part of the path-derivative objective that the stopped parameter enters).

        Mean-field families, ``u = (z - mu_s) / sigma_s = a + r eps`` with ``a = (mu - mu_s) / sigma_s``,
        ``r = sigma / sigma_s``, ``log q = sum_d phi(u_d) + const`` (``phi(u) = -u^2 / 2`` or the Student-t
        ``-(nu + 1) / 2 log(1 + u^2 / nu)``): per coordinate
        ``T_mumu = mean phi''(eps) / sigma^2``, ``T_mulam = mean phi''(eps) eps / sigma``,
        ``T_lamlam = mean [phi''(eps) eps^2 + phi'(eps) eps]``.
        Dense Gaussian, ``w = L_s^-1 (mu - mu_s) + L_s^-1 L eps``, ``T = -mean |w|^2 / 2``: the gradient is
        ``(-A' mean w, -tril(A' mean w eps'))`` with ``A = L_s^-1``, its derivative along ``x`` needs ``mean eps`` and the
        noise Gram matrix (``vb_noise_moments``), plus the chain rule through the log-diagonal."""
        approx = self.approx
        D = approx.dim
        if isinstance(approx, FullRankGaussian):
            colsum, gram = eng.noise_moments(_NOISE_SLOT, n_local, D, want_gram=True)      # summed over the ranks
            e_bar, M = colsum / n_total, gram / n_total
            _, L = approx._unpack(theta)
            x_mu = x[:D]
            X = np.zeros((D, D))
            X[np.tril_indices(D)] = x[D:]
            diag = np.diag(L).copy()
            dL = np.tril(X, -1) + np.diag(diag * np.diag(X))
            solve = lambda B, trans=0: _sla.solve_triangular(L, B, lower=True, trans=trans)     # noqa: E731
            w_mean = solve(x_mu) + solve(dL @ e_bar)                     # mean of dw/dt
            h_mu = -solve(w_mean, trans=1)
            G0 = -np.tril(solve(M, trans=1))                             # gradient w.r.t. L at theta_s
            G1 = -np.tril(solve(np.outer(solve(x_mu), e_bar) + solve(dL @ M), trans=1))
            H = np.tril(G1, -1) + np.diag(diag * np.diag(X) * np.diag(G0) + diag * np.diag(G1))
            return np.concatenate([h_mu, H[np.tril_indices(D)]])
        if eng.n_ranks > 1:
            raise NotImplementedError('_hessian_vector_product with use_path_deriv: mean-field families on one rank only')
        eps = eng.noise_get_host(_NOISE_SLOT, n_local, D)
        if isinstance(approx, MFStudentT):
            nu = float(approx.df)
            p1 = -(nu + 1.0) * eps / (nu + eps * eps)
            p2 = -(nu + 1.0) * (nu - eps * eps) / (nu + eps * eps) ** 2
        else:
            p1, p2 = -eps, -np.ones_like(eps)
        sigma = np.exp(theta[D:])
        t_mm = p2.mean(0) / sigma ** 2
        t_ml = (p2 * eps).mean(0) / sigma
        t_ll = (p2 * eps * eps + p1 * eps).mean(0)
        return np.concatenate([t_mm * x[:D] + t_ml * x[D:], t_ml * x[:D] + t_ll * x[D:]])

    # ---- device-resident optimiser loop ---------------------------------------------------------------
    def supports_device_fit(self):
        """True when a whole stochastic-gradient fit can run on the device without host round trips: a
        mean-field, full-rank or low-rank family drawing Philox noise (``rng='philox'``)."""
        approx = self.approx
        if isinstance(approx, LRGaussian):
            return (approx.rng == 'philox' and 1 <= approx.k <= 16 and not self._use_path_deriv
                    and self.hessian_approx_method is None)     # the path-derivative correction is host algebra
        if self.hessian_approx_method is not None and self.model.device_spec()[0] == _lib.MODEL_SOURCE:
            return False                 # control variates of a source model combine host-side (_source_model_cv)
        return (isinstance(approx, (MFGaussian, MFStudentT, FullRankGaussian)) and approx.rng == 'philox'
                and not (isinstance(approx, FullRankGaussian) and self.hessian_approx_method is not None))

    def device_fit(self, n_iters, init_param, opt_kind, hyper, state=None, hist_len=0, log_directions=False,
                   log_gradients=False):
        """Run ``n_iters`` iterations of ``theta <- theta - lr * descent_direction(grad)`` on the device
        (``vb_fit``): the loop of ``optimization.py:91-112`` with the noise of iteration k generated from the
        family's Philox stream exactly as ``n_iters`` consecutive objective calls would consume it.
        Returns (theta, value_history, iterate_history[-hist_len:], optimiser state, directions or None,
        gradients or None)."""
        if not self.supports_device_fit():
            raise NotImplementedError("device_fit needs a mean-field or full-rank family with rng='philox'")
        approx = self.approx
        init_param = np.asarray(init_param, dtype=np.float64)
        if init_param.shape != (approx.var_param_dim,):
            raise ValueError('var_param must have shape ({},)'.format(approx.var_param_dim))
        eng = self._engine()
        eng.set_model(self.model.device_spec())
        N = self.num_mc_samples
        begin, end = shard_rows(N, eng.n_ranks, eng.rank)
        family, df = approx._device_family()
        kind, noise_df = approx._philox_kind()
        first = approx._philox_calls
        approx._philox_calls += n_iters
        flags = _lib.FLAG_PATH_DERIV if self._use_path_deriv else 0
        return eng.fit(_NOISE_SLOT, end - begin, approx.dim, family, init_param, n_iters, opt_kind, hyper,
                       df=df, flags=flags, cv_mode=_lib.CV_MODES[self.hessian_approx_method], n_total=N,
                       row_offset=begin, noise_kind=kind, noise_df=noise_df, seed=approx._seed,
                       first_stream=first, state=state, hist_len=hist_len, log_directions=log_directions,
                       log_gradients=log_gradients, slot_aux=_LR_SLOT)

    def device_history_mean(self, rows):
        """``np.mean(history[-rows:], axis=0)`` of the iterates the last ``device_fit`` kept -- the iterate average
        ``optimization.py:120-126`` returns as ``opt_param`` -- formed on the device from the rows still resident there
        (``vb_fit_history_mean``): the same additions in the same order as numpy's, without its pass over ``rows`` iterates on
        the host (60 x 4.2 MB at D = 1024 dense: 7 ms)."""
        return self._engine().fit_history_mean(rows, self.approx.var_param_dim)

    def _mvt_exclusive_kl(self, approx):
        """Entropy-form ELBO for the multivariate t: sampling, model gradient and the D x D contraction
        sum_n g_n (z_n / s_n)' on the device; the O(D^3) chain rule through the symmetric root on the host (the
        reference differentiates ``sqrtm`` with autograd, ``approximations.py:348``)."""
        D, df = approx.dim, approx.df
        tril = np.tril_indices(D)
        path_deriv = self._use_path_deriv
        _lib.apply_host_blas_policy()      # before the first D x D host product

        def objective_and_grad(var_param):
            var_param = np.asarray(var_param, dtype=np.float64)
            if var_param.shape != (approx.var_param_dim,):
                raise ValueError('var_param must have shape ({},)'.format(approx.var_param_dim))
            eng = self._engine()
            eng.set_model(self.model.device_spec())
            N = self.num_mc_samples
            begin, end = shard_rows(N, eng.n_ranks, eng.rank)
            if approx.rng == 'philox' and not path_deriv:
                # throughput mode: chi-square draws and normals on the GPU, samples through the Cholesky factor
                # x = mu + (L z) / s instead of the symmetric root (approximations.py:348; same distribution, and no
                # reference noise stream is being reproduced in this mode).  With L itself in the sampler the chain
                # rule is d/dL = tril(sum g (z / s)'): the dense-Gaussian pipeline with scaled rows returns the
                # gradient in the flat layout -- no root, no Sylvester solve, no D^2 host work.
                stream = approx._next_philox_stream()
                eng.chisq_generate(df, end - begin, approx._seed, stream, row_offset=begin)
                eng.noise_generate(_NOISE_SLOT, end - begin, D, approx._seed, stream, row_offset=begin)
                value, grad = eng.elbo_grad_mvt_chol(_NOISE_SLOT, end - begin, D, var_param, df, n_total=N)
                # the device value carries the Gaussian entropy constant; the family's own drops the df-only terms
                return value + 0.5 * D * (1.0 + np.log(2.0 * np.pi)), grad
            if approx.rng == 'philox':
                chi = approx._rs.chisquare(df, N)
                eng.noise_generate(_NOISE_SLOT, end - begin, D, approx._seed, approx._next_philox_stream(),
                                   row_offset=begin)
            else:
                # chi-square draws first (approximations.py:345-347)
                want_resident = D > _RESIDENT_GATE
                chi = approx._stage_base_noise(eng, _NOISE_SLOT, N, begin, end, host_chi=not want_resident,
                                               device_chi=want_resident and D >= _RESIDENT_SMALL_N_MIN_DIM)
                if want_resident and getattr(approx, '_chi_on_device', False):
                    # the whole evaluation resident on the device: both noise streams are there already, the symmetric
                    # root and its Frechet derivative are device iterations, the chain rule to the free Cholesky
                    # parameters two more kernels (vb_elbo_grad_mvt_symroot; a buffer of its own -- an interleaved DIS
                    # objective's state samples are left alone; sharded jobs: the sample sums all-reduced on the
                    # device, the O(D^3) algebra redundantly on every rank); None: an iteration did not resolve
                    resident = eng.elbo_grad_mvt_symroot(_NOISE_SLOT, end - begin, D, df, var_param, path_deriv=path_deriv,
                                                         n_total=N)
                    if resident is not None:
                        return resident
                    chi = eng.chisq_get_host(N)          # the host route after all: the draws come down
            mu, L = approx._unpack(var_param)
            Sigma = L @ L.T
            inv_s = 1.0 / np.sqrt(chi / df)
            if path_deriv:
                eig = None
                if D > _HOST_ROOT_MAX_DIM:
                    root, inv_root, info = eng.sym_sqrt_inv(Sigma)
                if D <= _HOST_ROOT_MAX_DIM or not info[2] < _ROOT_TOL:
                    eig = symmetric_eig(Sigma)
                    root, inv_root = (eig[1] * np.sqrt(eig[0])) @ eig[1].T, (eig[1] / np.sqrt(eig[0])) @ eig[1].T
            else:
                root, eig = _device_root(eng, Sigma)
            f_sum, g_sum, C = eng.elbo_sums_mvt(_NOISE_SLOT, end - begin, D, mu, root, inv_s[begin:end], n_total=N)
            if path_deriv:
                # -dlog q/dx = c_n Sigma^(-1/2) z_n / s_n depends on the noise only: its part of the sums is
                # Sigma^(-1/2) m_w / Sigma^(-1/2) e_w, and the value takes the t log density of the samples
                m_w, e_w, l1p = eng.mvt_path_terms(_NOISE_SLOT, end - begin, D, df, inv_s[begin:end], n_total=N)
                C = C + inv_root @ m_w
                g_sum = g_sum + inv_root @ e_w
                lq_mean = (_special.gammaln(0.5 * (df + D)) - _special.gammaln(0.5 * df) - 0.5 * D * np.log(np.pi * df)
                           - np.sum(np.log(np.diag(L))) - 0.5 * (df + D) * l1p / N)
                value = -(f_sum / N - lq_mean)
            else:
                value = -(f_sum / N + approx.entropy(var_param))
            Gs = 0.5 * (C + C.T) / N                                 # d mean f / d root, symmetrised
            # root -> Sigma: the Sylvester equation  root X + X root = Gs
            X = None
            if eig is None:
                _, X, info = eng.sym_sqrt(Sigma, Gs)
                if not info[2] < _ROOT_TOL:
                    X, eig = None, symmetric_eig(Sigma)
            if X is None:
                w, U = eig
                r = np.sqrt(w)
                X = U @ ((U.T @ Gs @ U) / (r[:, None] + r[None, :])) @ U.T
            dL = np.tril(2.0 * X @ L)
            dL[np.diag_indices(D)] = np.diag(dL) * np.diag(L) + (0.0 if path_deriv else 1.0)   # free (log) diagonal,
            return value, -np.concatenate([g_sum / N, dL[tril]])                             # entropy gradient

        return objective_and_grad


def _lowrank_path_correction(eng, approx, var_param, value, grad, n_local, N):
    """Entropy-form (value, grad) of the low-rank family -> path-derivative form (objectives.py:156-159).

    The score Sigma^-1 (x - mu), x - mu = B z + sigma eps, is linear in the noise, so what it adds to the sums
    ``[sum g | sum g eps | sum g z']`` follows from second moments of the noise that the device forms with three
    skinny GEMMs (``vb_lowrank_path_terms``); Sigma^-1 is applied through the k x k capacitance matrix
    (Woodbury, approximations.py:559-607).  The entropy and its gradient, which the device result contains, are
    taken out again."""
    D, k = approx.dim, approx.k
    mu, ls, B = approx._unpack(var_param)
    sig, sig2 = np.exp(ls), np.exp(2.0 * ls)
    W = B / sig2[:, None]                                   # D^-2 B
    M = np.eye(k) + B.T @ W
    Minv = np.linalg.inv(M)
    Mm = M - np.eye(k)

    def sinv(X):                                            # Sigma^-1 X, X: (D, m)
        return X / sig2[:, None] - W @ (Minv @ (W.T @ X))

    ET, TT, es, ee, ts = eng.lowrank_path_terms(_NOISE_SLOT, _LR_SLOT, n_local, D, k, sig[:, None] * W, n_total=N)
    EZ, Q = ET[:, :k], ET[:, k:]
    Z2, ZU, UU = TT[:k, :k], TT[:k, k:], TT[k:, k:]
    zs = ts[:k]
    a1 = sinv((B @ zs + sig * es)[:, None])[:, 0]           # sum_n s_n
    a3 = sinv(B @ Z2 + sig[:, None] * EZ)                   # sum_n s_n z_n'
    bez = np.sum(B * EZ, axis=1)
    R = Mm @ EZ.T + Q.T                                     # sum_n (W' v_n) eps_n'   (k, D)
    a2 = (bez + sig * ee) / sig2 - np.sum((W @ Minv) * R.T, axis=1)        # sum_n s_n * eps_n
    maha = (np.sum((np.sum((B @ Z2) * B, axis=1) + 2.0 * sig * bez + sig2 * ee) / sig2)
            - np.trace(Minv @ (Mm @ Z2 @ Mm.T + Mm @ ZU + ZU.T @ Mm.T + UU)))
    logdet = 2.0 * np.sum(ls) + np.linalg.slogdet(M)[1]
    entropy = 0.5 * D * (np.log(2.0 * np.pi) + 1.0) + 0.5 * logdet
    mean_logq = -0.5 * (D * np.log(2.0 * np.pi) + logdet + maha / N)
    d_entropy = np.concatenate([np.zeros(D), (1.0 / sig2 - np.sum((W @ Minv) * W, axis=1)) * sig2,
                                sinv(B).reshape(-1)])
    corr = np.concatenate([a1, sig * a2, a3.reshape(-1)]) / N
    return value + entropy + mean_logq, grad + d_entropy - corr


def _lowrank_any_rank(eng, approx, var_param, n_local, N):
    """ExclusiveKL (entropy form, objectives.py:160-164) of an LRGaussian of any rank from the device's sums
    ``[sum f | sum g | sum g eps | sum g z']``: value = -(mean f + H), d/dmu = -mean g, d/dlog_sigma = -mean(g eps) sigma
    - dH/dlog_sigma, d/dB = -mean g z' - dH/dB, with H = D/2 (log 2 pi + 1) + sum log sigma + 1/2 log det M,
    M = I + B' D^-2 B (matrix determinant lemma, approximations.py:559-573, :646-652)."""
    D, k = approx.dim, approx.k
    mu, ls, B = approx._unpack(var_param)
    sig2 = np.exp(2.0 * ls)
    W = B / sig2[:, None]
    M = np.eye(k) + B.T @ W
    Minv = np.linalg.inv(M)
    Minv = 0.5 * (Minv + Minv.T)
    entropy = 0.5 * D * (np.log(2.0 * np.pi) + 1.0) + np.sum(ls) + 0.5 * np.linalg.slogdet(M)[1]
    WM = W @ Minv
    d_entropy = np.concatenate([np.zeros(D), 1.0 - np.sum(WM * W, axis=1) * sig2, WM.reshape(-1)])
    f, g, ge, gz = eng.elbo_sums_lowrank(_NOISE_SLOT, _LR_SLOT, n_local, D, k, var_param)
    grad = -np.concatenate([g, ge * np.exp(ls), gz.reshape(-1)]) / N - d_entropy
    return -(f / N + entropy), grad


def _lowrank_pieces(approx, var_param):
    """Pieces of the low-rank parameter the device entry points take, and the O(D k^2) quantities of the capacitance
    matrix M = I + Bs' Bs (Bs = B / sigma; approximations.py:559-607): ``(mu, ls, B, Bs, Minv, log_q_const, BsMinv)``."""
    mu, ls, B = approx._unpack(var_param)
    sig = np.exp(ls)
    Bs = B / sig[:, None]
    M = np.eye(approx.k) + Bs.T @ Bs
    Minv = np.linalg.inv(M)
    Minv = 0.5 * (Minv + Minv.T)
    logdet = 2.0 * np.sum(ls) + np.linalg.slogdet(M)[1]
    cq = -0.5 * (approx.dim * np.log(2.0 * np.pi) + logdet)
    return mu, ls, np.ascontiguousarray(B), Bs, Minv, cq, Bs @ Minv


_ROOT_TOL = 1e-12     # ||root root - Sigma|| / ||Sigma|| accepted from the Newton-Schulz iteration
# Up to this dimension the factor algebra of the parity mode stays on the host: one `eigh` (28 us at D = 10, 0.25 ms at
# D = 50) gives the root and the Sylvester solve, where the device's two GEMM iterations are ~40 dependent launches
# (0.5 ms whatever the size).  Measured per objective call (tools/mvt_root_bench.py, tools/small_shapes_bench.py):
# D = 10: 650 -> 223 us, 50: 1042 -> 328, 100: 1593 -> 867, 160: 2100 -> 1687, 200: equal, 256: the device wins
_HOST_ROOT_MAX_DIM = int(os.environ.get('VIABEL_AMD_HOST_ROOT_MAX_DIM', '160'))
# The RESIDENT reference-identical routes of the t family (vb_dis_refresh_mvt_symroot, vb_elbo_grad_mvt_symroot[_path],
# vb_alpha_grad_mvt_symroot) are taken for D > _RESIDENT_GATE whenever the chi-square draws are on the device (N >= 4096).
# Measured (tools/mvt_root_gate_probe.py, N = 4096 and 16 384): they beat the host-root route at every D from 2 upwards
# (D = 32: 0.58 against 0.83 ms, D = 160: 0.79 against 2.4 ms) -- the gate of 160 that round 4 chose for the HOST route's
# root (LAPACK below, device iteration above) does not apply to them.
_RESIDENT_GATE = int(os.environ.get('VIABEL_AMD_RESIDENT_GATE', '0'))
# Below N = 4096 the chi-square draws are cheaper on the host (17 against 70 us at 1000 draws) and the host route is taken
# with them -- unless the dimension makes the host's O(D^3) algebra the larger cost: from here on the draws are made on
# the device whatever N, for the resident route's sake (N = 1000: host route 1.1 ms at D = 100, 2.0-2.5 ms at D = 160;
# the resident route ~0.6 / ~0.8 ms)
_RESIDENT_SMALL_N_MIN_DIM = int(os.environ.get('VIABEL_AMD_RESIDENT_SMALL_N_MIN_DIM', '48'))


def _device_root(eng, Sigma):
    """Symmetric square root of the scale matrix (``scipy.linalg.sqrtm`` in ``approximations.py:348``) by GEMM
    iterations on the device (``vb_sym_sqrt``); scale matrices the iteration cannot resolve to ``_ROOT_TOL``
    (condition numbers beyond ~1e12) and small ones (``_HOST_ROOT_MAX_DIM``) take the LAPACK route.  Returns
    ``(root, eig)``; ``eig = (w, U)`` only when the eigen-decomposition had to be computed."""
    _lib.apply_host_blas_policy()
    if Sigma.shape[0] <= _HOST_ROOT_MAX_DIM:
        w, U = symmetric_eig(Sigma)
        return (U * np.sqrt(w)) @ U.T, (w, U)
    root, _, info = eng.sym_sqrt(Sigma)
    if info[2] < _ROOT_TOL:
        return root, None
    w, U = symmetric_eig(Sigma)
    return (U * np.sqrt(w)) @ U.T, (w, U)


class DISInclusiveKL(StochasticVariationalObjective):
    """Inclusive KL by distilled importance sampling (``viabel/objectives.py:280-416``)."""

    def __init__(self, approx, model, num_mc_samples, ess_target,
                 temper_prior, temper_prior_params, use_resampling=True,
                 num_resampling_batches=1, w_clip_threshold=10, psis_smooth=False):
        # psis_smooth (not in the reference; BASELINE configs[3] asks for "DISInclusiveKL with PSIS reweighting"):
        # the tempered state weights are Pareto-smoothed (`psislw`, viabel/_psis.py:113-209, on the device) before
        # they are clipped, keeping their sum; `_khat` holds the tail-shape estimate of the last refresh.
        self._psis_smooth = bool(psis_smooth)
        self._khat = None
        self._ess_target = ess_target
        self._w_clip_threshold = w_clip_threshold
        self._max_bisection_its = 50
        self._max_eps = self._eps = 1
        self._use_resampling = use_resampling
        self._num_resampling_batches = num_resampling_batches
        self._resampling_batch_size = max(1, self._ess_target // num_resampling_batches)
        self._objective_step = 0
        self._temper_prior = temper_prior
        self._temper_prior_params = np.asarray(temper_prior_params, dtype=np.float64)
        super().__init__(approx, model, num_mc_samples)

    # log p / log q of the state samples (objectives.py:394-395).  In throughput mode (rng='philox') they stay on the
    # device and are fetched on first access: nothing on the hot path reads them.
    def _set_state_logs(self, log_p, log_q, fetch=None):
        self._lp_cache, self._lq_cache, self._logs_fetch = log_p, log_q, fetch

    def _get_state_logs(self):
        if getattr(self, '_lp_cache', None) is None and getattr(self, '_logs_fetch', None) is not None:
            self._lp_cache, self._lq_cache = self._logs_fetch()
            self._logs_fetch = None
        return getattr(self, '_lp_cache', None), getattr(self, '_lq_cache', None)

    @property
    def _state_log_p_unnormalized(self):
        return self._get_state_logs()[0]

    @property
    def _state_log_q(self):
        return self._get_state_logs()[1]

    # the tempered weights: plain arrays on the two-call paths, fetched on first access after a device-resident step
    def _set_state_weights(self, w=None, fetch=None):
        self._w_cache, self._w_fetch, self._w_sum_cache, self._w_norm_cache = w, fetch, None, None

    @property
    def _state_w_clipped(self):
        if getattr(self, '_w_cache', None) is None and getattr(self, '_w_fetch', None) is not None:
            self._w_cache = self._w_fetch()
            self._w_fetch = None
        return getattr(self, '_w_cache', None)

    @_state_w_clipped.setter
    def _state_w_clipped(self, w):
        self._set_state_weights(w)

    @property
    def _state_w_sum(self):
        if getattr(self, '_w_sum_cache', None) is None:
            self._w_sum_cache = np.sum(self._state_w_clipped)
        return self._w_sum_cache

    @_state_w_sum.setter
    def _state_w_sum(self, value):
        self._w_sum_cache = value

    @property
    def _state_w_normalized(self):
        if getattr(self, '_w_norm_cache', None) is None:
            self._w_norm_cache = self._state_w_clipped / self._state_w_sum
        return self._w_norm_cache

    @_state_w_normalized.setter
    def _state_w_normalized(self, value):
        self._w_norm_cache = value

    def _claim_state(self, eng, kind):
        """Before a refresh overwrites the engine's state of `kind`: an objective that left its weights / per-sample logs
        on the device to be fetched on first access (the device-resident steps) fetches them NOW, so that they remain
        the arrays of ITS refresh -- as the reference's ``_state_*`` attributes do -- whoever refreshes next; one that is
        in the middle of a batch of kept-weights steps (``num_resampling_batches > 1``) parks its state samples."""
        self._drop_parked()             # (a state of our own that was parked: this refresh replaces it)
        _claim_engine_state(eng, kind, self)

    def _drop_parked(self):
        parked = getattr(self, '_parked', None)
        if parked is not None:
            eng, _, handle, fin = parked
            fin.detach()
            eng.dis_state_drop(handle)
            self._parked = None

    def _park_state(self, eng, kind):
        """Another objective is about to refresh over this one's state.  The reference keeps ``_state_samples`` per object
        (``objectives.py:391-403``): two objectives with ``num_resampling_batches > 1`` may take turns.  If this one's next
        call is a kept-weights step on a state that is still intact, the state leaves the context with it
        (``vb_dis_state_park``: buffers detached, no copies) and comes back in ``_own_state``."""
        if (not self._use_resampling or self._num_resampling_batches <= 1 or getattr(self, '_parked', None) is not None
                or self._objective_step % self._num_resampling_batches == 0
                or getattr(self, '_state_gen', None) != (id(eng), eng.dis_generation(kind))):
            return
        handle = eng.dis_state_park(kind, _DIS_SLOT if kind in (0, 1) else -1)
        fin = weakref.finalize(self, eng.dis_state_drop, handle)
        fin.atexit = False      # (not at interpreter exit: the HIP runtime may be gone by then)
        self._parked = (eng, kind, handle, fin)

    def _materialize_state(self):
        self._state_w_clipped           # (properties: the pending fetches run)
        self._get_state_logs()

    def _own_state(self, eng, kind, refreshed):
        """The state samples live in the engine, one set per family kind: after a refresh remember its generation,
        before a gradient on kept weights make sure nobody else refreshed in between (two interleaved
        objectives with num_resampling_batches > 1 used to compute on each other's samples silently)."""
        parked = getattr(self, '_parked', None)
        if parked is not None and not refreshed and parked[0] is eng and parked[1] == kind:
            # our state was parked when another objective refreshed: whoever holds the context's state now may be in the
            # middle of a batch too (it parks), then ours is installed again
            _claim_engine_state(eng, kind, self)
            parked[3].detach()
            eng.dis_state_unpark(parked[2])
            self._parked = None
        gen = eng.dis_generation(kind)
        if refreshed:
            self._state_gen = (id(eng), gen)
        elif getattr(self, '_state_gen', None) != (id(eng), gen):
            raise _lib.EngineError('the DIS state samples of this objective were overwritten by another objective on the '
                                   'same engine between two refreshes; give each interleaved objective its own engine '
                                   'or use num_resampling_batches=1')

    def _smooth_weights(self, eng, w):
        """``w -> sum(w) * exp(psislw(log w))``: the PSIS-smoothed weights on the scale of the raw ones."""
        if not self._psis_smooth:
            return w
        total = np.sum(w)
        with np.errstate(divide='ignore'):
            lw = np.log(w)
        smoothed, self._khat = eng.psis_smooth(w.size, lw)
        return total * np.exp(smoothed)

    def _clip_weights(self, w):
        """Clip weights to ``w_clip_threshold`` (``objectives.py:370-386``).

        With the default threshold 10 no weight can exceed ``10 * sum(w)``, so this is a no-op; the
        reference's clipping line (``:385``) calls a float and cannot run, and its literal recursion does not
        terminate in floating point (a clipped weight equals the next threshold up to rounding) -- the fixed point it
        aims at is computed instead: the clipped set only grows, unclipped weights are compared with
        ``thr * U / (1 - thr n)`` until none reaches it (``oracle.objectives.DISInclusiveKL._clip``; the same rounds as
        ``vb_dis_clip_mvt`` runs on the device-resident weights)."""
        thr = self._w_clip_threshold
        w = np.asarray(w, dtype=np.float64)
        S = np.sum(w)
        if not np.any(w > S * thr):
            return w
        clipped = np.zeros(w.shape, dtype=bool)
        while True:
            new = ~clipped & (w >= S * thr)
            if not np.any(new):
                break
            trial = clipped | new
            n, U = np.sum(trial), np.sum(w[~trial])
            if U == 0 or 1. - thr * n <= 0:
                break
            clipped, S = trial, U / (1. - thr * n)
        if not np.any(clipped):
            return w
        out = w.copy()
        out[clipped] = thr * np.sum(w[~clipped]) / (1. - thr * np.sum(clipped))
        return out

    def _build_prior_spec(self, D):
        """The tempering prior (``objectives.py:283-285``: any family; ``:317-319`` calls its ``log_density``) as the
        engine takes it: an MFGaussian parameter goes straight into the refresh calls (``tests/test_objectives.py:82-87``,
        fused with the model's row pass); MFStudentT is one more row pass; FullRankGaussian / MultivariateT -- and
        LRGaussian through the Cholesky factor of its covariance -- one more N x D x D product
        (``vb_dis_set_temper_prior``).  Returns ``(spec or None, prior argument of the refresh calls)``."""
        prior, params = self._temper_prior, self._temper_prior_params
        if not isinstance(prior, (MFGaussian, MFStudentT, MultivariateT, FullRankGaussian, LRGaussian)):
            raise NotImplementedError('temper_prior must be one of MFGaussian, MFStudentT, FullRankGaussian, '
                                      'MultivariateT, LRGaussian; got {}'.format(type(prior).__name__))
        if prior.dim != D:
            raise ValueError('temper_prior has dimension {}, the approximation {}'.format(prior.dim, D))
        if params.shape != (prior.var_param_dim,):
            raise ValueError('temper_prior_params must have shape ({},)'.format(prior.var_param_dim))
        if isinstance(prior, MFGaussian):
            return None, params
        if isinstance(prior, MFStudentT):
            loc, log_sigma = prior._unpack(params)
            spec = (_lib.PRIOR_DIAG_STUDENT_T, prior.df, np.array(loc), np.array(log_sigma), 0.0)
        else:
            if isinstance(prior, LRGaussian):
                loc, log_sigma, B = prior._unpack(params)
                L = np.linalg.cholesky(B @ B.T + np.diag(np.exp(2.0 * log_sigma)))
                df = 0.0
            else:
                loc, L = prior._unpack(params)
                df = prior.df if isinstance(prior, MultivariateT) else 0.0
            Linv = _sla.solve_triangular(L, np.eye(D), lower=True)
            spec = (_lib.PRIOR_DENSE, df, np.array(loc), np.ascontiguousarray(np.tril(Linv)),
                    float(np.sum(np.log(np.diag(L)))))
        # the refresh calls still take an MFGaussian parameter (its pass is overwritten by the installed prior's)
        return spec, np.zeros(2 * D)

    def _update_objective_and_grad(self):
        approx = self.approx
        self._require_device_model()
        if not isinstance(approx, (MFGaussian, MFStudentT, MultivariateT, FullRankGaussian, LRGaussian)):
            raise NotImplementedError('DISInclusiveKL on the HIP engine supports MFGaussian, MFStudentT, MultivariateT, '
                                      'FullRankGaussian and LRGaussian; got {}'.format(type(approx).__name__))
        if isinstance(approx, LRGaussian) and not 1 <= approx.k <= 64:
            raise NotImplementedError('LRGaussian under DISInclusiveKL on the HIP engine: 1 <= k <= 64')
        self._prior_spec, self._prior_arg = self._build_prior_spec(approx.dim)
        slot = _DIS_SLOT
        if isinstance(approx, (MultivariateT, FullRankGaussian)):
            self._objective_and_grad = self._mvt_objective(approx, slot)
            return
        if isinstance(approx, LRGaussian):
            self._objective_and_grad = self._lowrank_objective(approx, slot)
            return

        def variational_objective(var_param):
            var_param = np.asarray(var_param, dtype=np.float64)
            if var_param.shape != (approx.var_param_dim,):
                raise ValueError('var_param must have shape ({},)'.format(approx.var_param_dim))
            eng = self._engine()
            eng.set_model(self.model.device_spec())
            family, df = approx._device_family()
            N = self.num_mc_samples
            begin, end = shard_rows(N, eng.n_ranks, eng.rank)     # this rank's block of the samples
            n_local = end - begin
            if not self._use_resampling or self._objective_step % self._num_resampling_batches == 0:
                # state refresh (objectives.py:393-401): new samples (kept as noise on the device),
                # log q, log p, tempering bisection, clipping.  Sharded jobs gather the per-sample
                # vectors, so eps and the weights cover all N samples on every rank.
                self._claim_state(eng, 0)
                self._stage_noise(eng, N, slot=slot)
                eng.dis_set_temper_prior(self._prior_spec)
                self._eps, self._ess, w, log_p, log_q = eng.dis_refresh_meanfield(
                    slot, n_local, approx.dim, var_param, self._prior_arg, family, self._eps,
                    self._ess_target, self._max_bisection_its, df=df, n_total=N)
                self._set_state_logs(log_p, log_q)
                self._state_w_clipped = self._clip_weights(self._smooth_weights(eng, w))
                self._state_w_sum = np.sum(self._state_w_clipped)
                self._state_w_normalized = self._state_w_clipped / self._state_w_sum
                self._own_state(eng, 0, True)
            else:
                self._own_state(eng, 0, False)
            self._objective_step += 1
            if not self._use_resampling:     # :405-406
                return eng.dis_grad_meanfield(slot, n_local, approx.dim, var_param,
                                              self._state_w_clipped[begin:end], 1.0 / N, family, df=df)
            # global numpy RNG (:408): ranks of a sharded job must seed it identically
            indices = _shared_choice(eng, N, self._resampling_batch_size, self._state_w_normalized)
            counts = np.bincount(indices, minlength=N).astype(np.float64)
            scale = self._state_w_sum / N / self._resampling_batch_size       # :412-414
            return eng.dis_grad_meanfield(slot, n_local, approx.dim, var_param, counts[begin:end], scale,
                                          family, df=df)

        self._objective_and_grad = variational_objective


    def _lowrank_objective(self, approx, slot):
        """DIS for the low-rank Gaussian (``approximations.py:610-731``): sampling, the Woodbury log density
        (``:685-707``), tempering and the weighted score sums on the device (``vb_dis_*_lowrank``); the O(D k^2)
        algebra through the k x k capacitance matrix here.  With rho = (x - mu) / sigma, tau = M^-1 Bs' rho and
        a = Sigma^-1 (x - mu) = (rho - Bs tau) / sigma:  d log q / d mu = a,
        d/d log_sigma = -sigma^2 diag(Sigma^-1) + sigma^2 a^2,  d/dB = -Sigma^-1 B + a tau'."""
        D, k = approx.dim, approx.k

        def variational_objective(var_param):
            var_param = np.asarray(var_param, dtype=np.float64)
            if var_param.shape != (approx.var_param_dim,):
                raise ValueError('var_param must have shape ({},)'.format(approx.var_param_dim))
            eng = self._engine()
            eng.set_model(self.model.device_spec())
            N = self.num_mc_samples
            begin, end = shard_rows(N, eng.n_ranks, eng.rank)
            n_local = end - begin
            mu, ls, B, Bs, Minv, cq, BsMinv = _lowrank_pieces(approx, var_param)
            if not self._use_resampling or self._objective_step % self._num_resampling_batches == 0:
                self._claim_state(eng, 2)
                if approx.rng == 'philox':
                    approx._philox_noise(eng, n_local, None, begin, slot, _LR_SLOT)
                else:
                    # low-rank block first (approximations.py:639-640)
                    approx._stage_base_noise(eng, slot, N, begin, end, None, slot_aux=_LR_SLOT)
                eng.dis_set_temper_prior(self._prior_spec)
                self._eps, self._ess, w, log_p, log_q = eng.dis_refresh_lowrank(
                    slot, _LR_SLOT, n_local, D, k, mu, ls, B, Minv, cq, self._prior_arg, self._eps,
                    self._ess_target, self._max_bisection_its, n_total=N)
                self._set_state_logs(log_p, log_q)
                self._state_w_clipped = self._clip_weights(self._smooth_weights(eng, w))
                self._state_w_sum = np.sum(self._state_w_clipped)
                self._state_w_normalized = self._state_w_clipped / self._state_w_sum
                self._own_state(eng, 2, True)
            else:
                self._own_state(eng, 2, False)
            self._objective_step += 1
            if not self._use_resampling:
                weights, scale = self._state_w_clipped, 1.0 / N
            else:
                indices = _shared_choice(eng, N, self._resampling_batch_size, self._state_w_normalized)
                weights = np.bincount(indices, minlength=N).astype(np.float64)
                scale = self._state_w_sum / N / self._resampling_batch_size
            Srt, Stt, Sr, Srr, St, W, Wlq = eng.dis_grad_lowrank(n_local, D, k, mu, ls, B, Minv, cq, weights[begin:end])
            sig = np.exp(ls)
            d_mu = (Sr - Bs @ St) / sig
            quad = np.sum((Bs @ Stt) * Bs, axis=1)
            d_ls = -W * (1.0 - np.sum(BsMinv * Bs, axis=1)) + (Srr - 2.0 * np.sum(Bs * Srt, axis=1) + quad)
            d_B = (-W * BsMinv + Srt - Bs @ Stt) / sig[:, None]
            grad_logq = np.concatenate([d_mu, d_ls, d_B.reshape(-1)])
            return -scale * Wlq, -scale * grad_logq

        return variational_objective

    def _mvt_objective(self, approx, slot):
        """DIS for the dense families: O(D^3) factor algebra here (as the reference does with sqrtm / eigh on the
        host), O(N D^2) sampling / log-density / Gram work on the device.  The dense Gaussian is the df -> infinity
        member of the same kernels (``df = 0`` in the C ABI): no chi-square scaling, z = mu + L eps."""
        from scipy import linalg as sla
        gaussian = isinstance(approx, FullRankGaussian)
        D, df = approx.dim, (0.0 if gaussian else approx.df)
        tril = np.tril_indices(D)

        _lib.apply_host_blas_policy()      # before the first D x D host product

        from scipy.linalg import blas, lapack
        diag = np.diag_indices(D)
        philox = approx.rng == 'philox'

        def factors(var_param):
            # L from the flat parameter without the family's generic unpacking (index arrays cached), its inverse by
            # LAPACK dtrtri on the transposed (Fortran-ordered) view: no copies, no scipy wrapper overhead -- at
            # D = 256 the wrappers used to cost more than the D^3 / 3 flops
            L = np.zeros((D, D))
            L[tril] = var_param[D:]
            L[diag] = np.exp(L[diag])
            Ut, info = lapack.dtrtri(L.T, lower=0, overwrite_c=0)
            if info != 0:
                raise ValueError('singular Cholesky factor')
            return L, Ut.T

        def variational_objective(var_param):
            var_param = np.asarray(var_param, dtype=np.float64)
            if var_param.shape != (approx.var_param_dim,):
                raise ValueError('var_param must have shape ({},)'.format(approx.var_param_dim))
            eng = self._engine()
            eng.set_model(self.model.device_spec())
            N = self.num_mc_samples
            begin, end = shard_rows(N, eng.n_ranks, eng.rank)
            n_local = end - begin
            # throughput mode: mu, L, L^-1 and the chain rule of the gradient are formed on the device from var_param
            memo = []

            def host_factors():             # (L, L^-1) on the host, only where a host route needs them
                if not memo:
                    memo.append(factors(var_param))
                return memo[0]
            # ... and the weights never leave the device either -- Pareto smoothing (vb_dis_psis_mvt) and, for a clipping
            # threshold below 1 (objectives.py:370-386; the identity otherwise, the default is 10), the clipping
            # (vb_dis_clip_mvt) included.  Sharded jobs (round 6): every rank samples and scores its own
            # rows, [log p | log q | log prior] are all-gathered ON THE DEVICE, bisection / smoothing / clipping / the
            # multinomial draw run redundantly on the whole vectors on every rank, each rank forms the weighted sums of
            # its rows, one all-reduce, and the D^3 chain rule redundantly again: the same bits on every rank
            refresh_now = not self._use_resampling or self._objective_step % self._num_resampling_batches == 0
            # The reference-identical mode of the t family (rng='numpy') is resident on the device as well where the
            # symmetric root is the device's job anyway: numpy's chi-square and normal streams are generated there bit for
            # bit, the root of approximations.py:348 by vb_dis_refresh_mvt_symroot
            sym_try = not philox and not gaussian and D > _RESIDENT_GATE
            resident = philox or (not refresh_now and getattr(self, '_sym_resident', False))
            clip = self._w_clip_threshold < 1.0
            if refresh_now:
                self._claim_state(eng, 1)
                self._sym_resident = False
                eng.dis_set_temper_prior(self._prior_spec)
                if gaussian:
                    chi = np.ones(N)
                    if philox:
                        eng.noise_generate(slot, n_local, D, approx._seed, approx._next_philox_stream(),
                                           row_offset=begin)
                        root = None
                        if resident:     # as the t family below: df = 0 is the Gaussian member of the same kernels
                            eng.dis_refresh_mvt_deferred(slot, n_local, D, df, var_param, self._prior_arg,
                                                         self._eps, self._ess_target, self._max_bisection_its, n_total=N)
                            if self._psis_smooth:
                                eng.dis_psis_mvt(N)
                            if clip:
                                eng.dis_clip_mvt(N, self._w_clip_threshold)
                            self._set_state_logs(None, None, lambda: eng.dis_state_get(True, N))
                            self._set_state_weights(None, lambda: eng.dis_weights_get(N))
                            self._own_state(eng, 1, True)
                    else:
                        approx._stage_base_noise(eng, slot, N, begin, end)
                        # the dense Gaussian samples through L itself (x = mu + eps L'): with the family's own normal
                        # stream in the slot this IS the device's factor path -- resident like the throughput mode
                        # (the reference's resampling draw, when asked for, on the fetched weights as for the t family)
                        root = None
                        resident = self._sym_resident = True
                        eng.dis_refresh_mvt_deferred(slot, n_local, D, df, var_param, self._prior_arg,
                                                     self._eps, self._ess_target, self._max_bisection_its, n_total=N)
                        if self._psis_smooth:
                            eng.dis_psis_mvt(N)
                        if clip:
                            eng.dis_clip_mvt(N, self._w_clip_threshold)
                        self._set_state_logs(None, None, lambda: eng.dis_state_get(True, N))
                        self._set_state_weights(None, lambda: eng.dis_weights_get(N))
                        self._own_state(eng, 1, True)
                elif philox:
                    # throughput mode: chi-square draws and normals on the GPU, and x = mu + (z L') / s with the
                    # Cholesky factor instead of the reference's symmetric root (approximations.py:348).  The samples
                    # have the same distribution (z L' and z Sigma^1/2 are both N(0, Sigma)), DIS treats them as
                    # constants, and no noise stream of the reference is being reproduced in this mode -- so the
                    # D^3 root (0.9 ms of Newton-Schulz GEMMs at D = 256) is not computed at all.
                    stream = approx._next_philox_stream()
                    eng.chisq_generate(df, n_local, approx._seed, stream, row_offset=begin)
                    eng.noise_generate(slot, n_local, D, approx._seed, stream, row_offset=begin)
                    chi = None
                    root = None                                 # L' from var_param, on the device
                    if resident:
                        # device-resident step: the refresh only enqueues; weights, eps, ess stay on the device and
                        # come back (eps, ess) with the gradient after one synchronisation
                        eng.dis_refresh_mvt_deferred(slot, n_local, D, df, var_param, self._prior_arg, self._eps,
                                                     self._ess_target, self._max_bisection_its, n_total=N)
                        if self._psis_smooth:
                            eng.dis_psis_mvt(N)
                        if clip:
                            eng.dis_clip_mvt(N, self._w_clip_threshold)
                        self._set_state_logs(None, None, lambda: eng.dis_state_get(True, N))
                        self._set_state_weights(None, lambda: eng.dis_weights_get(N))
                        self._own_state(eng, 1, True)
                else:
                    # chi-square draws first (approximations.py:345-347)
                    chi = approx._stage_base_noise(eng, slot, N, begin, end, host_chi=not sym_try,
                                                   device_chi=sym_try and D >= _RESIDENT_SMALL_N_MIN_DIM)
                    if sym_try and getattr(approx, '_chi_on_device', False):
                        # the whole step on the device: the context holds numpy's chi-square draws, the slot its normals
                        info = eng.dis_refresh_mvt_symroot(slot, n_local, D, df, var_param, self._prior_arg, self._eps,
                                                           self._ess_target, self._max_bisection_its, n_total=N)
                        if info is not None:
                            resident = self._sym_resident = True
                            if self._psis_smooth:
                                eng.dis_psis_mvt(N)
                            if clip:
                                eng.dis_clip_mvt(N, self._w_clip_threshold)
                            self._set_state_logs(None, None, lambda: eng.dis_state_get(True, N))
                            self._set_state_weights(None, lambda: eng.dis_weights_get(N))
                            self._own_state(eng, 1, True)
                    if not resident:
                        if chi is None:
                            chi = eng.chisq_get_host(N)          # the host route after all: the draws come down
                        L = host_factors()[0]
                        root, _ = _device_root(eng, L @ L.T)        # symmetric square root, :348
                if not resident:
                    Linv = None if philox else host_factors()[1]
                    self._eps, self._ess, w, log_p, log_q = eng.dis_refresh_mvt(
                        slot, n_local, D, df, var_param, None if chi is None else chi[begin:end], root, Linv,
                        self._prior_arg, self._eps, self._ess_target, self._max_bisection_its, n_total=N,
                        fetch_logs=not philox)
                    self._set_state_logs(log_p, log_q, (lambda: eng.dis_state_get(True, N)) if philox else None)
                    self._state_w_clipped = self._clip_weights(self._smooth_weights(eng, w))
                    self._state_w_sum = np.sum(self._state_w_clipped)
                    self._state_w_normalized = self._state_w_clipped / self._state_w_sum
                    self._own_state(eng, 1, True)
            else:
                self._own_state(eng, 1, False)
            self._objective_step += 1
            if resident:
                if not self._use_resampling:
                    value, grad, self._eps, self._ess = eng.dis_step_mvt_packed(n_local, D, df, var_param, 1.0 / N)
                elif not philox:
                    # the reference's resampling draw (objectives.py:408: the global numpy generator) on the fetched
                    # weights; the weighted score and its chain rule stay on the device
                    if refresh_now:
                        self._eps, self._ess, khat = eng.dis_scalars_get()
                        if self._psis_smooth:
                            self._khat = khat
                    indices = _shared_choice(eng, N, self._resampling_batch_size, self._state_w_normalized)
                    weights = np.bincount(indices, minlength=N).astype(np.float64)
                    return eng.dis_grad_mvt_packed(n_local, D, df, var_param, weights[begin:end],
                                                   self._state_w_sum / N / self._resampling_batch_size)
                else:
                    # multinomial draw on the device from the family's Philox stream (objectives.py:408 draws from the
                    # global numpy RNG; this mode reproduces no reference stream)
                    M = self._resampling_batch_size
                    value, grad, self._eps, self._ess = eng.dis_step_mvt_packed(
                        n_local, D, df, var_param, 1.0 / N / M, resample_m=M, seed=approx._seed,
                        stream=approx._next_philox_stream())
                if self._psis_smooth:
                    self._khat = eng.last_khat
                return value, grad
            if not self._use_resampling:
                weights, scale = self._state_w_clipped, 1.0 / N
            else:
                indices = _shared_choice(eng, N, self._resampling_batch_size, self._state_w_normalized)
                weights = np.bincount(indices, minlength=N).astype(np.float64)
                scale = self._state_w_sum / N / self._resampling_batch_size
            if philox:
                return eng.dis_grad_mvt_packed(n_local, D, df, var_param, weights[begin:end], scale)
            L, Linv = host_factors()
            w_sum, w_logq, d_mu, gram = eng.dis_grad_mvt(n_local, D, df, var_param, Linv, weights[begin:end])
            # chain rule to the free Cholesky parameters
            # d log q / d Sigma = -1/2 w_sum Sigma^-1 + 1/2 S and Sigma = L L': d/dL = tril(2 (d/dSigma) L)
            #   = tril(S L) - w_sum tril(L^-T): L^-T is upper triangular, so only its diagonal 1 / L_ii survives --
            # one D x D product instead of three
            d_L = blas.dsymm(1.0, gram, L, side=0, lower=1)      # S L with S = S' given by its lower triangle; only
            d_L[diag] = d_L[diag] * L[diag] - w_sum              # the lower triangle of the product is read below
            grad_logq = np.concatenate([d_mu, d_L[tril]])
            return -scale * w_logq, -scale * grad_logq

        return variational_objective


class AlphaDivergence(StochasticVariationalObjective):
    """Log of the alpha-divergence (``viabel/objectives.py:419-463``)."""

    def __init__(self, approx, model, num_mc_samples, alpha):
        self._alpha = alpha
        super().__init__(approx, model, num_mc_samples)

    @property
    def alpha(self):
        return self._alpha

    def _update_objective_and_grad(self):
        approx = self.approx
        self._require_device_model()
        if not isinstance(approx, (MFGaussian, MFStudentT, FullRankGaussian, MultivariateT, LRGaussian)):
            raise NotImplementedError('AlphaDivergence on the HIP engine supports MFGaussian, MFStudentT, '
                                      'MultivariateT, FullRankGaussian and LRGaussian; got {}'.format(
                                          type(approx).__name__))
        alpha = self.alpha
        if isinstance(approx, MultivariateT):
            self._objective_and_grad = self._mvt_alpha(approx, alpha)
            return
        if isinstance(approx, LRGaussian):
            if not 1 <= approx.k <= 64:
                raise NotImplementedError('LRGaussian under AlphaDivergence on the HIP engine: 1 <= k <= 64')
            self._objective_and_grad = self._lowrank_alpha(approx, alpha)
            return

        def objective_grad_and_log_norm(var_param):
            var_param = np.asarray(var_param, dtype=np.float64)
            if var_param.shape != (approx.var_param_dim,):
                raise ValueError('var_param must have shape ({},)'.format(approx.var_param_dim))
            # the reference draws a shared seed from the GLOBAL numpy RNG (objectives.py:455) and
            # samples from a fresh RandomState(seed) (approximations.py:213)
            eng = self._engine()
            seed = _shared_randint(eng)
            eng.set_model(self.model.device_spec())
            n_local, n_total = self._stage_noise(eng, self.num_mc_samples, seed=seed)
            if approx.rng == 'philox':
                _hint_next_seed(eng, 1 << _NOISE_SLOT)
            if isinstance(approx, FullRankGaussian):
                return eng.alpha_grad_fullrank(_NOISE_SLOT, n_local, approx.dim, var_param, alpha, n_total=n_total)
            family, df = approx._device_family()
            return eng.alpha_grad_meanfield(_NOISE_SLOT, n_local, approx.dim, var_param, family, alpha, df=df,
                                            n_total=n_total)

        self._objective_and_grad = objective_grad_and_log_norm

    def _lowrank_alpha(self, approx, alpha):
        """AlphaDivergence over the low-rank Gaussian: weights, value and the weighted sums on the device
        (``vb_alpha_sums_lowrank``), the capacitance-matrix algebra here.  Total derivative of
        lw_n = f(x_n(theta)) - log q(x_n(theta); theta) with x = mu + B z + sigma eps, t = M^-1 (z - Bs' eps),
        c = Bs t:  d/dmu = g,  d/d log_sigma = g sigma eps + sigma^2 diag(Sigma^-1) - c^2 - c eps,
        d/dB = g z' + ((c + eps) / sigma) t' + Sigma^-1 B."""
        D, k = approx.dim, approx.k

        def objective_grad_and_log_norm(var_param):
            var_param = np.asarray(var_param, dtype=np.float64)
            if var_param.shape != (approx.var_param_dim,):
                raise ValueError('var_param must have shape ({},)'.format(approx.var_param_dim))
            eng = self._engine()
            seed = _shared_randint(eng)                    # objectives.py:455
            eng.set_model(self.model.device_spec())
            N = self.num_mc_samples
            begin, end = shard_rows(N, eng.n_ranks, eng.rank)
            if approx.rng == 'philox':
                approx._philox_noise(eng, end - begin, seed, begin, _NOISE_SLOT, _LR_SLOT)
                _hint_next_seed(eng, (1 << _NOISE_SLOT) | (1 << _LR_SLOT))
            else:
                approx._stage_base_noise(eng, _NOISE_SLOT, N, begin, end, seed, slot_aux=_LR_SLOT)
            mu, ls, B, Bs, Minv, cq, BsMinv = _lowrank_pieces(approx, var_param)
            value, S, Sgz, Set, Stt, Sg, Sge = eng.alpha_sums_lowrank(_NOISE_SLOT, _LR_SLOT, end - begin, D, k, alpha,
                                                                      mu, ls, B, Minv, cq, n_total=N)
            sig = np.exp(ls)
            g_ls = (sig * Sge - np.sum((Bs @ Stt) * Bs, axis=1) - np.sum(Bs * Set, axis=1)
                    + S * (1.0 - np.sum(BsMinv * Bs, axis=1)))
            g_B = Sgz + (Bs @ Stt + Set + S * BsMinv) / sig[:, None]
            return value, alpha * np.concatenate([Sg, g_ls, g_B.reshape(-1)]) / N          # objectives.py:460

        return objective_grad_and_log_norm

    def _mvt_alpha(self, approx, alpha):
        """AlphaDivergence over the multivariate t: weights, value and the weighted sums on the device
        (``vb_alpha_sums_mvt``); the chain rule through the symmetric root as in ``ExclusiveKL._mvt_exclusive_kl``.
        Of ``log q(x(theta); theta)`` only ``-sum log L_ii`` moves (the Mahalanobis distance of a sample is a
        function of its noise), which puts ``sum w`` on the free diagonal."""
        D, df = approx.dim, approx.df
        tril = np.tril_indices(D)
        _lib.apply_host_blas_policy()

        def objective_grad_and_log_norm(var_param):
            var_param = np.asarray(var_param, dtype=np.float64)
            if var_param.shape != (approx.var_param_dim,):
                raise ValueError('var_param must have shape ({},)'.format(approx.var_param_dim))
            eng = self._engine()
            seed = _shared_randint(eng)                    # objectives.py:455
            eng.set_model(self.model.device_spec())
            N = self.num_mc_samples
            begin, end = shard_rows(N, eng.n_ranks, eng.rank)
            if approx.rng == 'philox':
                # throughput mode, as ExclusiveKL's: chi-square draws and normals on the device, samples through the
                # Cholesky factor x = mu + L z / s (same distribution as the symmetric root of approximations.py:348; no
                # reference stream is reproduced in this mode) -- the chain rule is tril(sum w g (z / s)') itself: no
                # matrix root, no Sylvester solve
                eng.chisq_generate(df, end - begin, seed, 0, row_offset=begin)
                eng.noise_generate(_NOISE_SLOT, end - begin, D, seed, 0, row_offset=begin)
                _hint_next_seed(eng, 1 << _NOISE_SLOT, with_chi=True)
                # the dense family's weighted pipeline with the rows scaled by 1 / s_n (vb_alpha_grad_mvt_chol): the
                # gradient arrives in the flat layout, alpha / N applied on the device
                return eng.alpha_grad_mvt_chol(_NOISE_SLOT, end - begin, D, df, var_param, alpha, n_total=N)
            else:
                # chi-square draws first (approximations.py:345-347)
                want_resident = D > _RESIDENT_GATE
                chi = approx._stage_base_noise(eng, _NOISE_SLOT, N, begin, end, seed, host_chi=not want_resident,
                                               device_chi=want_resident and D >= _RESIDENT_SMALL_N_MIN_DIM)
                if want_resident and getattr(approx, '_chi_on_device', False):
                    # the whole evaluation resident on the device (vb_alpha_grad_mvt_symroot), as ExclusiveKL's (its own
                    # buffer; sharded jobs: maximum and weighted sums all-reduced on the device); None: a root
                    # iteration did not resolve
                    resident = eng.alpha_grad_mvt_symroot(_NOISE_SLOT, end - begin, D, df, alpha, var_param, n_total=N)
                    if resident is not None:
                        return resident
                    chi = eng.chisq_get_host(N)          # the host route after all: the draws come down
            mu, L = approx._unpack(var_param)
            Sigma = L @ L.T
            inv_s = 1.0 / np.sqrt(chi / df)
            root, eig = _device_root(eng, Sigma)
            value, w_sum, g_sum, C = eng.alpha_sums_mvt(_NOISE_SLOT, end - begin, D, df, alpha, mu, root,
                                                        inv_s[begin:end], np.sum(np.log(np.diag(L))), n_total=N)
            Gs = 0.5 * (C + C.T)
            X = None
            if eig is None:
                _, X, info = eng.sym_sqrt(Sigma, Gs)
                if not info[2] < _ROOT_TOL:
                    X, eig = None, symmetric_eig(Sigma)
            if X is None:
                w, U = eig
                r = np.sqrt(w)
                X = U @ ((U.T @ Gs @ U) / (r[:, None] + r[None, :])) @ U.T
            dL = np.tril(2.0 * X @ L)
            dL[np.diag_indices(D)] = np.diag(dL) * np.diag(L) + w_sum
            return value, alpha * np.concatenate([g_sum, dL[tril]]) / N          # objectives.py:460

        return objective_grad_and_log_norm
