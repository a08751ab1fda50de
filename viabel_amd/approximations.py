"""Approximation families (host side of the reference's ``ApproximationFamily`` API).

API surface of ``viabel/approximations.py:26-182`` kept: ``init_param``, ``sample(var_param,
n_samples, seed=None)``, ``entropy``, ``kl``, ``log_density``, ``mean_and_cov``,
``pth_moment``, ``supports_pth_moment``, ``dim``, ``var_param_dim``, ``supports_entropy``,
``supports_kl``.  These are O(P) / O(N D) host conveniences used by optimisers and
diagnostics; the Monte-Carlo objective itself (``objective(var_param)``) never goes through
them -- it hands the family's *noise* and flat parameter to the HIP engine.

Flat parameter layouts (paragami 0.42 "free" flattening, ``approximations.py:185-189`` and
``:315-319``): ``[mu | log_sigma]`` for the mean-field families, ``[mu | vec(L)]`` for the
dense ones, where ``L`` is the Cholesky factor of the scale matrix with the *log* of its
diagonal, lower triangle in ``numpy.tril_indices`` order.

Noise sources.  ``rng='numpy'`` (default) reproduces the reference bit for bit: a persistent
``numpy.random.RandomState(seed)`` stream held by the family and advanced by every draw
(``approximations.py:203``, ``:213-216``) -- drawn by the library's own restatement of numpy's legacy generator
(``_legacy_rng.LegacyRandomState``: same values, same state, the big normal matrices on all host threads).  ``rng='philox'`` is the throughput mode: noise
is generated on the GPU by a counter-based Philox stream and never touches the host.
"""
from abc import ABC, abstractmethod

import numpy as np
from scipy import linalg as _sla
from scipy import special as _special

from . import _lib
from ._legacy_rng import LegacyRandomState

__all__ = ['ApproximationFamily', 'MFGaussian', 'MFStudentT', 'MultivariateT', 'FullRankGaussian', 'LRGaussian']

_LOG_2PI = float(np.log(2.0 * np.pi))


class ApproximationFamily(ABC):
    """Abstract variational family (``viabel/approximations.py:26-182``)."""

    def __init__(self, dim, var_param_dim, supports_entropy, supports_kl):
        self._dim = dim
        self._var_param_dim = var_param_dim
        self._supports_entropy = supports_entropy
        self._supports_kl = supports_kl

    def init_param(self):
        return np.zeros(self.var_param_dim)

    @abstractmethod
    def sample(self, var_param, n_samples, seed=None):
        """Draw ``(n_samples, dim)`` samples from the variational distribution."""

    def entropy(self, var_param):
        if self.supports_entropy:
            return self._entropy(var_param)
        raise NotImplementedError()

    def _entropy(self, var_param):
        raise NotImplementedError()

    @property
    def supports_entropy(self):
        return self._supports_entropy

    def kl(self, var_param0, var_param1):
        if self.supports_kl:
            return self._kl(var_param0, var_param1)
        raise NotImplementedError()

    def _kl(self, var_param0, var_param1):
        raise NotImplementedError()

    @property
    def supports_kl(self):
        return self._supports_kl

    @abstractmethod
    def log_density(self, var_param, x):
        """Log density of the variational distribution at ``x``."""

    @abstractmethod
    def mean_and_cov(self, var_param):
        """Mean and covariance of the variational distribution."""

    def pth_moment(self, var_param, p):
        if self.supports_pth_moment(p):
            return self._pth_moment(var_param, p)
        raise ValueError('p = {} is not a supported moment'.format(p))

    @abstractmethod
    def _pth_moment(self, var_param, p):
        """Absolute p-th moment."""

    @abstractmethod
    def supports_pth_moment(self, p):
        """Whether the p-th moment is available in closed form."""

    @property
    def dim(self):
        return self._dim

    @property
    def var_param_dim(self):
        return self._var_param_dim


class _NoiseMixin:
    """Noise bookkeeping shared by the device-backed families."""

    def _init_rng(self, seed, rng):
        if rng not in ('numpy', 'philox'):
            raise ValueError("rng must be 'numpy' or 'philox'")
        self._seed = seed
        self._rng_kind = rng
        self._rs = LegacyRandomState(seed)
        self._philox_calls = 0

    @property
    def rng(self):
        return self._rng_kind

    def _random_state(self, seed):
        return self._rs if seed is None else LegacyRandomState(seed)

    def _next_philox_stream(self):
        k = self._philox_calls
        self._philox_calls += 1
        return k

    # -- parity-mode noise into a device slot ---------------------------------------------------------------------------
    # values; below these the host draw + upload beats the device path's ~20 launches (tools/legacy_draw_gate_probe.py, late
    # round 5: randn host 83 / device 71 us at 8 192 values, 147 / 81 at 16 384, 453 / 100 at 32 768; standard_t 130 / 117
    # at 4 096, 231 / 145 at 8 192, 1 664 / 224 at 65 536 -- the gate stood at 65 536 before the draws lost their host
    # round trips)
    _DEVICE_DRAW_FROM = 1 << 13
    _DEVICE_T_FROM = 1 << 12
    _DEVICE_CHI_FROM = 1 << 12       # chi-square draws; the sequential host loop costs ~60 ns a draw, the device path ~0.2 ms

    def _stage_normals(self, eng, rs, slot, n_total, d, begin, end):
        """Rows ``[begin, end)`` of ``rs.randn(n_total, d)`` into ``slot``.  Big draws are generated ON THE DEVICE from the
        generator's own state, bit for bit numpy's stream (``vb_legacy_rng_randn_device``: MT19937 in parallel streams,
        the polar method's attempts and their prefix sums as kernels), which leaves ``rs`` where ``randn`` would have left
        it; small ones, and anything outside the device path's range, are drawn on the host and uploaded."""
        if (n_total * d >= self._DEVICE_DRAW_FROM and isinstance(rs, LegacyRandomState)
                and eng.noise_legacy_randn(slot, rs._h, n_total, d, begin, end - begin)):
            return
        eng.noise_set_host(slot, rs.randn(n_total, d)[begin:end])

    def _round_end(self, eng, rs, seed):
        """This call's draws from the family's PERSISTENT generator are done: the engine may start the next call's -- the
        same requests from the generator's current state -- beside the objective's kernels (``vb_legacy_round_end``).  A
        fresh ``RandomState(seed)`` (``seed=`` calls, AlphaDivergence) is used once: nothing to look ahead to."""
        if seed is None and isinstance(rs, LegacyRandomState):
            eng.legacy_round_end(rs._h)

    def _stage_base_noise(self, eng, slot, n_total, begin, end, seed=None, slot_aux=None):
        """This rank's rows of the family's base noise (what ``sample`` would consume) into ``slot``; returns what stays
        on the host (nothing here).  Default: host draw + upload."""
        eng.noise_set_host(slot, self._base_noise(n_total, seed)[begin:end])
        return None

    def _philox_kind(self):
        """(noise kind, df) of the family's base noise for the device generator."""
        if getattr(self, '_family_id', None) == _lib.FAMILY_MF_STUDENT_T:
            return _lib.NOISE_STUDENT_T, float(self._df)
        return _lib.NOISE_NORMAL, 0.0


def _as_rows(x):
    x = np.asarray(x, dtype=np.float64)
    return x[np.newaxis, :] if x.ndim == 1 else x


# ------------------------------------------------------------------------------------------
class MFGaussian(_NoiseMixin, ApproximationFamily):
    """Mean-field Gaussian, ``var_param = [mu | log_sigma]`` (``approximations.py:192-251``)."""

    _family_id = _lib.FAMILY_MF_GAUSSIAN

    def __init__(self, dim, seed=1, rng='numpy'):
        self._init_rng(seed, rng)
        super().__init__(dim, 2 * dim, True, True)

    # -- engine hooks ------------------------------------------------------------------------
    def _device_family(self):
        return self._family_id, 0.0

    def _base_noise(self, n_samples, seed=None):
        """The base draws the reference's ``sample`` would consume (``:216``)."""
        rs = self._random_state(seed)
        noise = _legacy_host_copy(rs, 'n', 0.0, n_samples, self.dim)
        return rs.randn(n_samples, self.dim) if noise is None else noise

    def _stage_base_noise(self, eng, slot, n_total, begin, end, seed=None, slot_aux=None):
        rs = self._random_state(seed)
        self._stage_normals(eng, rs, slot, n_total, self.dim, begin, end)
        self._round_end(eng, rs, seed)

    def _unpack(self, var_param):
        var_param = np.asarray(var_param, dtype=np.float64)
        return var_param[:self.dim], var_param[self.dim:]

    # -- reference API -----------------------------------------------------------------------
    def init_param(self):
        return np.concatenate([np.zeros(self.dim), np.full(self.dim, 2.0)])

    def sample(self, var_param, n_samples, seed=None):
        mu, log_sigma = self._unpack(var_param)
        if self._rng_kind == 'philox':
            noise = _philox_host_copy(self, n_samples, seed)
        else:
            noise = self._base_noise(n_samples, seed)
        return mu + np.exp(log_sigma) * noise

    def _entropy(self, var_param):
        _, log_sigma = self._unpack(var_param)
        return 0.5 * self.dim * (1.0 + _LOG_2PI) + np.sum(log_sigma)

    def _kl(self, var_param0, var_param1):
        mu0, ls0 = self._unpack(var_param0)
        mu1, ls1 = self._unpack(var_param1)
        dls = ls0 - ls1
        return 0.5 * np.sum(np.exp(2 * dls) + (mu0 - mu1) ** 2 * np.exp(-2 * ls1) - 2 * dls - 1)

    def log_density(self, var_param, x):
        mu, log_sigma = self._unpack(var_param)
        r = (_as_rows(x) - mu) * np.exp(-log_sigma)
        return np.sum(-0.5 * r * r - log_sigma - 0.5 * _LOG_2PI, axis=-1)

    def mean_and_cov(self, var_param):
        mu, log_sigma = self._unpack(var_param)
        return mu, np.diag(np.exp(2 * log_sigma))

    def _pth_moment(self, var_param, p):
        _, log_sigma = self._unpack(var_param)
        var = np.exp(2 * log_sigma)
        if p == 2:
            return np.sum(var)
        return 2 * np.sum(var ** 2) + np.sum(var) ** 2

    def supports_pth_moment(self, p):
        return p in [2, 4]


class MFStudentT(_NoiseMixin, ApproximationFamily):
    """Mean-field Student's t, ``var_param = [mu | log_sigma]`` (``approximations.py:254-312``)."""

    _family_id = _lib.FAMILY_MF_STUDENT_T

    def __init__(self, dim, df, seed=1, rng='numpy'):
        if df <= 2:
            raise ValueError('df must be greater than 2')
        self._df = df
        self._init_rng(seed, rng)
        super().__init__(dim, 2 * dim, True, False)

    def _device_family(self):
        return self._family_id, float(self._df)

    def _base_noise(self, n_samples, seed=None):
        rs = self._random_state(seed)
        noise = _legacy_host_copy(rs, 't', self.df, n_samples, self.dim)
        return rs.standard_t(self.df, size=(n_samples, self.dim)) if noise is None else noise   # :273-274

    def _stage_base_noise(self, eng, slot, n_total, begin, end, seed=None, slot_aux=None):
        """Rows ``[begin, end)`` of ``standard_t(df, (n_total, dim))`` into ``slot``: big draws ON THE DEVICE from the
        generator's own state, values and state bit for bit numpy's (``vb_legacy_rng_standard_t_device``); small ones, and
        anything the device path does not take, on the host."""
        rs = self._random_state(seed)
        if (n_total * self.dim >= self._DEVICE_T_FROM and isinstance(rs, LegacyRandomState)
                and eng.noise_legacy_standard_t(slot, rs._h, self.df, n_total, self.dim, begin, end - begin)):
            self._round_end(eng, rs, seed)
            return None
        eng.noise_set_host(slot, rs.standard_t(self.df, size=(n_total, self.dim))[begin:end])
        return None

    def _unpack(self, var_param):
        var_param = np.asarray(var_param, dtype=np.float64)
        return var_param[:self.dim], var_param[self.dim:]

    def init_param(self):
        return np.concatenate([np.zeros(self.dim), np.full(self.dim, 2.0)])

    def sample(self, var_param, n_samples, seed=None):
        mu, log_sigma = self._unpack(var_param)
        if self._rng_kind == 'philox':
            noise = _philox_host_copy(self, n_samples, seed)
        else:
            noise = self._base_noise(n_samples, seed)
        return mu + np.exp(log_sigma) * noise

    def entropy(self, var_param):
        # the reference drops the df-only constants (approximations.py:276-279)
        _, log_sigma = self._unpack(var_param)
        return np.sum(log_sigma)

    def log_density(self, var_param, x):
        mu, log_sigma = self._unpack(var_param)
        df = self.df
        r = (_as_rows(x) - mu) * np.exp(-log_sigma)
        const = (_special.gammaln(0.5 * (df + 1)) - _special.gammaln(0.5 * df)
                 - 0.5 * np.log(df * np.pi))
        return np.sum(const - 0.5 * (df + 1) * np.log1p(r * r / df) - log_sigma, axis=-1)

    def mean_and_cov(self, var_param):
        mu, log_sigma = self._unpack(var_param)
        return mu, self.df / (self.df - 2) * np.diag(np.exp(2 * log_sigma))

    def _pth_moment(self, var_param, p):
        df = self.df
        if df <= p:
            raise ValueError('df must be greater than p')
        _, log_sigma = self._unpack(var_param)
        s2 = np.exp(2 * log_sigma)
        c = df / (df - 2)
        if p == 2:
            return c * np.sum(s2)
        return c ** 2 * (2 * (df - 1) / (df - 4) * np.sum(s2 ** 2) + np.sum(s2) ** 2)

    def supports_pth_moment(self, p):
        return p in [2, 4] and p < self.df

    @property
    def df(self):
        return self._df


# ------------------------------------------------------------------------------------------
def _chol_from_free(vec, dim):
    L = np.zeros((dim, dim))
    L[np.tril_indices(dim)] = vec
    L[np.diag_indices(dim)] = np.exp(np.diag(L))
    return L


def _free_from_chol(L):
    dim = L.shape[0]
    out = np.array(L, dtype=np.float64)
    out[np.diag_indices(dim)] = np.log(np.diag(L))
    return out[np.tril_indices(dim)]


class FullRankGaussian(_NoiseMixin, ApproximationFamily):
    """Gaussian with a dense covariance ``Sigma = L L'``; ``var_param = [mu | vec(L)]``.

    Not in the reference: it has no dense *Gaussian* family.  Layout follows the
    reference's other dense family (``approximations.py:315-319``); ``z = mu + L eps``.
    """

    _family_id = _lib.FAMILY_FULLRANK_GAUSSIAN

    def __init__(self, dim, seed=1, rng='numpy'):
        self._init_rng(seed, rng)
        super().__init__(dim, dim + dim * (dim + 1) // 2, True, True)

    def _device_family(self):
        return self._family_id, 0.0

    def _base_noise(self, n_samples, seed=None):
        rs = self._random_state(seed)
        noise = _legacy_host_copy(rs, 'n', 0.0, n_samples, self.dim)
        return rs.randn(n_samples, self.dim) if noise is None else noise

    def _stage_base_noise(self, eng, slot, n_total, begin, end, seed=None, slot_aux=None):
        rs = self._random_state(seed)
        self._stage_normals(eng, rs, slot, n_total, self.dim, begin, end)
        self._round_end(eng, rs, seed)

    def _unpack(self, var_param):
        var_param = np.asarray(var_param, dtype=np.float64)
        return var_param[:self.dim], _chol_from_free(var_param[self.dim:], self.dim)

    def pack(self, mu, L):
        """Flat parameter from a mean and a lower-triangular factor with positive diagonal."""
        return np.concatenate([np.asarray(mu, dtype=np.float64), _free_from_chol(np.asarray(L))])

    def init_param(self):
        return self.pack(np.zeros(self.dim), np.exp(2.0) * np.eye(self.dim))

    def sample(self, var_param, n_samples, seed=None):
        mu, L = self._unpack(var_param)
        if self._rng_kind == 'philox':
            noise = _philox_host_copy(self, n_samples, seed)
        else:
            noise = self._base_noise(n_samples, seed)
        return mu + noise @ L.T

    def _entropy(self, var_param):
        _, L = self._unpack(var_param)
        return 0.5 * self.dim * (1.0 + _LOG_2PI) + np.sum(np.log(np.diag(L)))

    def _kl(self, var_param0, var_param1):
        mu0, L0 = self._unpack(var_param0)
        mu1, L1 = self._unpack(var_param1)
        A = _sla.solve_triangular(L1, L0, lower=True)
        dm = _sla.solve_triangular(L1, mu1 - mu0, lower=True)
        return (0.5 * (np.sum(A * A) + dm @ dm - self.dim)
                + np.sum(np.log(np.diag(L1))) - np.sum(np.log(np.diag(L0))))

    def log_density(self, var_param, x):
        mu, L = self._unpack(var_param)
        e = _sla.solve_triangular(L, (_as_rows(x) - mu).T, lower=True).T
        return -0.5 * np.sum(e * e, axis=-1) - np.sum(np.log(np.diag(L))) - 0.5 * self.dim * _LOG_2PI

    def mean_and_cov(self, var_param):
        mu, L = self._unpack(var_param)
        return mu, L @ L.T

    def _pth_moment(self, var_param, p):
        _, L = self._unpack(var_param)
        S = L @ L.T
        if p == 2:
            return np.trace(S)
        return 2 * np.sum(S * S) + np.trace(S) ** 2

    def supports_pth_moment(self, p):
        return p in [2, 4]


def symmetric_eig(S):
    """``eigh`` of a small symmetric matrix on one BLAS thread (see ``_lib.small_lapack``)."""
    with _lib.small_lapack(S.shape[0]):
        return np.linalg.eigh(S)


def symmetric_root(S):
    """The symmetric square root the reference takes with ``scipy.linalg.sqrtm`` (``approximations.py:348``),
    through ``eigh``: the same matrix to rounding (3e-15 at D=256) in a twelfth of the time (10 vs 120 ms)."""
    w, U = symmetric_eig(S)
    return (U * np.sqrt(w)) @ U.T


class MultivariateT(_NoiseMixin, ApproximationFamily):
    """Full-rank multivariate t, ``var_param = [mu | vec(chol Sigma)]``
    (``approximations.py:322-382``, log pdf ``_distributions.py:7-38``)."""

    _family_id = _lib.FAMILY_MULTIVARIATE_T

    def __init__(self, dim, df, seed=1, rng='numpy'):
        if df <= 2:
            raise ValueError('df must be greater than 2')
        self._df = df
        self._init_rng(seed, rng)
        super().__init__(dim, dim + dim * (dim + 1) // 2, True, False)

    def _device_family(self):
        return self._family_id, float(self._df)

    def _base_noise(self, n_samples, seed=None):
        """Chi-square draws FIRST, then the normals (``approximations.py:345-347``)."""
        rs = self._random_state(seed)
        chi = rs.chisquare(self.df, n_samples)
        z = _legacy_host_copy(rs, 'n', 0.0, n_samples, self.dim)
        return chi, (rs.randn(n_samples, self.dim) if z is None else z)

    def _stage_base_noise(self, eng, slot, n_total, begin, end, seed=None, slot_aux=None, host_chi=True, device_chi=False):
        """The normals into ``slot``; returns the chi-square draws (all ``n_total`` of them).  ``host_chi=False``: a caller
        that means to stay on the device takes None when the draws were generated there (``_chi_on_device``; they are
        resident in the context, ``eng.chisq_get_host(n_total)`` fetches them should the host route be needed after all).
        ``device_chi``: draw them on the device below ``_DEVICE_CHI_FROM`` too (slower than the host loop by ~50 us there,
        but the resident routes need them on the device and save far more at larger dimensions)."""
        rs = self._random_state(seed)
        chi = None                                         # first, as ``sample`` draws them (:345-347)
        on_device = False
        if (n_total >= self._DEVICE_CHI_FROM or device_chi) and isinstance(rs, LegacyRandomState):
            # on the device, bit for bit numpy's (None: not this path's case)
            chi = eng.chisq_legacy(rs._h, self.df, n_total, to_host=host_chi)
            on_device = chi is not None
            if on_device and not host_chi:
                chi = None
        self._chi_on_device = on_device                    # ... and resident in the context (vb_dis_refresh_mvt_symroot)
        if chi is None and not on_device:
            chi = rs.chisquare(self.df, n_total)
        self._stage_normals(eng, rs, slot, n_total, self.dim, begin, end)
        self._round_end(eng, rs, seed)
        return chi

    def _unpack(self, var_param):
        var_param = np.asarray(var_param, dtype=np.float64)
        L = _chol_from_free(var_param[self.dim:], self.dim)
        return var_param[:self.dim], L

    def init_param(self):
        return np.concatenate([np.zeros(self.dim),
                               _free_from_chol(np.sqrt(10.0) * np.eye(self.dim))])

    def sample(self, var_param, n_samples, seed=None):
        if self._rng_kind == 'philox':          # chi-square on the host (N draws), normals on the GPU
            chi = self._random_state(seed).chisquare(self.df, n_samples)
            z = _philox_host_copy(self, n_samples, seed)
        else:
            chi, z = self._base_noise(n_samples, seed)
        mu, L = self._unpack(var_param)
        return mu + (z @ symmetric_root(L @ L.T)) / np.sqrt(chi / self.df)[:, np.newaxis]

    def entropy(self, var_param):
        # df-only constants dropped, as the reference does (:351-354)
        _, L = self._unpack(var_param)
        return np.sum(np.log(np.diag(L)))

    def log_density(self, var_param, x):
        mu, L = self._unpack(var_param)
        df, d = self.df, self.dim
        e = _sla.solve_triangular(L, (_as_rows(x) - mu).T, lower=True).T
        maha = np.sum(e * e, axis=-1)
        const = (_special.gammaln(0.5 * (df + d)) - _special.gammaln(0.5 * df)
                 - 0.5 * d * np.log(np.pi * df) - np.sum(np.log(np.diag(L))))
        return const - 0.5 * (df + d) * np.log1p(maha / df)

    def mean_and_cov(self, var_param):
        mu, L = self._unpack(var_param)
        return mu, self.df / (self.df - 2.) * (L @ L.T)

    def _pth_moment(self, var_param, p):
        df = self.df
        if df <= p:
            raise ValueError('df must be greater than p')
        _, L = self._unpack(var_param)
        with _lib.small_lapack(self.dim):
            ev = np.linalg.eigvalsh(L @ L.T)
        c = df / (df - 2)
        if p == 2:
            return c * np.sum(ev)
        return c ** 2 * (2 * (df - 1) / (df - 4) * np.sum(ev ** 2) + np.sum(ev) ** 2)

    def supports_pth_moment(self, p):
        return p in [2, 4] and p < self.df

    @property
    def df(self):
        return self._df


class LRGaussian(_NoiseMixin, ApproximationFamily):
    """Gaussian with covariance ``B B' + diag(sigma^2)`` (``viabel/approximations.py:610-731``).

    ``var_param = [mu (D) | log_sigma (D) | B (D x k, row-major)]`` (``:551-556``).  All dense algebra goes
    through the k x k capacitance matrix ``M = I + B' D^-1 B`` (Woodbury / determinant lemma), as in the
    reference's helpers (``:559-607``).
    """

    _family_id = _lib.FAMILY_LOWRANK_GAUSSIAN

    def __init__(self, dim, seed=1, k=0, rng='numpy'):
        self._init_rng(seed, rng)
        self._k = int(k)
        super().__init__(dim, 2 * dim + dim * self._k, True, True)

    @property
    def k(self):
        return self._k

    def _unpack(self, var_param):
        v = np.asarray(var_param, dtype=np.float64)
        D, k = self.dim, self._k
        return v[:D], v[D:2 * D], v[2 * D:].reshape(D, k)

    def pack(self, mu, log_sigma, B):
        return np.concatenate([np.asarray(mu, dtype=np.float64), np.asarray(log_sigma, dtype=np.float64),
                               np.asarray(B, dtype=np.float64).reshape(-1)])

    def _base_noise(self, n_samples, seed=None):
        """``(z, eps)``: the low-rank block is drawn first (``:639-640``)."""
        if self._rng_kind == 'philox':
            return self._philox_noise(_lib.default_engine(), n_samples, seed, 0, _lib.MAX_SLOTS - 1,
                                      _lib.MAX_SLOTS - 2, read_back=True)
        rs = self._random_state(seed)
        z = rs.randn(n_samples, self._k)
        return z, rs.randn(n_samples, self.dim)

    def _stage_base_noise(self, eng, slot, n_total, begin, end, seed=None, slot_aux=None):
        """The low-rank block (drawn first, ``:639-640``) into ``slot_aux``, the diagonal block into ``slot``."""
        if self._rng_kind == 'philox':
            self._philox_noise(eng, end - begin, seed, begin, slot, slot_aux)
            return None
        rs = self._random_state(seed)
        if self._k > 0:
            self._stage_normals(eng, rs, slot_aux, n_total, self._k, begin, end)
        self._stage_normals(eng, rs, slot, n_total, self.dim, begin, end)
        self._round_end(eng, rs, seed)
        return None

    def _device_family(self):
        return self._family_id, 0.0

    def _philox_noise(self, eng, n_rows, seed, row_offset, slot_eps, slot_z, read_back=False):
        """Throughput mode: call c of the family fills the n x D block from Philox stream 2 c and the n x k block
        from stream 2 c + 1 (the convention of ``vb_fit``); an explicit ``seed`` uses streams 0 / 1 of that seed."""
        if seed is None:
            seed, c = self._seed, self._next_philox_stream()
        else:
            c = 0
        eng.noise_generate(slot_eps, n_rows, self.dim, seed, 2 * c, row_offset=row_offset)
        if self._k > 0:
            eng.noise_generate(slot_z, n_rows, self._k, seed, 2 * c + 1, row_offset=row_offset)
        if not read_back:
            return None
        z = eng.noise_get_host(slot_z, n_rows, self._k) if self._k > 0 else np.zeros((n_rows, 0))
        return z, eng.noise_get_host(slot_eps, n_rows, self.dim)

    def init_param(self):          # :630-634 (advances the family's stream by D k draws)
        return self.pack(np.zeros(self.dim), np.ones(self.dim), self._rs.randn(self.dim, self._k))

    def sample(self, var_param, n_samples, seed=None):
        mu, ls, B = self._unpack(var_param)
        z, eps = self._base_noise(n_samples, seed)
        return mu + z @ B.T + np.exp(ls) * eps

    def _capacitance(self, ls, B):
        w = B * np.exp(-2.0 * ls)[:, np.newaxis]                 # D^-1 B
        return w, np.eye(self._k) + B.T @ w

    def _log_det(self, ls, B):
        _, M = self._capacitance(ls, B)
        return 2.0 * np.sum(ls) + np.linalg.slogdet(M)[1]

    def _entropy(self, var_param):
        _, ls, B = self._unpack(var_param)
        return 0.5 * self.dim * (_LOG_2PI + 1.0) + 0.5 * self._log_det(ls, B)

    def _solve(self, ls, B, rhs):
        """``Sigma^-1 rhs`` for ``rhs`` of shape (D, m)."""
        w, M = self._capacitance(ls, B)
        y = rhs * np.exp(-2.0 * ls)[:, np.newaxis]
        return y - w @ np.linalg.solve(M, w.T @ rhs)

    def _kl(self, var_param0, var_param1):
        mu0, ls0, B0 = self._unpack(var_param0)
        mu1, ls1, B1 = self._unpack(var_param1)
        dm = (mu0 - mu1)[:, np.newaxis]
        # tr(Sigma1^-1 Sigma0) = tr(Sigma1^-1 diag(s0^2)) + tr(B0' Sigma1^-1 B0)
        w1, M1 = self._capacitance(ls1, B1)
        inv_diag = np.exp(-2.0 * ls1) - np.sum(w1 * np.linalg.solve(M1, w1.T).T, axis=1)
        trace = np.sum(inv_diag * np.exp(2.0 * ls0)) + np.sum(B0 * self._solve(ls1, B1, B0))
        maha = (dm.T @ self._solve(ls1, B1, dm)).item()
        return 0.5 * (self._log_det(ls1, B1) - self._log_det(ls0, B0) - self.dim + maha + trace)

    def log_density(self, var_param, x):
        mu, ls, B = self._unpack(var_param)
        diff = (_as_rows(x) - mu).T
        maha = np.sum(diff * self._solve(ls, B, diff), axis=0)
        return -0.5 * (self.dim * _LOG_2PI + self._log_det(ls, B) + maha)

    def mean_and_cov(self, var_param):
        mu, ls, B = self._unpack(var_param)
        return mu, B @ B.T + np.diag(np.exp(2.0 * ls))

    def _pth_moment(self, var_param, p):
        ev = np.linalg.eigvalsh(self.mean_and_cov(var_param)[1])
        return np.sum(ev) if p == 2 else 2 * np.sum(ev ** 2) + np.sum(ev) ** 2

    def supports_pth_moment(self, p):
        return p in [2, 4]


# values: from here on a host `sample()` draws its legacy noise on the GPU and reads it back (host randn 453 us against 100 us
# on the device + a 256-KB copy at 32 768 values, tools/legacy_draw_gate_probe.py; the gate stood at 262 144 while a
# device draw still cost several host round trips)
_HOST_SAMPLE_DEVICE_FROM = 1 << 15


def _legacy_host_copy(rs, kind, df, n_samples, dim):
    """``rs.randn(n, d)`` / ``rs.standard_t(df, (n, d))`` for a HOST caller (``sample``, ``vi_diagnostics``: 10^5 draws),
    generated on the device from the generator's own state -- values and state bit for bit numpy's -- and read back; None
    when there is no engine, the draw is small, or the device path declines (the caller draws on the host)."""
    if n_samples * dim < _HOST_SAMPLE_DEVICE_FROM or not isinstance(rs, LegacyRandomState):
        return None
    try:
        eng = _lib.default_engine()
    except Exception:
        return None
    slot = _lib.MAX_SLOTS - 1
    ok = (eng.noise_legacy_standard_t(slot, rs._h, df, n_samples, dim) if kind == 't'
          else eng.noise_legacy_randn(slot, rs._h, n_samples, dim))
    return eng.noise_get_host(slot, n_samples, dim) if ok else None


def _philox_host_copy(family, n_samples, seed):
    """Generate Philox noise on the GPU and read it back (host ``sample`` in throughput mode)."""
    eng = _lib.default_engine()
    slot = _lib.MAX_SLOTS - 1
    kind, df = family._philox_kind()
    if seed is None:
        eng.noise_generate(slot, n_samples, family.dim, family._seed, family._next_philox_stream(), kind=kind, df=df)
    else:
        eng.noise_generate(slot, n_samples, family.dim, seed, 0, kind=kind, df=df)
    return eng.noise_get_host(slot, n_samples, family.dim)
