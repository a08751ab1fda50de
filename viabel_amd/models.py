"""Target models.

Mirrors ``viabel/models.py`` (``Model`` ``:11-77``).  The reference model is an arbitrary
Python callable differentiated by autograd; a GPU engine needs targets whose log density
and derivatives exist as device code, so the hot path accepts
:class:`DeviceModel` instances.  Calling a device model evaluates ``f(x_n)`` on the GPU
through the C ABI (``vb_model_logp``) -- same contract as ``Model.__call__``
(``models.py:27-39``): ``(N, D) -> (N,)``, and ``(D,)`` is promoted to one row
(``viabel/tests/test_models.py:14-15``).
"""
import numpy as np

from . import _lib

__all__ = ['Model', 'DeviceModel', 'GaussianModel', 'FunnelModel', 'CorrelatedGaussianModel',
           'LogisticRegressionModel', 'PoissonRegressionModel', 'LinearRegressionModel', 'SourceModel',
           'CallableModel']


class Model(object):
    """Base class for representing a model (``viabel/models.py:11-77``).

    A plain ``Model(log_density)`` wraps a host callable.  The objectives bind it as a
    :class:`CallableModel` (the callable is evaluated on the host between the device's sampling and
    reduction kernels, its gradient by central differences); the built-in device models and
    :class:`SourceModel` are the fast routes.
    """

    def __init__(self, log_density):
        self._log_density = log_density

    def __call__(self, model_param):
        return self._log_density(model_param)

    def constrain(self, model_param):
        raise NotImplementedError()

    @property
    def supports_tempering(self):
        return False

    def set_inverse_temperature(self, inverse_temp):
        raise NotImplementedError()


class DeviceModel(Model):
    """A target whose log density / gradient / Hessian products are HIP device code."""

    def __init__(self, dim):
        self._dim = int(dim)
        super().__init__(self._device_log_density)

    @property
    def dim(self):
        return self._dim

    def device_spec(self):
        """``(model_id, dim, dparams, iparams)`` for ``vb_set_model`` (built once, then cached)."""
        spec = getattr(self, '_spec_cache', None)
        if spec is None:
            spec = self._spec_cache = self._build_spec()
        return spec

    def _build_spec(self):
        raise NotImplementedError()

    def _device_log_density(self, x):
        x = np.asarray(x, dtype=np.float64)
        one = x.ndim == 1
        if one:
            x = x[np.newaxis, :]
        if x.ndim != 2 or x.shape[1] != self._dim:
            raise ValueError('model_param must have shape (N, {0}) or ({0},)'.format(self._dim))
        eng = _lib.default_engine()
        eng.set_model(self.device_spec())
        return eng.model_logp(x)

    def _rows(self, x):
        x = np.asarray(x, dtype=np.float64)
        one = x.ndim == 1
        if one:
            x = x[np.newaxis, :]
        if x.ndim != 2 or x.shape[1] != self._dim:
            raise ValueError('model_param must have shape (N, {0}) or ({0},)'.format(self._dim))
        return np.ascontiguousarray(x), one

    def grad(self, x):
        """Gradient of the log density at each row of ``x`` -- what ``autograd.elementwise_grad(model)`` gives in the
        reference (``objectives.py:191``; ``tests/test_models.py:13-15`` checks it) -- from the device code the
        objectives use (``vb_model_grad``).  ``x``: (N, D) or (D,); returns the same shape."""
        x, one = self._rows(x)
        eng = _lib.default_engine()
        eng.set_model(self.device_spec())
        g = eng.model_grad(x)[1]
        return g[0] if one else g

    def check_gradient(self, x, step=1e-6):
        """Largest deviation of the device gradient from central differences of the device log density at the rows of
        ``x``, relative to the largest gradient entry there: the check the reference runs on its models with
        ``autograd.test_util.check_vjp`` (``tests/test_models.py:13-15``).  For a ``SourceModel`` the gradient is
        hand-written code and nothing else verifies it: a value above ~1e-6 means ``grad`` does not match ``f``."""
        x, _ = self._rows(x)
        n, d = x.shape
        h = step * np.maximum(1.0, np.abs(x))                          # (N, D) per-coordinate steps
        pts = np.repeat(x[:, np.newaxis, :], 2 * d, axis=1)          # (N, 2D, D): x + h e_j, then x - h e_j
        idx = np.arange(d)
        pts[:, idx, idx] += h
        pts[:, d + idx, idx] -= h
        f = self(pts.reshape(n * 2 * d, d)).reshape(n, 2 * d)
        fd = (f[:, :d] - f[:, d:]) / (2.0 * h)
        g = self.grad(x)
        return float(np.max(np.abs(g - fd)) / max(np.max(np.abs(g)), np.max(np.abs(fd)), 1e-300))


class SourceModel(DeviceModel):
    """A log density outside the built-in set, given as HIP device code -- the adaptor for what the reference
    does with an arbitrary Python callable and autograd (``viabel/models.py:17-39``,
    ``convenience.py:75`` ``bbvi(dim, log_density=...)``).

    ``source`` is HIP C++ that defines::

        __device__ double vb_log_density(const double* z, int d, const double* params, double* grad);

    returning ``f(z)`` for one sample ``z[0..d)`` and writing its gradient to ``grad[0..d)`` unless ``grad`` is
    NULL; ``params`` is the array given here (data, hyper-parameters), resident on the device.  With
    ``grad='auto'`` the source gives only the density, generic in its scalar type --
    ``template <class T> __device__ T vb_log_density(vb::vec<T> z, int d, const double* params);`` (``z[j]`` is a ``T``;
    ``+ - * /``, comparisons, ``log exp sqrt log1p expm1 pow tanh sin cos atan erf fabs fmin fmax lgamma`` work on ``T``
    and mix with ``double``; ``vb::dot(row, z, d)`` is the inner product of a ``const double*`` row with the whole sample
    as one operation -- the cheap way to write a regression's linear predictor) -- and the engine differentiates it like
    autograd does the reference's callable
    (``models.py:17-39``): forward-mode dual numbers carrying 8 derivatives in registers, ``ceil(dim / 8)`` threads
    per sample, exact to rounding.  The source is
    compiled for the GPU with hiprtc when the model is first bound; a source that does not compile raises
    ``ValueError`` with the compiler's log.  One thread evaluates one sample; a density that is a sum over data
    can have K threads per sample instead (``dim <= 128``): ``#define VB_LOG_DENSITY_PARTS K`` (a power of two up to
    64) and define ``vb_log_density_part(z, d, params, grad, part, n_parts)`` returning the share of ``f`` and adding
    the share of the gradient (``grad`` arrives zeroed) of, say, the observations ``part, part + K, ...`` with the
    prior in part 0 -- several times faster for a few hundred observations (``DESIGN.md`` 4.8).  ``ExclusiveKL`` (both estimator forms; with the mean-field families also the four RGE control variates,
    the model's Hessian terms taken as central differences of its device gradient),
    ``AlphaDivergence`` and ``DISInclusiveKL`` take it with every family; the model can be called on host samples, and ``vi_diagnostics`` forms its importance weights on the device."""

    def __init__(self, dim, source, params=None, grad='explicit'):
        if not isinstance(source, (str, bytes)) or not source:
            raise ValueError('source must be a non-empty string of HIP code')
        if grad not in ('explicit', 'auto'):
            raise ValueError("grad must be 'explicit' (the source writes the gradient) or 'auto'")
        self._source = source.encode() if isinstance(source, str) else bytes(source)
        if grad == 'auto':
            # the density alone, generic in its scalar type:
            #     template <class T> __device__ T vb_log_density(vb::vec<T> z, int d, const double* params);
            # differentiated on the device by forward-mode dual numbers, ceil(dim / 8) threads per sample
            self._source = b'#define VB_AUTO_GRAD 1\n' + self._source
        self.grad_mode = grad
        self.params = np.ascontiguousarray(np.zeros(0) if params is None else params, dtype=np.float64).ravel()
        super().__init__(dim)

    def _build_spec(self):
        return (_lib.MODEL_SOURCE, self._dim, self.params, np.zeros(0, dtype=np.int64), self._source)


class CallableModel(DeviceModel):
    """A target that exists only as a host (Python) callable -- the reference's front door ``Model(log_density)``
    (``viabel/models.py:17-39``, ``convenience.py:69-75``) and, with ``grad_log_density``, the contract of its
    ``StanModel`` (``models.py:80-104``: ``log_prob`` with ``grad_log_prob`` as its vector-Jacobian product).

    ``log_density``: ``(N, D) ndarray -> (N,)``; ``grad_log_density``: ``(N, D) -> (N, D)``.  Alternatively
    ``value_and_grad``: ``(N, D) -> ((N,), (N, D))`` in one call.  Without a gradient the engine differentiates the
    callable numerically (central differences, ``2 D`` extra calls per evaluation, relative error ~1e-9; the reference
    uses autograd, which is not available to a foreign callable here) and says so once in a ``UserWarning``.

    The estimator stays on the GPU: the engine draws the samples, hands them to the callable through pinned host
    memory (``vb_set_model_callback``), takes ``f`` and ``grad f`` back and runs the variational log density, the
    weights and every Monte-Carlo reduction on the device as for a :class:`SourceModel` -- so every objective x family
    combination a source model supports is supported.  Each evaluation contains a host round trip of ``3 N D``
    doubles plus the callable's own time; :class:`SourceModel` is the fast route."""

    def __init__(self, dim, log_density=None, grad_log_density=None, value_and_grad=None, fd_step=1e-6):
        if log_density is None and value_and_grad is None:
            raise ValueError('give log_density or value_and_grad')
        for fn in (log_density, grad_log_density, value_and_grad):
            if fn is not None and not callable(fn):
                raise TypeError('log_density / grad_log_density / value_and_grad must be callables')
        self._f, self._g, self._fg = log_density, grad_log_density, value_and_grad
        self._fd_step = float(fd_step)
        self._warned = False
        self._error = [None]
        super().__init__(dim)
        self._callback = _lib.MODEL_CALLBACK_TYPE(self._trampoline)

    # -- host evaluation ------------------------------------------------------------------------------------------
    def _values(self, z):
        f = self._f(z) if self._f is not None else self._fg(z)[0]
        f = np.asarray(f, dtype=np.float64)
        if f.shape != (z.shape[0],):
            raise ValueError('log_density returned shape {}, expected ({},)'.format(f.shape, z.shape[0]))
        return f

    def _values_and_grads(self, z):
        n, d = z.shape
        if self._fg is not None:
            f, g = self._fg(z)
        elif self._g is not None:
            f, g = self._f(z), self._g(z)
        else:
            if not self._warned:
                import warnings
                warnings.warn('CallableModel without a gradient: differentiating the callable by central differences '
                              '(2 D extra calls per evaluation); pass grad_log_density or value_and_grad, or write '
                              'the density as a SourceModel', UserWarning, stacklevel=3)
                self._warned = True
            f = self._f(z)
            h = self._fd_step * np.maximum(1.0, np.abs(z))
            g = np.empty((n, d))
            for j in range(d):                      # one batched call per coordinate and side
                zp, zm = z.copy(), z.copy()
                zp[:, j] += h[:, j]
                zm[:, j] -= h[:, j]
                g[:, j] = (np.asarray(self._f(zp), dtype=np.float64) - np.asarray(self._f(zm), dtype=np.float64)) \
                    / (zp[:, j] - zm[:, j])
        f = np.asarray(f, dtype=np.float64)
        g = np.asarray(g, dtype=np.float64)
        if f.shape != (n,) or g.shape != (n, d):
            raise ValueError('model callables returned shapes {} / {}, expected ({},) / ({}, {})'.format(
                f.shape, g.shape, n, n, d))
        return f, g

    def _trampoline(self, _user, zp, n, d, fp, gp):
        try:
            z = np.ctypeslib.as_array(zp, shape=(n, d)).copy()       # the callable may keep or modify its argument
            if gp:
                f, g = self._values_and_grads(z)
                np.ctypeslib.as_array(gp, shape=(n, d))[...] = g
            else:
                f = self._values(z)
            np.ctypeslib.as_array(fp, shape=(n,))[...] = f
            return 0
        except BaseException as exc:      # noqa: B902 -- an exception must not unwind through the C frames
            self._error[0] = exc
            return 1

    def _build_spec(self):
        empty = np.zeros(0)
        return (_lib.MODEL_SOURCE, self._dim, empty, np.zeros(0, dtype=np.int64), self._callback, self._error)


def as_device_model(model, dim):
    """What the objectives bind: a :class:`DeviceModel` as it is; the reference's ``Model(log_density)`` or a bare
    callable as a :class:`CallableModel` of dimension ``dim`` (gradient by central differences unless the caller
    builds the ``CallableModel`` with one)."""
    if isinstance(model, DeviceModel):
        return model
    if isinstance(model, Model):
        return CallableModel(dim, model._log_density)
    if callable(model):
        return CallableModel(dim, model)
    raise TypeError('model must be a viabel_amd device model (GaussianModel, FunnelModel, CorrelatedGaussianModel, a '
                    'regression model, SourceModel), a Model(log_density) or a callable; got %r'
                    % type(model).__name__)


class GaussianModel(DeviceModel):
    """``sum_d norm.logpdf(x_d; mean_d, stdev_d)`` -- the target of the reference's own
    objective / convenience tests (``tests/test_objectives.py:15-19``)."""

    def __init__(self, mean, stdev):
        mean = np.asarray(mean, dtype=np.float64).ravel()
        stdev = np.broadcast_to(np.asarray(stdev, dtype=np.float64).ravel(), mean.shape).copy()
        if np.any(stdev <= 0):
            raise ValueError('stdev must be positive')
        self.mean, self.stdev = mean, stdev
        super().__init__(mean.size)

    def _build_spec(self):
        return (_lib.MODEL_GAUSS_DIAG, self._dim, np.concatenate([self.mean, self.stdev]),
                np.zeros(0, dtype=np.int64))


class FunnelModel(DeviceModel):
    """D-dimensional funnel generalising ``docs/source/quickstart.ipynb:23-29``.

    Coordinate ``scale_index`` (default: the last; index 1 at D=2 as in the notebook) is the
    log-scale ``v ~ N(0, log_sigma_stdev)``; every other coordinate is ``N(0, exp(v))``.
    """

    def __init__(self, dim, scale_index=None, log_sigma_stdev=1.0):
        dim = int(dim)
        if dim < 2:
            raise ValueError('the funnel needs at least 2 dimensions')
        self.scale_index = dim - 1 if scale_index is None else int(scale_index)
        if not 0 <= self.scale_index < dim:
            raise ValueError('scale_index out of range')
        if log_sigma_stdev <= 0:
            raise ValueError('log_sigma_stdev must be positive')
        self.log_sigma_stdev = float(log_sigma_stdev)
        super().__init__(dim)

    def _build_spec(self):
        return (_lib.MODEL_FUNNEL, self._dim, np.array([self.log_sigma_stdev]),
                np.array([self.scale_index], dtype=np.int64))


class CorrelatedGaussianModel(DeviceModel):
    """``N(mean, covariance)`` with a dense covariance (target of the full-rank configs)."""

    def __init__(self, mean, covariance=None, precision=None):
        mean = np.asarray(mean, dtype=np.float64).ravel()
        if (covariance is None) == (precision is None):
            raise ValueError('give exactly one of covariance / precision')
        if precision is None:
            precision = np.linalg.inv(np.asarray(covariance, dtype=np.float64))
        precision = np.asarray(precision, dtype=np.float64)
        if precision.shape != (mean.size, mean.size):
            raise ValueError('precision must be (D, D)')
        self.mean = mean
        self.precision = 0.5 * (precision + precision.T)
        sign, self.logdet_precision = np.linalg.slogdet(self.precision)
        if sign <= 0:
            raise ValueError('precision must be positive definite')
        super().__init__(mean.size)

    def _build_spec(self):
        return (_lib.MODEL_GAUSS_FULL, self._dim,
                np.concatenate([self.mean, self.precision.ravel(), [self.logdet_precision]]),
                np.zeros(0, dtype=np.int64))


class LogisticRegressionModel(DeviceModel):
    """Bayesian logistic regression ``y_i ~ Bernoulli(sigmoid(x_i' b))`` with a ``N(0, prior_sd)`` prior.

    Not in the reference (BASELINE configs[4] names a logistic-regression model that the
    upstream tree does not contain); the prior scale 10 follows the Stan model of the reference's tests
    (``viabel/tests/test_models.py:41``).  Its gradient couples all coordinates through ``X``, so the engine
    evaluates it with two fp64 MFMA GEMMs per objective call.
    """

    def __init__(self, X, y, prior_sd=10.0):
        X = np.ascontiguousarray(X, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64).ravel()
        if X.ndim != 2 or y.shape != (X.shape[0],):
            raise ValueError('X must be (n_data, dim) and y (n_data,)')
        if prior_sd <= 0:
            raise ValueError('prior_sd must be positive')
        self.X, self.y, self.prior_sd = X, y, float(prior_sd)
        super().__init__(X.shape[1])

    def _build_spec(self):
        return (_lib.MODEL_LOGISTIC, self._dim, np.concatenate([self.X.ravel(), self.y, [self.prior_sd]]),
                np.array([self.X.shape[0]], dtype=np.int64))


class PoissonRegressionModel(LogisticRegressionModel):
    """Bayesian Poisson regression ``y_i ~ Poisson(exp(x_i' b))`` with a ``N(0, prior_sd)`` prior: the same two-GEMM
    pipeline as the logistic target with the log link in the GEMM epilogue (not in the reference)."""

    def __init__(self, X, y, prior_sd=10.0):
        super().__init__(X, y, prior_sd)
        if np.any(self.y < 0):
            raise ValueError('counts must be non-negative')

    def _build_spec(self):
        return (_lib.MODEL_LOGISTIC, self._dim, np.concatenate([self.X.ravel(), self.y, [self.prior_sd]]),
                np.array([self.X.shape[0], _lib.GLM_POISSON], dtype=np.int64))


class LinearRegressionModel(LogisticRegressionModel):
    """Bayesian linear regression ``y_i ~ N(x_i' b, noise_sd)`` with a ``N(0, prior_sd)`` prior (known noise scale);
    the identity-link member of the same pipeline (not in the reference)."""

    def __init__(self, X, y, prior_sd=10.0, noise_sd=1.0):
        super().__init__(X, y, prior_sd)
        if noise_sd <= 0:
            raise ValueError('noise_sd must be positive')
        self.noise_sd = float(noise_sd)

    def _build_spec(self):
        return (_lib.MODEL_LOGISTIC, self._dim,
                np.concatenate([self.X.ravel(), self.y, [self.prior_sd, self.noise_sd]]),
                np.array([self.X.shape[0], _lib.GLM_GAUSSIAN], dtype=np.int64))
