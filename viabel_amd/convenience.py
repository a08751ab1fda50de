"""``bbvi``: one-call black-box variational inference (``viabel/convenience.py:14-94``).

Same keyword interface, validation errors and optimiser wiring as the reference:
``adaptive and not fixed_lr`` -> RAABBVI(RMSProp), ``adaptive and fixed_lr`` -> FASO(RMSProp),
``not adaptive and fixed_lr`` -> RMSProp.  ``log_density`` is a device model
(``viabel_amd.models.DeviceModel``), HIP source, or a Python callable (bound as a ``CallableModel``: evaluated on the
host between the device's sampling and reduction kernels); ``fit`` (PyStan) is not supported.
"""
import numpy as np

from . import _lib
from ._psis import psislw
from .approximations import MFGaussian, MFStudentT
from .diagnostics import all_diagnostics
from .models import CallableModel, DeviceModel, SourceModel
from .objectives import ExclusiveKL
from .optimization import FASO, RAABBVI, RMSProp

__all__ = ['bbvi', 'vi_diagnostics', 'psis_correction', 'samples_and_log_weights']


def bbvi(dimension, *, n_iters=10000, num_mc_samples=10, log_density=None, approx=None, objective=None,
         fit=None, adaptive=True, fixed_lr=False, init_var_param=None, learning_rate=0.01,
         RMS_kwargs=dict(), FASO_kwargs=dict(), RAABBVI_kwargs=dict(), grad_log_density=None):
    """Fit a model with black-box variational inference; returns the optimiser's result dict plus
    ``'objective'`` (``convenience.py:92-94``).  ``log_density``: a device model, HIP source, or -- the reference's
    own form -- a callable ``(N, D) -> (N,)``, optionally with ``grad_log_density`` ``(N, D) -> (N, D)`` (an addition:
    the reference differentiates the callable with autograd; without a gradient the engine uses central differences)."""
    if objective is not None:
        if fit is not None or log_density is not None or approx is not None:
            raise ValueError('if objective is specified, cannot specify fit, log_density, or approx')
        approx = objective.approx
    else:
        if log_density is None:
            if fit is None:
                raise ValueError('either log_density or fit must be specified if objective not given')
            raise NotImplementedError('PyStan fits are not supported by the HIP engine; pass a device model')
        elif fit is not None:
            raise ValueError('log_density and fit cannot both be specified')
        if isinstance(log_density, (str, bytes)):      # the log density as HIP source (models.SourceModel)
            src = log_density.encode() if isinstance(log_density, str) else bytes(log_density)
            # a density written generically over vb::vec<T> carries no gradient: the engine differentiates it
            log_density = SourceModel(dimension, log_density, grad='auto' if b'vb::vec' in src else 'explicit')
        if not isinstance(log_density, DeviceModel):
            # the reference's front door (convenience.py:69-75: Model(log_density) + autograd): a host callable, with
            # `grad_log_density` when the caller has one (the StanModel contract, models.py:80-104)
            if not callable(log_density):
                raise TypeError('log_density must be a viabel_amd device model (GaussianModel, FunnelModel, '
                                'CorrelatedGaussianModel, a regression model, a SourceModel / HIP source string '
                                'defining vb_log_density) or a callable (N, D) -> (N,)')
            log_density = CallableModel(dimension, log_density, grad_log_density)
        elif grad_log_density is not None:
            raise ValueError('grad_log_density goes with a callable log_density')
        if approx is None:
            approx = MFGaussian(dimension)
        objective = ExclusiveKL(approx, log_density, num_mc_samples)
    if init_var_param is None:
        init_var_param = approx.init_param()
    base_opt = RMSProp(learning_rate, diagnostics=True, **RMS_kwargs)
    if adaptive and not fixed_lr:
        opt = RAABBVI(base_opt, **RAABBVI_kwargs)
    elif adaptive and fixed_lr:
        opt = FASO(base_opt, **FASO_kwargs)
    elif not adaptive and fixed_lr:
        opt = base_opt
    else:
        raise ValueError('if fixed_lr is False, adaptive must be True')
    results = opt.optimize(n_iters, objective, init_var_param)
    results['objective'] = objective
    return results


def vi_diagnostics(var_param, *, objective=None, model=None, approx=None, n_samples=100000):
    """Pareto k-hat and 2-divergence diagnostics with mean / std / covariance error bounds
    (``convenience.py:97-133``).  Returns a dict that also holds the samples and smoothed log weights."""
    if objective is None:
        if model is None or approx is None:
            raise ValueError('either objective or both model and approx must be specified')
    elif model is not None or approx is not None:
        raise ValueError('model and/or approx cannot be specified if objective is')
    else:
        model, approx = objective.model, objective.approx
    if n_samples <= 0:
        raise ValueError('n_samples must be positive')
    return _vi_diagnostics(var_param, model, approx, n_samples)


def _vi_diagnostics(var_param, model, approx, n_samples):      # convenience.py:136-163
    samples, smoothed_log_weights, khat = psis_correction(var_param, model, approx, n_samples)
    results = dict(samples=samples, smoothed_log_weights=smoothed_log_weights, khat=khat)
    print('Pareto k is estimated to be khat = {:.2f}'.format(khat))
    if khat > 0.7:
        print('WARNING: khat > 0.7 means importance sampling is not feasible.')
        print('WARNING: not running further diagnostics')
        return results
    print()
    moment_bound_fn = None
    if approx.supports_pth_moment(2) and approx.supports_pth_moment(4):
        def moment_bound_fn(p):
            return approx.pth_moment(var_param, p)
    _, q_var = approx.mean_and_cov(var_param)
    # NB the reference hands all_diagnostics the (D, n) transpose that psis_correction returns; with
    # pth_moment available (every in-scope family) the samples are not touched, so the orientation is moot
    results.update(all_diagnostics(smoothed_log_weights, samples=samples, moment_bound_fn=moment_bound_fn,
                                   q_var=q_var))
    print('The 2-divergence is estimated to be d2 = {:.2g}'.format(results['d2']))
    if results['d2'] > 4.6:
        print('WARNING: d2 > 4.6 means the approximation is very inaccurate')
    elif results['d2'] > 0.1:
        print('WARNING: 0.1 < d2 < 4.6 means the approximation is somewhat '
              'inaccurate. Use importance sampling to decrease error.')
    else:
        print('\nAll diagnostics pass.')
    return results


def psis_correction(var_param, model, approx, n_samples):
    """Samples (transposed, as the reference returns them), PSIS-smoothed log weights and k-hat
    (``convenience.py:166-169``).  For the mean-field families on an elementwise / funnel target the log
    weights never leave the GPU between their evaluation and the smoothing."""
    var_param = np.asarray(var_param, dtype=np.float64)
    if _on_device_weights(model, approx):
        eng = _lib.default_engine()
        eng.set_model(model.device_spec())
        if approx.rng == 'philox':          # base noise drawn on the GPU; read back only for the returned samples
            kind, kdf = approx._philox_kind()
            eng.noise_generate(_DIAG_SLOT, n_samples, approx.dim, approx._seed, approx._next_philox_stream(),
                               kind=kind, df=kdf)
            noise = eng.noise_get_host(_DIAG_SLOT, n_samples, approx.dim)
        else:
            noise = approx._base_noise(n_samples)
            eng.noise_set_host(_DIAG_SLOT, noise)
        family, df = approx._device_family()
        eng.log_weights_meanfield(_DIAG_SLOT, n_samples, approx.dim, var_param, family, df=df, fetch=False)
        smoothed, khat = eng.psis_smooth(n_samples)
        samples = var_param[:approx.dim] + np.exp(var_param[approx.dim:]) * noise
        return samples.T, smoothed, khat
    samples, log_weights = samples_and_log_weights(var_param, model, approx, n_samples)
    smoothed, khat = psislw(log_weights, overwrite_lw=True)
    return samples.T, smoothed, khat


_DIAG_SLOT = 2      # device noise slot used by the diagnostics (objectives use 0 and 1)


def _on_device_weights(model, approx):
    return (isinstance(approx, (MFGaussian, MFStudentT)) and isinstance(model, DeviceModel)
            and model.device_spec()[0] in (_lib.MODEL_GAUSS_DIAG, _lib.MODEL_FUNNEL, _lib.MODEL_SOURCE))


def samples_and_log_weights(var_param, model, approx, n_samples):
    """Samples from the approximation and their importance log weights (``convenience.py:176-179``);
    the model log density is evaluated on the GPU."""
    samples = approx.sample(var_param, n_samples)
    return samples, model(samples) - approx.log_density(var_param, samples)
