"""``bbvi``: one-call black-box variational inference (``viabel/convenience.py:14-94``).

Same keyword interface, validation errors and optimiser wiring as the reference:
``adaptive and not fixed_lr`` -> RAABBVI(RMSProp), ``adaptive and fixed_lr`` -> FASO(RMSProp),
``not adaptive and fixed_lr`` -> RMSProp.  Differences forced by the GPU engine: ``log_density`` must be
a device model (``viabel_amd.models.DeviceModel``) rather than a Python callable, and ``fit`` (PyStan)
is not supported.
"""
from .approximations import MFGaussian
from .models import DeviceModel
from .objectives import ExclusiveKL
from .optimization import FASO, RAABBVI, RMSProp

__all__ = ['bbvi', 'samples_and_log_weights']


def bbvi(dimension, *, n_iters=10000, num_mc_samples=10, log_density=None, approx=None, objective=None,
         fit=None, adaptive=True, fixed_lr=False, init_var_param=None, learning_rate=0.01,
         RMS_kwargs=dict(), FASO_kwargs=dict(), RAABBVI_kwargs=dict()):
    """Fit a model with black-box variational inference; returns the optimiser's result dict plus
    ``'objective'`` (``convenience.py:92-94``)."""
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
        if not isinstance(log_density, DeviceModel):
            raise TypeError('log_density must be a viabel_amd device model (GaussianModel, FunnelModel, '
                            'CorrelatedGaussianModel): Python callables cannot run on the GPU')
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


def samples_and_log_weights(var_param, model, approx, n_samples):
    """Samples from the approximation and their importance log weights (``convenience.py:176-179``);
    the model log density is evaluated on the GPU."""
    samples = approx.sample(var_param, n_samples)
    return samples, model(samples) - approx.log_density(var_param, samples)
