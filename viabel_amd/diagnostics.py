"""Post-fit accuracy diagnostics (``viabel/diagnostics.py``): 2-divergence (CUBO - ELBO) bound, Wasserstein
bounds and the mean / standard-deviation / covariance error bounds derived from them.

These are O(n_samples) scalar reductions of log weights that the GPU path has already produced
(``convenience.vi_diagnostics`` -> ``vb_log_weights_meanfield`` / ``vb_psis_smooth``); they run on the host.
"""
from warnings import warn

import numpy as np

__all__ = ['all_diagnostics', 'error_bounds', 'wasserstein_bounds', 'divergence_bound']


def _mc_mean(a, name, atol=0.01):
    """Mean with the reference's Monte Carlo error warning (``diagnostics.py:189-198``)."""
    a = np.asarray(a, dtype=np.float64)
    m = a.mean()
    se = a.std() / np.sqrt(a.size)
    if se > atol:
        warn('significant Monte Carlo error when computing {} (mean = {}, standard deviation = {})'.format(name, m, se))
    return m


def divergence_bound(log_weights, *, alpha=2., log_norm_bound=None, return_log_norm_bound=False):
    """Bound on the alpha-divergence between p and q from q-samples' log weights (``diagnostics.py:140-186``):
    ``alpha / (alpha - 1) * (CUBO_alpha - log_norm_bound)``, the ELBO standing in for the bound when none is given."""
    if alpha <= 1:
        raise ValueError('alpha must be greater than 1')
    lw = np.asarray(log_weights, dtype=np.float64)
    shift = lw.max()
    cubo = np.log(_mc_mean(np.exp(lw - shift) ** alpha, 'CUBO')) / alpha + shift
    if log_norm_bound is None:
        log_norm_bound = _mc_mean(lw, 'ELBO')
    dalpha = alpha / (alpha - 1) * (cubo - log_norm_bound)
    return (dalpha, log_norm_bound) if return_log_norm_bound else dalpha


def wasserstein_bounds(d2, *, samples=None, moment_bound_fn=None):
    """1- and 2-Wasserstein bounds from the 2-divergence and the 2nd / 4th central moments of q
    (``diagnostics.py:99-137``)."""
    if moment_bound_fn is None:
        if samples is None:
            raise ValueError('must provides samples if moment_bound_fn not given')
        x = np.asarray(samples, dtype=np.float64)
        if x.ndim == 1:
            x = x[:, np.newaxis]
        centred = x - x.mean(axis=0, keepdims=True)

        def moment_bound_fn(p):
            return np.mean(np.sum(centred ** p, axis=1))
    return {'W{}'.format(p): 2 * moment_bound_fn(2 * p) ** (.5 / p) * np.expm1(d2) ** (.5 / p) for p in (1, 2)}


def _spectral_norm(var):
    return np.linalg.norm(var, ord=2) if np.asarray(var).ndim == 2 else var


def error_bounds(*, W1=np.inf, W2=np.inf, q_var=np.inf, p_var=np.inf):
    """Mean / std / covariance error bounds from Wasserstein bounds (``diagnostics.py:66-96``, ``:201-219``)."""
    qv, pv = _spectral_norm(q_var), _spectral_norm(p_var)
    min_std = np.sqrt(qv if pv is None else np.min([qv, pv], axis=0))
    return dict(mean_error=min(W1, W2), std_error=W2, cov_error=2 * (min_std * W2 + W2 ** 2))


def all_diagnostics(log_weights, *, samples=None, moment_bound_fn=None, q_var=None, p_var=None,
                    log_norm_bound=None):
    """All bounds at once (``diagnostics.py:13-55``); returns ``mean_error``, ``std_error``, ``cov_error``,
    ``W1``, ``W2``, ``d2``, ``log_norm_bound``."""
    d2, log_norm_bound = divergence_bound(log_weights, log_norm_bound=log_norm_bound, return_log_norm_bound=True)
    results = wasserstein_bounds(d2, samples=samples, moment_bound_fn=moment_bound_fn)
    if q_var is None and samples is not None:
        q_var = np.cov(np.asarray(samples).T)
    results.update(error_bounds(q_var=q_var, p_var=p_var, **results))
    results['d2'] = d2
    results['log_norm_bound'] = log_norm_bound
    return results
