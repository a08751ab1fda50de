"""Pareto smoothed importance sampling (``viabel/_psis.py``), smoothed on the GPU.

``psislw`` keeps the reference's signature and return convention (``_psis.py:113-209``): log weights of shape
``(n,)`` or ``(n, m)`` (m sets, one per column), smoothed weights normalised so that each set's log-sum-exp is 0,
and the Pareto tail indices.  The tail selection, the generalised-Pareto fit (``gpdfitnew``, ``_psis.py:212-332``)
and the quantile replacement (``gpinv``, ``:335-377``) run in one HIP kernel (``csrc/vb_psis.hip``) through
``vb_psis_smooth``; they are not re-exported as host functions.
"""
import numpy as np

from . import _lib

__all__ = ['psislw', 'psisloo', 'sumlogs']


def sumlogs(x, axis=None):
    """``log(sum(exp(x)))`` along ``axis`` without overflow (``_psis.py:380-396``)."""
    x = np.asarray(x, dtype=np.float64)
    m = np.max(x, axis=axis, keepdims=True)
    return np.log(np.sum(np.exp(x - m), axis=axis)) + np.squeeze(m, axis=axis)


def psislw(lw, Reff=1.0, overwrite_lw=False):
    """Pareto smoothed importance sampling of log weights (``_psis.py:113-209``)."""
    lw = np.asarray(lw, dtype=np.float64)
    if lw.ndim not in (1, 2):
        raise ValueError('Argument `lw` must be 1 or 2 dimensional.')
    n = lw.shape[0]
    if n <= 1:
        raise ValueError('More than one log-weight needed.')
    eng = _lib.default_engine()
    if lw.ndim == 1:
        out, k = eng.psis_smooth(n, lw, reff=Reff)
        if overwrite_lw:
            lw[...] = out
            out = lw
        return out, k
    out = lw if overwrite_lw else np.empty_like(lw, order='F')
    ks = np.empty(lw.shape[1])
    for j in range(lw.shape[1]):
        out[:, j], ks[j] = eng.psis_smooth(n, np.ascontiguousarray(lw[:, j]), reff=Reff)
    return out, ks


def psisloo(log_lik, **kwargs):
    """PSIS leave-one-out log predictive densities (``_psis.py:70-110``): ``log_lik`` is ``(n, m)``."""
    log_lik = np.asarray(log_lik, dtype=np.float64)
    kwargs.pop('overwrite_lw', None)
    lw, ks = psislw(-log_lik, **kwargs)
    loos = sumlogs(lw + log_lik, axis=0)
    return loos.sum(), loos, ks
