"""Markov-chain statistics of the iterate history, used by FASO / RAABBVI between objective calls.

Host-side (numpy) equivalents of the reference's ``viabel/_mc_diagnostics.py``: FFT autocovariance
(``:7-37``), effective sample size with Geyer's initial-positive / monotone sequence (``:40-99``), Monte
Carlo standard error (``:102-121``), split R-hat of a single chain (``:124-160``) and the trailing-window
search (``:163-184``).  O(window x var_param_dim) work, not on the gradient path.  Written vectorised
(pair sums + running minimum) and pinned against the reference's functions by
``tests/golden/chainstats.npz``.
"""
import warnings

import numpy as np
from scipy.fft import next_fast_len

__all__ = ['autocov', 'ess', 'MCSE', 'compute_R_hat', 'R_hat_convergence_check']


def autocov(samples, axis=-1):
    """Biased autocovariance at every lag along ``axis``, via the power spectrum."""
    x = np.asarray(samples, dtype=np.float64)
    axis = axis % x.ndim
    n = x.shape[axis]
    nfft = next_fast_len(2 * n)
    x = x - x.mean(axis, keepdims=True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        spectrum = np.fft.rfft(x, n=nfft, axis=axis)
        acov = np.fft.irfft(spectrum * np.conjugate(spectrum), n=nfft, axis=axis)
    return np.take(acov, np.arange(n), axis=axis) / n


def ess(samples):
    """Effective sample size of ``samples`` with shape ``(n_chain, n_draw)`` (the callers pass one chain).

    Autocorrelations are grouped in pairs (0,1), (2,3), ...; the sum over pairs stops at the first
    non-positive pair sum and the retained pair sums are made non-increasing."""
    x = np.asarray(samples, dtype=np.float64)
    n_chain, n = x.shape
    acov_mean = autocov(x, axis=1).mean(axis=0)
    chain_var = acov_mean[0] * n / (n - 1.0)
    var_plus = chain_var * (n - 1.0) / n
    with np.errstate(invalid='ignore', divide='ignore'):      # constant chain: 0/0 -> nan, reported as nan
        r = 1.0 - (chain_var - acov_mean) / var_plus
    r[0] = 1.0
    last_pair = max((n - 3) // 2, 0)
    pair_sum = r[0:2 * last_pair + 1:2] + r[1:2 * last_pair + 2:2]
    stop = np.flatnonzero(pair_sum <= 0)
    jl = int(stop[0]) if stop.size else last_pair            # index of the last pair looked at
    if np.isnan(r[:2 * jl + 2]).any():
        return np.nan
    kept = np.minimum.accumulate(pair_sum[:jl]) if jl > 0 else np.zeros(0)
    if jl == 0:
        tail = 1.0
    elif pair_sum[jl] >= 0 or r[2 * jl] > 0:
        tail = r[2 * jl]
    else:
        tail = 0.0
    total = n_chain * n
    tau = max(-1.0 + 2.0 * kept.sum() + tail, 1.0 / np.log10(total))
    return total / tau


def MCSE(sample):
    """``(ess list, mcse array)`` for iterates of shape ``(n_iters, d)``."""
    sample = np.asarray(sample, dtype=np.float64)
    n_iters, d = sample.shape
    spread = np.sqrt(np.var(sample, ddof=1, axis=0))
    eff = [ess(sample[:, i][np.newaxis, :]) for i in range(d)]
    return eff, spread / np.sqrt(eff)


def compute_R_hat(chains, warmup=0, jitter=1e-8):
    """Split R-hat: the single chain ``(n_iters, d)`` is cut into two halves that play the chains."""
    chains = np.asarray(chains, dtype=np.float64)[warmup:]
    usable = chains.shape[0] - chains.shape[0] % 2
    half = usable // 2
    halves = chains[:usable].reshape(2, half, -1)
    means = halves.mean(axis=1)
    within = ((halves - means[:, np.newaxis, :]) ** 2).sum(axis=1) / (half - 1)
    between = half * ((means - means.mean(axis=0)) ** 2).sum(axis=0)      # / (2 - 1) half-chains
    W = np.nanmean(within, axis=0) + jitter
    return np.sqrt((half - 1) / half + between / (half * W))


def R_hat_convergence_check(samples, windows, Rhat_threshold=1.1):
    """Max R-hat over parameters for each trailing window; returns ``(converged, best_window)``."""
    worst = [np.max(compute_R_hat(np.array(samples[-w:]))) for w in windows]
    best = int(np.argmin(worst))
    return worst[best] <= Rhat_threshold, windows[best]
