"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's Pareto-smoothed importance sampling and
post-fit diagnostics.  Nothing in `viabel_amd/` may import this module.

Follows `viabel/_psis.py:113-209` (psislw), `:212-332` (gpdfitnew, Zhang & Stephens empirical-Bayes GPD
fit), `:335-377` (gpinv), `:380-396` (sumlogs) and `viabel/diagnostics.py:13-219`.  Pinned against the
reference's own functions (importable with numpy only) by `tests/golden/psis.npz`
(`tests/golden/make_golden.py::gen_psis`).
"""
import numpy as np

K_MIN = 1.0 / 3.0            # _psis.py:161


def log_sum_exp(x):          # sumlogs, _psis.py:380-396
    x = np.asarray(x, dtype=np.float64)
    m = np.max(x)
    return np.log(np.sum(np.exp(x - m))) + m


def tail_size(n, reff=1.0):  # _psis.py:158 (cutoff_ind = -tail_size - 1)
    return int(np.ceil(min(0.2 * n, 3.0 * np.sqrt(n / reff))))


def gpd_fit(x_sorted):
    """(k, sigma) of the generalised Pareto fit to ascending data (_psis.py:266-325)."""
    x = np.asarray(x_sorted, dtype=np.float64)
    n = x.size
    m = 30 + int(np.sqrt(n))
    j = np.arange(1, m + 1, dtype=np.float64)
    b_grid = (1.0 - np.sqrt(m / (j - 0.5))) / (3.0 * x[int(n / 4 + 0.5) - 1]) + 1.0 / x[-1]
    k_grid = np.array([np.mean(np.log1p(-b * x)) for b in b_grid])
    L = n * (np.log(-(b_grid / k_grid)) - k_grid - 1.0)
    w = 1.0 / np.array([np.sum(np.exp(L - Lj)) for Lj in L])
    keep = w >= 10 * np.finfo(float).eps
    w, b_grid = w[keep], b_grid[keep]
    w = w / w.sum()
    b = np.sum(b_grid * w)
    k = np.mean(np.log1p(-b * x))
    sigma = -k / b
    a = 10.0
    k = k * n / (n + a) + a * 0.5 / (n + a)
    return k, sigma


def gpd_quantile(p, k, sigma):          # gpinv for 0 < p < 1, _psis.py:335-352
    p = np.asarray(p, dtype=np.float64)
    if sigma <= 0:
        return np.full(p.shape, np.nan)
    if abs(k) < np.finfo(float).eps:
        return -np.log1p(-p) * sigma
    return np.expm1(-k * np.log1p(-p)) / k * sigma


def psis_smooth(lw, reff=1.0):
    """Smoothed, normalised log weights and the Pareto k-hat of one weight vector (_psis.py:164-204)."""
    x = np.array(lw, dtype=np.float64)
    n = x.size
    if n <= 1:
        raise ValueError('More than one log-weight needed.')
    x -= np.max(x)
    order = np.argsort(x, kind='stable')
    xcut = max(x[order[-tail_size(n, reff) - 1]], np.log(np.finfo(float).tiny))
    tail = np.where(x > xcut)[0]
    n2 = tail.size
    if n2 <= 4:
        k = np.inf
    else:
        tail = tail[np.argsort(x[tail], kind='stable')]          # ascending, ties by index
        k, sigma = gpd_fit(np.exp(x[tail]) - np.exp(xcut))
    if k >= K_MIN and not np.isinf(k):
        q = gpd_quantile((np.arange(n2) + 0.5) / n2, k, sigma) + np.exp(xcut)
        x[tail] = np.log(q)
        x[x > 0] = 0.0
    x -= log_sum_exp(x)
    return x, k


# ---- diagnostics.py -----------------------------------------------------------------------------------
def divergence_bound(log_weights, alpha=2.0, log_norm_bound=None):      # diagnostics.py:140-186
    lw = np.asarray(log_weights, dtype=np.float64)
    shift = np.max(lw)
    cubo = np.log(np.mean(np.exp(lw - shift) ** alpha)) / alpha + shift
    if log_norm_bound is None:
        log_norm_bound = np.mean(lw)
    return alpha / (alpha - 1.0) * (cubo - log_norm_bound), log_norm_bound


def wasserstein_bounds(d2, moment_fn):                                  # diagnostics.py:99-137
    return {'W%d' % p: 2.0 * moment_fn(2 * p) ** (0.5 / p) * np.expm1(d2) ** (0.5 / p) for p in (1, 2)}


def error_bounds(W1, W2, q_var, p_var=None):                            # diagnostics.py:66-96, :208-219
    def norm2(v):
        return np.linalg.norm(v, ord=2) if np.asarray(v).ndim == 2 else v
    qv = norm2(q_var)
    min_var = qv if p_var is None else np.min([qv, norm2(p_var)], axis=0)
    return {'mean_error': min(W1, W2), 'std_error': W2,
            'cov_error': 2.0 * (np.sqrt(min_var) * W2 + W2 ** 2)}
