"""Oracle approximation families (numpy fp64).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Restates ``viabel/approximations.py`` (MFGaussian ``:192-251``, MFStudentT
``:254-312``, MultivariateT ``:322-382``) and ``viabel/_distributions.py:7-38``.
The noise draws are separated from the arithmetic (``draw_noise`` /
``sample_from_noise``) so the same noise can be handed to the device engine.

Parameter layout (``paragami`` 0.42, a third-party dependency that is not under
the reference tree; ``approximations.py:185-189`` / ``:315-319``): members are
concatenated in insertion order, unconstrained vectors are stored as they are,
and ``PSDSymmetricMatrixPattern`` stores the Cholesky factor with the *log* of
its diagonal, lower triangle in ``numpy.tril_indices`` (row-major) order.
"""
import numpy as np
from scipy import linalg as sla
from scipy import special

LOG_2PI = np.log(2.0 * np.pi)


# --------------------------------------------------------------------------
# free-Cholesky packing (paragami PSDSymmetricMatrixPattern, free=True)
# --------------------------------------------------------------------------
def chol_to_free(L):
    """Lower-triangular factor -> flat free vector (log diagonal, tril order)."""
    L = np.array(L, dtype=np.float64)
    D = L.shape[0]
    Lf = L.copy()
    Lf[np.diag_indices(D)] = np.log(np.diag(L))
    return Lf[np.tril_indices(D)]


def free_to_chol(v, D):
    L = np.zeros((D, D))
    L[np.tril_indices(D)] = np.asarray(v, dtype=np.float64)
    L[np.diag_indices(D)] = np.exp(np.diag(L))
    return L


def psd_to_free(S):
    return chol_to_free(np.linalg.cholesky(np.asarray(S, dtype=np.float64)))


def free_to_psd(v, D):
    L = free_to_chol(v, D)
    return L @ L.T


# --------------------------------------------------------------------------
class MFGaussian:
    """``viabel/approximations.py:192-251``; theta = [mu | log_sigma]."""

    def __init__(self, dim):
        self.dim = int(dim)
        self.var_param_dim = 2 * self.dim

    def split(self, theta):
        theta = np.asarray(theta, dtype=np.float64)
        return theta[:self.dim], theta[self.dim:]

    def init_param(self):                      # :207-210
        return np.concatenate([np.zeros(self.dim), 2.0 * np.ones(self.dim)])

    def draw_noise(self, rs, n):               # :216  rs.randn(n, dim)
        return rs.randn(n, self.dim)

    def sample_from_noise(self, theta, eps):   # :215-216
        mu, ls = self.split(theta)
        return mu + np.exp(ls) * eps

    def entropy(self, theta):                  # :218-220
        _, ls = self.split(theta)
        return 0.5 * self.dim * (1.0 + LOG_2PI) + np.sum(ls)

    def kl(self, theta0, theta1):              # :222-229
        mu0, ls0 = self.split(theta0)
        mu1, ls1 = self.split(theta1)
        md = mu0 - mu1
        lsd = ls0 - ls1
        return 0.5 * np.sum(np.exp(2 * lsd) + md ** 2 / np.exp(2 * ls1) - 2 * lsd - 1)

    def log_density(self, theta, x):           # :231-236
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x[np.newaxis, :]
        mu, ls = self.split(theta)
        r = (x - mu) / np.exp(ls)
        return np.sum(-0.5 * r * r - ls - 0.5 * LOG_2PI, axis=-1)

    def log_density_grad_weighted(self, theta, x, w):
        """sum_n w_n d/dtheta log q(x_n; theta) (x held fixed)."""
        mu, ls = self.split(theta)
        sig = np.exp(ls)
        r = (np.atleast_2d(x) - mu) / sig
        w = np.asarray(w, dtype=np.float64)
        return np.concatenate([(w[:, None] * r).sum(0) / sig,
                               (w[:, None] * (r * r - 1.0)).sum(0)])

    def mean_and_cov(self, theta):             # :238-240
        mu, ls = self.split(theta)
        return mu, np.diag(np.exp(2 * ls))

    def scale(self, theta):
        """sqrt(diag(cov)) as used by the RGE path (objectives.py:172-173)."""
        _, ls = self.split(theta)
        return np.exp(ls)

    def pth_moment(self, theta, p):            # :242-251
        _, ls = self.split(theta)
        v = np.exp(2 * ls)
        if p == 2:
            return np.sum(v)
        if p == 4:
            return 2 * np.sum(v ** 2) + np.sum(v) ** 2
        raise ValueError('p = {} is not a supported moment'.format(p))


class MFStudentT(MFGaussian):
    """``viabel/approximations.py:254-312``; theta = [mu | log_sigma]."""

    def __init__(self, dim, df):
        if df <= 2:
            raise ValueError('df must be greater than 2')
        super().__init__(dim)
        self.df = df

    def draw_noise(self, rs, n):               # :273-274
        return rs.standard_t(self.df, size=(n, self.dim))

    def entropy(self, theta):                  # :276-279 (df-only terms dropped)
        _, ls = self.split(theta)
        return np.sum(ls)

    def kl(self, theta0, theta1):
        raise NotImplementedError()

    def log_density(self, theta, x):           # :281-286  t.logpdf(x, df, mu, sigma)
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x[np.newaxis, :]
        mu, ls = self.split(theta)
        df = self.df
        r = (x - mu) / np.exp(ls)
        c = (special.gammaln(0.5 * (df + 1)) - special.gammaln(0.5 * df)
             - 0.5 * np.log(df * np.pi))
        return np.sum(c - 0.5 * (df + 1) * np.log1p(r * r / df) - ls, axis=-1)

    def log_density_grad_weighted(self, theta, x, w):
        mu, ls = self.split(theta)
        sig = np.exp(ls)
        df = self.df
        r = (np.atleast_2d(x) - mu) / sig
        sc = (df + 1.0) * r / (df + r * r)
        w = np.asarray(w, dtype=np.float64)
        return np.concatenate([(w[:, None] * sc).sum(0) / sig,
                               (w[:, None] * (sc * r - 1.0)).sum(0)])

    def mean_and_cov(self, theta):             # :288-292
        mu, ls = self.split(theta)
        return mu, self.df / (self.df - 2) * np.diag(np.exp(2 * ls))

    def scale(self, theta):
        _, ls = self.split(theta)
        return np.sqrt(self.df / (self.df - 2)) * np.exp(ls)

    def pth_moment(self, theta, p):            # :294-304
        df = self.df
        if p not in (2, 4) or p >= df:
            raise ValueError('p = {} is not a supported moment'.format(p))
        _, ls = self.split(theta)
        s = np.exp(ls)
        c = df / (df - 2)
        if p == 2:
            return c * np.sum(s ** 2)
        return c ** 2 * (2 * (df - 1) / (df - 4) * np.sum(s ** 4) + np.sum(s ** 2) ** 2)


# --------------------------------------------------------------------------
class FullRankGaussian:
    """New family: theta = [mu | free-Cholesky of Sigma = L L'].

    No reference class exists.  The layout is MultivariateT's
    (``approximations.py:315-319``); ``z = mu + L eps``.  Pinned by reduction to
    MFGaussian when L is diagonal and by closed-form Gaussian identities.
    """

    def __init__(self, dim):
        self.dim = int(dim)
        self.var_param_dim = self.dim + self.dim * (self.dim + 1) // 2

    def split(self, theta):
        theta = np.asarray(theta, dtype=np.float64)
        return theta[:self.dim], free_to_chol(theta[self.dim:], self.dim)

    def pack(self, mu, L):
        return np.concatenate([np.asarray(mu, dtype=np.float64), chol_to_free(L)])

    def init_param(self):
        # same spread as MFGaussian.init_param: log-scale 2 on the diagonal
        return self.pack(np.zeros(self.dim), np.exp(2.0) * np.eye(self.dim))

    def draw_noise(self, rs, n):
        return rs.randn(n, self.dim)

    def sample_from_noise(self, theta, eps):
        mu, L = self.split(theta)
        return mu + eps @ L.T

    def entropy(self, theta):
        _, L = self.split(theta)
        return 0.5 * self.dim * (1.0 + LOG_2PI) + np.sum(np.log(np.diag(L)))

    def log_density(self, theta, x):
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x[np.newaxis, :]
        mu, L = self.split(theta)
        e = sla.solve_triangular(L, (x - mu).T, lower=True).T
        return (-0.5 * np.sum(e * e, axis=-1) - np.sum(np.log(np.diag(L)))
                - 0.5 * self.dim * LOG_2PI)

    def log_density_grad_weighted(self, theta, x, w):
        mu, L = self.split(theta)
        w = np.asarray(w, dtype=np.float64)
        e = sla.solve_triangular(L, (np.atleast_2d(x) - mu).T, lower=True).T   # (N, D)
        u = sla.solve_triangular(L.T, e.T, lower=False).T                      # L^-T e
        dmu = (w[:, None] * u).sum(0)
        dL = np.tril((w[:, None] * u).T @ e) - np.sum(w) * np.diag(1.0 / np.diag(L))
        dfree = dL.copy()
        dfree[np.diag_indices(self.dim)] = np.diag(dL) * np.diag(L)
        return np.concatenate([dmu, dfree[np.tril_indices(self.dim)]])

    def kl(self, theta0, theta1):
        mu0, L0 = self.split(theta0)
        mu1, L1 = self.split(theta1)
        A = sla.solve_triangular(L1, L0, lower=True)
        dm = sla.solve_triangular(L1, mu1 - mu0, lower=True)
        return 0.5 * (np.sum(A * A) + dm @ dm - self.dim) \
            + np.sum(np.log(np.diag(L1))) - np.sum(np.log(np.diag(L0)))

    def mean_and_cov(self, theta):
        mu, L = self.split(theta)
        return mu, L @ L.T

    def pth_moment(self, theta, p):
        _, L = self.split(theta)
        S = L @ L.T
        if p == 2:
            return np.trace(S)
        if p == 4:
            return 2 * np.sum(S * S) + np.trace(S) ** 2
        raise ValueError('p = {} is not a supported moment'.format(p))


# --------------------------------------------------------------------------
def multivariate_t_logpdf(x, m, S, df):
    """``viabel/_distributions.py:7-38`` (finite df branch), literal restatement."""
    d = m.shape[-1]
    s, u = np.linalg.eigh(S)                                     # :26
    eps = 1e-10
    s_pinv = np.array([0 if abs(v) <= eps else 1 / v for v in s], dtype=float)   # :28
    U = np.multiply(u, np.sqrt(s_pinv))                          # :29
    log_pdet = np.sum(np.log(s))                                 # :30
    log_pdf = (special.gammaln(.5 * (df + d)) - special.gammaln(.5 * df)
               - .5 * d * np.log(np.pi * df))                    # :32-33
    log_pdf += -.5 * log_pdet                                    # :34
    dev = x - m                                                  # :35
    maha = np.sum(np.square(np.dot(dev, U)), axis=-1)            # :36
    log_pdf = log_pdf + -.5 * (df + d) * np.log(1 + maha / df)   # :37
    return log_pdf


class MultivariateT:
    """``viabel/approximations.py:322-382``; theta = [mu | free-Cholesky of Sigma]."""

    def __init__(self, dim, df):
        if df <= 2:
            raise ValueError('df must be greater than 2')
        self.dim = int(dim)
        self.df = df
        self.var_param_dim = self.dim + self.dim * (self.dim + 1) // 2

    def split(self, theta):
        theta = np.asarray(theta, dtype=np.float64)
        return theta[:self.dim], free_to_psd(theta[self.dim:], self.dim)

    def init_param(self):                      # :337-340
        return np.concatenate([np.zeros(self.dim), psd_to_free(10 * np.eye(self.dim))])

    def draw_noise(self, rs, n):               # :345-347: chi-square first, then normals
        chi = rs.chisquare(self.df, n)
        z = rs.randn(n, self.dim)
        return chi, z

    def sample_from_noise(self, theta, noise):  # :345-349 (symmetric square root)
        chi, z = noise
        mu, S = self.split(theta)
        s = np.sqrt(chi / self.df)
        return mu + np.dot(z, sla.sqrtm(S).real) / s[:, np.newaxis]

    def entropy(self, theta):                  # :351-354
        _, S = self.split(theta)
        return .5 * np.log(np.linalg.det(S))

    def log_density(self, theta, x):           # :356-357 -> _distributions.py
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x[np.newaxis, :]
        mu, S = self.split(theta)
        return multivariate_t_logpdf(x, mu, S, self.df)

    def log_density_grad_weighted(self, theta, x, w):
        """sum_n w_n d/dtheta log q(x_n; theta), SURVEY Appendix A.5."""
        theta = np.asarray(theta, dtype=np.float64)
        D, df = self.dim, self.df
        mu = theta[:D]
        L = free_to_chol(theta[D:], D)
        w = np.asarray(w, dtype=np.float64)
        dev = np.atleast_2d(x) - mu
        e = sla.solve_triangular(L, dev.T, lower=True).T          # L^-1 dev
        maha = np.sum(e * e, axis=1)
        u = sla.solve_triangular(L.T, e.T, lower=False).T          # Sigma^-1 dev
        c = (df + D) / (df + maha)
        wc = w * c
        dmu = (wc[:, None] * u).sum(0)
        Sinv = sla.cho_solve((L, True), np.eye(D))
        dS = -0.5 * np.sum(w) * Sinv + 0.5 * (wc[:, None] * u).T @ u
        dL = np.tril(2.0 * dS @ L)          # dS symmetric
        dfree = dL.copy()
        dfree[np.diag_indices(D)] = np.diag(dL) * np.diag(L)
        return np.concatenate([dmu, dfree[np.tril_indices(D)]])

    def kl(self, theta0, theta1):
        raise NotImplementedError()

    def mean_and_cov(self, theta):             # :359-362
        mu, S = self.split(theta)
        return mu, self.df / (self.df - 2.) * S

    def pth_moment(self, theta, p):            # :364-374
        df = self.df
        if p not in (2, 4) or p >= df:
            raise ValueError('p = {} is not a supported moment'.format(p))
        _, S = self.split(theta)
        sq = np.linalg.eigvalsh(S)
        c = df / (df - 2)
        if p == 2:
            return c * np.sum(sq)
        return c ** 2 * (2 * (df - 1) / (df - 4) * np.sum(sq ** 2) + np.sum(sq) ** 2)


class LRGaussian:
    """approximations.py:610-731: theta = [mu | log_sigma | B (D x k row-major)], x = mu + z B' + sigma eps."""

    def __init__(self, dim, k):
        self.dim, self.k = dim, k
        self.var_param_dim = 2 * dim + dim * k

    def split(self, theta):
        theta = np.asarray(theta, dtype=np.float64)
        D = self.dim
        return theta[:D], theta[D:2 * D], theta[2 * D:].reshape(D, self.k)

    def draw_noise(self, rs, n):                  # :639-640: z first, then eps
        z = rs.randn(n, self.k)
        return z, rs.randn(n, self.dim)

    def sample_from_noise(self, theta, noise):    # :636-644
        mu, ls, B = self.split(theta)
        z, eps = noise
        return mu + z @ B.T + np.exp(ls) * eps

    def cov(self, theta):                         # :709-713
        _, ls, B = self.split(theta)
        return B @ B.T + np.diag(np.exp(2 * ls))

    def entropy(self, theta):                     # :646-652
        return 0.5 * self.dim * (np.log(2 * np.pi) + 1) + 0.5 * np.linalg.slogdet(self.cov(theta))[1]

    def log_density(self, theta, x):              # :685-707 (dense inverse instead of Woodbury)
        mu = self.split(theta)[0]
        S = self.cov(theta)
        diff = np.atleast_2d(x) - mu
        maha = np.sum(diff * np.linalg.solve(S, diff.T).T, axis=1)
        return -0.5 * (self.dim * np.log(2 * np.pi) + np.linalg.slogdet(S)[1] + maha)

    def log_density_grad_weighted(self, theta, x, w):
        """sum_n w_n d/dtheta log q(x_n; theta) for fixed samples (what autograd gives DISInclusiveKL,
        objectives.py:405-416, through approximations.py:685-707).  With a_n = Sigma^-1 (x_n - mu):
        d/dmu = a, d/dSigma = -1/2 Sigma^-1 + 1/2 a a', Sigma = B B' + diag(sigma^2) so
        d/dlog_sigma = 2 sigma^2 diag(d/dSigma) and d/dB = 2 (d/dSigma) B."""
        mu, ls, B = self.split(theta)
        Sinv = np.linalg.inv(self.cov(theta))
        a = (np.atleast_2d(x) - mu) @ Sinv
        w = np.asarray(w, dtype=np.float64)
        dS = -0.5 * np.sum(w) * Sinv + 0.5 * (a * w[:, None]).T @ a
        return np.concatenate([(a * w[:, None]).sum(0), 2.0 * np.exp(2 * ls) * np.diag(dS), (2.0 * dS @ B).reshape(-1)])

    def kl(self, theta0, theta1):                 # :654-682
        mu0, mu1 = self.split(theta0)[0], self.split(theta1)[0]
        S0, S1 = self.cov(theta0), self.cov(theta1)
        dm = mu0 - mu1
        return 0.5 * (np.linalg.slogdet(S1)[1] - np.linalg.slogdet(S0)[1] - self.dim
                      + dm @ np.linalg.solve(S1, dm) + np.trace(np.linalg.solve(S1, S0)))

    def entropy_grad(self, theta):
        """d entropy / d theta: [0 | diag(Sigma^-1) sigma^2 | Sigma^-1 B]."""
        _, ls, B = self.split(theta)
        Sinv = np.linalg.inv(self.cov(theta))
        return np.concatenate([np.zeros(self.dim), np.diag(Sinv) * np.exp(2 * ls), (Sinv @ B).reshape(-1)])
