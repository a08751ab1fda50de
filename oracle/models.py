"""Oracle target models: log density, gradient, Hessian products (numpy fp64).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

The reference takes the model as an arbitrary Python callable differentiated by
autograd (``viabel/models.py:17-39``).  The device engine needs a closed set of
targets with analytic derivatives; these are the same targets restated on the
CPU.  Shapes follow the reference's contract: ``x`` is ``(N, D)`` (a ``(D,)``
input is promoted), the log density is ``(N,)``.
"""
import numpy as np

LOG_2PI = np.log(2.0 * np.pi)


def _as2d(x):
    x = np.asarray(x, dtype=np.float64)
    return x[np.newaxis, :] if x.ndim == 1 else x


class GaussDiag:
    """``sum(norm.logpdf(x, loc=mean, scale=stdev), axis=1)``.

    Target of the reference's objective tests (``viabel/tests/test_objectives.py:15-19``)
    and convenience tests (``viabel/tests/test_convenience.py:12-16``).
    """

    def __init__(self, mean, stdev):
        self.mean = np.asarray(mean, dtype=np.float64).ravel()
        self.stdev = np.asarray(stdev, dtype=np.float64).ravel()
        self.dim = self.mean.size

    def logp(self, x):
        x = _as2d(x)
        r = (x - self.mean) / self.stdev
        return np.sum(-0.5 * r * r - np.log(self.stdev) - 0.5 * LOG_2PI, axis=1)

    def grad(self, x):
        x = _as2d(x)
        return -(x - self.mean) / self.stdev ** 2

    def hessian(self, m):
        return np.diag(-1.0 / self.stdev ** 2)

    def hvp(self, m, v):
        return _as2d(v) * (-1.0 / self.stdev ** 2)


class Funnel:
    """D-dimensional generalisation of the quickstart funnel.

    ``docs/source/quickstart.ipynb:23-29``: ``x[:,1]`` is the log-scale,
    ``N(0, log_sigma_stdev)``; ``x[:,0] ~ N(0, exp(x[:,1]))``.  Generalised: the
    coordinate ``scale_index`` (default: the last one, which is index 1 at D=2
    and reproduces the notebook exactly) is the log-scale ``v``; every other
    coordinate is ``N(0, exp(v))``.
    """

    def __init__(self, dim, scale_index=None, log_sigma_stdev=1.0):
        self.dim = int(dim)
        self.k = self.dim - 1 if scale_index is None else int(scale_index)
        self.tau = float(log_sigma_stdev)

    def logp(self, x):
        x = _as2d(x)
        v = x[:, self.k]
        others = np.delete(x, self.k, axis=1)
        lp_v = -0.5 * (v / self.tau) ** 2 - np.log(self.tau) - 0.5 * LOG_2PI
        lp_o = np.sum(-0.5 * others ** 2 * np.exp(-2.0 * v)[:, None]
                      - v[:, None] - 0.5 * LOG_2PI, axis=1)
        return lp_v + lp_o

    def grad(self, x):
        x = _as2d(x)
        v = x[:, self.k]
        w = np.exp(-2.0 * v)
        g = -x * w[:, None]
        others_sq = np.sum(x ** 2, axis=1) - v ** 2
        g[:, self.k] = -v / self.tau ** 2 + w * others_sq - (self.dim - 1)
        return g

    def hessian(self, m):
        m = np.asarray(m, dtype=np.float64).ravel()
        k, D = self.k, self.dim
        v = m[k]
        w = np.exp(-2.0 * v)
        H = np.zeros((D, D))
        idx = np.arange(D) != k
        H[idx, idx] = -w
        H[idx, k] = 2.0 * m[idx] * w
        H[k, idx] = 2.0 * m[idx] * w
        H[k, k] = -1.0 / self.tau ** 2 - 2.0 * w * np.sum(m[idx] ** 2)
        return H

    def hvp(self, m, v):
        return _as2d(v) @ self.hessian(m).T


class GaussFull:
    """Correlated Gaussian ``N(mean, S)`` parameterised by the precision ``P = S^-1``.

    Target used for the full-rank configurations (C2 / headline).
    ``f(x) = -1/2 (x-m)' P (x-m) + 1/2 logdet P - D/2 log 2pi``.
    """

    def __init__(self, mean, precision):
        self.mean = np.asarray(mean, dtype=np.float64).ravel()
        self.P = np.asarray(precision, dtype=np.float64)
        self.dim = self.mean.size
        sign, ld = np.linalg.slogdet(self.P)
        self.const = 0.5 * ld - 0.5 * self.dim * LOG_2PI

    def logp(self, x):
        d = _as2d(x) - self.mean
        return -0.5 * np.sum((d @ self.P) * d, axis=1) + self.const

    def grad(self, x):
        return -(_as2d(x) - self.mean) @ self.P

    def hessian(self, m):
        return -self.P

    def hvp(self, m, v):
        return -_as2d(v) @ self.P


class Logistic:
    """Bayesian logistic regression (not in the reference: SURVEY F3).

    ``f(b) = sum_i [y_i eta_i - log(1 + exp(eta_i))] + sum_d norm.logpdf(b_d, 0, prior_sd)``
    with ``eta = X b``.  Prior scale 10 follows the Stan test model of
    ``viabel/tests/test_models.py:41``.
    """

    def __init__(self, X, y, prior_sd=10.0):
        self.X = np.asarray(X, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64).ravel()
        self.prior_sd = float(prior_sd)
        self.dim = self.X.shape[1]

    def logp(self, b):
        b = _as2d(b)
        eta = b @ self.X.T                                   # (N, n_data)
        ll = np.sum(self.y * eta - np.logaddexp(0.0, eta), axis=1)
        pr = np.sum(-0.5 * (b / self.prior_sd) ** 2 - np.log(self.prior_sd)
                    - 0.5 * LOG_2PI, axis=1)
        return ll + pr

    def grad(self, b):
        b = _as2d(b)
        eta = b @ self.X.T
        p = 1.0 / (1.0 + np.exp(-eta))
        return (self.y - p) @ self.X - b / self.prior_sd ** 2

    def hessian(self, m):
        m = np.asarray(m, dtype=np.float64).ravel()
        p = 1.0 / (1.0 + np.exp(-(self.X @ m)))
        return -(self.X.T * (p * (1 - p))) @ self.X - np.eye(self.dim) / self.prior_sd ** 2

    def hvp(self, m, v):
        return _as2d(v) @ self.hessian(m).T


class Poisson(Logistic):
    """Poisson regression with log link, ``N(0, prior_sd)`` prior (test oracle for ``PoissonRegressionModel``)."""

    def logp(self, b):
        from scipy.special import gammaln
        b = _as2d(b)
        eta = b @ self.X.T
        ll = np.sum(self.y * eta - np.exp(eta), axis=1) - np.sum(gammaln(self.y + 1.0))
        pr = np.sum(-0.5 * (b / self.prior_sd) ** 2 - np.log(self.prior_sd) - 0.5 * LOG_2PI, axis=1)
        return ll + pr

    def grad(self, b):
        b = _as2d(b)
        return (self.y - np.exp(b @ self.X.T)) @ self.X - b / self.prior_sd ** 2


class LinearRegression(Logistic):
    """Linear regression with known noise scale (test oracle for ``LinearRegressionModel``)."""

    def __init__(self, X, y, prior_sd=10.0, noise_sd=1.0):
        super().__init__(X, y, prior_sd)
        self.noise_sd = float(noise_sd)

    def logp(self, b):
        b = _as2d(b)
        r = (self.y - b @ self.X.T) / self.noise_sd
        ll = np.sum(-0.5 * r ** 2, axis=1) - self.y.size * (np.log(self.noise_sd) + 0.5 * LOG_2PI)
        pr = np.sum(-0.5 * (b / self.prior_sd) ** 2 - np.log(self.prior_sd) - 0.5 * LOG_2PI, axis=1)
        return ll + pr

    def grad(self, b):
        b = _as2d(b)
        return ((self.y - b @ self.X.T) / self.noise_sd ** 2) @ self.X - b / self.prior_sd ** 2
