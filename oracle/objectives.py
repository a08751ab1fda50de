"""Oracle objectives: value and gradient of viabel's stochastic objectives (numpy fp64).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Restates ``viabel/objectives.py``: ExclusiveKL plain path ``:150-168``, the RGE
control-variate path ``:170-273``, DISInclusiveKL ``:317-416`` and
AlphaDivergence ``:443-463``.  Where the reference calls autograd, the
closed-form derivative is written out.

All functions take the noise explicitly (``noise`` is what
``family.draw_noise(RandomState, N)`` returns) so that the oracle and the device
engine can be driven with the very same draws.
"""
import numpy as np
from scipy import linalg as sla

from . import families as fam


# ==========================================================================
# ExclusiveKL, plain path  (objectives.py:154-168)
# ==========================================================================
def _t_score(r, df):
    return (df + 1.0) * r / (df + r * r)


def exclusive_kl(family, model, theta, noise, use_path_deriv=False):
    """Returns ``(value, grad)`` = what ``value_and_grad(variational_objective)`` returns.

    ``value = -lower_bound`` with the entropy form (``:160-161``) or the
    path-derivative form (``:156-159``, log q evaluated at stopped parameters).
    """
    theta = np.asarray(theta, dtype=np.float64)
    if isinstance(family, fam.FullRankGaussian):
        return _exclusive_kl_fullrank(family, model, theta, noise, use_path_deriv)
    if isinstance(family, fam.MultivariateT):
        return _exclusive_kl_mvt(family, model, theta, noise, use_path_deriv)
    if isinstance(family, fam.LRGaussian):
        return _exclusive_kl_lowrank(family, model, theta, noise, use_path_deriv)
    D = family.dim
    mu, ls = family.split(theta)
    sig = np.exp(ls)
    z = family.sample_from_noise(theta, noise)                     # :155
    f = model.logp(z)
    g = model.grad(z)
    if use_path_deriv:                                             # :156-159
        logq = family.log_density(theta, z)
        value = -np.mean(f - logq)
        score = _t_score(noise, family.df) if isinstance(family, fam.MFStudentT) else noise
        gt = g + score / sig                                       # d/dz [f - log q]
        grad = -np.concatenate([gt.mean(0), (gt * noise * sig).mean(0)])
    else:                                                          # :160-161
        value = -(np.mean(f) + family.entropy(theta))
        grad = -np.concatenate([g.mean(0), (g * noise * sig).mean(0) + 1.0])
    return value, grad


def sqrt_root_vjp(S, G):
    """dF/dS given dF/dR = G for R = S^{1/2} (symmetric): solve R X + X R = sym(G) in R's eigenbasis."""
    w, U = np.linalg.eigh(S)
    r = np.sqrt(w)
    Gs = 0.5 * (G + G.T)
    return U @ ((U.T @ Gs @ U) / (r[:, None] + r[None, :])) @ U.T


def _exclusive_kl_mvt(family, model, theta, noise, use_path_deriv=False):
    """Entropy form (objectives.py:160-164) for MultivariateT: x = mu + (z R) / s, R = sqrtm(L L')
    (approximations.py:342-349), entropy = sum log L_ii (:351-354).  Chain rule: dF/dR = mean g (z / s)',
    R -> Sigma by the Sylvester solve above, Sigma = L L' -> dL = tril(2 X L), free diagonal x L_ii.

    Path derivative (:156-159): value = -mean(f(x) - log q(x; stop(theta))); only x moves, so the model gradient
    is replaced by g - dlog q/dx = g + c_n R^-1 z_n / s_n with c_n = (df + D) / (df + maha_n) and
    maha_n = |z_n|^2 / s_n^2 (the Mahalanobis distance of a sample depends on its noise only)."""
    theta = np.asarray(theta, dtype=np.float64)
    D = family.dim
    chi, z = noise
    mu, Sigma = family.split(theta)
    L = fam.free_to_chol(theta[D:], D)
    x = family.sample_from_noise(theta, noise)
    g = model.grad(x)
    N = x.shape[0]
    zs = z / np.sqrt(chi / family.df)[:, None]
    if use_path_deriv:
        df = family.df
        value = -np.mean(model.logp(x) - family.log_density(theta, x))
        w, U = np.linalg.eigh(Sigma)
        Rinv = (U / np.sqrt(w)) @ U.T
        c = (df + D) / (df + np.sum(zs * zs, axis=1))
        g = g + c[:, None] * (zs @ Rinv)
        ent = 0.0
    else:
        value = -(np.mean(model.logp(x)) + family.entropy(theta))
        ent = 1.0
    X = sqrt_root_vjp(Sigma, g.T @ zs / N)
    dL = np.tril(2.0 * X @ L)
    dL[np.diag_indices(D)] = np.diag(dL) * np.diag(L) + ent          # free (log) diagonal (+ entropy)
    return value, -np.concatenate([g.mean(0), dL[np.tril_indices(D)]])


def _exclusive_kl_lowrank(family, model, theta, noise, use_path_deriv=False):
    """Entropy form (objectives.py:160-164) for LRGaussian: d/dmu = -mean g, d/dlog_sigma = -mean(g eps) sigma,
    d/dB = -mean g z', minus the entropy gradient.  Path derivative (:156-159): log q at stopped parameters, so
    g is replaced by g - dlog q/dx = g + Sigma^-1 (x - mu) and the entropy gradient drops out."""
    theta = np.asarray(theta, dtype=np.float64)
    z, eps = noise
    _, ls, _ = family.split(theta)
    x = family.sample_from_noise(theta, noise)
    g = model.grad(x)
    N = x.shape[0]
    if use_path_deriv:
        mu = family.split(theta)[0]
        g = g + np.linalg.solve(family.cov(theta), (x - mu).T).T
        value = -np.mean(model.logp(x) - family.log_density(theta, x))
        data = np.concatenate([g.sum(0), (g * eps).sum(0) * np.exp(ls), (g.T @ z).reshape(-1)]) / N
        return value, -data
    value = -(np.mean(model.logp(x)) + family.entropy(theta))
    data = np.concatenate([g.sum(0), (g * eps).sum(0) * np.exp(ls), (g.T @ z).reshape(-1)]) / N
    return value, -(data + family.entropy_grad(theta))


def _exclusive_kl_fullrank(family, model, theta, eps, use_path_deriv):
    D = family.dim
    mu, L = family.split(theta)
    z = mu + eps @ L.T
    f = model.logp(z)
    g = model.grad(z)
    N = eps.shape[0]
    if use_path_deriv:
        logq = family.log_density(theta, z)
        value = -np.mean(f - logq)
        gt = g + sla.solve_triangular(L.T, eps.T, lower=False).T   # g - dlogq/dz
        dmu = -gt.mean(0)
        dL = -np.tril(gt.T @ eps) / N
        dfree = dL.copy()
        dfree[np.diag_indices(D)] = np.diag(dL) * np.diag(L)
    else:
        value = -(np.mean(f) + family.entropy(theta))
        dmu = -g.mean(0)
        dL = -np.tril(g.T @ eps) / N
        dfree = dL.copy()
        dfree[np.diag_indices(D)] = np.diag(dL) * np.diag(L) - 1.0
    return value, np.concatenate([dmu, dfree[np.tril_indices(D)]])


# ==========================================================================
# ExclusiveKL, RGE control-variate path  (objectives.py:170-271)
# ==========================================================================
def _lower_bound(family, model, theta, z, use_path_deriv):
    if use_path_deriv:                                             # :176-179
        return np.mean(model.logp(z) - family.log_density(theta, z))
    return np.mean(model.logp(z)) + family.entropy(theta)          # :180-181


def rge_literal(family, model, theta, noise, method, use_path_deriv=False):
    """Per-sample restatement of ``RGE`` (``objectives.py:170-271``), line by line.

    autograd's ``elementwise_grad`` / ``grad`` / ``hessian`` / ``make_hvp`` of the
    model are replaced by the model's analytic derivatives; nothing else changes.
    """
    theta = np.asarray(theta, dtype=np.float64)
    z_samples = family.sample_from_noise(theta, noise)             # :171
    m_mean, cov = family.mean_and_cov(theta)                       # :172
    s_scale = np.sqrt(np.diag(cov))                                # :173
    epsilon_sample = (z_samples - m_mean) / s_scale                # :174
    lower_bound = _lower_bound(family, model, theta, z_samples, use_path_deriv)
    N = z_samples.shape[0]
    dLdm = model.grad(z_samples)                                   # :193
    dLdlns = dLdm * epsilon_sample * s_scale + 1                   # :196
    g_hat_rprm_grad = np.column_stack([dLdm, dLdlns])              # :198
    if method == 'full':                                           # :200-216
        gmu = model.grad(m_mean)[0]
        H = model.hessian(m_mean)
        Hdiag = np.diag(H)
        dLdz = gmu + np.dot(H, (s_scale * epsilon_sample).T).T
        dLds = dLdz * epsilon_sample * s_scale + 1.
        elbo_gsamps_tilde = np.column_stack([dLdz, dLds])
        dLds_mu = (Hdiag * s_scale + 1 / s_scale) * s_scale
        gsamps_tilde_mean = np.concatenate([gmu, dLds_mu])
        elbo_gsamps_cv = g_hat_rprm_grad - (elbo_gsamps_tilde - gsamps_tilde_mean)
        g_hat_rv = np.mean(elbo_gsamps_cv, axis=0)
    elif method == 'mean_only':                                    # :217-233
        scaled_samples = np.multiply(s_scale, epsilon_sample)
        a = model.grad(m_mean * np.ones_like(z_samples))
        b = model.hvp(m_mean, scaled_samples)
        g_tilde_mean_approx = a + b
        g_tilde_scale_approx_ln = np.zeros_like(g_tilde_mean_approx)
        E_g_tilde_mean = model.grad(m_mean)[0]
        E_g_tilde_scale_ln = np.zeros_like(E_g_tilde_mean)
        g_tilde = np.column_stack([g_tilde_mean_approx, g_tilde_scale_approx_ln])
        E_g_tilde = np.concatenate([E_g_tilde_mean, E_g_tilde_scale_ln])
        E_g_tilde = np.multiply(E_g_tilde, np.ones_like(g_tilde))
        g_hat_rv = np.mean(g_hat_rprm_grad - (g_tilde - E_g_tilde), axis=0)
    elif method == 'loo_diag_approx':                              # :234-255
        hvps = model.hvp(m_mean, s_scale * epsilon_sample)
        gmu = model.grad(m_mean * np.ones_like(z_samples))
        dLdz = gmu + hvps
        dLds = dLdz * (epsilon_sample * s_scale) + 1
        Hdiag_sum = np.sum(epsilon_sample * hvps, axis=0)
        Hdiag_s = (Hdiag_sum[None, :] - epsilon_sample * hvps) / float(N - 1)
        dLds_mu = (Hdiag_s + 1 / s_scale[None, :]) * s_scale
        D = int(0.5 * g_hat_rprm_grad.shape[1])
        g_hat_rv = g_hat_rprm_grad.copy()
        g_hat_rv[:, :D] -= hvps
        g_hat_rv[:, D:] -= (dLds - dLds_mu)
        g_hat_rv = np.mean(g_hat_rv, axis=0)
    elif method == 'loo_direct_approx':                            # :256-268
        gmu = model.grad(m_mean * np.ones_like(z_samples))
        hvps = model.hvp(m_mean, s_scale * epsilon_sample)
        dLdz = gmu + hvps
        dLds = (dLdz * epsilon_sample + 1 / s_scale[None, :]) * s_scale
        dLds_sum = np.sum(dLds, axis=0)
        dLds_mu = (dLds_sum[None, :] - dLds) / float(N - 1)
        elbo_gsamps_tilde_centered = np.column_stack([hvps, dLds - dLds_mu])
        g_hat_rv = np.mean(g_hat_rprm_grad - elbo_gsamps_tilde_centered, axis=0)
    else:
        raise RuntimeError("Invalid hessian approximation method!")
    return -lower_bound, -g_hat_rv                                 # :271


def rge_reduced(family, model, theta, noise, method, use_path_deriv=False):
    """Single-pass algebraic reduction of ``rge_literal``.

    With gbar = mean g, ge = mean g*eps, ebar = mean eps, M2 = E'E/N:
      mean block (all methods)        gbar - H (s * ebar)
      scale block mean_only/loo_direct  ge*s + 1
      scale block loo_diag              ge*s + 1 - gmu*s*ebar
      scale block full   ge*s + 1 - gmu*s*ebar - s_i sum_j H_ij s_j M2_ij + H_ii s_i^2
    This is the form the device kernels accumulate.
    """
    theta = np.asarray(theta, dtype=np.float64)
    z = family.sample_from_noise(theta, noise)
    m, cov = family.mean_and_cov(theta)
    s = np.sqrt(np.diag(cov))
    eps = (z - m) / s
    lower_bound = _lower_bound(family, model, theta, z, use_path_deriv)
    g = model.grad(z)
    gbar = g.mean(0)
    ge = (g * eps).mean(0)
    ebar = eps.mean(0)
    gmu = model.grad(m)[0]
    mean_block = gbar - model.hvp(m, (s * ebar)[None, :])[0]
    scale_block = ge * s + 1.0
    if method in ('mean_only', 'loo_direct_approx'):
        pass
    elif method == 'loo_diag_approx':
        scale_block = scale_block - gmu * s * ebar
    elif method == 'full':
        H = model.hessian(m)
        M2 = eps.T @ eps / eps.shape[0]
        scale_block = (scale_block - gmu * s * ebar
                       - s * np.sum(H * M2 * s[None, :], axis=1) + np.diag(H) * s * s)
    else:
        raise RuntimeError("Invalid hessian approximation method!")
    return -lower_bound, -np.concatenate([mean_block, scale_block])


# ==========================================================================
# AlphaDivergence  (objectives.py:443-463)
# ==========================================================================
def alpha_divergence(family, model, theta, noise, alpha):
    """``objective_grad_and_log_norm`` (``:453-461``) with the VJP written out.

    ``grad = alpha/N * sum_n s_n d/dtheta [f(z_n(theta)) - log q(z_n(theta); theta)]``
    (total derivative; note ``s`` is *not* normalised by ``mean(s)``, ``:460``).
    """
    theta = np.asarray(theta, dtype=np.float64)
    z = family.sample_from_noise(theta, noise)                     # :444
    lw = model.logp(z) - family.log_density(theta, z)              # :445
    log_norm = np.max(lw)                                          # :457
    sv = np.exp(lw - log_norm) ** alpha                            # :458
    value = np.log(np.mean(sv)) / alpha + log_norm                 # :459
    g = model.grad(z)
    N = z.shape[0]
    if isinstance(family, fam.FullRankGaussian):
        D = family.dim
        mu, L = family.split(theta)
        # log q(z(theta); theta) = -1/2|eps|^2 - sum log L_ii - c  => only the log-det term moves
        dmu = (sv[:, None] * g).sum(0)
        dL = np.tril((sv[:, None] * g).T @ noise)
        dfree = dL.copy()
        dfree[np.diag_indices(D)] = np.diag(dL) * np.diag(L) + np.sum(sv)
        grad = alpha * np.concatenate([dmu, dfree[np.tril_indices(D)]]) / N
        return value, grad
    if isinstance(family, fam.MultivariateT):
        # z = mu + (e R) / s, R = sqrtm(L L'): the Mahalanobis distance of a sample is |e / s|^2 whatever theta
        # is, so of log q(z(theta); theta) again only -sum log L_ii moves; R -> L as in _exclusive_kl_mvt
        D = family.dim
        chi, e = noise
        _, Sigma = family.split(theta)
        L = fam.free_to_chol(theta[D:], D)
        es = e / np.sqrt(chi / family.df)[:, None]
        sg = sv[:, None] * g
        X = sqrt_root_vjp(Sigma, sg.T @ es)
        dL = np.tril(2.0 * X @ L)
        dL[np.diag_indices(D)] = np.diag(dL) * np.diag(L) + np.sum(sv)
        grad = alpha * np.concatenate([sg.sum(0), dL[np.tril_indices(D)]]) / N
        return value, grad
    if isinstance(family, fam.LRGaussian):
        # x = mu + z B' + sigma eps; total derivative of lw_n = f(x_n(theta)) - log q(x_n(theta); theta).  With
        # r = x - mu, a = Sigma^-1 r, Q = r' a:  dQ = 2 a' dr - a' dSigma a, dr = dB z + sigma eps dls,
        # dSigma = dB B' + B dB' + 2 diag(sigma^2 dls), d logdet = tr(Sigma^-1 dSigma)
        mu, ls, B = family.split(theta)
        zl, eps = noise
        sig = np.exp(ls)
        Sinv = np.linalg.inv(family.cov(theta))
        a = (z - mu) @ Sinv
        sw = sv[:, None]
        dmu = (sw * g).sum(0)
        dls = (sw * (g * sig * eps + a * sig * eps - a ** 2 * sig ** 2)).sum(0) + np.sum(sv) * sig ** 2 * np.diag(Sinv)
        dB = (sw * g).T @ zl + (sw * a).T @ (zl - a @ B) + np.sum(sv) * Sinv @ B
        grad = alpha * np.concatenate([dmu, dls, dB.reshape(-1)]) / N
        return value, grad
    mu, ls = family.split(theta)
    sig = np.exp(ls)
    # log q(z(theta);theta) = sum_d base_logpdf(noise) - sum(ls): d/dls = -1, d/dmu = 0
    dmu = (sv[:, None] * g).sum(0)
    dls = (sv[:, None] * (g * noise * sig + 1.0)).sum(0)
    grad = alpha * np.concatenate([dmu, dls]) / N                  # :460
    return value, grad


# ==========================================================================
# DISInclusiveKL  (objectives.py:283-416)
# ==========================================================================
class DISInclusiveKL:
    """Stateful restatement of the reference class.

    The caller injects the noise for each state refresh and the resampling
    indices (``np.random.choice`` on the global numpy RNG in the reference,
    ``:408``), so the oracle and the engine consume identical randomness.
    """

    def __init__(self, family, model, num_mc_samples, ess_target, temper_family,
                 temper_prior_params, use_resampling=True, num_resampling_batches=1,
                 w_clip_threshold=10):
        self.family, self.model = family, model
        self.num_mc_samples = num_mc_samples
        self._ess_target = ess_target                               # :308
        self._w_clip_threshold = w_clip_threshold
        self._max_bisection_its = 50
        self._max_eps = self._eps = 1
        self._use_resampling = use_resampling
        self._num_resampling_batches = num_resampling_batches
        self._resampling_batch_size = max(1, ess_target // num_resampling_batches)   # :314
        self._objective_step = 0
        self.temper_family = temper_family
        self.temper_prior_params = np.asarray(temper_prior_params, dtype=np.float64)

    def needs_refresh(self):                                        # :392
        return (not self._use_resampling
                or self._objective_step % self._num_resampling_batches == 0)

    def _weights(self, eps, log_prior, log_p, log_q):               # :317-331
        logw = eps * log_prior + (1 - eps) * log_p - log_q
        if np.max(logw) == -np.inf:
            raise ValueError('All weights zero! Suggests overflow in importance density.')
        return np.exp(logw)

    @staticmethod
    def _ess(w):                                                    # :333-336
        return (np.sum(w) ** 2.0) / np.sum(w ** 2.0)

    def _eps_and_weights(self, eps_guess, log_prior, log_p, log_q):  # :338-368
        lower, upper = 0., eps_guess
        eps_guess = (lower + upper) / 2.
        for _ in range(self._max_bisection_its):
            w = self._weights(eps_guess, log_prior, log_p, log_q)
            if self._ess(w) > self._ess_target:
                upper = eps_guess
            else:
                lower = eps_guess
            eps_guess = (lower + upper) / 2.
        w = self._weights(eps_guess, log_prior, log_p, log_q)
        ess = self._ess(w)
        if lower == 0.:
            eps_guess = 0.
        if upper == self._max_eps:
            eps_guess = self._max_eps
        return eps_guess, ess, w

    def _clip(self, w):                                             # :370-386
        """Weight clipping.  The reference's recursion (``:370-386``) cannot run (``:385`` calls a float), and its
        evident intent -- set every weight at or above ``thr * sum(w)`` to the value that makes it exactly
        ``thr * sum(w_new)``, repeat -- is not well defined in floating point as a literal recursion: a clipped weight
        EQUALS the next level's threshold up to rounding, so ``np.any(w > S * thr)`` re-triggers on rounding noise and
        the recursion does not terminate for about one weight vector in ten (measured, thr in {0.01, 0.05, 0.2}).
        Restated as the fixed point the recursion aims at: the clipped set only grows; each round the UNCLIPPED
        weights are compared with ``thr * S``, ``S = U / (1 - thr n)`` the total after clipping ``n`` weights with
        ``U`` the sum of the unclipped ones; stop when no unclipped weight reaches it.  Equal to the recursion in exact
        arithmetic; the default threshold 10 never triggers (``w <= sum w``).  PARITY UNPINNED for this branch: no
        reference output can exist for it."""
        thr = self._w_clip_threshold
        w = np.asarray(w, dtype=np.float64)
        clipped = np.zeros(w.shape, dtype=bool)
        S = np.sum(w)
        if not np.any(w > S * thr):                                 # :373-374
            return w
        while True:
            new = ~clipped & (w >= S * thr)                         # :375
            if not np.any(new):
                break
            trial = clipped | new
            n = np.sum(trial)                                       # :376
            U = np.sum(w[~trial])                                   # :377
            if U == 0 or 1. - thr * n <= 0:                         # :378-379
                break
            clipped = trial
            S = U / (1. - thr * n)
        if not np.any(clipped):
            return w
        out = w.copy()
        out[clipped] = thr * np.sum(w[~clipped]) / (1. - thr * np.sum(clipped))     # :385 (intent)
        return out

    def refresh(self, theta, noise):                                # :393-401
        theta = np.asarray(theta, dtype=np.float64)
        self._state_samples = self.family.sample_from_noise(theta, noise)
        self._state_log_q = self.family.log_density(theta, self._state_samples)
        self._state_log_p = self.model.logp(self._state_samples)
        log_prior = self.temper_family.log_density(self.temper_prior_params,
                                                   self._state_samples)
        self._eps, self._ess_val, w = self._eps_and_weights(
            self._eps, log_prior, self._state_log_p, self._state_log_q)
        self._state_w_clipped = self._clip(w)
        self._state_w_sum = np.sum(self._state_w_clipped)
        self._state_w_normalized = self._state_w_clipped / self._state_w_sum

    def __call__(self, theta, noise=None, indices=None):
        """``noise`` is consumed only on refresh steps; ``indices`` only when resampling."""
        theta = np.asarray(theta, dtype=np.float64)
        if self.needs_refresh():
            self.refresh(theta, noise)
        self._objective_step += 1                                   # :403
        N = self.num_mc_samples
        if not self._use_resampling:                                # :405-406
            # log q here is the *state* log q, which autograd still tracks through
            # log_density's theta (samples are stopped)
            lq = self.family.log_density(theta, self._state_samples)
            value = -np.inner(self._state_w_clipped, lq) / N
            grad = -self.family.log_density_grad_weighted(
                theta, self._state_samples, self._state_w_clipped) / N
            return value, grad
        xs = self._state_samples[indices]                           # :410
        M = len(indices)
        scale = self._state_w_sum / N                               # :414
        value = np.mean(-self.family.log_density(theta, xs)) * scale    # :412-414
        grad = -self.family.log_density_grad_weighted(theta, xs, np.ones(M)) / M * scale
        return value, grad
