"""Stochastic optimisers that drive the HIP objectives: the callers of the hot path.

Host-side counterparts of ``viabel/optimization.py`` with the same class names, constructor
arguments, result dictionaries and messages: plain SGD loop (``:83-127``), RMSProp (``:147-197``),
AveragedRMSProp (``:200-258``), Adam (``:260-326``), AveragedAdam (``:328-396``), Adagrad (``:398-433``),
WindowedAdagrad (``:435-476``), FASO (``:479-633``) and RAABBVI (``:635-931``).  Per iteration they do
O(var_param_dim) numpy arithmetic around one ``objective(var_param)`` call, which is where the GPU
work happens.

Deviation: RAABBVI fits its three-parameter weighted regression (``stan_models/weighted_lin_regression*.stan``)
with PyStan NUTS (``optimization.py:677-725``).  PyStan is not available; the same posterior is sampled
here by a seeded, adaptive random-walk Metropolis sampler (``_weighted_regression_posterior``), which
like Stan returns posterior means of ``kappa`` and ``log c``.
"""
from abc import ABC, abstractmethod
from collections import defaultdict
import time

import numpy as np
import tqdm

from . import _lib
from ._chain_stats import MCSE, R_hat_convergence_check
from .approximations import MFGaussian

__all__ = [
    'Optimizer',
    'StochasticGradientOptimizer',
    'RMSProp',
    'Adam',
    'Adagrad',
    'WindowedAdagrad',
    'AveragedRMSProp',
    'AveragedAdam',
    'FASO',
    'RAABBVI'
]


class _Stopwatch:
    def __enter__(self):
        self._t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        self.interval = time.perf_counter() - self._t0


class Optimizer(ABC):
    """Abstract optimiser: ``optimize`` returns a dict with at least ``opt_param``."""

    @abstractmethod
    def optimize(self, n_iters, objective, init_param, **kwargs):
        """Run ``n_iters`` iterations from ``init_param`` on ``objective``."""


class StochasticGradientOptimizer(Optimizer):
    """Stochastic gradient descent with optional tail averaging of the iterates."""

    def __init__(self, learning_rate, *, weight_decay=0, iterate_avg_prop=0.2, diagnostics=False):
        self._learning_rate = learning_rate
        self._weight_decay = weight_decay
        if iterate_avg_prop is not None and not 0.0 < iterate_avg_prop <= 1.0:
            raise ValueError('"iterate_avg_prop" must be None or between 0 and 1')
        self._iterate_avg_prop = iterate_avg_prop
        self._diagnostics = diagnostics
        self.reset_state()

    def reset_state(self):
        """Forget the optimiser's running statistics."""

    def optimize(self, n_iters, objective, init_param, init_hamflow_model_param=None,
                 init_hamflow_rho_param=None, on_device=None):
        """The reference's loop (``optimization.py:83-127``).  ``on_device``: run the whole loop on the GPU
        without host round trips (``VariationalObjective.device_fit``); the default ``None`` does so whenever
        the objective and the optimiser support it -- the trajectory is the same bit for bit."""
        if on_device is None:
            on_device = self._device_fit_possible(objective, init_param)
        elif on_device and not self._device_fit_possible(objective, init_param):
            raise NotImplementedError('this optimiser / objective pair has no device-resident loop')
        if on_device:
            return self._optimize_on_device(n_iters, objective, init_param)
        param = init_param.copy()
        tail = self._iterate_avg_prop
        log = defaultdict(list)
        k = 0
        with tqdm.trange(n_iters) as bar:
            try:
                for k in bar:
                    value, grad = objective(param)
                    direction = self.descent_direction(grad)
                    param = objective.update(param, self._learning_rate * direction)
                    if param.ndim == 2:
                        param *= (1 - self._weight_decay)
                    log['value_history'].append(value)
                    if self._diagnostics or tail is not None:
                        log['variational_param_history'].append(param.copy())
                        if tail is not None and len(log['variational_param_history']) > tail * k:
                            log['variational_param_history'].pop(0)
                    if self._diagnostics:
                        log['descent_dir_history'].append(direction)
                    if k % 10 == 0:
                        recent = np.mean(log['value_history'][max(0, k - 1000):k + 1])
                        bar.set_description('average loss = {:,.5g}'.format(recent))
            except (KeyboardInterrupt, StopIteration):  # pragma: no cover
                pass
            finally:
                bar.close()
        if tail is not None:
            window = max(1, int(k * tail))
            log['opt_param'] = np.mean(log['variational_param_history'][-window:], axis=0)
        else:
            log['opt_param'] = param.copy()
        return {name: np.array(h) for name, h in log.items()}

    def descent_direction(self, grad):
        """Direction to step against; plain SGD uses the gradient itself."""
        return grad

    # ---- device-resident loop -------------------------------------------------------------------------
    _device_kind = _lib.OPT_SGD

    def _device_hyper(self):
        """[learning_rate, beta / beta1, beta2, jitter] for ``vb_fit``."""
        return [self._learning_rate, 0.0, 0.0, 0.0]

    def _device_state(self, p):
        """Optimiser state as ``[second moment (p) | momentum (p)]`` or None before the first step."""
        return None

    def _set_device_state(self, state, p):
        pass

    def _device_fit_possible(self, objective, init_param):
        return (self._device_kind is not None
                and np.ndim(init_param) == 1 and np.ndim(self._learning_rate) == 0
                and getattr(objective, 'supports_device_fit', lambda: False)())

    def _optimize_on_device(self, n_iters, objective, init_param):
        tail = self._iterate_avg_prop
        # length of the iterate history the host loop ends with (append, then pop while len > tail * k)
        kept = 0
        if self._diagnostics or tail is not None:
            for k in range(n_iters):
                kept += 1
                if tail is not None and kept > tail * k:
                    kept -= 1
        p = np.size(init_param)
        theta, values, history, state, directions, _ = objective.device_fit(
            n_iters, init_param, self._device_kind, self._device_hyper(), state=self._device_state(p),
            hist_len=kept, log_directions=self._diagnostics)
        self._set_device_state(state, p)
        log = {'value_history': values}
        if self._diagnostics or tail is not None:
            log['variational_param_history'] = history
        if self._diagnostics:
            log['descent_dir_history'] = directions
        if tail is not None:
            window = max(1, int((n_iters - 1) * tail))
            # (the same mean from the rows still on the device: numpy's order of additions, no host pass over `window` iterates)
            if window <= kept and window * p >= _DEVICE_MEAN_MIN:
                log['opt_param'] = np.asarray(objective.device_history_mean(window))
            else:
                log['opt_param'] = np.mean(history[-window:], axis=0)
        else:
            log['opt_param'] = theta
        return log


# iterate averages of at least this many numbers are formed on the device (vb_fit_history_mean); below, numpy's pass is cheaper
# than a launch and a copy
_DEVICE_MEAN_MIN = 1 << 20


def _ema_update(state, decay, grad_sq):
    """state <- decay * state + (1 - decay) * grad_sq, starting from grad_sq."""
    state = grad_sq.copy() if state is None else state
    state *= decay
    state += (1.0 - decay) * grad_sq
    return state


def _adam_moments(momentum, avg_grad_sq, beta1, beta2, grad):
    """One update of (momentum, second moment) for the reference's Adam variants.

    On the FIRST call the reference aliases ``momentum = grad`` and scales it in place
    (``optimization.py:310-320``, ``:375-386``): the gradient itself becomes ``beta1 grad`` before
    ``(1 - beta1) grad`` is added, so the momentum is ``beta1 (2 - beta1) grad`` rather than ``grad``, and
    -- because the aliased gradient was overwritten before the second moment is refreshed -- the squared
    *momentum* enters the second-moment update.  Both are reproduced operation by operation so that
    trajectories match the reference; the caller's gradient array is left untouched."""
    if momentum is None:
        momentum = grad * beta1
        momentum += (1.0 - beta1) * momentum
        avg_grad_sq = grad ** 2
        avg_grad_sq *= beta2
        avg_grad_sq += (1.0 - beta2) * momentum ** 2
        return momentum, avg_grad_sq
    momentum *= beta1
    momentum += (1.0 - beta1) * grad
    avg_grad_sq *= beta2
    avg_grad_sq += (1.0 - beta2) * grad ** 2
    return momentum, avg_grad_sq


class RMSProp(StochasticGradientOptimizer):
    """RMSProp: gradient scaled by an exponential moving average of its square."""

    def __init__(self, learning_rate, *, weight_decay=0, iterate_avg_prop=0.2, beta=0.9, jitter=1e-8,
                 diagnostics=False):
        self._beta = beta
        self._jitter = jitter
        super().__init__(learning_rate, weight_decay=weight_decay, iterate_avg_prop=iterate_avg_prop,
                         diagnostics=diagnostics)

    def reset_state(self):
        self._avg_grad_sq = None

    def descent_direction(self, grad):
        self._avg_grad_sq = _ema_update(self._avg_grad_sq, self._beta, grad ** 2)
        return grad / np.sqrt(self._jitter + self._avg_grad_sq)

    _device_kind = _lib.OPT_RMSPROP

    def _device_hyper(self):
        return [self._learning_rate, self._beta, 0.0, self._jitter]

    def _device_state(self, p):
        if self._avg_grad_sq is None:
            return None
        return np.concatenate([self._avg_grad_sq, np.zeros(p)])

    def _set_device_state(self, state, p):
        self._avg_grad_sq = state[:p].copy()


class AveragedRMSProp(StochasticGradientOptimizer):
    """RMSProp with ``beta_k = 1 - 1/k``: the plain average of all squared gradients."""

    _device_kind = None      # host loop only

    def __init__(self, learning_rate, *, jitter=1e-8, diagnostics=False, component_wise=True):
        self._jitter = jitter
        self._component_wise = component_wise
        super().__init__(learning_rate, diagnostics=diagnostics)

    def reset_state(self):
        self._avg_grad_sq = None
        self._t = None

    def descent_direction(self, grad):
        self._t = 1 if self._avg_grad_sq is None else self._t + 1
        self._avg_grad_sq = _ema_update(self._avg_grad_sq, 1 - 1 / self._t, grad ** 2)
        scale = self._avg_grad_sq if self._component_wise else np.sum(self._avg_grad_sq)
        return grad / np.sqrt(self._jitter + scale)


class Adam(StochasticGradientOptimizer):
    """Adam without bias correction, as the reference implements it."""

    def __init__(self, learning_rate, *, beta1=0.9, beta2=0.999, jitter=1e-8, iterate_avg_prop=0.2,
                 diagnostics=False):
        self._beta1 = beta1
        self._beta2 = beta2
        self._jitter = jitter
        super().__init__(learning_rate, iterate_avg_prop=iterate_avg_prop, diagnostics=diagnostics)

    def reset_state(self):
        self._momentum = None
        self._avg_grad_sq = None

    def descent_direction(self, grad):
        self._momentum, self._avg_grad_sq = _adam_moments(self._momentum, self._avg_grad_sq, self._beta1,
                                                          self._beta2, grad)
        return self._momentum / np.sqrt(self._jitter + self._avg_grad_sq)

    _device_kind = _lib.OPT_ADAM

    def _device_hyper(self):
        return [self._learning_rate, self._beta1, self._beta2, self._jitter]

    def _device_state(self, p):
        if self._momentum is None:
            return None
        return np.concatenate([self._avg_grad_sq, self._momentum])

    def _set_device_state(self, state, p):
        self._avg_grad_sq, self._momentum = state[:p].copy(), state[p:].copy()


class AveragedAdam(StochasticGradientOptimizer):
    """Adam whose second moment is the plain average of all squared gradients."""

    _device_kind = None      # host loop only

    def __init__(self, learning_rate, *, beta1=0.9, jitter=1e-8, diagnostics=False, component_wise=True):
        self._beta1 = beta1
        self._jitter = jitter
        self._component_wise = component_wise
        super().__init__(learning_rate, diagnostics=diagnostics)

    def reset_state(self):
        self._momentum = None
        self._avg_grad_sq = None
        self._t = None

    def descent_direction(self, grad):
        self._t = 1 if self._avg_grad_sq is None else self._t + 1
        self._momentum, self._avg_grad_sq = _adam_moments(self._momentum, self._avg_grad_sq, self._beta1,
                                                          1 - 1 / self._t, grad)
        scale = self._avg_grad_sq if self._component_wise else np.sum(self._avg_grad_sq)
        return self._momentum / np.sqrt(self._jitter + scale)


class Adagrad(StochasticGradientOptimizer):
    """Adagrad: gradient scaled by the root of the accumulated squared gradients."""

    def __init__(self, learning_rate, *, weight_decay=0, jitter=1e-8, iterate_avg_prop=0.2,
                 diagnostics=False):
        self._jitter = jitter
        super().__init__(learning_rate, weight_decay=weight_decay, iterate_avg_prop=iterate_avg_prop,
                         diagnostics=diagnostics)

    def reset_state(self):
        self._sum_grad_sq = 0

    def descent_direction(self, grad):
        self._sum_grad_sq = self._sum_grad_sq + grad ** 2
        return grad / np.sqrt(self._jitter + self._sum_grad_sq)

    _device_kind = _lib.OPT_ADAGRAD

    def _device_hyper(self):
        return [self._learning_rate, 0.0, 0.0, self._jitter]

    def _device_state(self, p):
        if np.ndim(self._sum_grad_sq) == 0:
            return None
        return np.concatenate([self._sum_grad_sq, np.zeros(p)])

    def _set_device_state(self, state, p):
        self._sum_grad_sq = state[:p].copy()


class WindowedAdagrad(StochasticGradientOptimizer):
    """Adagrad over a sliding window of the last ``window_size`` squared gradients."""

    _device_kind = None      # host loop only

    def __init__(self, learning_rate, *, weight_decay=0, window_size=10, jitter=1e-8, diagnostics=False):
        self._window_size = window_size
        self._jitter = jitter
        super().__init__(learning_rate, weight_decay=weight_decay, diagnostics=diagnostics)

    def reset_state(self):
        self._history = []

    def descent_direction(self, grad):
        self._history.append(grad ** 2)
        del self._history[:-self._window_size]
        return grad / np.sqrt(self._jitter + np.mean(self._history, axis=0))


class FASO(Optimizer):
    """Fixed-learning-rate automated stochastic optimisation (https://arxiv.org/abs/2203.15945).

    Runs ``sgo`` at a fixed learning rate; detects stationarity with split R-hat over trailing windows
    of the iterates and stops once the Monte Carlo standard error of the iterate average is below
    ``mcse_threshold`` with at least ``ESS_min`` effective samples.
    """

    def __init__(self, sgo, *, mcse_threshold=0.1, W_min=200, ESS_min=None, k_check=None):
        if not isinstance(sgo, StochasticGradientOptimizer):
            raise ValueError('sgo must be a subclass of StochasticGradientOptimizer')
        self._sgo = sgo
        self._mcse_threshold = mcse_threshold
        self._W_min = W_min
        self._ESS_min = W_min // 8 if ESS_min is None else ESS_min
        self._k_check = W_min if k_check is None else k_check
        if mcse_threshold <= 0:
            raise ValueError('"mcse_threshold" must be greater than zero')
        if W_min <= 0:
            raise ValueError('"W_min" must be greater than zero')
        if self._k_check <= 0:
            raise ValueError('"k_check" must be greater than zero')
        if self._ESS_min <= 0:
            raise ValueError('"ESS_min" must be greater than zero')

    def _iterate_mcse(self, objective, iterates, dim_hint):
        """ESS and MCSE of the converged iterates; mean-field Gaussians report MCSE(mu)/sigma."""
        W = iterates.shape[0]
        if isinstance(objective.approx, MFGaussian):
            dim = int(dim_hint / 2)
            frozen = (iterates[W - 2, :] - iterates[W - 1, :]) == 0
            if np.any(frozen):          # constant coordinates carry no Monte Carlo error
                iterates = np.delete(iterates, np.argwhere(frozen), 1)
            mean_log_sd = np.mean(iterates[:, -dim:], axis=0)
            ess, mcse = MCSE(iterates)
            mcse = np.concatenate((mcse[:dim] / np.exp(mean_log_sd), mcse[-dim:]))
            return ess, mcse
        return MCSE(iterates)

    def _next_check(self, k, k_conv, W_check, n_iters):
        """First iteration >= k whose bookkeeping can do anything: the next stationarity check (multiples of
        k_check) before convergence, the next MCSE check (k_conv + W_check) after it."""
        if k_conv is None:
            k_end = -(-k // self._k_check) * self._k_check
        else:
            k_end = max(k, k_conv + W_check)
        return min(k_end, n_iters - 1)

    def optimize(self, n_iters, objective, init_param, on_device=None):
        """``optimization.py:521-633``.  ``on_device`` (default: whenever possible): the iterations between two
        convergence checks run as one device-resident chunk (``vb_fit``) instead of one blocking objective
        call + numpy step each; iterates, gradients and values are the same bit for bit, only the wall-clock
        ratio that paces the MCSE re-checks differs."""
        sgo = self._sgo
        if on_device is None:
            on_device = sgo._device_fit_possible(objective, init_param)
        elif on_device and not sgo._device_fit_possible(objective, init_param):
            raise NotImplementedError('this optimiser / objective pair has no device-resident loop')
        done_until = -1      # device mode: iterations up to here are already in `hist`
        diagnostics = self._sgo._diagnostics
        k_conv = k_stopped = k_Rhat = None
        lr = self._sgo._learning_rate
        param = init_param.copy()
        hist = defaultdict(list)
        iterate_average = param.copy()
        if diagnostics:
            hist['iterate_average_k_history'].append(0)
            hist['iterate_average_history'].append(iterate_average)
        opt_time = 0.0
        mcse = ess = None
        W_check = None
        with tqdm.trange(n_iters) as bar:
            try:
                for k in bar:
                    if on_device:
                        if k > done_until:
                            done_until = self._next_check(k, k_conv, W_check, n_iters)
                            count = done_until - k + 1
                            with _Stopwatch() as sw:
                                param, values, iterates, state, dirs, grads = objective.device_fit(
                                    count, param, sgo._device_kind, sgo._device_hyper(),
                                    state=sgo._device_state(param.size), hist_len=count,
                                    log_directions=diagnostics, log_gradients=True)
                                sgo._set_device_state(state, param.size)
                                hist['value_history'].extend(values)
                                hist['grad_history'].extend(grads)
                                hist['variational_param_history'].extend(iterates)
                                if diagnostics:
                                    hist['descent_dir_history'].extend(dirs)
                            opt_time += sw.interval
                        if k < done_until:
                            continue
                    else:
                        with _Stopwatch() as sw:
                            value, grad = objective(param)
                            hist['value_history'].append(value)
                            hist['grad_history'].append(grad)
                            direction = self._sgo.descent_direction(grad)
                            param = objective.update(param, lr * direction)
                            hist['variational_param_history'].append(param.copy())
                            if diagnostics:
                                hist['descent_dir_history'].append(direction)
                        opt_time += sw.interval
                    # stationarity: split R-hat over five trailing windows
                    if k_conv is None and k % self._k_check == 0:
                        W_upper = int(0.95 * k)
                        if W_upper > self._W_min:
                            windows = np.linspace(self._W_min, W_upper, num=5, dtype=int)
                            ok, best_W = R_hat_convergence_check(hist['variational_param_history'], windows)
                            iterate_average = np.mean(hist['variational_param_history'][-best_W:], axis=0)
                            if diagnostics:
                                hist['iterate_average_k_history'].append(k)
                                hist['iterate_average_history'].append(iterate_average)
                            if ok:
                                k_Rhat = k
                                k_conv = k - best_W
                                W_check = best_W
                    # after stationarity: Monte Carlo standard error of the iterate average
                    if k_conv is not None and k - k_conv == W_check:
                        W = W_check
                        converged = np.array(hist['variational_param_history'][-W:])
                        iterate_average = np.mean(converged, axis=0)
                        if diagnostics and k not in hist['iterate_average_k_history']:
                            hist['iterate_average_k_history'].append(k)
                            hist['iterate_average_history'].append(iterate_average)
                        with _Stopwatch() as sw_mcse:
                            ess, mcse = self._iterate_mcse(objective, converged, init_param.size)
                        if diagnostics:
                            hist['ess_and_mcse_k_history'].append(k)
                            hist['ess_history'].append(ess)
                            hist['mcse_history'].append(mcse)
                        if np.max(mcse) < self._mcse_threshold and np.min(ess) > self._ESS_min:
                            k_stopped = k
                            break
                        # re-check later, the more so the costlier the check is relative to a step
                        ratio = (opt_time / k) / (sw_mcse.interval / W)
                        W_check = int(max(1.05, 1 + 1 / np.sqrt(1 + ratio)) * W_check + 1)
                    if k % self._k_check == 0:
                        recent = np.mean(hist['value_history'][max(0, k - 1000):k + 1])
                        state = 'converged' if k_conv is not None else 'not converged'
                        bar.set_description('average loss = {:,.5g} | R hat {}|'.format(recent, state))
            except (KeyboardInterrupt, StopIteration):  # pragma: no cover
                pass
            finally:
                bar.close()
        if k_stopped is None:
            if k_conv is None:
                print('WARNING: stationarity not reached after maximum number of iterations')
                print('WARNING: try incresing the learning rate or the maximum number of '
                      'iterations')
            else:
                print('WARNING: stationarity reached but MCSE too large and/or ESS too small')
                print('WARNING: maximum MCSE = {:.3g}'.format(np.max(mcse)))
                print('WARNING: minimum ESS = {:.1f}'.format(np.min(ess)))
        else:
            print('Convergence reached at iteration', k_stopped)
        results = {name: np.array(h) for name, h in hist.items()}
        results['k_conv'] = k_conv
        results['k_Rhat'] = k_Rhat
        results['k_stopped'] = k_stopped
        results['opt_param'] = iterate_average
        return results


def _weighted_regression_posterior(y, x, w, rho, fixed_kappa, n_chains=4, n_warmup=1500, n_draws=2500,
                                   seed=20220331):
    """Posterior means of (kappa, log c) for the weighted regression of RAABBVI.

    Model (``viabel/stan_models/weighted_lin_regression.stan:19-29`` and ``_sgd.stan``):
    ``y_n ~ Normal(log c + 2 log(rho^-kappa - 1) + 2 kappa x_n, sigma)`` with each log-likelihood term
    weighted by ``w_n``; priors ``kappa ~ U(0,1)``, ``log c ~ Cauchy(0,10)``, ``sigma ~ HalfCauchy(0,10)``.
    With ``fixed_kappa`` the power is 1 (the Averaged* optimisers, ``_sgd`` variant).

    The reference samples this with PyStan NUTS; here: a vectorised random-walk Metropolis on
    ``(logit kappa, log c, log sigma)`` with per-chain step-size adaptation during warm-up, seeded.
    """
    y, x, w = (np.asarray(a, dtype=np.float64) for a in (y, x, w))
    rs = np.random.RandomState(seed)

    def log_post(u):                       # u: (chains, 3)
        lk, log_c, ls = u[:, 0], u[:, 1], u[:, 2]
        kappa = np.ones_like(lk) if fixed_kappa else 1.0 / (1.0 + np.exp(-lk))
        sigma = np.exp(ls)
        mu = (log_c + 2 * np.log(rho ** (-kappa) - 1.0))[:, None] + 2 * kappa[:, None] * x[None, :]
        resid = (y[None, :] - mu) / sigma[:, None]
        ll = np.sum(w[None, :] * (-0.5 * resid ** 2 - ls[:, None] - 0.5 * np.log(2 * np.pi)), axis=1)
        lp = -np.log1p((log_c / 10.0) ** 2) - np.log1p((sigma / 10.0) ** 2) + ls      # priors + |d sigma/d ls|
        if not fixed_kappa:
            lp = lp + np.log(kappa) + np.log1p(-kappa)                                   # |d kappa/d lk|
        out = ll + lp
        return np.where(np.isfinite(out), out, -np.inf)

    u = np.tile(np.array([0.0 if fixed_kappa else np.log(0.8 / 0.2), 0.0, np.log(5.0)]), (n_chains, 1))
    u[:, 1] = np.mean(y) if y.size else 0.0
    u = u + 0.1 * rs.randn(n_chains, 3)
    step = np.tile(np.array([0.0 if fixed_kappa else 0.5, 1.0, 0.5]), (n_chains, 1))
    cur = log_post(u)
    draws = np.empty((n_draws, n_chains, 3))
    accepted = np.zeros(n_chains)
    for it in range(n_warmup + n_draws):
        prop = u + step * rs.randn(n_chains, 3)
        new = log_post(prop)
        take = np.log(rs.rand(n_chains)) < new - cur
        u[take] = prop[take]
        cur[take] = new[take]
        accepted += take
        if it < n_warmup and (it + 1) % 50 == 0:      # aim at ~30 % acceptance
            rate = accepted / 50.0
            step *= np.exp(rate - 0.3)[:, None]
            accepted[:] = 0
        if it >= n_warmup:
            draws[it - n_warmup] = u
    lk = draws[:, :, 0].ravel()
    kappa_draws = np.ones_like(lk) if fixed_kappa else 1.0 / (1.0 + np.exp(-lk))
    return kappa_draws, draws[:, :, 1].ravel()


class RAABBVI(FASO):
    """Robust, automated and accurate BBVI (https://arxiv.org/abs/2203.15945).

    Repeats FASO at learning rates ``gamma, rho gamma, rho^2 gamma, ...``; after each epoch it regresses
    the symmetrised KL between successive iterate averages on the learning rate and stops when the
    predicted accuracy gain no longer justifies the predicted number of iterations.
    """

    def __init__(self, sgo, *, rho=0.5, iters0=1000, accuracy_threshold=0.1, inefficiency_threshold=1.0,
                 init_rmsprop=False, **kwargs):
        super().__init__(sgo, **kwargs)
        self._iters0 = iters0
        self._rho = rho
        self._accuracy_threshold = accuracy_threshold
        self._inefficiency_threshold = inefficiency_threshold
        self._init_rmsprop = init_rmsprop
        if rho < 0 or rho > 1:
            raise ValueError('"rho" must be between zero and one')

    def _averaged_sgo(self):
        return isinstance(self._sgo, (AveragedRMSProp, AveragedAdam))

    def weighted_linear_regression(self, model, y, x, s=9, a=0.25, n_chains=4):
        """Posterior-mean ``kappa`` and ``c`` of log SKL vs log learning rate.

        ``model`` is ignored (kept for signature compatibility with the PyStan-based reference);
        returns ``(fit, kappa, c)`` where ``fit`` is a dict of posterior draws.
        """
        y = np.asarray(y, dtype=np.float64)
        N = len(y)
        w = np.array(1 / (1 + np.arange(N)[::-1] ** 2 / s) ** a)
        kappa_draws, log_c_draws = _weighted_regression_posterior(y, x, w, self._rho, self._averaged_sgo(),
                                                                  n_chains=n_chains)
        kappa = 1 if self._averaged_sgo() else np.mean(kappa_draws)
        fit = {'kappa': kappa_draws, 'log_c': log_c_draws}
        return fit, kappa, np.exp(np.mean(log_c_draws))

    def wls(self, x, y, s=9, a=0.25):
        """Weighted least squares ``y ~ b0 + b1 x`` with weights decaying into the past."""
        y = np.asarray(y, dtype=np.float64)
        n = y.size
        X = np.column_stack((np.ones(n), x))
        w = 1 / (1 + np.arange(n)[::-1] ** 2 / s ** 2) ** a
        XtW = X.T * w
        beta = np.linalg.inv(XtW @ X) @ (XtW @ y.reshape(n, 1))
        return beta[0], beta[1]

    def convg_iteration_trend_detection(self, slope):
        """True when fewer iterations were needed at larger learning rates (negative slope)."""
        return bool(slope < 0)

    def optimize(self, K_max, objective, init_param, on_device=None):
        if not objective.approx.supports_kl:
            print('WARNING: approximation family does not support KL. Using FASO.', flush=True)
            return super().optimize(K_max, objective, init_param, on_device=on_device)
        k_new = -1           # iterations spent at the current learning rate
        epoch = 0
        k_total = 0
        k_add = 0
        k_stopped_final = None
        sgo = self._sgo
        diagnostics = sgo._diagnostics
        average = init_param.copy()
        hist = defaultdict(list)
        hist['iterate_average_curr_hist'].append(average)
        hist['k_mcse'].append(0)
        stopped = False
        inefficiency = None
        try:
            while not stopped:
                K_max -= (k_new + 1)
                previous = average
                if epoch == 0 and self._init_rmsprop:
                    opt = FASO(sgo=RMSProp(learning_rate=sgo._learning_rate, diagnostics=diagnostics)) \
                        .optimize(K_max, objective, average, on_device=on_device)
                else:
                    opt = super().optimize(K_max, objective, average, on_device=on_device)
                if opt['k_stopped'] is not None and epoch != 0:
                    hist['conv_iters_hist'].append(opt['k_stopped'])
                average = opt['opt_param']
                hist['iterate_average_curr_hist'].append(average)
                k_new = opt['k_stopped']
                shift = k_add if k_new is not None else None
                hist['k_Rhat'].append(opt['k_Rhat'] + shift if opt['k_Rhat'] is not None and shift is not None
                                      else opt['k_Rhat'])
                hist['k_conv'].append(opt['k_conv'] + shift if opt['k_conv'] is not None and shift is not None
                                      else opt['k_conv'])
                hist['k_mcse'].append(k_new + k_add if k_new is not None else k_new)
                for name in ('variational_param_history', 'value_history', 'grad_history'):
                    hist[name].extend(opt[name])
                if diagnostics:
                    hist['descent_dir_history'].extend(opt['descent_dir_history'])
                    if opt['k_conv'] is not None:
                        hist['ess_history'].extend(opt['ess_history'])
                        hist['mcse_history'].extend(opt['mcse_history'])
                        hist['final_mcse_history'].append(hist['mcse_history'][-1] if hist['mcse_history']
                                                          else hist['mcse_history'])
                    if epoch == 0:
                        hist['iterate_average_k_history'].extend(opt['iterate_average_k_history'])
                        hist['iterate_average_history'].extend(opt['iterate_average_history'])
                    else:
                        hist['iterate_average_k_history'].extend(opt['iterate_average_k_history'][1:] + k_add)
                        hist['iterate_average_history'].extend(opt['iterate_average_history'][1:, :])
                if hist['iterate_average_k_history']:
                    k_add = hist['iterate_average_k_history'][-1]
                if k_new is None:          # iteration budget exhausted
                    break
                k_total += k_new
                sgo._learning_rate *= self._rho
                self._mcse_threshold *= self._rho
                if self._averaged_sgo():
                    sgo.reset_state()
                if len(hist['learning_rate_hist']) > 0:
                    skl = (objective.approx.kl(previous, average) + objective.approx.kl(average, previous))
                    hist['SKL_history'].append(skl)
                    fit, kappa, c = self.weighted_linear_regression(None, np.log(hist['SKL_history']),
                                                                    np.log(hist['learning_rate_hist']))
                    if diagnostics:
                        hist['c_sample_hist'].append(np.exp(fit['log_c']))
                        if self._averaged_sgo():
                            hist['kappa_sample_hist'] = None
                        else:
                            hist['kappa_sample_hist'].append(fit['kappa'])
                    hist['kappa_hist'].append(kappa)
                    hist['c_hist'].append(c)
                    if len(hist['learning_rate_hist']) > 1:
                        lr_last = hist['learning_rate_hist'][-1]
                        relative_skl = self._rho ** kappa + self._accuracy_threshold / (np.sqrt(c) * lr_last ** kappa)
                        curr_iters = hist['conv_iters_hist'][-1]
                        _, slope = self.wls(np.log(hist['learning_rate_hist']), np.log(hist['conv_iters_hist']))
                        if self.convg_iteration_trend_detection(float(slope[0])):
                            y_wls, x_wls = hist['conv_iters_hist'], hist['learning_rate_hist']
                        else:              # drop the first (transient) epoch
                            y_wls, x_wls = hist['conv_iters_hist'][1:], hist['learning_rate_hist'][1:]
                        b0, b1 = self.wls(np.log(x_wls), np.log(y_wls))
                        pred_iters = int(np.exp(float(b0[0])) * (self._rho * lr_last) ** float(b1[0]))
                        hist['predicted_iters_hist'].append(pred_iters)
                        inefficiency = relative_skl * pred_iters / (curr_iters + self._iters0)
                        hist['stopping_crt'].append(inefficiency)
                        if inefficiency > self._inefficiency_threshold:
                            stopped = True
                            k_stopped_final = k_total
                            hist['k_stopped_final_hist'].append(k_total)
                            break
                hist['learning_rate_hist'].append(sgo._learning_rate)
                epoch += 1
        except (KeyboardInterrupt, StopIteration):  # pragma: no cover
            pass
        if stopped:
            print('Termination rule reached at iteration', k_total)
            print('Inefficiency Index:', inefficiency)
        else:
            print('WARNING: maximum number of iterations reached before '
                  'stopping rule was triggered')
        results = {name: np.array(h) for name, h in hist.items() if name not in ('k_Rhat', 'k_mcse', 'k_conv')}
        results['opt_param'] = average
        results['k_stopped_final'] = k_stopped_final
        results['k_Rhat'] = hist['k_Rhat']
        results['k_mcse'] = hist['k_mcse']
        results['k_conv'] = hist['k_conv']
        return results
