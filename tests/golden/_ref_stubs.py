"""Container-only import shims that let the upstream reference's Python run.

BUILD-CONTAINER TOOLING for ``make_golden.py`` -- never imported by the product,
by the tests, or on the GPU box (the reference tree does not exist there).

The reference needs ``autograd``, ``paragami`` and ``pystan``; none is installed
here and there is no network.  These shims do NOT re-implement those packages:

* ``autograd.numpy`` / ``autograd.scipy`` are aliased to numpy / scipy, so the
  reference's own forward arithmetic (``sample``, ``log_density``, ``entropy``,
  the objective closures, the RGE control-variate code) runs unchanged;
* ``value_and_grad`` / ``vector_jacobian_product`` differentiate the reference's
  own closure by Richardson-extrapolated central differences, with
  ``autograd.core.getval`` emulated as a stop-gradient (values recorded during
  the unperturbed evaluation are replayed during perturbed ones);
* ``elementwise_grad`` / ``grad`` / ``hessian`` / ``make_hvp`` -- which the
  reference only ever applies to the *model* inside ``RGE`` -- return the analytic
  derivatives of the model registered in ``STATE['model']`` (those derivatives
  are separately checked against ``torch.autograd`` fp64);
* ``paragami`` is a ~60-line layout shim (flatten/fold in insertion order,
  free PSD matrix = Cholesky with log-diagonal in tril order) -- the layout
  recalled from paragami 0.42; it is the one part the reference tree cannot pin.
"""
import sys
import types

import numpy as np
import scipy
import scipy.linalg
import scipy.special
import scipy.stats

STATE = {
    'model': None,          # oracle model whose analytic derivatives back grad/hessian/hvp
    'before_eval': None,    # callable restoring RNG / objective state before each evaluation
    'getval_mode': 'off',   # 'off' | 'record' | 'replay'
    'getval_tape': [],
    'getval_pos': 0,
}


# ------------------------------------------------------------------ autograd
def _getval(x):
    mode = STATE['getval_mode']
    if mode == 'record':
        STATE['getval_tape'].append(np.array(x, copy=True))
        return x
    if mode == 'replay':
        v = STATE['getval_tape'][STATE['getval_pos']]
        STATE['getval_pos'] += 1
        return v
    return x


def _eval(fun, x, *args):
    if STATE['before_eval'] is not None:
        STATE['before_eval']()
    STATE['getval_pos'] = 0
    return fun(x, *args)


def _fd_grad(scalar_fun, x, h_rel=1e-3):
    """Richardson-extrapolated central differences (O(h^4)) of a scalar function."""
    x = np.asarray(x, dtype=np.float64)
    g = np.zeros_like(x)
    for i in range(x.size):
        h = h_rel * max(1.0, abs(x[i]))
        d = []
        for hh in (h, h / 2):
            xp = x.copy(); xp[i] += hh
            xm = x.copy(); xm[i] -= hh
            d.append((scalar_fun(xp) - scalar_fun(xm)) / (2 * hh))
        g[i] = (4 * d[1] - d[0]) / 3
    return g


def value_and_grad(fun):
    def vg(x, *args):
        x = np.asarray(x, dtype=np.float64)
        STATE['getval_mode'] = 'record'
        STATE['getval_tape'] = []
        val = _eval(fun, x, *args)
        STATE['getval_mode'] = 'replay'
        try:
            g = _fd_grad(lambda xx: float(_eval(fun, xx, *args)), x)
        finally:
            STATE['getval_mode'] = 'off'
        # leave every piece of mutable state as ONE evaluation would
        _eval_state_after_single(fun, x, *args)
        return val, g
    return vg


def _eval_state_after_single(fun, x, *args):
    STATE['getval_mode'] = 'off'
    _eval(fun, x, *args)


def vector_jacobian_product(fun):
    def vjp(x, *rest):
        *args, v = rest
        x = np.asarray(x, dtype=np.float64)
        v = np.asarray(v, dtype=np.float64)
        return _fd_grad(lambda xx: float(np.sum(v * fun(xx, *args))), x)
    return vjp


def elementwise_grad(fun):
    return lambda x: STATE['model'].grad(x).reshape(np.shape(x))


def grad(fun):
    return lambda x: STATE['model'].grad(x)[0]


def hessian(fun):
    return lambda x: STATE['model'].hessian(x)[np.newaxis]


def make_hvp(fun):
    def at(x):
        return (lambda v: STATE['model'].hvp(x, v)[0],)
    return at


def install(reference_root='/root/reference'):
    ag = types.ModuleType('autograd')
    ag.numpy = np
    ag.value_and_grad = value_and_grad
    ag.vector_jacobian_product = vector_jacobian_product
    ag.make_hvp = make_hvp
    ag.elementwise_grad = elementwise_grad
    ag.grad = grad
    ag.hessian = hessian
    core = types.ModuleType('autograd.core'); core.getval = _getval
    ext = types.ModuleType('autograd.extend')
    ext.primitive = lambda f: f
    ext.defvjp = lambda *a, **k: None
    agsp = types.ModuleType('autograd.scipy')
    agsp.stats, agsp.special, agsp.linalg = scipy.stats, scipy.special, scipy.linalg
    ag.scipy, ag.core, ag.extend = agsp, core, ext
    mods = {
        'autograd': ag, 'autograd.numpy': np, 'autograd.numpy.random': np.random,
        'autograd.numpy.linalg': np.linalg,
        'autograd.core': core, 'autograd.extend': ext, 'autograd.scipy': agsp,
        'autograd.scipy.stats': scipy.stats, 'autograd.scipy.stats.norm': scipy.stats.norm,
        'autograd.scipy.stats.t': scipy.stats.t, 'autograd.scipy.special': scipy.special,
        'autograd.scipy.linalg': scipy.linalg,
        'pystan': types.ModuleType('pystan'),
        'paragami': _paragami_module(),
    }
    sys.modules.update(mods)
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)


# ------------------------------------------------------------------ paragami
class _VecPattern:
    def __init__(self, length=None, shape=None):
        self.shape = (length,) if shape is None else tuple(shape)

    def flat_length(self, free=True):
        return int(np.prod(self.shape))

    def flatten(self, val, free=True):
        return np.asarray(val, dtype=np.float64).reshape(-1)

    def fold(self, vec, free=True):
        return np.asarray(vec).reshape(self.shape)


class _PSDPattern:
    def __init__(self, size):
        self.size = size

    def flat_length(self, free=True):
        return self.size * (self.size + 1) // 2

    def flatten(self, val, free=True):
        L = np.linalg.cholesky(np.asarray(val, dtype=np.float64))
        L[np.diag_indices(self.size)] = np.log(np.diag(L))
        return L[np.tril_indices(self.size)]

    def fold(self, vec, free=True):
        L = np.zeros((self.size, self.size))
        L[np.tril_indices(self.size)] = vec
        L[np.diag_indices(self.size)] = np.exp(np.diag(L))
        return L @ L.T


class _PatternDict:
    def __init__(self, free_default=True):
        self._p = {}

    def __setitem__(self, k, v):
        self._p[k] = v

    def flat_length(self, free=True):
        return sum(p.flat_length(free) for p in self._p.values())

    def flatten(self, d, free=True):
        return np.concatenate([p.flatten(d[k], free) for k, p in self._p.items()])

    def fold(self, vec, free=True):
        out, o = {}, 0
        for k, p in self._p.items():
            n = p.flat_length(free)
            out[k] = p.fold(vec[o:o + n], free)
            o += n
        return out


def _flatten_function_input(fun, patterns, free=True, argnums=0):
    def wrapped(*args):
        args = list(args)
        args[argnums] = patterns.fold(args[argnums], free)
        return fun(*args)
    return wrapped


def _paragami_module():
    m = types.ModuleType('paragami')
    m.PatternDict = _PatternDict
    m.NumericVectorPattern = lambda length: _VecPattern(length=length)
    m.NumericArrayPattern = lambda shape: _VecPattern(shape=shape)
    m.PSDSymmetricMatrixPattern = lambda size: _PSDPattern(size)
    m.FlattenFunctionInput = _flatten_function_input
    return m
