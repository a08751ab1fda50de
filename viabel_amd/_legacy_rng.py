"""``numpy.random.RandomState`` as the reference's families use it (``approximations.py:203``, ``:273-274``,
``:342-345``), drawn by the library's own C++ restatement of the legacy generator (``vb_legacy_rng.cpp``,
``include/viabel_hip.h``): MT19937, polar-method normals, Marsaglia-Tsang gammas.  Same values and same generator
state as numpy, bit for bit (``tests/test_legacy_rng_cpu.py``); the large normal matrices of the parity mode are
produced on all host threads instead of one."""
import ctypes

import numpy as np

from . import _lib

_u32_p = ctypes.POINTER(ctypes.c_uint32)
_dbl_p = ctypes.POINTER(ctypes.c_double)


def _size(shape):
    if shape is None:
        return (), 1
    if isinstance(shape, (int, np.integer)):
        shape = (int(shape),)
    shape = tuple(int(s) for s in shape)
    if any(s < 0 for s in shape):
        raise ValueError('negative dimensions are not allowed')
    n = 1
    for s in shape:      # (np.prod costs microseconds per call: this runs once per objective call at small shapes)
        n *= s
    return shape, n


class LegacyRandomState:
    """The subset of ``numpy.random.RandomState`` the hot path draws from: ``randn``, ``standard_normal``,
    ``standard_t``, ``chisquare``, ``random_sample``, ``get_state`` / ``set_state``."""

    def __init__(self, seed=None):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        plain = isinstance(seed, (int, np.integer)) and not isinstance(seed, bool) and 0 <= int(seed) <= 0xFFFFFFFF
        if self._lib.vb_legacy_rng_create(int(seed) if plain else 0, ctypes.byref(self._h)) != _lib.VB_OK:
            raise MemoryError('vb_legacy_rng_create failed')
        if not plain:       # None (OS entropy), arrays, out-of-range values: numpy's own seeding rules and errors
            self.set_state(np.random.RandomState(seed).get_state())

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            self._lib.vb_legacy_rng_destroy(h)

    # the handle is a pointer into the library: copies and pickles carry the state in numpy's format instead
    def __getstate__(self):
        return {'state': self.get_state()}

    def __setstate__(self, d):
        self.__init__(0)
        self.set_state(d['state'])

    def __deepcopy__(self, memo):
        other = LegacyRandomState(0)
        other.set_state(self.get_state())
        return other

    __copy__ = lambda self: self.__deepcopy__({})      # noqa: E731

    def _check(self, rc):
        if rc != _lib.VB_OK:
            raise ValueError('legacy generator: invalid argument')

    # ---- draws ------------------------------------------------------------------------------------------------
    def standard_normal(self, size=None):
        shape, n = _size(size)
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.vb_legacy_rng_randn(self._h, out.ctypes.data_as(_dbl_p), n, 0))
        return float(out[0]) if size is None else out.reshape(shape)

    def randn(self, *dims):
        return self.standard_normal(dims if dims else None)

    def standard_t(self, df, size=None):
        if not np.isscalar(df):
            raise NotImplementedError('array-valued df')
        if not df > 0:
            raise ValueError('df <= 0')
        shape, n = _size(size)
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.vb_legacy_rng_standard_t(self._h, float(df), out.ctypes.data_as(_dbl_p), n))
        return float(out[0]) if size is None else out.reshape(shape)

    def chisquare(self, df, size=None):
        if not np.isscalar(df):
            raise NotImplementedError('array-valued df')
        if not df > 0:
            raise ValueError('df <= 0')
        shape, n = _size(size)
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.vb_legacy_rng_chisquare(self._h, float(df), out.ctypes.data_as(_dbl_p), n))
        return float(out[0]) if size is None else out.reshape(shape)

    def random_sample(self, size=None):
        shape, n = _size(size)
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.vb_legacy_rng_random_sample(self._h, out.ctypes.data_as(_dbl_p), n))
        return float(out[0]) if size is None else out.reshape(shape)

    # ---- state, in numpy's own format ----------------------------------------------------------------------------
    def get_state(self):
        key = np.empty(624, dtype=np.uint32)
        pos, has, g = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_double(0.0)
        self._check(self._lib.vb_legacy_rng_get_state(self._h, key.ctypes.data_as(_u32_p), ctypes.byref(pos),
                                                      ctypes.byref(has), ctypes.byref(g)))
        return ('MT19937', key, pos.value, has.value, g.value)

    def set_state(self, state):
        name, key, pos = state[0], np.ascontiguousarray(state[1], dtype=np.uint32), int(state[2])
        if name != 'MT19937' or key.shape != (624,):
            raise ValueError('state must be a RandomState.get_state() tuple')
        has, g = (int(state[3]), float(state[4])) if len(state) > 3 else (0, 0.0)
        self._check(self._lib.vb_legacy_rng_set_state(self._h, key.ctypes.data_as(_u32_p), pos, has, g))
