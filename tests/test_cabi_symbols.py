"""CPU-side checks of the C-ABI boundary: the library loads and exports every declared symbol."""
import os
import re

import pytest

from viabel_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'viabel_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(vb_[a-z_0-9]+)\s*\(', text)))


def test_header_symbols_exported(lib):
    names = _declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), 'libviabel_hip.so does not export %s' % name


def test_binding_covers_header(lib):
    assert sorted(_lib.SIGNATURES) == _declared_symbols()


def test_version_and_error_text(lib):
    assert lib.vb_version().decode().startswith('viabel_hip')
    assert isinstance(lib.vb_last_error(None), bytes)


def test_no_gpu_fails_loudly():
    """Without a GPU the engine must raise, never fall back to a CPU path."""
    import ctypes
    n = ctypes.c_int(0)
    lib = _lib.load()
    rc = lib.vb_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(_lib.EngineError):
        _lib.Engine(0)
    import numpy as np
    import viabel_amd as vb
    _lib.set_default_engine(None)
    obj = vb.ExclusiveKL(vb.MFGaussian(4), vb.GaussianModel(np.zeros(4), np.ones(4)), 8)
    with pytest.raises(_lib.EngineError):
        obj(np.zeros(8))
