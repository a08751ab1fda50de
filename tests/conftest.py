import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


# Load the product library before any test module is collected: some CPU-side test modules import torch (gloo
# world-size-2 tests, fp64 autograd cross-checks), whose wheel bundles its own librccl / HIP runtime under the same
# sonames -- whichever is loaded first serves the whole process.  The library under test must run on the ROCm
# libraries it links (/opt/rocm/lib, Makefile rpath), as it does in bench.py and in production, not on torch's copies.
try:
    from viabel_amd import _lib as _vb_lib
    _vb_lib.load()
except Exception:      # not built yet: the tests that need it fail with the loader's own message
    pass
