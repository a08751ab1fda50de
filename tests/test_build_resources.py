"""Build-time resource check (CPU: hipcc cross-compiles gfx950 without a GPU): no fp64 GEMM kernel of the library may
use scratch memory.  A spilling 128 x 128 epilogue once turned an 8 ms evaluation into 24 ms without failing a single
parity test -- results stay right, only the time goes."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'),
                    reason='hipcc not available')
def test_no_gemm_kernel_spills():
    r = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'check_scratch.sh')], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'gemm_f64' not in r.stdout
