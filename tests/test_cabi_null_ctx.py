"""CPU: every entry point of the C ABI validates its context before using it (tests/cabi_null_probe.py in a child
process, so a crash shows up as a failed test instead of killing the run)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def test_entry_points_reject_null_context():
    res = subprocess.run([sys.executable, os.path.join(HERE, 'cabi_null_probe.py')], capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0, (res.returncode, res.stdout[-2000:], res.stderr[-2000:])
    assert 'rejected a NULL context' in res.stdout
