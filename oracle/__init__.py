"""CPU oracle: a numpy fp64 restatement of viabel's BBVI hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (``viabel_amd/``)
may import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker / the timed CPU
baseline -- never as the thing measured or shipped.

Every function cites the reference file:line (relative to the upstream
``viabel`` tree) whose arithmetic it restates.  The reference differentiates
with ``autograd``; the oracle writes the same derivatives in closed form.

Pinning (see ``tests/golden/make_golden.py``): the oracle is checked in the
build container against the reference's own Python code imported from the
read-only reference tree (forward values of ``sample`` / ``log_density`` /
``entropy`` / the objective closures, and the literal ``RGE`` control-variate
code driven with analytic model derivatives), against central finite
differences of the reference's own objective closures, and against
``torch.autograd`` in fp64.  The resulting vectors are committed under
``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py`` on every
run.  The reference's own test-suite holds no golden vectors for this path
(only statistical convergence tests), so that is the strongest pin available.
"""
from . import families, models, objectives  # noqa: F401
