"""viabel_amd: MI355X-native black-box variational inference gradient engine.

Drop-in for the hot path of jhuggins/viabel -- ``objective(var_param) -> (value, grad)`` --
with the reference's Python API (``VariationalObjective`` / ``ExclusiveKL`` /
``DISInclusiveKL`` / ``AlphaDivergence``, ``MFGaussian`` / ``MFStudentT`` /
``MultivariateT``, ``Model``, ``bbvi``) on top of hand-written gfx950 kernels reached
through the C ABI in ``include/viabel_hip.h``.
"""
from viabel_amd.approximations import *  # noqa: F401,F403
from viabel_amd.models import *  # noqa: F401,F403
from viabel_amd.objectives import *  # noqa: F401,F403
from viabel_amd.optimization import *  # noqa: F401,F403
from viabel_amd.convenience import *  # noqa: F401,F403
from viabel_amd.diagnostics import *  # noqa: F401,F403
from viabel_amd._lib import set_host_blas_threads  # noqa: F401

__version__ = '0.1.0'
