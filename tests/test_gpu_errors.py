"""GPU: error behaviour at the C-ABI boundary -- status codes surface as the exceptions the reference raises
(ValueError for bad arguments / numerics, NotImplementedError for unsupported combinations, EngineError
otherwise), the context stays usable afterwards, and nothing falls back to a CPU path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import viabel_amd as vb
    from viabel_amd import _lib
    return vb, _lib, _lib.default_engine()


def test_state_and_shape_errors(env):
    vb, _lib, eng = env
    fresh = _lib.Engine(eng.device)
    theta = np.zeros(8)
    with pytest.raises(_lib.EngineError):                       # no model bound
        fresh.elbo_grad_meanfield(0, 4, 4, theta, _lib.FAMILY_MF_GAUSSIAN)
    fresh.set_model(vb.GaussianModel(np.zeros(4), np.ones(4)).device_spec())
    with pytest.raises(_lib.EngineError):                       # empty noise slot
        fresh.elbo_grad_meanfield(0, 4, 4, theta, _lib.FAMILY_MF_GAUSSIAN)
    fresh.noise_set_host(0, np.zeros((4, 4)))
    with pytest.raises(ValueError):                             # more rows than the slot holds
        fresh.elbo_grad_meanfield(0, 5, 4, theta, _lib.FAMILY_MF_GAUSSIAN)
    with pytest.raises(ValueError):                             # slot out of range
        fresh.noise_set_host(_lib.MAX_SLOTS, np.zeros((4, 4)))
    with pytest.raises(ValueError):                             # dimension mismatch with the bound model
        fresh.noise_set_host(1, np.zeros((4, 3)))
        fresh.elbo_grad_meanfield(1, 4, 3, np.zeros(6), _lib.FAMILY_MF_GAUSSIAN)
    with pytest.raises(ValueError):                             # Student-t needs df > 2
        fresh.elbo_grad_meanfield(0, 4, 4, theta, _lib.FAMILY_MF_STUDENT_T, df=1.5)
    with pytest.raises(ValueError):
        fresh.elbo_grad_meanfield(0, 4, 4, theta, _lib.FAMILY_MF_GAUSSIAN, cv_mode=9)
    with pytest.raises(_lib.EngineError):                       # no pending result in that slot
        fresh.result_get(5, 8)
    # the context survives all of the above
    v, g = fresh.elbo_grad_meanfield(0, 4, 4, theta, _lib.FAMILY_MF_GAUSSIAN)
    assert np.isfinite(v) and g.shape == (8,)
    fresh.close()


def test_objective_level_errors(env):
    vb, _lib, eng = env
    model = vb.GaussianModel(np.zeros(3), np.ones(3))
    with pytest.raises(ValueError):
        vb.MFStudentT(3, 2)                                     # approximations.py:258-259
    with pytest.raises(ValueError):
        vb.MultivariateT(3, 1.0)
    with pytest.raises(TypeError):                              # neither a model nor a callable
        vb.ExclusiveKL(vb.MFGaussian(3), 3.0, 10)
    with pytest.raises(ValueError):                             # wrong parameter length
        vb.ExclusiveKL(vb.MFGaussian(3), model, 10)(np.zeros(5))
    class Unset(vb.VariationalObjective):                       # a subclass that never builds its closure
        def _update_objective_and_grad(self):
            pass
    with pytest.raises(RuntimeError):                           # objectives.py:42-43
        Unset(vb.MFGaussian(3), model)(np.zeros(6))
    with pytest.raises(NotImplementedError):                    # control variates need the [mean | log-scale] layout
        vb.ExclusiveKL(vb.FullRankGaussian(3), model, 10, hessian_approx_method='full')


def test_nonfinite_parameters_do_not_crash(env):
    vb, _lib, eng = env
    obj = vb.ExclusiveKL(vb.MFGaussian(4), vb.FunnelModel(4), 32)
    theta = np.array([0, 0, 0, 0, 800.0, 0, 0, 0])             # sigma overflows: the result is non-finite, no fault
    v, g = obj(theta)
    assert not np.isfinite(v) or not np.all(np.isfinite(g))
    v, g = obj(np.zeros(8))                                     # and the engine still works afterwards
    assert np.isfinite(v) and np.all(np.isfinite(g))
