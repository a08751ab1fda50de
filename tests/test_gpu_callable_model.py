"""GPU: host-callable models (vb_set_model_callback / CallableModel) -- the reference's front door
``Model(log_density)`` (viabel/models.py:17-39, convenience.py:69-75) and the StanModel contract
``log_prob`` + ``grad_log_prob`` (models.py:80-104).

The callable is evaluated on the host between the device's sampling and reduction kernels; a callable that
restates a built-in device target must reproduce that target's objective value and gradient (tolerances: the
two routes differ only in the rounding of f and grad f: 1e-12 / 1e-11; 1e-6 when the gradient is numerical)."""
import numpy as np
import pytest

import _golden as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import viabel_amd
    from viabel_amd import _lib
    _lib.default_engine()
    return viabel_amd


def _gauss(D, seed=0):
    rng = np.random.RandomState(seed)
    mean, sd = rng.randn(D), np.exp(0.3 * rng.randn(D))

    def f(z):
        r = (z - mean) / sd
        return np.sum(-0.5 * r * r - np.log(sd) - 0.5 * np.log(2 * np.pi), axis=1)

    def g(z):
        return -(z - mean) / (sd * sd)
    return mean, sd, f, g


def _families(vb, D, rng_kind):
    return {
        'mf_gaussian': lambda: vb.MFGaussian(D, seed=3, rng=rng_kind),
        'mf_student_t': lambda: vb.MFStudentT(D, 9, seed=3, rng=rng_kind),
        'fullrank': lambda: vb.FullRankGaussian(D, seed=3, rng=rng_kind),
        'multivariate_t': lambda: vb.MultivariateT(D, 30, seed=3, rng=rng_kind),
        'lr_gaussian': lambda: vb.LRGaussian(D, seed=3, k=3, rng=rng_kind),
    }


def _theta(approx, seed=5):
    rng = np.random.RandomState(seed)
    theta = approx.init_param().copy()
    theta[:approx.dim] = 0.2 * rng.randn(approx.dim)
    theta[approx.dim:] = theta[approx.dim:] * 0.2 + 0.05 * rng.randn(theta.size - approx.dim)
    return theta


@pytest.mark.parametrize('family', ['mf_gaussian', 'mf_student_t', 'fullrank', 'multivariate_t', 'lr_gaussian'])
@pytest.mark.parametrize('use_path_deriv', [False, True])
def test_exclusive_kl_callable_equals_device_model(vb, family, use_path_deriv):
    D, N = 11, 300
    mean, sd, f, g = _gauss(D)
    out = []
    for model in (vb.GaussianModel(mean, sd), vb.CallableModel(D, f, g),
                  vb.CallableModel(D, value_and_grad=lambda z: (f(z), g(z)))):
        approx = _families(vb, D, 'numpy')[family]()
        obj = vb.ExclusiveKL(approx, model, N, use_path_deriv=use_path_deriv)
        out.append(obj(_theta(approx)))
    for value, grad in out[1:]:
        assert G.rel_err(value, out[0][0]) < 1e-12
        assert G.rel_err(grad, out[0][1]) < 1e-11


def test_numerical_gradient_and_reference_front_door(vb):
    """``ExclusiveKL(approx, Model(log_density), N)`` as in the reference (models.py:17-39): no gradient given, the
    engine differentiates the callable by central differences (and says so once)."""
    D, N = 6, 200
    mean, sd, f, g = _gauss(D, seed=2)
    approx = vb.MFGaussian(D, seed=4)
    theta = _theta(approx)
    ref = vb.ExclusiveKL(vb.MFGaussian(D, seed=4), vb.GaussianModel(mean, sd), N)(theta)
    with pytest.warns(UserWarning, match='central differences'):
        value, grad = vb.ExclusiveKL(approx, vb.Model(f), N)(theta)
    assert G.rel_err(value, ref[0]) < 1e-12
    assert G.rel_err(grad, ref[1]) < 1e-6


@pytest.mark.parametrize('family', ['mf_gaussian', 'fullrank', 'multivariate_t', 'lr_gaussian'])
def test_alpha_and_dis_callable_equals_device_model(vb, family):
    D, N = 9, 400
    mean, sd, f, g = _gauss(D, seed=1)
    prior = np.concatenate([np.zeros(D), 0.4 * np.ones(D)])
    res = {}
    for name, model in (('device', vb.GaussianModel(mean, sd)), ('callable', vb.CallableModel(D, f, g))):
        approx = _families(vb, D, 'numpy')[family]()
        theta = _theta(approx)
        np.random.seed(12)
        alpha = vb.AlphaDivergence(approx, model, N, 2.0)(theta)
        approx = _families(vb, D, 'numpy')[family]()
        dis = vb.DISInclusiveKL(approx, model, N, ess_target=60, temper_prior=vb.MFGaussian(D),
                                temper_prior_params=prior, use_resampling=False)
        res[name] = (alpha, dis(theta), dis._eps)
    for k in range(2):
        assert G.rel_err(res['callable'][k][0], res['device'][k][0]) < 1e-11
        assert G.rel_err(res['callable'][k][1], res['device'][k][1]) < 1e-10
    assert G.rel_err(res['callable'][2], res['device'][2]) < 1e-10


def test_callable_model_call_grad_and_errors(vb):
    D = 5
    mean, sd, f, g = _gauss(D, seed=3)
    model = vb.CallableModel(D, f, g)
    x = np.random.RandomState(0).randn(17, D)
    np.testing.assert_allclose(model(x), f(x), rtol=0, atol=1e-13)
    np.testing.assert_allclose(model(x[0]), f(x[:1]), rtol=0, atol=1e-13)
    np.testing.assert_allclose(model.grad(x), g(x), rtol=0, atol=1e-13)
    assert model.check_gradient(x) < 1e-6

    class Boom(RuntimeError):
        pass

    def bad(z):
        raise Boom('inside the model')
    with pytest.raises(Boom):                                   # the callable's own exception comes back
        vb.ExclusiveKL(vb.MFGaussian(D), vb.CallableModel(D, bad, g), 10)(np.zeros(2 * D))
    with pytest.raises(ValueError):                             # wrong shape from the callable
        vb.ExclusiveKL(vb.MFGaussian(D), vb.CallableModel(D, lambda z: np.zeros(3), g), 10)(np.zeros(2 * D))
    # the engine still works afterwards
    v, gr = vb.ExclusiveKL(vb.MFGaussian(D), model, 10)(np.zeros(2 * D))
    assert np.isfinite(v) and np.all(np.isfinite(gr))


def test_bbvi_with_a_python_callable(vb):
    """viabel/tests/test_convenience.py:10-37: bbvi(dim, log_density=callable) recovers a Gaussian target."""
    D = 3
    mean, sd, f, g = _gauss(D, seed=7)
    np.random.seed(851)
    res = vb.bbvi(D, log_density=f, grad_log_density=g, num_mc_samples=50, n_iters=1500, learning_rate=0.05,
                  adaptive=False, fixed_lr=True)
    opt = res['opt_param']
    np.testing.assert_allclose(opt[:D], mean, atol=0.1)
    np.testing.assert_allclose(np.exp(opt[D:]), sd, rtol=0.15)
    # device-resident fit with rng='philox': the callable sits inside vb_fit's loop
    approx = vb.MFGaussian(D, seed=2, rng='philox')
    res = vb.bbvi(D, log_density=vb.CallableModel(D, f, g), approx=approx, num_mc_samples=50, n_iters=1500,
                  learning_rate=0.05, adaptive=False, fixed_lr=True)
    np.testing.assert_allclose(res['opt_param'][:D], mean, atol=0.1)
