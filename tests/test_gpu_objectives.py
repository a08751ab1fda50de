"""GPU parity of AlphaDivergence and DISInclusiveKL against the reference-derived golden vectors
and the oracle, through the product's Python classes (which call the C ABI).

Tolerances: value 1e-12 relative, gradient 1e-11 relative to max|grad|; 2e-7 / 2e-6 against the
reference's finite-difference gradients (as in tests/test_oracle_golden.py).
"""
import numpy as np
import pytest

import _golden as G
from oracle import families as ofam
from oracle import models as omod
from oracle import objectives as oobj

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import viabel_amd
    from viabel_amd import _lib
    _lib.default_engine()
    return viabel_amd


def product_family(vb, fx, seed=1):
    kind, D = str(fx['family_kind']), int(fx['dim'])
    if kind == 'mf_gaussian':
        return vb.MFGaussian(D, seed=seed)
    if kind == 'mf_student_t':
        return vb.MFStudentT(D, float(fx['df']), seed=seed)
    if kind == 'multivariate_t':
        return vb.MultivariateT(D, float(fx['df']), seed=seed)
    raise ValueError(kind)


def product_model(vb, fx):
    if str(fx['model_kind']) == 'gauss_diag':
        return vb.GaussianModel(fx['model_mean'], fx['model_stdev'])
    return vb.FunnelModel(int(fx['dim']), int(fx['model_scale_index']),
                          float(fx['model_log_sigma_stdev']))


@pytest.mark.parametrize('path', G.fixtures('alpha_'), ids=lambda p: p.split('/')[-1][:-4])
def test_alpha_golden(vb, path):
    fx = G.load(path)
    obj = vb.AlphaDivergence(product_family(vb, fx), product_model(vb, fx), int(fx['n']), float(fx['alpha']))
    np.random.seed(int(fx['np_seed']))          # the objective draws its noise seed from the global RNG
    value, grad = obj(fx['theta'])
    assert G.rel_err(value, fx['value']) < 1e-12
    assert G.rel_err(grad, fx['grad']) < 1e-11
    assert G.rel_err(grad, fx['grad_fd']) < 2e-7


@pytest.mark.parametrize('D,N', [(1024, 4096), (77, 333), (300, 1000)])
@pytest.mark.parametrize('family', ['gauss', 't'])
def test_alpha_against_oracle(vb, D, N, family):
    rng = np.random.RandomState(D + N)
    theta = np.concatenate([0.3 * rng.randn(D), -1.0 + 0.2 * rng.randn(D)])
    for model, omodel in ((vb.FunnelModel(D, 5), omod.Funnel(D, 5)),
                          (vb.GaussianModel(np.ones(D), 2 * np.ones(D)), omod.GaussDiag(np.ones(D), 2 * np.ones(D)))):
        if family == 'gauss':
            approx, ofamily = vb.MFGaussian(D), ofam.MFGaussian(D)
        else:
            approx, ofamily = vb.MFStudentT(D, 12), ofam.MFStudentT(D, 12)
        for alpha in (2.0, 0.5):
            np.random.seed(7)
            value, grad = vb.AlphaDivergence(approx, model, N, alpha)(theta)
            np.random.seed(7)
            seed = np.random.randint(2 ** 32)
            noise = ofamily.draw_noise(np.random.RandomState(seed), N)
            ov, og = oobj.alpha_divergence(ofamily, omodel, theta, noise, alpha)
            assert G.rel_err(value, ov) < 1e-12
            assert G.rel_err(grad, og) < 1e-11
