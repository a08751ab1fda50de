"""GPU: the Newton-Schulz symmetric square root / Sylvester solve (vb_sym_sqrt) against LAPACK.

`scipy.linalg.sqrtm(Sigma)` is what MultivariateT.sample takes (approximations.py:348); the eigen-decomposition
gives the same matrix and, for the derivative, the closed form X~_ij = E~_ij / (r_i + r_j) in the eigenbasis."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _spd(d, cond, seed):
    rng = np.random.RandomState(seed)
    Q, _ = np.linalg.qr(rng.randn(d, d))
    ev = np.exp(np.linspace(0.0, np.log(cond), d)) * 0.37
    return (Q * ev) @ Q.T


def _reference(a, e=None):
    w, U = np.linalg.eigh(a)
    r = np.sqrt(w)
    root = (U * r) @ U.T
    if e is None:
        return root, None
    x = U @ ((U.T @ e @ U) / (r[:, None] + r[None, :])) @ U.T
    return root, x


@pytest.mark.parametrize('d,cond', [(1, 1.0), (5, 10.0), (64, 1e3), (130, 1e4), (256, 1e2), (256, 1e6), (300, 1e8)])
def test_symmetric_root_matches_eigh(d, cond):
    from viabel_amd import _lib
    eng = _lib.default_engine()
    a = _spd(d, cond, d)
    root, x, info = eng.sym_sqrt(a)
    want, _ = _reference(a)
    assert x is None
    np.testing.assert_array_equal(root, root.T)
    scale = np.linalg.norm(want)
    # forward error and residual grow (mildly) with sqrt(cond)
    assert np.linalg.norm(root - want) / scale < 2e-15 * max(10.0, np.sqrt(cond)) * np.sqrt(d)
    resid_tol = max(5e-14, 3e-16 * np.sqrt(cond * d))
    assert np.linalg.norm(root @ root - a) / np.linalg.norm(a) < resid_tol
    assert info[2] < resid_tol and 0 <= info[0] < 60


@pytest.mark.parametrize('d,cond', [(3, 5.0), (40, 1e2), (128, 1e4), (256, 1e3)])
def test_root_derivative_solves_the_sylvester_equation(d, cond):
    from viabel_amd import _lib
    eng = _lib.default_engine()
    a = _spd(d, cond, 100 + d)
    rng = np.random.RandomState(d)
    e = rng.randn(d, d)
    e = 0.5 * (e + e.T) * 37.0
    root, x, info = eng.sym_sqrt(a, e)
    want_root, want_x = _reference(a, e)
    np.testing.assert_allclose(root, want_root, rtol=0, atol=1e-13 * np.linalg.norm(want_root))
    np.testing.assert_allclose(x, want_x, rtol=0, atol=1e-11 * np.linalg.norm(want_x))
    resid = root @ x + x @ root - e
    assert np.linalg.norm(resid) / np.linalg.norm(e) < 1e-11
    # general (non-symmetric) directions work as well: the equation is linear
    e2 = rng.randn(d, d)
    _, x2, _ = eng.sym_sqrt(a, e2)
    assert np.linalg.norm(want_root @ x2 + x2 @ want_root - e2) / np.linalg.norm(e2) < 1e-11


def test_sym_sqrt_errors():
    from viabel_amd import _lib
    eng = _lib.default_engine()
    with pytest.raises(Exception):
        eng.sym_sqrt(np.zeros((4, 4)))
    root, _, info = eng.sym_sqrt(np.diag([4.0, 9.0, 16.0]))
    np.testing.assert_allclose(root, np.diag([2.0, 3.0, 4.0]), atol=1e-14)
