"""GPU: the mean-field evaluation as ONE launch (mf_one_kernel: prep + streaming pass + finalize behind tickets,
north_star "... fused into one launch") against the three-launch chain it replaces (VB_MF_ONE=0) AND against the oracle
(`test_one_launch_against_the_oracle`: `oracle.objectives.exclusive_kl` on the noise read back from the device, value
1e-12, gradient 1e-11 relative to max |grad| -- the tolerances of tests/test_gpu_meanfield.py).

The column sums are formed in the same order by both, so every gradient entry except the funnel's coupling column is
bit-identical; the value and the coupling column add the per-row-block scalar partials in a different grouping (row
blocks instead of 256-row prep blocks): equal to rounding."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import viabel_amd as vb
    from viabel_amd import _lib
    return vb, _lib.default_engine(), _lib


def _both(call):
    os.environ['VB_MF_ONE'] = '0'
    try:
        chain = call()
    finally:
        os.environ['VB_MF_ONE'] = '1'
    one = call()
    return one, chain


SHAPES = [(10, 100), (1024, 4096), (300, 257), (1000, 4099), (64, 16384), (130, 8), (2049, 515)]


@pytest.mark.parametrize('d,n', SHAPES)
@pytest.mark.parametrize('target', ['gauss', 'funnel'])
@pytest.mark.parametrize('family', ['gaussian', 'student'])
@pytest.mark.parametrize('path_deriv', [False, True])
def test_one_launch_equals_launch_chain_noise_in_memory(env, d, n, target, family, path_deriv):
    vb, eng, _lib = env
    rng = np.random.RandomState(d + n)
    model = vb.GaussianModel(rng.randn(d), np.exp(0.2 * rng.randn(d))) if target == 'gauss' else vb.FunnelModel(d, scale_index=d // 3)
    eng.set_model(model.device_spec())
    fam = _lib.FAMILY_MF_GAUSSIAN if family == 'gaussian' else _lib.FAMILY_MF_STUDENT_T
    df = 0.0 if family == 'gaussian' else 7.0
    theta = np.concatenate([0.3 * rng.randn(d), -1.0 + 0.2 * rng.randn(d)])
    eng.noise_generate(3, n, d, seed=5, stream=1, kind=_lib.NOISE_NORMAL if family == 'gaussian' else _lib.NOISE_STUDENT_T, df=df)
    flags = _lib.FLAG_PATH_DERIV if path_deriv else 0
    (v1, g1), (v0, g0) = _both(lambda: eng.elbo_grad_meanfield(3, n, d, theta, fam, df=df, flags=flags))
    assert abs(v1 - v0) <= 1e-13 * abs(v0)
    k = d // 3 if target == 'funnel' else -1
    keep = np.ones(2 * d, dtype=bool)
    if k >= 0:
        keep[[k, d + k]] = False
    np.testing.assert_array_equal(g1[keep], g0[keep])
    np.testing.assert_allclose(g1, g0, rtol=1e-12, atol=1e-13 * np.max(np.abs(g0)))


@pytest.mark.parametrize('d,n', SHAPES)
@pytest.mark.parametrize('target', ['gauss', 'funnel'])
@pytest.mark.parametrize('family', ['gaussian', 'student'])
@pytest.mark.parametrize('path_deriv', [False, True])
def test_one_launch_against_the_oracle(env, d, n, target, family, path_deriv):
    """K1: the one-launch kernel's result against the CPU restatement of objectives.py:150-168 on the
    very noise the kernel read -- not against the launch chain."""
    from oracle import families as ofam, models as omod, objectives as oobj
    vb, eng, _lib = env
    rng = np.random.RandomState(7 * d + n)
    if target == 'gauss':
        mean, sd = rng.randn(d), np.exp(0.2 * rng.randn(d))
        model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    else:
        model, omodel = vb.FunnelModel(d, scale_index=d // 3), omod.Funnel(d, d // 3)
    eng.set_model(model.device_spec())
    student = family == 'student'
    fam, df = (_lib.FAMILY_MF_STUDENT_T, 7.0) if student else (_lib.FAMILY_MF_GAUSSIAN, 0.0)
    ofamily = ofam.MFStudentT(d, df) if student else ofam.MFGaussian(d)
    theta = np.concatenate([0.3 * rng.randn(d), -1.0 + 0.2 * rng.randn(d)])
    eng.noise_generate(3, n, d, seed=11, stream=2, kind=_lib.NOISE_STUDENT_T if student else _lib.NOISE_NORMAL, df=df)
    noise = eng.noise_get_host(3, n, d)
    os.environ['VB_MF_ONE'] = '1'
    value, grad = eng.elbo_grad_meanfield(3, n, d, theta, fam, df=df, flags=_lib.FLAG_PATH_DERIV if path_deriv else 0)
    ov, og = oobj.exclusive_kl(ofamily, omodel, theta, noise, path_deriv)
    assert abs(value - ov) <= 1e-12 * abs(ov), (value, ov)
    assert np.max(np.abs(grad - og)) <= 1e-11 * np.max(np.abs(og)), np.max(np.abs(grad - og)) / np.max(np.abs(og))


@pytest.mark.parametrize('d,n', [(1024, 4096), (10, 100), (257, 1000)])
@pytest.mark.parametrize('target', ['gauss', 'funnel'])
def test_one_launch_against_the_oracle_noise_in_registers(env, d, n, target):
    """The same with the noise generated in registers (the device loop's kernel): the oracle runs on the Philox matrix of the
    same (seed, stream) written to a slot by vb_noise_generate -- the counter-based values are the ones the registers held."""
    from oracle import families as ofam, models as omod, objectives as oobj
    vb, eng, _lib = env
    rng = np.random.RandomState(d)
    if target == 'gauss':
        mean, sd = rng.randn(d), np.exp(0.2 * rng.randn(d))
        model, omodel = vb.GaussianModel(mean, sd), omod.GaussDiag(mean, sd)
    else:
        model, omodel = vb.FunnelModel(d), omod.Funnel(d)
    eng.set_model(model.device_spec())
    theta = np.concatenate([0.3 * rng.randn(d), -1.0 + 0.2 * rng.randn(d)])
    os.environ['VB_MF_ONE'] = '1'
    value, grad = eng.elbo_grad_meanfield_philox(0, n, d, theta, _lib.FAMILY_MF_GAUSSIAN, 9, 77)
    eng.noise_generate(4, n, d, seed=9, stream=77)
    ov, og = oobj.exclusive_kl(ofam.MFGaussian(d), omodel, theta, eng.noise_get_host(4, n, d))
    assert abs(value - ov) <= 1e-12 * abs(ov), (value, ov)
    assert np.max(np.abs(grad - og)) <= 1e-11 * np.max(np.abs(og))


@pytest.mark.parametrize('d,n', [(1024, 4096), (10, 100), (257, 1000)])
@pytest.mark.parametrize('target', ['gauss', 'funnel'])
def test_one_launch_equals_launch_chain_noise_in_registers(env, d, n, target):
    vb, eng, _lib = env
    rng = np.random.RandomState(d)
    model = vb.GaussianModel(rng.randn(d), np.exp(0.2 * rng.randn(d))) if target == 'gauss' else vb.FunnelModel(d)
    eng.set_model(model.device_spec())
    theta = np.concatenate([0.3 * rng.randn(d), -1.0 + 0.2 * rng.randn(d)])
    (v1, g1), (v0, g0) = _both(lambda: eng.elbo_grad_meanfield_philox(0, n, d, theta, _lib.FAMILY_MF_GAUSSIAN, 9, 77))
    assert abs(v1 - v0) <= 1e-13 * abs(v0)
    np.testing.assert_allclose(g1, g0, rtol=1e-12, atol=1e-13 * np.max(np.abs(g0)))


def test_one_launch_repeated_calls_and_changing_shapes(env):
    """The tickets and publication flags live on from launch to launch: many evaluations back to back, shapes and targets
    changing in between, every one against the launch chain."""
    vb, eng, _lib = env
    rng = np.random.RandomState(0)
    for it in range(40):
        d, n = [(1024, 4096), (10, 100), (513, 300), (64, 2048)][it % 4]
        model = vb.FunnelModel(d) if it % 3 else vb.GaussianModel(np.zeros(d), np.ones(d))
        eng.set_model(model.device_spec())
        theta = np.concatenate([0.1 * rng.randn(d), -1.0 + 0.1 * rng.randn(d)])
        eng.noise_generate(2, n, d, seed=it, stream=0)
        (v1, g1), (v0, g0) = _both(lambda: eng.elbo_grad_meanfield(2, n, d, theta, _lib.FAMILY_MF_GAUSSIAN))
        assert abs(v1 - v0) <= 1e-13 * abs(v0), it
        np.testing.assert_allclose(g1, g0, rtol=1e-12, atol=1e-13 * np.max(np.abs(g0)), err_msg=str(it))


@pytest.mark.parametrize('target', ['gauss', 'funnel'])
def test_device_fit_loop_one_launch_per_iteration(env, target):
    """vb_fit: the optimiser step applied by the tail of the same launch; the trajectory against the launch chain's."""
    vb, eng, _lib = env
    from viabel_amd.optimization import RMSProp
    d, n = 256, 1024
    model = vb.FunnelModel(d) if target == 'funnel' else vb.GaussianModel(np.linspace(-1, 1, d), np.ones(d))
    theta = np.concatenate([np.zeros(d), -np.ones(d)])

    def run():
        obj = vb.ExclusiveKL(vb.MFGaussian(d, seed=3, rng='philox'), model, n)
        opt = RMSProp(0.01)
        th, values = obj.device_fit(60, theta, opt._device_kind, opt._device_hyper())[:2]
        return th, np.asarray(values)
    (th1, v1), (th0, v0) = _both(run)
    np.testing.assert_allclose(v1, v0, rtol=1e-11)
    np.testing.assert_allclose(th1, th0, rtol=1e-9, atol=1e-11)
