"""GPU: the reference's family self-consistency t-tests (viabel/tests/test_approximations.py:11-113) for the
throughput-mode families (rng='philox': normals / Student-t draws from the device generator).

Same statistics, same seeds (341 / 226 / 56), same size of the test (p > 1e-4); MC_SAMPLES reduced from 1e6 to
2.5e5 per SURVEY 8(c) (a smaller sample makes the test weaker, never spuriously green: a wrong entropy, KL, covariance
or moment shows as p ~ 0 long before 2.5e5 draws)."""
import numpy as np
import pytest
from scipy import stats

pytestmark = pytest.mark.gpu

MC_SAMPLES = 250000
test_size = 0.0001


@pytest.fixture(scope='module')
def vb():
    import viabel_amd
    from viabel_amd import _lib
    _lib.default_engine()
    return viabel_amd


def _test_entropy(approx, var_param, entropy_offset):              # :11-16
    entropy = approx.entropy(var_param) + entropy_offset
    log_probs = approx.log_density(var_param, approx.sample(var_param, MC_SAMPLES))
    p_value = stats.ttest_1samp(log_probs, -entropy)[1]
    assert p_value > test_size, "expected: {}, estimated: {}".format(entropy, -np.mean(log_probs))


def _test_kl(approx, var_param0, var_param1):                      # :19-26
    kl = approx.kl(var_param0, var_param1)
    samples = approx.sample(var_param0, MC_SAMPLES)
    diffs = approx.log_density(var_param0, samples) - approx.log_density(var_param1, samples)
    assert stats.ttest_1samp(diffs, kl)[1] > test_size


def _test_mean_and_cov(approx, var_param):                         # :29-40
    mean, cov = approx.mean_and_cov(var_param)
    second_moments = np.outer(mean, mean) + cov
    samples = approx.sample(var_param, MC_SAMPLES)
    samples_outer = np.einsum('ij,ik->ijk', samples, samples)
    np.testing.assert_array_less(test_size, stats.ttest_1samp(samples, mean, axis=0)[1])
    np.testing.assert_array_less(test_size, stats.ttest_1samp(samples_outer, second_moments, axis=0)[1])


def _test_pth_moment(approx, var_param, p):                        # :43-52
    pth_moment = approx.pth_moment(var_param, p)
    samples = approx.sample(var_param, MC_SAMPLES)
    norms = np.linalg.norm(samples - np.mean(samples, axis=0), axis=1, ord=2)
    p_value = stats.ttest_1samp(norms ** p, pth_moment)[1]
    assert p_value > test_size, "expected: {}, estimated: {}".format(pth_moment, np.mean(norms ** p))


def _test_family(approx, var_param0, var_param1, should_support=(), entropy_offset=0):      # :55-75
    if approx.supports_entropy:
        _test_entropy(approx, var_param0, entropy_offset)
    else:
        with pytest.raises(NotImplementedError):
            approx.entropy(var_param0)
    if approx.supports_kl:
        _test_kl(approx, var_param0, var_param1)
    else:
        with pytest.raises(NotImplementedError):
            approx.kl(var_param0, var_param1)
    _test_mean_and_cov(approx, var_param0)
    for p in set([1, 2, 4]) | set(should_support):
        if p in should_support:
            assert approx.supports_pth_moment(p)
        if approx.supports_pth_moment(p):
            _test_pth_moment(approx, var_param0, p)
        else:
            with pytest.raises(ValueError):
                approx.pth_moment(var_param0, p)


def test_MFGaussian(vb):                                           # :78-86
    np.random.seed(341)
    for dim in [1, 3]:
        approx = vb.MFGaussian(dim, rng='philox')
        for i in range(3):
            var_param0 = np.random.randn(approx.var_param_dim)
            var_param1 = np.random.randn(approx.var_param_dim)
            _test_family(approx, var_param0, var_param1, [2, 4])


def test_MFStudentT(vb):                                           # :89-100
    np.random.seed(226)
    df = 20
    for dim in [1, 3]:
        approx = vb.MFStudentT(dim, df, rng='philox')
        for i in range(3):
            var_param0 = np.random.randn(approx.var_param_dim)
            var_param1 = np.random.randn(approx.var_param_dim)
            _test_family(approx, var_param0, var_param1, [2, 4], dim * stats.t.entropy(df))


def test_MultivariateT(vb):                                        # :103-114
    np.random.seed(56)
    df = 100
    for dim in [1, 3]:
        approx = vb.MultivariateT(dim, df, rng='philox')
        for i in range(3):
            var_param0 = np.random.randn(approx.var_param_dim)
            var_param1 = np.random.randn(approx.var_param_dim)
            _test_family(approx, var_param0, var_param1, [2, 4], dim * stats.t.entropy(df))


def test_FullRankGaussian(vb):
    """The dense Gaussian family (no reference class) through the same self-consistency statistics."""
    np.random.seed(77)
    for dim in [1, 3]:
        approx = vb.FullRankGaussian(dim, rng='philox')
        for i in range(3):
            var_param0 = 0.5 * np.random.randn(approx.var_param_dim)
            var_param1 = 0.5 * np.random.randn(approx.var_param_dim)
            _test_family(approx, var_param0, var_param1, [2, 4])
