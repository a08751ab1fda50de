"""GPU: Monte-Carlo self-consistency of the throughput-mode families (rng='philox').

Every closed form a family publishes (entropy, KL, mean / covariance, p-th central moment) is compared with a
Monte-Carlo estimate from the family's own device draws by a one-sample t statistic.  The statistics, the seeds
(341 / 226 / 56) and the rejection level (p > 1e-4) are those of the reference's family checks
(viabel/tests/test_approximations.py:11-113); the code below is this repository's: one table of statistics, one
parametrised test per (family, dim, draw of the parameters).  250 000 draws instead of 10^6: a smaller
sample only weakens the test, a wrong closed form still shows as p ~ 0.
"""
import numpy as np
import pytest
from scipy import stats

pytestmark = pytest.mark.gpu

N_DRAWS = 250000
P_MIN = 1e-4
MOMENTS = (1, 2, 4)

# family name -> (seed, constructor kwargs, parameter scale, per-dimension entropy offset, orders that must be supported)
FAMILIES = {
    'MFGaussian': (341, {}, 1.0, 0.0, (2, 4)),
    'MFStudentT': (226, {'df': 20}, 1.0, float(stats.t.entropy(20)), (2, 4)),
    'MultivariateT': (56, {'df': 100}, 1.0, float(stats.t.entropy(100)), (2, 4)),
    'FullRankGaussian': (77, {}, 0.5, 0.0, (2, 4)),          # no reference class: same statistics, own seed
}
DIMS = (1, 3)
REPEATS = 3


@pytest.fixture(scope='module')
def vb():
    import viabel_amd
    from viabel_amd import _lib
    _lib.default_engine()
    return viabel_amd


def _parameter_pairs(vb, name):
    """The (family object, theta0, theta1) triples of one family, drawn in the reference's order from its seed:
    for each dim, REPEATS times two standard-normal parameter vectors."""
    seed, kwargs, scale, offset, must = FAMILIES[name]
    gen = np.random.RandomState(seed)
    out = []
    for dim in DIMS:
        fam = getattr(vb, name)(dim, rng='philox', **kwargs)
        for _ in range(REPEATS):
            t0 = scale * gen.randn(fam.var_param_dim)
            t1 = scale * gen.randn(fam.var_param_dim)
            out.append((fam, t0, t1, dim * offset, must))
    return out


def _p_of_mean(x, expected):
    """Two-sided p-value of H0: E[x] = expected, along axis 0."""
    return stats.ttest_1samp(x, expected, axis=0).pvalue


def _check(label, x, expected):
    p = np.asarray(_p_of_mean(x, expected))
    assert np.all(p > P_MIN), '%s: closed form %s, Monte-Carlo %s, p = %s' % (
        label, np.asarray(expected), np.mean(x, axis=0), p)


@pytest.mark.parametrize('name', sorted(FAMILIES))
def test_family_closed_forms_match_own_draws(vb, name):
    for fam, t0, t1, offset, must in _parameter_pairs(vb, name):
        z = fam.sample(t0, N_DRAWS)
        logq0 = fam.log_density(t0, z)

        # entropy: E_q0[-log q0] (the t families publish theirs up to the constant `offset`)
        if fam.supports_entropy:
            _check('entropy', -logq0, fam.entropy(t0) + offset)
        else:
            with pytest.raises(NotImplementedError):
                fam.entropy(t0)

        # KL(q0 || q1) = E_q0[log q0 - log q1]
        if fam.supports_kl:
            _check('kl', logq0 - fam.log_density(t1, z), fam.kl(t0, t1))
        else:
            with pytest.raises(NotImplementedError):
                fam.kl(t0, t1)

        # first and (raw) second moments
        mean, cov = fam.mean_and_cov(t0)
        _check('mean', z, mean)
        _check('second moments', z[:, :, None] * z[:, None, :], cov + mean[:, None] * mean[None, :])

        # E || z - mean ||_2^p about the sample mean
        radius = np.sqrt(np.sum((z - z.mean(axis=0)) ** 2, axis=1))
        for p in sorted(set(MOMENTS) | set(must)):
            if p in must:
                assert fam.supports_pth_moment(p)
            if fam.supports_pth_moment(p):
                _check('moment %d' % p, radius ** p, fam.pth_moment(t0, p))
            else:
                with pytest.raises(ValueError):
                    fam.pth_moment(t0, p)
