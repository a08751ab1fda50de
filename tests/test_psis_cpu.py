"""CPU checks of the PSIS / diagnostics oracle (oracle/psis.py) and of the host-side diagnostics module against
vectors produced by the reference's own viabel/_psis.py and viabel/diagnostics.py (tests/golden/psis.npz)."""
import warnings

import numpy as np
import pytest

import _golden as G
from oracle import psis as opsis
from viabel_amd import diagnostics as diag

FX = G.load(G.fixtures('psis')[0])
NAMES = [str(n) for n in FX['names']]


def _close_k(k, ref):
    if np.isinf(ref):
        return np.isinf(k)
    return abs(k - ref) <= 1e-10 * max(1.0, abs(ref))


def assert_smoothed_equal(raw, got, want, atol=1e-10):
    """Equal up to the order in which tied raw weights receive their quantiles: the reference orders ties with
    numpy's unstable argsort (implementation-defined), this build by index."""
    if np.unique(raw).size == raw.size:
        np.testing.assert_allclose(got, want, rtol=0, atol=atol)
        return
    order = np.argsort(raw, kind='stable')
    bounds = np.flatnonzero(np.diff(raw[order])) + 1
    for grp in np.split(order, bounds):
        np.testing.assert_allclose(np.sort(got[grp]), np.sort(want[grp]), rtol=0, atol=atol)


@pytest.mark.parametrize('name', NAMES)
def test_oracle_psis_matches_reference(name):
    sm, k = opsis.psis_smooth(FX[name + '_lw'])
    assert _close_k(k, float(FX[name + '_khat']))
    assert_smoothed_equal(FX[name + '_lw'], sm, FX[name + '_smoothed'])
    assert abs(opsis.log_sum_exp(sm)) < 1e-12


def test_oracle_psis_reff_and_columns():
    sm, k = opsis.psis_smooth(FX['normal_heavy_lw'], reff=float(FX['reff_value']))
    assert _close_k(k, float(FX['reff_khat']))
    np.testing.assert_allclose(sm, FX['reff_smoothed'], rtol=0, atol=1e-10)
    for j in range(2):
        sm, k = opsis.psis_smooth(FX['two_lw'][:, j])
        assert _close_k(k, float(FX['two_khat'][j]))
        np.testing.assert_allclose(sm, FX['two_smoothed'][:, j], rtol=0, atol=1e-10)


def test_oracle_gpd_fit():
    k, sigma = opsis.gpd_fit(FX['gpd_x'])
    assert abs(k - float(FX['gpd_k'])) < 1e-12 and abs(sigma - float(FX['gpd_sigma'])) < 1e-12


def test_tail_size_rule():
    assert opsis.tail_size(16384) == 384 and opsis.tail_size(100000) == 949 and opsis.tail_size(10) == 2


@pytest.mark.parametrize('impl', ['oracle', 'host'])
def test_diagnostics_match_reference(impl):
    lw, samples = FX['diag_lw'], FX['diag_samples']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if impl == 'host':
            res = diag.all_diagnostics(lw, samples=samples)
            res_q = diag.all_diagnostics(lw, samples=samples, q_var=9.0 * np.eye(2), p_var=4.0 * np.eye(2),
                                         log_norm_bound=0.0)
            d3 = diag.divergence_bound(lw, alpha=3.0)
        else:
            centred = samples - samples.mean(0, keepdims=True)

            def moments(p):
                return np.mean(np.sum(centred ** p, axis=1))
            res, res_q = {}, {}
            for r, lnb, qv, pv in ((res, None, np.cov(samples.T), None), (res_q, 0.0, 9.0 * np.eye(2), 4.0 * np.eye(2))):
                r['d2'], r['log_norm_bound'] = opsis.divergence_bound(lw, log_norm_bound=lnb)
                r.update(opsis.wasserstein_bounds(r['d2'], moments))
                r.update(opsis.error_bounds(r['W1'], r['W2'], qv, pv))
            d3 = opsis.divergence_bound(lw, alpha=3.0)[0]
    for key in ('d2', 'log_norm_bound', 'W1', 'W2', 'mean_error', 'std_error', 'cov_error'):
        np.testing.assert_allclose(res[key], float(FX['diag_' + key]), rtol=1e-12)
        np.testing.assert_allclose(res_q[key], float(FX['diagq_' + key]), rtol=1e-12)
    np.testing.assert_allclose(d3, float(FX['diag_d3']), rtol=1e-12)


def test_diagnostics_validation():
    with pytest.raises(ValueError):
        diag.divergence_bound(np.zeros(4), alpha=1.0)
    with pytest.raises(ValueError):
        diag.wasserstein_bounds(1.0)
    with pytest.raises(ValueError):
        opsis.psis_smooth(np.zeros(1))
