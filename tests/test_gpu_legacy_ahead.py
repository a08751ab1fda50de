"""GPU: look-ahead generation of numpy's legacy streams (round 6; LegacySpec in vb_api.hip, vb_legacy_round_end).

A family in the reference-identical mode makes the same device draws call after call from one persistent RandomState
(viabel/approximations.py:213-216, :270-274, :342-349); when a call's draws are done the engine starts the NEXT call's --
the same requests from the generator's current state -- on a stream of its own, into shadow buffers, beside the objective's
kernels.  A request that finds the generator exactly where the speculation started adopts the shadow and the speculated end
state.  The bar is the integer one of tests/test_gpu_legacy_rng.py: every value and every generator state ARRAY-EQUAL to
numpy.random.RandomState itself, whether a draw was adopted or made on the spot -- plain loops, host draws in between,
changed shapes, reseeds, set_state, two generators taking turns, sharded row blocks."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import viabel_amd as vb
    from viabel_amd import _lib
    from viabel_amd._legacy_rng import LegacyRandomState
    return vb, _lib.default_engine(), LegacyRandomState


def _same_state(ours, ref):
    a, b = ours.get_state(), ref.get_state()
    np.testing.assert_array_equal(a[1], b[1])
    assert a[2:] == b[2:]


def _round(eng, ours, ref, reqs, slot0=20):
    """One call's draws (device) against numpy's, then the round end."""
    for k, r in enumerate(reqs):
        if r[0] == 'n':
            _, n, d = r
            assert eng.noise_legacy_randn(slot0 + k, ours._h, n, d)
            np.testing.assert_array_equal(eng.noise_get_host(slot0 + k, n, d), ref.randn(n, d))
        elif r[0] == 't':
            _, df, n, d = r
            assert eng.noise_legacy_standard_t(slot0 + k, ours._h, df, n, d)
            np.testing.assert_array_equal(eng.noise_get_host(slot0 + k, n, d), ref.standard_t(df, size=(n, d)))
        else:
            _, df, n = r
            got = eng.chisq_legacy(ours._h, df, n, to_host=True)
            assert got is not None
            np.testing.assert_array_equal(got, ref.chisquare(df, n))
            np.testing.assert_array_equal(eng.chisq_get_host(n), got)
        _same_state(ours, ref)
    eng.legacy_round_end(ours._h)


@pytest.mark.parametrize('reqs', [
    [('n', 4096, 1024)],                          # MFGaussian / FullRankGaussian at the C1 / headline shape
    [('n', 1001, 77)],                            # odd count: a cached normal crosses every round boundary
    [('t', 7.0, 2048, 256)],                      # MFStudentT
    [('c', 100.0, 16384), ('n', 16384, 256)],     # MultivariateT at the C3 shape: chi-square draws, then the normals
    [('c', 9.0, 5001), ('n', 5001, 33)],
    [('n', 3000, 5), ('n', 3000, 300)],           # LRGaussian: the low-rank block first
])
def test_plain_loops_adopt_and_stay_numpy(env, reqs):
    vb, eng, LegacyRandomState = env
    ours, ref = LegacyRandomState(31), np.random.RandomState(31)
    before = eng.legacy_ahead_stats()
    for it in range(6):
        _round(eng, ours, ref, reqs)
    launched, adopted, discarded = (b - a for a, b in zip(before, eng.legacy_ahead_stats()))
    # rounds 0 and 1 establish the pattern; the job launched at the end of round 1 serves round 2, and so on
    assert launched >= 4 and adopted >= 4 * len(reqs), (launched, adopted, discarded)
    # the generator goes on where numpy's does (a host draw behind an adopted device draw)
    np.testing.assert_array_equal(ours.randn(7), ref.randn(7))
    eng.sync()


def test_a_generator_that_moved_in_between_discards_the_speculation(env):
    vb, eng, LegacyRandomState = env
    ours, ref = LegacyRandomState(5), np.random.RandomState(5)
    reqs = [('n', 2048, 64)]
    for it in range(12):
        if it in (3, 4, 7):               # host draws between two calls: the speculated start state is stale
            np.testing.assert_array_equal(ours.randn(3), ref.randn(3))
        if it == 9:                       # ... a reseed
            ref.seed(77)
            ours.set_state(np.random.RandomState(77).get_state())
        if it == 10:                      # ... a state handed back and forth
            ours.set_state(ref.get_state())
        _round(eng, ours, ref, reqs)
    np.testing.assert_array_equal(ours.standard_t(5.0, size=4), ref.standard_t(5.0, size=4))


def test_changing_shapes_and_two_generators_taking_turns(env):
    vb, eng, LegacyRandomState = env
    a, ra = LegacyRandomState(1), np.random.RandomState(1)
    b, rb = LegacyRandomState(2), np.random.RandomState(2)
    shapes = [[('n', 2048, 64)], [('n', 2048, 64)], [('n', 2048, 64)], [('n', 1024, 64)], [('n', 1024, 64)], [('n', 1024, 64)],
              [('t', 6.0, 1024, 64)], [('t', 6.0, 1024, 64)], [('t', 6.0, 1024, 64)]]
    for it, reqs in enumerate(shapes):
        _round(eng, a, ra, reqs)
        _round(eng, b, rb, [('c', 50.0, 4100), ('n', 4100, 16)], slot0=24)      # another generator in between, every time
    _same_state(a, ra), _same_state(b, rb)


def test_sharded_row_blocks_are_adopted_too(env):
    """A rank of a sharded job draws ITS rows of randn(N, D) (the generator still ends where numpy's does)."""
    vb, eng, LegacyRandomState = env
    n, d, begin, rows = 4099, 130, 2050, 2049
    ours, ref = LegacyRandomState(9), np.random.RandomState(9)
    before = eng.legacy_ahead_stats()
    for it in range(5):
        assert eng.noise_legacy_randn(21, ours._h, n, d, begin, rows)
        np.testing.assert_array_equal(eng.noise_get_host(21, rows, d), ref.randn(n, d)[begin:begin + rows])
        _same_state(ours, ref)
        eng.legacy_round_end(ours._h)
    assert eng.legacy_ahead_stats()[1] - before[1] >= 3


@pytest.mark.parametrize('family', ['mf', 'student', 'fullrank', 'mvt_dis', 'mvt_ekl', 'lowrank'])
def test_objective_loops_equal_the_loops_without_look_ahead(env, family):
    """Whole objective calls in the default (rng='numpy') mode, with and without the look-ahead: the same values, gradients
    and generator states, bit for bit -- and the look-ahead really served the calls."""
    vb, eng, LegacyRandomState = env
    rng = np.random.RandomState(3)
    D, N = (64, 8192) if family != 'mf' else (256, 4096)
    model = vb.GaussianModel(0.2 * rng.randn(D), np.exp(0.1 * rng.randn(D)))
    th_mf = np.concatenate([0.1 * rng.randn(D), -0.5 + 0.1 * rng.randn(D)])
    from oracle import families as ofam
    A = rng.randn(D, D)
    th_ch = np.concatenate([0.1 * rng.randn(D), ofam.psd_to_free(0.7 * (A @ A.T / D + np.eye(D)))])

    def loop():
        np.random.seed(4)
        if family == 'mf':
            obj, th = vb.ExclusiveKL(vb.MFGaussian(D, seed=5), model, N), th_mf
        elif family == 'student':
            obj, th = vb.ExclusiveKL(vb.MFStudentT(D, 7.0, seed=5), model, N), th_mf
        elif family == 'fullrank':
            obj, th = vb.ExclusiveKL(vb.FullRankGaussian(D, seed=5), model, N), th_ch
        elif family == 'mvt_ekl':
            obj, th = vb.ExclusiveKL(vb.MultivariateT(D, 9.0, seed=5), model, N), th_ch
        elif family == 'lowrank':
            fam = vb.LRGaussian(D, seed=5, k=4)
            obj, th = vb.ExclusiveKL(fam, model, N), fam.pack(0.1 * rng.randn(D) * 0, -0.5 * np.ones(D), 0.1 * np.ones((D, 4)))
        else:
            prior = np.concatenate([np.zeros(D), 0.3 * np.ones(D)])
            obj = vb.DISInclusiveKL(vb.MultivariateT(D, 9.0, seed=5), model, N, ess_target=1000, temper_prior=vb.MFGaussian(D),
                                    temper_prior_params=prior, use_resampling=True)
            th = th_ch
        out = []
        for it in range(6):
            v, g = obj(th)
            out.append((v, g.copy()))
            th = th - 0.01 * g / (1.0 + np.abs(g))
        return out, obj.approx._rs.get_state()
    old = os.environ.get('VB_LEGACY_AHEAD')
    try:
        os.environ['VB_LEGACY_AHEAD'] = '0'
        plain, st_plain = loop()
        os.environ['VB_LEGACY_AHEAD'] = '1'
        before = eng.legacy_ahead_stats()
        ahead, st_ahead = loop()
        adopted = eng.legacy_ahead_stats()[1] - before[1]
    finally:
        if old is None:
            os.environ.pop('VB_LEGACY_AHEAD', None)
        else:
            os.environ['VB_LEGACY_AHEAD'] = old
    assert adopted >= 3, adopted
    for (v0, g0), (v1, g1) in zip(plain, ahead):
        assert v0 == v1
        np.testing.assert_array_equal(g0, g1)
    np.testing.assert_array_equal(st_plain[1], st_ahead[1])
    assert st_plain[2:] == st_ahead[2:]


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_random_sequences_of_draws_rounds_and_host_draws_stay_numpy(env, seed):
    """Fuzz: 60 random steps on two generators -- device draws of random kinds and shapes (repeated often enough for the
    look-ahead to start and to be adopted), round ends at random, host draws of all kinds, get_state / set_state round trips --
    every value and every generator state against numpy.random.RandomState."""
    vb, eng, LegacyRandomState = env
    rnd = np.random.RandomState(1000 + seed)
    gens = [(LegacyRandomState(10 + seed), np.random.RandomState(10 + seed)), (LegacyRandomState(20 + seed), np.random.RandomState(20 + seed))]
    menus = [[('n', 2048, 48)], [('c', 12.0, 4500), ('n', 4500, 16)], [('t', 5.0, 1500, 32)], [('n', 999, 7), ('n', 999, 41)]]
    before = eng.legacy_ahead_stats()
    current = [menus[rnd.randint(len(menus))] for _ in gens]
    for step in range(60):
        g = rnd.randint(len(gens))
        ours, ref = gens[g]
        action = rnd.rand()
        if action < 0.62:                       # the generator's usual round (so that speculation gets going)
            _round(eng, ours, ref, current[g], slot0=30 + 4 * g)
        elif action < 0.72:                     # another round shape from now on
            current[g] = menus[rnd.randint(len(menus))]
        elif action < 0.82:                     # a host draw in between
            k = 1 + rnd.randint(5)
            kind = rnd.randint(3)
            if kind == 0:
                np.testing.assert_array_equal(ours.randn(k), ref.randn(k))
            elif kind == 1:
                np.testing.assert_array_equal(ours.standard_t(4.0, size=k), ref.standard_t(4.0, size=k))
            else:
                np.testing.assert_array_equal(ours.chisquare(3.0, k), ref.chisquare(3.0, k))
        elif action < 0.9:                      # a device draw outside any round pattern, no round end
            n, d = 1 + rnd.randint(3000), 1 + rnd.randint(40)
            if n * d >= 2:
                assert eng.noise_legacy_randn(29, ours._h, n, d)
                np.testing.assert_array_equal(eng.noise_get_host(29, n, d), ref.randn(n, d))
        else:                                   # state out and in again (and into the OTHER generator now and then)
            st = ref.get_state()
            ours.set_state(st)
            if rnd.rand() < 0.3:
                o2, r2 = gens[1 - g]
                o2.set_state(st), r2.set_state(st)
        _same_state(ours, ref)
    launched, adopted, discarded = (b - a for a, b in zip(before, eng.legacy_ahead_stats()))
    assert launched > 0 and adopted > 0, (launched, adopted, discarded)      # the fuzz did exercise the look-ahead
    eng.sync()
