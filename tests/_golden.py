"""Helpers shared by the oracle tests and the GPU parity tests: load fixtures, rebuild specs."""
import glob
import os

import numpy as np

from oracle import families as ofam
from oracle import models as omod

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def fixtures(prefix):
    paths = sorted(glob.glob(os.path.join(GOLDEN_DIR, prefix + '*.npz')))
    assert paths, 'no golden fixtures named %s*' % prefix
    return paths


def load(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: (z[k].item() if z[k].ndim == 0 else z[k]) for k in z.files}


def oracle_family(fx):
    kind, D = str(fx['family_kind']), int(fx['dim'])
    if kind == 'mf_gaussian':
        return ofam.MFGaussian(D)
    if kind == 'mf_student_t':
        return ofam.MFStudentT(D, float(fx['df']))
    if kind == 'multivariate_t':
        return ofam.MultivariateT(D, float(fx['df']))
    if kind == 'lr_gaussian':
        return ofam.LRGaussian(D, int(fx['rank']))
    raise ValueError(kind)


def oracle_prior_family(fx):
    """The tempering prior of a `disprior_*` fixture as an oracle family."""
    kind, D = str(fx['prior_kind']), int(fx['dim'])
    if kind == 'mf_student_t':
        return ofam.MFStudentT(D, float(fx['prior_df']))
    if kind == 'multivariate_t':
        return ofam.MultivariateT(D, float(fx['prior_df']))
    if kind == 'lr_gaussian':
        return ofam.LRGaussian(D, int(fx['prior_rank']))
    raise ValueError(kind)


def oracle_model(fx):
    kind = str(fx['model_kind'])
    if kind == 'gauss_diag':
        return omod.GaussDiag(fx['model_mean'], fx['model_stdev'])
    if kind == 'funnel':
        return omod.Funnel(int(fx['dim']), int(fx['model_scale_index']),
                           float(fx['model_log_sigma_stdev']))
    raise ValueError(kind)


def noise_of(fx):
    if 'noise' in fx:
        return fx['noise']
    if 'noise_eps' in fx:                      # LRGaussian: low-rank block first (approximations.py:639-640)
        return fx['noise_z'], fx['noise_eps']
    return fx['noise_chi'], fx['noise_z']


def rel_err(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def ids(paths):
    return [os.path.splitext(os.path.basename(p))[0] for p in paths]
