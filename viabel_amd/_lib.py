"""ctypes binding of ``libviabel_hip.so`` (the C ABI in ``include/viabel_hip.h``).

The shared library is the product: there is no CPU fallback.  If it is missing, or
no MI355X is visible, every compute call raises -- loudly -- instead of silently
computing something else.
"""
import ctypes
import weakref
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VIABEL_AMD_LIB: load another build of the same C ABI (the host-sanitizer build libviabel_hip_asan.so, `make asan`)
LIB_PATH = os.environ.get('VIABEL_AMD_LIB') or os.path.join(_HERE, 'libviabel_hip.so')
CSRC_DIR = os.path.join(_HERE, 'csrc')

VB_OK, VB_ERR_INVALID, VB_ERR_HIP, VB_ERR_UNSUPPORTED, VB_ERR_STATE, VB_ERR_NUMERIC, VB_ERR_COMM, VB_ERR_CALLBACK = range(8)

FAMILY_MF_GAUSSIAN, FAMILY_MF_STUDENT_T, FAMILY_FULLRANK_GAUSSIAN, FAMILY_MULTIVARIATE_T, FAMILY_LOWRANK_GAUSSIAN = \
    range(5)
MODEL_GAUSS_DIAG, MODEL_FUNNEL, MODEL_GAUSS_FULL, MODEL_LOGISTIC, MODEL_SOURCE = range(5)
GLM_BERNOULLI_LOGIT, GLM_POISSON, GLM_GAUSSIAN = range(3)
NOISE_NORMAL, NOISE_STUDENT_T = range(2)
FLAG_PATH_DERIV = 1
CV_MODES = {None: 0, 'full': 1, 'mean_only': 2, 'loo_diag_approx': 3, 'loo_direct_approx': 4}
MAX_SLOTS = 64
OPT_SGD, OPT_RMSPROP, OPT_ADAM, OPT_ADAGRAD = range(4)
PRIOR_DIAG_GAUSSIAN, PRIOR_DIAG_STUDENT_T, PRIOR_DENSE = range(3)
# int fn(void* user, const double* z, int64 n, int64 d, double* f, double* grad)   (include/viabel_hip.h: vb_model_callback)
MODEL_CALLBACK_TYPE = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int64,
                                       ctypes.c_int64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double))
COMM_ID_BYTES = 128
PROF_MF_ACCUM, PROF_FR_SAMPLE_GEMM, PROF_FR_MODEL_GEMM, PROF_FR_GRAD_GEMM = range(4)

# `const double*` / `double*` parameters are declared void*: ctypes then takes the array's address as a plain integer
# (`_dptr`), half the cost of building a POINTER(c_double) per argument on calls that last tens of microseconds
_c_double_p = ctypes.c_void_p
_c_int64_p = ctypes.POINTER(ctypes.c_int64)
_ctx_p = ctypes.c_void_p

# name -> (restype, argtypes); every symbol declared in include/viabel_hip.h
SIGNATURES = {
    'vb_version': (ctypes.c_char_p, []),
    'vb_device_count': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    'vb_create': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_ctx_p)]),
    'vb_destroy': (ctypes.c_int, [_ctx_p]),
    'vb_last_error': (ctypes.c_char_p, [_ctx_p]),
    'vb_device_info': (ctypes.c_int, [_ctx_p, ctypes.c_char_p, ctypes.c_size_t,
                                      ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64)]),
    'vb_sync': (ctypes.c_int, [_ctx_p]),
    'vb_noise_set_host': (ctypes.c_int, [_ctx_p, ctypes.c_int, _c_double_p, ctypes.c_int64, ctypes.c_int64]),
    'vb_noise_generate': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                         ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64,
                                         ctypes.c_int64, ctypes.c_int64]),
    'vb_noise_hint_seed': (ctypes.c_int, [_ctx_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64]),
    'vb_noise_ahead_stats': (ctypes.c_int, [_ctx_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    'vb_chisq_generate': (ctypes.c_int, [_ctx_p, ctypes.c_double, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64,
                                         ctypes.c_int64]),
    'vb_chisq_get_host': (ctypes.c_int, [_ctx_p, _c_double_p, ctypes.c_int64]),
    'vb_noise_get_host': (ctypes.c_int, [_ctx_p, ctypes.c_int, _c_double_p, ctypes.c_int64, ctypes.c_int64]),
    'vb_set_model': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, _c_double_p, ctypes.c_size_t,
                                    _c_int64_p, ctypes.c_size_t]),
    'vb_set_model_source': (ctypes.c_int, [_ctx_p, ctypes.c_int64, ctypes.c_char_p, _c_double_p, ctypes.c_size_t]),
    'vb_set_model_callback': (ctypes.c_int, [_ctx_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    'vb_model_logp': (ctypes.c_int, [_ctx_p, _c_double_p, ctypes.c_int64, ctypes.c_int64, _c_double_p]),
    'vb_model_grad': (ctypes.c_int, [_ctx_p, _c_double_p, ctypes.c_int64, ctypes.c_int64, _c_double_p, _c_double_p]),
    'vb_elbo_sums_lowrank': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                            _c_double_p, _c_double_p]),
    'vb_elbo_grad_mvt_chol': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_double,
                                             _c_double_p, _c_double_p, _c_double_p]),
    'vb_dis_step_mvt_packed': (ctypes.c_int, [_ctx_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_double, _c_double_p, ctypes.c_int64,
                                              ctypes.c_uint64, ctypes.c_uint64, ctypes.c_double, _c_double_p, _c_double_p,
                                              _c_double_p, _c_double_p, _c_double_p]),
    'vb_dis_psis_mvt': (ctypes.c_int, [_ctx_p, ctypes.c_int64, ctypes.c_double]),
    'vb_dis_weights_get': (ctypes.c_int, [_ctx_p, _c_double_p, ctypes.c_int64, ctypes.c_int]),
    'vb_noise_moments': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, _c_double_p, _c_double_p]),
    'vb_elbo_grad_meanfield': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                              ctypes.c_int64, ctypes.c_int, ctypes.c_double, _c_double_p,
                                              ctypes.c_uint, ctypes.c_int, _c_double_p, _c_double_p]),
    'vb_elbo_grad_meanfield_async': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                                    ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                                    _c_double_p, ctypes.c_uint, ctypes.c_int, ctypes.c_int]),
    'vb_elbo_grad_meanfield_batch_async': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                                          ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                                          ctypes.c_int, ctypes.c_double, _c_double_p,
                                                          ctypes.c_uint, ctypes.c_int,
                                                          ctypes.POINTER(ctypes.c_int)]),
    'vb_elbo_sums_mvt': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                        _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p,
                                        _c_double_p]),
    'vb_elbo_grad_lowrank': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                            ctypes.c_int64, ctypes.c_int64, _c_double_p, ctypes.c_uint, _c_double_p,
                                            _c_double_p]),
    'vb_sym_sqrt': (ctypes.c_int, [_ctx_p, _c_double_p, _c_double_p, ctypes.c_int64, _c_double_p, _c_double_p,
                                   _c_double_p]),
    'vb_alpha_grad_fullrank': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                              _c_double_p, ctypes.c_double, _c_double_p, _c_double_p]),
    'vb_alpha_grad_mvt_chol': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                              ctypes.c_double, _c_double_p, ctypes.c_double, _c_double_p, _c_double_p]),
    'vb_sym_sqrt_inv': (ctypes.c_int, [_ctx_p, _c_double_p, _c_double_p, ctypes.c_int64, _c_double_p, _c_double_p,
                                       _c_double_p, _c_double_p]),
    'vb_lowrank_path_terms': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.c_int64, ctypes.c_int64, _c_double_p, _c_double_p]),
    'vb_mvt_path_terms': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                         ctypes.c_double, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    'vb_alpha_sums_mvt': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                         ctypes.c_double, ctypes.c_double, _c_double_p, _c_double_p, _c_double_p,
                                         ctypes.c_double, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    'vb_elbo_grad_meanfield_philox': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                                     ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                                     _c_double_p, ctypes.c_uint, ctypes.c_int, ctypes.c_uint64,
                                                     ctypes.c_uint64, _c_double_p, _c_double_p]),
    'vb_fit': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                              ctypes.c_int, ctypes.c_double, ctypes.c_uint, ctypes.c_int, ctypes.c_int,
                              ctypes.c_double, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, _c_double_p,
                              ctypes.c_int64, _c_double_p, ctypes.c_int64, _c_double_p, ctypes.c_int,
                              _c_double_p, _c_double_p, ctypes.c_int64, _c_double_p, _c_double_p]),
    'vb_fit_history_mean': (ctypes.c_int, [_ctx_p, ctypes.c_int64, ctypes.c_int64, _c_double_p]),
    'vb_mvt_route_stats': (ctypes.c_int, [_ctx_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    'vb_log_weights_meanfield': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                                ctypes.c_double, _c_double_p, _c_double_p]),
    'vb_psis_smooth': (ctypes.c_int, [_ctx_p, _c_double_p, ctypes.c_int64, ctypes.c_double, _c_double_p,
                                      _c_double_p]),
    'vb_alpha_grad_meanfield': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                               ctypes.c_int, ctypes.c_double, _c_double_p, ctypes.c_double,
                                               _c_double_p, _c_double_p]),
    'vb_dis_clip_mvt': (ctypes.c_int, [_ctx_p, ctypes.c_int64, ctypes.c_double]),
    'vb_dis_set_temper_prior': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double, _c_double_p,
                                               _c_double_p, ctypes.c_double]),
    'vb_dis_refresh_meanfield': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                                ctypes.c_int, ctypes.c_double, _c_double_p, _c_double_p,
                                                ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                                _c_double_p, _c_double_p, _c_double_p, _c_double_p,
                                                _c_double_p]),
    'vb_dis_grad_meanfield': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.c_int, ctypes.c_double, _c_double_p, _c_double_p,
                                             ctypes.c_double, _c_double_p, _c_double_p]),
    'vb_dis_refresh_mvt': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_double,
                                          _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p,
                                          ctypes.c_double, ctypes.c_double, ctypes.c_int, _c_double_p,
                                          _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    'vb_dis_scalars_get': (ctypes.c_int, [_ctx_p, _c_double_p]),
    'vb_elbo_grad_mvt_symroot': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_double, _c_double_p,
                                                ctypes.POINTER(ctypes.c_double), _c_double_p, _c_double_p]),
    'vb_alpha_grad_mvt_symroot': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_double,
                                                 ctypes.c_double, _c_double_p, ctypes.POINTER(ctypes.c_double), _c_double_p,
                                                 _c_double_p]),
    'vb_elbo_grad_mvt_symroot_path': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_double,
                                                     _c_double_p, ctypes.POINTER(ctypes.c_double), _c_double_p, _c_double_p]),
    'vb_dis_refresh_mvt_symroot': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_double,
                                                  _c_double_p, _c_double_p, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                                  _c_double_p]),
    'vb_dis_refresh_lowrank': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                              ctypes.c_int64, ctypes.c_int64, _c_double_p, _c_double_p, _c_double_p,
                                              _c_double_p, ctypes.c_double, _c_double_p, ctypes.c_double,
                                              ctypes.c_double, ctypes.c_int, _c_double_p, _c_double_p, _c_double_p,
                                              _c_double_p, _c_double_p]),
    'vb_dis_grad_lowrank': (ctypes.c_int, [_ctx_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _c_double_p,
                                           _c_double_p, _c_double_p, _c_double_p, ctypes.c_double, _c_double_p,
                                           _c_double_p]),
    'vb_alpha_sums_lowrank': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.c_int64, ctypes.c_int64, ctypes.c_double, _c_double_p, _c_double_p,
                                             _c_double_p, _c_double_p, ctypes.c_double, _c_double_p, _c_double_p,
                                             _c_double_p]),
    'vb_dis_grad_mvt_packed': (ctypes.c_int, [_ctx_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_double, _c_double_p,
                                              _c_double_p, ctypes.c_double, _c_double_p, _c_double_p]),
    'vb_dis_generation': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]),
    'vb_dis_state_get': (ctypes.c_int, [_ctx_p, ctypes.c_int, _c_double_p, _c_double_p, ctypes.c_int64]),
    'vb_dis_grad_mvt': (ctypes.c_int, [_ctx_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_double, _c_double_p,
                                       _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p,
                                       _c_double_p]),
    'vb_elbo_grad_fullrank': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.c_int64, _c_double_p, ctypes.c_uint, _c_double_p,
                                             _c_double_p]),
    'vb_fullrank_set_theta': (ctypes.c_int, [_ctx_p, _c_double_p, ctypes.c_int64]),
    'vb_elbo_grad_fullrank_enqueue': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                                     ctypes.c_int64, ctypes.c_uint]),
    'vb_fullrank_get': (ctypes.c_int, [_ctx_p, _c_double_p, _c_double_p, ctypes.c_int64]),
    'vb_result_get': (ctypes.c_int, [_ctx_p, ctypes.c_int, _c_double_p, _c_double_p, ctypes.c_int64]),
    'vb_legacy_rng_create': (ctypes.c_int, [ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]),
    'vb_legacy_rng_destroy': (None, [ctypes.c_void_p]),
    'vb_legacy_rng_randn': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_int64, ctypes.c_int]),
    'vb_legacy_rng_standard_t': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, _c_double_p, ctypes.c_int64]),
    'vb_legacy_rng_chisquare': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, _c_double_p, ctypes.c_int64]),
    'vb_legacy_rng_random_sample': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_int64]),
    'vb_legacy_rng_get_state': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_int),
                                               ctypes.POINTER(ctypes.c_int), _c_double_p]),
    'vb_legacy_rng_set_state': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, ctypes.c_int,
                                               ctypes.c_double]),
    'vb_legacy_rng_randn_device': (ctypes.c_int, [_ctx_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                                  ctypes.c_int64, ctypes.c_int64]),
    'vb_legacy_rng_standard_t_device': (ctypes.c_int, [_ctx_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_int64,
                                                       ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]),
    'vb_legacy_rng_chisquare_device': (ctypes.c_int, [_ctx_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_int64, _c_double_p]),
    'vb_legacy_rng_log_proven': (ctypes.c_int, []),
    'vb_legacy_rng_uid': (ctypes.c_uint64, [ctypes.c_void_p]),
    'vb_comm_ipc_window': (ctypes.c_int, [_ctx_p, ctypes.c_size_t, ctypes.c_char_p]),
    'vb_comm_init_ipc': (ctypes.c_int, [_ctx_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]),
    'vb_comm_check': (ctypes.c_int, [_ctx_p]),
    'vb_legacy_round_end': (ctypes.c_int, [_ctx_p, ctypes.c_void_p]),
    'vb_legacy_ahead_stats': (ctypes.c_int, [_ctx_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                            ctypes.POINTER(ctypes.c_uint64)]),
    'vb_dis_state_park': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    'vb_dis_state_unpark': (ctypes.c_int, [_ctx_p, ctypes.c_void_p]),
    'vb_dis_state_drop': (ctypes.c_int, [ctypes.c_void_p]),
    'vb_fullrank_upload_stats': (ctypes.c_int, [_ctx_p, ctypes.POINTER(ctypes.c_uint64)]),
    'vb_host_alloc': (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    'vb_host_free': (ctypes.c_int, [ctypes.c_void_p]),
    'vb_comm_allreduce_time': (ctypes.c_int, [_ctx_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    'vb_comm_unique_id': (ctypes.c_int, [ctypes.c_char_p]),
    'vb_comm_init': (ctypes.c_int, [_ctx_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]),
    'vb_comm_init_host': (ctypes.c_int, [_ctx_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    'vb_comm_destroy': (ctypes.c_int, [_ctx_p]),
    'vb_comm_info': (ctypes.c_int, [_ctx_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    'vb_profile_enable': (ctypes.c_int, [_ctx_p, ctypes.c_int]),
    'vb_profile_read': (ctypes.c_int, [_ctx_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                       ctypes.POINTER(ctypes.c_double), ctypes.c_int]),
    'vb_profile_read_kernel': (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64),
                                              ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double),
                                              ctypes.c_int]),
}

_lib = None
_lib_lock = threading.Lock()


HOST_COLLECTIVE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_size_t,
                                      ctypes.c_int)


class EngineError(RuntimeError):
    """The HIP engine is unavailable or a HIP call failed."""


def build(verbose=False):
    """Compile ``libviabel_hip.so`` for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ['make', '-C', CSRC_DIR, '-j8']
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise EngineError('building libviabel_hip.so failed (see output above)')
    return LIB_PATH


def load():
    """Load the shared library and declare every entry point (no GPU needed for this)."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise EngineError(
                'libviabel_hip.so not found at %s: build it with '
                '`python -c "import __graft_entry__ as g; g.build()"` or `make -C viabel_amd/csrc`. '
                'There is no CPU fallback.' % LIB_PATH)
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as exc:
            raise EngineError('cannot load %s: %s' % (LIB_PATH, exc))
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
        return lib


_DPTR_CHECKS = bool(os.environ.get('VIABEL_AMD_CHECK_POINTERS'))


def _dptr(a):
    """Address of a contiguous float64 array (the caller keeps the array alive across the call: every call site
    passes a local NAME, never a temporary -- ``tests/test_host_logic.py::test_dptr_call_sites_pass_names`` enforces it,
    and ``VIABEL_AMD_CHECK_POINTERS=1`` checks dtype and contiguity on every call)."""
    if _DPTR_CHECKS:
        assert isinstance(a, np.ndarray) and a.flags.c_contiguous and a.dtype in (np.float64, np.int64, np.int32), \
            'C-ABI array argument must be a contiguous float64 / int array'
    return a.__array_interface__['data'][0]


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def device_count():
    """GPUs visible to HIP (0 without one)."""
    n = ctypes.c_int(0)
    return n.value if load().vb_device_count(ctypes.byref(n)) == VB_OK else 0


class PinnedPool:
    """Page-locked host blocks (``vb_host_alloc``) handed out as the float64 arrays the engine RETURNS.

    The reference returns a freshly allocated gradient per call (``objectives.py:32-44``); so does this binding, but a
    4.2-MB gradient copied into pageable memory is staged by the runtime (118 us at D = 1024: 36 GB/s), while the copy
    engine writes straight into a pinned block.  ``array(n)`` is an ordinary ``ndarray`` whose memory is such a block; when
    the array (and every view of it) is gone the block returns to the free list of its size and the next call reuses it --
    an optimiser loop that drops its gradients cycles through two or three blocks.  At most ``keep`` idle blocks per size
    stay allocated; arrays smaller than ``MIN_BYTES`` are plain ``np.empty`` (the small-result path copies through the
    engine's own mapped buffer anyway)."""

    MIN_BYTES = 1 << 20
    MAX_OUTSTANDING = 1 << 30      # page-locked bytes in the callers' hands; beyond it the arrays are pageable again (a caller
                                   # that KEEPS every gradient -- FASO's history -- must not pin the host's memory)

    def __init__(self, lib, keep=4):
        self._lib, self._keep, self._free, self._out = lib, keep, {}, 0

    def array(self, n):
        nbytes = 8 * int(n)
        if nbytes < self.MIN_BYTES or self._out + nbytes > self.MAX_OUTSTANDING:
            return np.empty(n, dtype=np.float64)
        free = self._free.setdefault(nbytes, [])
        if free:
            ptr = free.pop()
        else:
            box = ctypes.c_void_p()
            if self._lib.vb_host_alloc(nbytes, ctypes.byref(box)) != VB_OK or not box.value:
                return np.empty(n, dtype=np.float64)          # (no pinned memory left: a pageable array still works)
            ptr = box.value
        buf = (ctypes.c_double * int(n)).from_address(ptr)
        fin = weakref.finalize(buf, self._give_back, nbytes, ptr)     # buf is the array's base: it dies with the last view
        fin.atexit = False      # (at interpreter exit the HIP runtime may be gone before the arrays: the OS takes the pages back)
        self._out += nbytes
        return np.frombuffer(buf, dtype=np.float64)

    def _give_back(self, nbytes, ptr):
        self._out -= nbytes
        free = self._free.setdefault(nbytes, [])
        if len(free) < self._keep:
            free.append(ptr)
        else:
            self._lib.vb_host_free(ptr)

    def idle_blocks(self):
        return sum(len(v) for v in self._free.values())


_pinned_pool = None


def pinned_array(n):
    """A float64 result array of ``n`` entries in page-locked memory (see :class:`PinnedPool`)."""
    global _pinned_pool
    if _pinned_pool is None:
        _pinned_pool = PinnedPool(load())
    return _pinned_pool.array(n)


class Engine:
    """One HIP context on one GPU (``vb_ctx``).  Calls are synchronous unless named ``*_async``."""

    def __init__(self, device=None):
        self._lib = load()
        from_env = device is None
        if from_env:
            device = int(os.environ.get('LOCAL_RANK', '0'))
        n = ctypes.c_int(0)
        rc = self._lib.vb_device_count(ctypes.byref(n))
        if rc != VB_OK or n.value == 0:
            raise EngineError('no MI355X visible to HIP (%s); the engine has no CPU fallback'
                              % self._lib.vb_last_error(None).decode())
        if from_env and device >= n.value == 1 and any(
                os.environ.get(v) for v in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES')):
            # a launcher that masks the devices per rank: this rank's one visible GPU is its own
            device = 0
        ctx = _ctx_p()
        # no wrap-around: a rank whose LOCAL_RANK has no GPU must fail, not share another rank's device
        rc = self._lib.vb_create(int(device), ctypes.byref(ctx))
        if rc != VB_OK:
            raise EngineError('vb_create(device=%d) failed: %s (%d device(s) visible)'
                              % (int(device), self._lib.vb_last_error(None).decode(), n.value))
        self._ctx = ctx
        self.device = int(device)
        self._model_key = None
        self.n_ranks, self.rank = 1, 0

    def close(self):
        if getattr(self, '_ctx', None):
            self._lib.vb_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ errors
    def _check(self, rc):
        if rc == VB_OK:
            # IPC transport: a device-side wait that gave up has poisoned the results with NaN -- every call says so
            if getattr(self, '_ipc_on', False) and self._lib.vb_comm_check(self._ctx) != VB_OK:
                rc = VB_ERR_COMM
            else:
                return
        msg = self._lib.vb_last_error(self._ctx).decode()
        if rc == VB_ERR_CALLBACK:                    # a host model callable raised: hand its own exception on
            holder = getattr(self, '_callback_error', None)
            if holder and holder[0] is not None:
                exc, holder[0] = holder[0], None
                raise exc
        if rc == VB_ERR_INVALID or rc == VB_ERR_NUMERIC:
            raise ValueError(msg)
        if rc == VB_ERR_UNSUPPORTED:
            raise NotImplementedError(msg)
        raise EngineError(msg)

    # ------------------------------------------------------------------ info
    def device_info(self):
        name = ctypes.create_string_buffer(256)
        cu = ctypes.c_int(0)
        hbm = ctypes.c_uint64(0)
        self._check(self._lib.vb_device_info(self._ctx, name, 256, ctypes.byref(cu), ctypes.byref(hbm)))
        return {'name': name.value.decode(), 'cus': cu.value, 'hbm_bytes': hbm.value}

    def sync(self):
        self._check(self._lib.vb_sync(self._ctx))

    # ------------------------------------------------------------------ noise
    def noise_set_host(self, slot, eps):
        eps = _f64(eps)
        n, d = eps.shape
        self._check(self._lib.vb_noise_set_host(self._ctx, slot, _dptr(eps), n, d))

    def noise_hint_seed(self, slot_mask, seed, with_chi=False):
        """Tell the look-ahead generator the seed of the NEXT call's Philox requests (``vb_noise_hint_seed``): a
        prediction only -- a shadow that does not match the request is never adopted."""
        self._check(self._lib.vb_noise_hint_seed(self._ctx, int(slot_mask), 1 if with_chi else 0, int(seed)))

    def noise_ahead_stats(self):
        """``(generated, adopted)``: look-ahead buffers this engine has generated / requests that adopted one."""
        g, a = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self._lib.vb_noise_ahead_stats(self._ctx, ctypes.byref(g), ctypes.byref(a)))
        return int(g.value), int(a.value)

    def noise_generate(self, slot, n, d, seed, stream=0, row_offset=0, kind=NOISE_NORMAL, df=0.0):
        self._check(self._lib.vb_noise_generate(self._ctx, slot, kind, float(df), int(seed), int(stream),
                                                int(row_offset), int(n), int(d)))

    def chisq_generate(self, df, n, seed, stream=0, row_offset=0):
        """``n`` chi-square(df) draws on the device (kept in the context for ``dis_refresh_mvt(chi=None)``)."""
        self._check(self._lib.vb_chisq_generate(self._ctx, float(df), int(seed), int(stream), int(row_offset), int(n)))

    def chisq_get_host(self, n):
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.vb_chisq_get_host(self._ctx, _dptr(out), n))
        return out

    def noise_legacy_randn(self, slot, rng_handle, n_total, d, row_begin=0, rows=None):
        """Rows ``[row_begin, row_begin + rows)`` of ``RandomState.randn(n_total, d)`` generated ON THE DEVICE into
        ``slot``, bit for bit numpy's stream (``vb_legacy_rng_randn_device``); ``rng_handle`` is the ``vb_legacy_rng*``
        of a ``LegacyRandomState`` and is advanced as numpy would advance it.  Returns False when the request is outside
        the device path's range (the generator is untouched then: draw on the host and upload)."""
        rows = n_total - row_begin if rows is None else rows
        rc = self._lib.vb_legacy_rng_randn_device(self._ctx, rng_handle, slot, n_total, d, row_begin, rows)
        if rc == VB_ERR_UNSUPPORTED:
            return False
        self._check(rc)
        return True

    def noise_legacy_standard_t(self, slot, rng_handle, df, n_total, d, row_begin=0, rows=None):
        """Rows ``[row_begin, row_begin + rows)`` of ``RandomState.standard_t(df, (n_total, d))`` generated ON THE DEVICE
        into ``slot``, values and generator state bit for bit numpy's (``vb_legacy_rng_standard_t_device``).  False: the
        request is outside the device path's range, the generator is untouched."""
        rows = n_total - row_begin if rows is None else rows
        rc = self._lib.vb_legacy_rng_standard_t_device(self._ctx, rng_handle, float(df), slot, n_total, d, row_begin, rows)
        if rc == VB_ERR_UNSUPPORTED:
            return False
        self._check(rc)
        return True

    def chisq_legacy(self, rng_handle, df, n, to_host=True):
        """``RandomState.chisquare(df, n)`` generated ON THE DEVICE into the context's chi-square buffer (where
        ``chisq_generate`` puts the throughput mode's), bit for bit numpy's; returns the host copy (``to_host``), True, or
        None when the request is outside the device path's range (generator untouched)."""
        out = np.empty(n, dtype=np.float64) if to_host else None
        rc = self._lib.vb_legacy_rng_chisquare_device(self._ctx, rng_handle, float(df), n,
                                                      _dptr(out) if to_host else None)
        if rc == VB_ERR_UNSUPPORTED:
            return None
        self._check(rc)
        return out if to_host else True

    def legacy_round_end(self, rng_handle):
        """A call's device draws from ``rng_handle`` are done (``vb_legacy_round_end``): the engine may start the next call's
        -- the same requests, from the generator's current state -- beside the kernels that follow."""
        self._check(self._lib.vb_legacy_round_end(self._ctx, rng_handle))

    def legacy_ahead_stats(self):
        """``(launched, adopted, discarded)`` of the look-ahead generation of numpy's streams."""
        a, b, c = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self._lib.vb_legacy_ahead_stats(self._ctx, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def noise_get_host(self, slot, n, d):
        out = np.empty((n, d), dtype=np.float64)
        self._check(self._lib.vb_noise_get_host(self._ctx, slot, _dptr(out), n, d))
        return out

    # ------------------------------------------------------------------ model
    def set_model(self, spec):
        """``spec`` = (model_id, dim, dparams ndarray, iparams ndarray) from ``DeviceModel.device_spec``."""
        model_id, dim, dparams, iparams = spec[:4]
        # models hand out the same (cached) parameter arrays every time: identity is the cache key
        key = (model_id, dim, id(dparams), id(iparams)) + tuple(id(x) for x in spec[4:])
        if key == self._model_key:
            return
        self._model_arrays = tuple(spec[2:])         # keep them alive so the ids stay unique
        if model_id == MODEL_SOURCE and isinstance(spec[4], MODEL_CALLBACK_TYPE):
            # spec = (id, dim, -, -, ctypes callback, error holder): a host callable with its gradient
            self._check(self._lib.vb_set_model_callback(self._ctx, dim, ctypes.cast(spec[4], ctypes.c_void_p), None))
            self._callback_error = spec[5]
            self._model_key = key
            return
        if model_id == MODEL_SOURCE:                 # spec = (id, dim, params, iparams (unused), source bytes)
            params = _f64(dparams)
            self._check(self._lib.vb_set_model_source(self._ctx, dim, spec[4], _dptr(params) if params.size else None,
                                                      params.size))
            self._model_key = key
            return
        dparams = _f64(dparams)
        iparams = np.ascontiguousarray(iparams, dtype=np.int64)
        self._check(self._lib.vb_set_model(
            self._ctx, model_id, dim, _dptr(dparams) if dparams.size else None, dparams.size,
            iparams.ctypes.data_as(_c_int64_p) if iparams.size else None, iparams.size))
        self._model_key = key

    def model_logp(self, x):
        x = _f64(x)
        n, d = x.shape
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.vb_model_logp(self._ctx, _dptr(x), n, d, _dptr(out)))
        return out

    def model_grad(self, x):
        """(f(x_n), grad f(x_n)) of the bound model for host points x (N x D)."""
        x = _f64(x)
        n, d = x.shape
        f = np.empty(n, dtype=np.float64)
        g = np.empty((n, d), dtype=np.float64)
        self._check(self._lib.vb_model_grad(self._ctx, _dptr(x), n, d, _dptr(f), _dptr(g)))
        return f, g

    def elbo_sums_lowrank(self, slot_eps, slot_z, n, d, k, theta):
        """(sum f, sum g (d), sum g eps (d), sum g z' (d x k)) of the low-rank family's ExclusiveKL at any rank."""
        theta = _f64(theta)
        out = np.empty(1 + 2 * d + d * k, dtype=np.float64)
        self._check(self._lib.vb_elbo_sums_lowrank(self._ctx, slot_eps, slot_z, n, d, k, _dptr(theta), _dptr(out)))
        return out[0], out[1:1 + d], out[1 + d:1 + 2 * d], out[1 + 2 * d:].reshape(d, k)

    def elbo_grad_mvt_chol(self, slot, n, d, theta, df, n_total=None):
        """Throughput-mode ExclusiveKL of the multivariate t (Cholesky sampling, device chi-square draws)."""
        theta = _f64(theta)
        value = ctypes.c_double(0.0)
        grad = np.empty(d + d * (d + 1) // 2, dtype=np.float64)
        self._check(self._lib.vb_elbo_grad_mvt_chol(self._ctx, slot, n, d, n if n_total is None else n_total, float(df),
                                                    _dptr(theta), ctypes.byref(value), _dptr(grad)))
        return value.value, grad

    def noise_moments(self, slot, n, d, want_gram=False):
        """(sum_n eps_n (d), sum_n eps_n eps_n' (d x d) or None) of the first n rows of a noise slot, summed over the
        ranks of a sharded job."""
        colsum = np.empty(d, dtype=np.float64)
        gram = np.empty((d, d), dtype=np.float64) if want_gram else None
        self._check(self._lib.vb_noise_moments(self._ctx, slot, n, d, _dptr(colsum),
                                               _dptr(gram) if want_gram else None))
        return colsum, gram

    # ------------------------------------------------------------------ ExclusiveKL, mean field
    def elbo_grad_meanfield(self, slot, n, d, theta, family, df=0.0, flags=0, cv_mode=0, n_total=None):
        theta = _f64(theta)
        value = ctypes.c_double(0.0)
        grad = np.empty(2 * d, dtype=np.float64)
        self._check(self._lib.vb_elbo_grad_meanfield(
            self._ctx, slot, n, d, n if n_total is None else n_total, family, float(df), _dptr(theta),
            flags, cv_mode, ctypes.byref(value), _dptr(grad)))
        return value.value, grad

    def elbo_grad_meanfield_async(self, slot, n, d, theta, family, rslot, df=0.0, flags=0, cv_mode=0,
                                  n_total=None):
        theta = _f64(theta)
        self._check(self._lib.vb_elbo_grad_meanfield_async(
            self._ctx, slot, n, d, n if n_total is None else n_total, family, float(df), _dptr(theta),
            flags, cv_mode, rslot))

    def elbo_grad_meanfield_batch_async(self, slots, n, d, thetas, family, rslots, df=0.0, flags=0,
                                        cv_mode=0, n_total=None):
        """Enqueue ``len(slots)`` independent evaluations; ``thetas`` is ``(count, 2d)``."""
        thetas = _f64(thetas)
        count = len(slots)
        if thetas.shape != (count, 2 * d):
            raise ValueError('thetas must have shape (count, 2 * d)')
        c_slots = (ctypes.c_int * count)(*slots)
        c_rslots = (ctypes.c_int * count)(*rslots)
        self._check(self._lib.vb_elbo_grad_meanfield_batch_async(
            self._ctx, count, c_slots, n, d, n if n_total is None else n_total, family, float(df),
            _dptr(thetas), flags, cv_mode, c_rslots))

    def result_get(self, rslot, p):
        value = ctypes.c_double(0.0)
        grad = np.empty(p, dtype=np.float64)
        self._check(self._lib.vb_result_get(self._ctx, rslot, ctypes.byref(value), _dptr(grad), p))
        return value.value, grad

    # ------------------------------------------------------------------ ExclusiveKL, multivariate t
    def elbo_sums_mvt(self, slot, n, d, mu, sqrt_sigma, inv_s, n_total=None):
        """(sum f, sum g (D,), sum g (z / s)' (D, D)) over the samples x = mu + (z sqrt_sigma) / s."""
        mu, sqrt_sigma, inv_s = _f64(mu), _f64(sqrt_sigma), _f64(inv_s)
        f = ctypes.c_double(0.0)
        g = np.empty(d, dtype=np.float64)
        c = np.empty((d, d), dtype=np.float64)
        self._check(self._lib.vb_elbo_sums_mvt(self._ctx, slot, n, d, n if n_total is None else n_total,
                                               _dptr(mu), _dptr(sqrt_sigma), _dptr(inv_s), ctypes.byref(f),
                                               _dptr(g), _dptr(c)))
        return f.value, g, c

    def elbo_grad_meanfield_philox(self, slot, n, d, theta, family, seed, stream, df=0.0, flags=0, cv_mode=0,
                                   n_total=None, row_offset=0):
        """ExclusiveKL on fresh Philox noise generated inside the streaming kernel
        (``vb_elbo_grad_meanfield_philox``)."""
        theta = _f64(theta)
        value = ctypes.c_double(0.0)
        grad = np.empty(2 * d, dtype=np.float64)
        self._check(self._lib.vb_elbo_grad_meanfield_philox(
            self._ctx, slot, n, d, n if n_total is None else n_total, int(row_offset), family, float(df),
            _dptr(theta), flags, cv_mode, int(seed), int(stream), ctypes.byref(value), _dptr(grad)))
        return value.value, grad

    # ------------------------------------------------------------------ ExclusiveKL, low-rank Gaussian
    def elbo_grad_lowrank(self, slot_eps, slot_z, n, d, k, theta, flags=0, n_total=None):
        theta = _f64(theta)
        value = ctypes.c_double(0.0)
        grad = np.empty(2 * d + d * k, dtype=np.float64)
        self._check(self._lib.vb_elbo_grad_lowrank(self._ctx, slot_eps, slot_z, n, d, k,
                                                   n if n_total is None else n_total, _dptr(theta), flags,
                                                   ctypes.byref(value), _dptr(grad)))
        return value.value, grad

    # ------------------------------------------------------------------ importance weights, PSIS
    def log_weights_meanfield(self, slot, n, d, theta, family, df=0.0, fetch=True):
        """log p(z_n) - log q(z_n) for the staged noise; the weights also stay on the device for
        :meth:`psis_smooth`."""
        theta = _f64(theta)
        lw = np.empty(n, dtype=np.float64) if fetch else None
        self._check(self._lib.vb_log_weights_meanfield(self._ctx, slot, n, d, family, float(df), _dptr(theta),
                                                       _dptr(lw) if fetch else None))
        return lw

    def psis_smooth(self, n, log_weights=None, reff=1.0):
        """Pareto-smoothed, normalised log weights and k-hat; ``log_weights=None`` smooths the weights
        left on the device by :meth:`log_weights_meanfield`."""
        out = np.empty(n, dtype=np.float64)
        khat = ctypes.c_double(0.0)
        src = None
        if log_weights is not None:
            log_weights = _f64(log_weights)
            if log_weights.shape != (n,):
                raise ValueError('log_weights must have shape ({},)'.format(n))
            src = _dptr(log_weights)
        self._check(self._lib.vb_psis_smooth(self._ctx, src, n, float(reff), _dptr(out), ctypes.byref(khat)))
        return out, khat.value

    # ------------------------------------------------------------------ AlphaDivergence, mean field
    def alpha_grad_meanfield(self, slot, n, d, theta, family, alpha, df=0.0, n_total=None):
        theta = _f64(theta)
        value = ctypes.c_double(0.0)
        grad = np.empty(2 * d, dtype=np.float64)
        self._check(self._lib.vb_alpha_grad_meanfield(self._ctx, slot, n, d, n if n_total is None else n_total,
                                                      family, float(df), _dptr(theta),
                                                      float(alpha), ctypes.byref(value), _dptr(grad)))
        return value.value, grad

    # ------------------------------------------------------------------ DISInclusiveKL
    def dis_set_temper_prior(self, spec):
        """Install a tempering prior that is not an MFGaussian (``vb_dis_set_temper_prior``); ``spec`` =
        ``(kind, df, loc, scale, log_det_l)`` or ``None`` for the refresh calls' own ``prior_theta`` argument.  The
        spec object's identity is the cache key: objectives hand out the same tuple every time."""
        if spec is getattr(self, '_temper_spec', None):
            return
        if spec is None:
            self._check(self._lib.vb_dis_set_temper_prior(self._ctx, PRIOR_DIAG_GAUSSIAN, 0, 0.0, None, None, 0.0))
        else:
            kind, df, loc, scale, log_det_l = spec
            loc, scale = _f64(loc), _f64(scale)
            self._check(self._lib.vb_dis_set_temper_prior(self._ctx, int(kind), loc.size, float(df), _dptr(loc),
                                                          _dptr(scale), float(log_det_l)))
        self._temper_spec = spec

    def dis_refresh_meanfield(self, slot, n, d, theta, prior_theta, family, eps_prev, ess_target,
                              max_bisection_its=50, df=0.0, n_total=None):
        """``n`` local rows; the returned weights / log p / log q cover all ``n_total`` samples."""
        theta, prior_theta = _f64(theta), _f64(prior_theta)
        n_total = n if n_total is None else n_total
        eps, ess = ctypes.c_double(0.0), ctypes.c_double(0.0)
        w, lp, lq = (np.empty(n_total, dtype=np.float64) for _ in range(3))
        self._check(self._lib.vb_dis_refresh_meanfield(
            self._ctx, slot, n, d, n_total, family, float(df), _dptr(theta), _dptr(prior_theta), float(eps_prev),
            float(ess_target), int(max_bisection_its), ctypes.byref(eps), ctypes.byref(ess), _dptr(w),
            _dptr(lp), _dptr(lq)))
        return eps.value, ess.value, w, lp, lq

    def dis_grad_meanfield(self, slot, n, d, theta, weights, scale, family, df=0.0):
        theta, weights = _f64(theta), _f64(weights)
        value = ctypes.c_double(0.0)
        grad = np.empty(2 * d, dtype=np.float64)
        self._check(self._lib.vb_dis_grad_meanfield(self._ctx, slot, n, d, family, float(df), _dptr(theta),
                                                    _dptr(weights), float(scale), ctypes.byref(value),
                                                    _dptr(grad)))
        return value.value, grad

    # ------------------------------------------------------------------ DISInclusiveKL, multivariate t
    # ------------------------------------------------------------------ low-rank Gaussian under DIS / alpha
    def dis_refresh_lowrank(self, slot_eps, slot_z, n, d, k, mu, log_sigma, B, m_inv, log_q_const, prior_theta,
                            eps_prev, ess_target, max_bisection_its=50, n_total=None):
        mu, log_sigma, B, m_inv, prior_theta = (_f64(a) for a in (mu, log_sigma, B, m_inv, prior_theta))
        n_total = n if n_total is None else n_total
        eps, ess = ctypes.c_double(0.0), ctypes.c_double(0.0)
        w, lp, lq = (np.empty(n_total, dtype=np.float64) for _ in range(3))
        self._check(self._lib.vb_dis_refresh_lowrank(
            self._ctx, slot_eps, slot_z, n, d, k, n_total, _dptr(mu), _dptr(log_sigma), _dptr(B), _dptr(m_inv),
            float(log_q_const), _dptr(prior_theta), float(eps_prev), float(ess_target), int(max_bisection_its),
            ctypes.byref(eps), ctypes.byref(ess), _dptr(w), _dptr(lp), _dptr(lq)))
        return eps.value, ess.value, w, lp, lq

    def dis_grad_lowrank(self, n, d, k, mu, log_sigma, B, m_inv, log_q_const, weights):
        """``(sum w rho tau' (d, k), sum w tau tau' (k, k), sum w rho, sum w rho^2, sum w tau, sum w, sum w log q)``."""
        mu, log_sigma, B, m_inv, weights = (_f64(a) for a in (mu, log_sigma, B, m_inv, weights))
        out = np.empty(d * k + k * k + 2 * d + k + 2, dtype=np.float64)
        self._check(self._lib.vb_dis_grad_lowrank(self._ctx, n, d, k, _dptr(mu), _dptr(log_sigma), _dptr(B),
                                                  _dptr(m_inv), float(log_q_const), _dptr(weights), _dptr(out)))
        a, b = d * k, d * k + k * k
        return (out[:a].reshape(d, k), out[a:b].reshape(k, k), out[b:b + d], out[b + d:b + 2 * d],
                out[b + 2 * d:b + 2 * d + k], out[-2], out[-1])

    def alpha_sums_lowrank(self, slot_eps, slot_z, n, d, k, alpha, mu, log_sigma, B, m_inv, log_q_const, n_total=None):
        """``(value, sum s, sum s g z' (d, k), sum s eps t' (d, k), sum s t t' (k, k), sum s g, sum s g eps)``."""
        mu, log_sigma, B, m_inv = (_f64(a) for a in (mu, log_sigma, B, m_inv))
        out = np.empty(2 * d * k + k * k + 2 * d, dtype=np.float64)
        value, w_sum = ctypes.c_double(0.0), ctypes.c_double(0.0)
        self._check(self._lib.vb_alpha_sums_lowrank(
            self._ctx, slot_eps, slot_z, n, d, k, n if n_total is None else n_total, float(alpha), _dptr(mu),
            _dptr(log_sigma), _dptr(B), _dptr(m_inv), float(log_q_const), ctypes.byref(value), ctypes.byref(w_sum),
            _dptr(out)))
        a, b, c = d * k, 2 * d * k, 2 * d * k + k * k
        return (value.value, w_sum.value, out[:a].reshape(d, k), out[a:b].reshape(d, k), out[b:c].reshape(k, k),
                out[c:c + d], out[c + d:])

    def dis_generation(self, kind):
        """Refresh counter of the DIS state of one family kind (0 mean-field, 1 dense, 2 low-rank)."""
        g = ctypes.c_uint64(0)
        self._check(self._lib.vb_dis_generation(self._ctx, int(kind), ctypes.byref(g)))
        return g.value

    def dis_state_park(self, kind, slot=-1):
        """Detach the context's DIS state of ``kind`` (and noise slot ``slot``) into a handle (``vb_dis_state_park``)."""
        h = ctypes.c_void_p()
        self._check(self._lib.vb_dis_state_park(self._ctx, int(kind), int(slot), ctypes.byref(h)))
        return h

    def dis_state_unpark(self, handle):
        self._check(self._lib.vb_dis_state_unpark(self._ctx, handle))

    def dis_state_drop(self, handle):
        self._lib.vb_dis_state_drop(handle)

    def dis_state_get(self, dense, n_total):
        """(log p, log q) of the state samples of the last DIS refresh."""
        lp, lq = np.empty(n_total, dtype=np.float64), np.empty(n_total, dtype=np.float64)
        self._check(self._lib.vb_dis_state_get(self._ctx, int(bool(dense)), _dptr(lp), _dptr(lq), n_total))
        return lp, lq

    def dis_refresh_mvt(self, slot, n, d, df, theta, chi, sqrt_sigma, l_inv, prior_theta, eps_prev, ess_target,
                        max_bisection_its=50, n_total=None, fetch_logs=True):
        theta, prior_theta = _f64(theta), _f64(prior_theta)
        sqrt_sigma = None if sqrt_sigma is None else _f64(sqrt_sigma)   # both None: factors from theta on the device
        l_inv = None if l_inv is None else _f64(l_inv)
        chi = None if chi is None else _f64(chi)       # None: the device draws of chisq_generate
        n_total = n if n_total is None else n_total
        eps, ess = ctypes.c_double(0.0), ctypes.c_double(0.0)
        w = np.empty(n_total, dtype=np.float64)
        lp, lq = ((np.empty(n_total, dtype=np.float64), np.empty(n_total, dtype=np.float64)) if fetch_logs
                  else (None, None))
        self._check(self._lib.vb_dis_refresh_mvt(
            self._ctx, slot, n, d, n_total, float(df), _dptr(theta), None if chi is None else _dptr(chi),
            None if sqrt_sigma is None else _dptr(sqrt_sigma), None if l_inv is None else _dptr(l_inv),
            _dptr(prior_theta), float(eps_prev), float(ess_target), int(max_bisection_its), ctypes.byref(eps),
            ctypes.byref(ess), _dptr(w), None if lp is None else _dptr(lp), None if lq is None else _dptr(lq)))
        return eps.value, ess.value, w, lp, lq

    def dis_refresh_mvt_deferred(self, slot, n, d, df, theta, prior_theta, eps_prev, ess_target, max_bisection_its=50,
                                 n_total=None):
        """Throughput-mode refresh that only ENQUEUES: samples, log p / log q, tempering bisection and the weights stay
        on the device; ``dis_step_mvt_packed`` reads them and synchronises once.  Sharded jobs: ``n`` rows of ``n_total``
        (the three per-sample vectors are gathered on the device, the weights formed redundantly on every rank)."""
        theta, prior_theta = _f64(theta), _f64(prior_theta)
        self._check(self._lib.vb_dis_refresh_mvt(
            self._ctx, slot, n, d, n if n_total is None else n_total, float(df), _dptr(theta), None, None, None,
            _dptr(prior_theta), float(eps_prev), float(ess_target), int(max_bisection_its), None, None, None, None, None))

    def elbo_grad_mvt_symroot(self, slot, n, d, df, theta, path_deriv=False, n_total=None):
        """``(value, grad)`` of the t family's ExclusiveKL in the reference-identical mode, resident on the device
        (``vb_elbo_grad_mvt_symroot`` / ``_path``: symmetric root and its Frechet derivative by device iterations), or None
        when an iteration did not resolve (take the host route)."""
        theta = _f64(theta)
        value = ctypes.c_double(0.0)
        grad = np.empty(d + d * (d + 1) // 2, dtype=np.float64)
        info = np.zeros(4, dtype=np.float64)
        fn = self._lib.vb_elbo_grad_mvt_symroot_path if path_deriv else self._lib.vb_elbo_grad_mvt_symroot
        rc = fn(self._ctx, slot, n, d, n if n_total is None else n_total, float(df), _dptr(theta), ctypes.byref(value),
                _dptr(grad), _dptr(info))
        if rc == VB_ERR_UNSUPPORTED:
            return None
        self._check(rc)
        self.last_root_info = info
        return value.value, grad

    def alpha_grad_mvt_symroot(self, slot, n, d, df, alpha, theta, n_total=None):
        """``(value, grad)`` of the t family's AlphaDivergence in the reference-identical mode, resident on the device
        (``vb_alpha_grad_mvt_symroot``), or None when a root iteration did not resolve (take the host route)."""
        theta = _f64(theta)
        value = ctypes.c_double(0.0)
        grad = np.empty(d + d * (d + 1) // 2, dtype=np.float64)
        info = np.zeros(4, dtype=np.float64)
        rc = self._lib.vb_alpha_grad_mvt_symroot(self._ctx, slot, n, d, n if n_total is None else n_total, float(df),
                                                 float(alpha), _dptr(theta), ctypes.byref(value), _dptr(grad), _dptr(info))
        if rc == VB_ERR_UNSUPPORTED:
            return None
        self._check(rc)
        self.last_root_info = info
        return value.value, grad

    def dis_refresh_mvt_symroot(self, slot, n, d, df, theta, prior_theta, eps_prev, ess_target, max_bisection_its=50,
                                n_total=None):
        """The reference-identical refresh resident on the device (``vb_dis_refresh_mvt_symroot``): the noise in ``slot``
        and the context's chi-square draws are numpy's stream, the samples go through the symmetric root of ``Sigma``
        formed on the device.  Returns ``info = [steps, residual, accuracy]`` of the root, or None when the iteration did
        not resolve it (nothing was installed: take the host route)."""
        theta, prior_theta = _f64(theta), _f64(prior_theta)
        info = np.zeros(3, dtype=np.float64)
        rc = self._lib.vb_dis_refresh_mvt_symroot(self._ctx, slot, n, d, n if n_total is None else n_total, float(df),
                                                  _dptr(theta), _dptr(prior_theta), float(eps_prev), float(ess_target),
                                                  int(max_bisection_its), _dptr(info))
        if rc == VB_ERR_UNSUPPORTED:
            return None
        self._check(rc)
        return info

    def dis_step_mvt_packed(self, n, d, df, theta, scale, resample_m=0, seed=0, stream=0):
        """``(value, grad, eps, ess)``: gradient of ``-scale sum w log q`` on the device-resident tempered weights, or
        on ``resample_m`` multinomial draws from them (then ``scale`` multiplies ``sum w`` on the device)."""
        theta = _f64(theta)
        value, eps, ess, khat = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_double(0.0)
        grad = np.empty(theta.size, dtype=np.float64)
        self._check(self._lib.vb_dis_step_mvt_packed(
            self._ctx, n, d, float(df), _dptr(theta), int(resample_m), int(seed), int(stream), float(scale),
            ctypes.byref(eps), ctypes.byref(ess), ctypes.byref(khat), ctypes.byref(value), _dptr(grad)))
        self.last_khat = khat.value          # tail shape of the last dis_psis_mvt (0.0 without one)
        return value.value, grad, eps.value, ess.value

    def dis_psis_mvt(self, n_total, reff=1.0):
        """Enqueue the Pareto smoothing of the device-resident tempered weights (between ``dis_refresh_mvt_deferred``
        and ``dis_step_mvt_packed``; ``last_khat`` after the step)."""
        self._check(self._lib.vb_dis_psis_mvt(self._ctx, int(n_total), float(reff)))

    def dis_clip_mvt(self, n_total, threshold):
        """Enqueue the weight clipping (``objectives.py:370-386``) of the device-resident weights, in place."""
        self._check(self._lib.vb_dis_clip_mvt(self._ctx, int(n_total), float(threshold)))

    def dis_scalars_get(self):
        """``(eps, ess, khat)`` of the last device-resident refresh."""
        out = np.zeros(4, dtype=np.float64)
        self._check(self._lib.vb_dis_scalars_get(self._ctx, _dptr(out)))
        return out[0], out[1], out[3]

    def dis_weights_get(self, n_total, resampled=False):
        w = np.empty(n_total, dtype=np.float64)
        self._check(self._lib.vb_dis_weights_get(self._ctx, _dptr(w), n_total, 1 if resampled else 0))
        return w

    def dis_grad_mvt_packed(self, n, d, df, theta, weights, scale):
        """``(value, grad)`` of ``-scale sum_n w_n log q(x_n; theta)`` with the factor algebra and the chain rule on
        the device (``vb_dis_grad_mvt_packed``)."""
        theta, weights = _f64(theta), _f64(weights)
        value = ctypes.c_double(0.0)
        grad = np.empty(theta.size, dtype=np.float64)
        self._check(self._lib.vb_dis_grad_mvt_packed(self._ctx, n, d, float(df), _dptr(theta), _dptr(weights),
                                                     float(scale), ctypes.byref(value), _dptr(grad)))
        return value.value, grad

    def dis_grad_mvt(self, n, d, df, theta, l_inv, weights):
        theta, l_inv, weights = _f64(theta), _f64(l_inv), _f64(weights)
        w_sum, w_logq = ctypes.c_double(0.0), ctypes.c_double(0.0)
        d_mu = np.empty(d, dtype=np.float64)
        gram = np.empty((d, d), dtype=np.float64)
        self._check(self._lib.vb_dis_grad_mvt(self._ctx, n, d, float(df), _dptr(theta), _dptr(l_inv),
                                              _dptr(weights), ctypes.byref(w_sum), ctypes.byref(w_logq),
                                              _dptr(d_mu), _dptr(gram)))
        return w_sum.value, w_logq.value, d_mu, gram

    # ------------------------------------------------------------------ ExclusiveKL, full rank
    def elbo_grad_fullrank(self, slot, n, d, theta, flags=0, n_total=None):
        theta = _f64(theta)
        p = d + d * (d + 1) // 2
        value = ctypes.c_double(0.0)
        grad = pinned_array(p)
        self._check(self._lib.vb_elbo_grad_fullrank(
            self._ctx, slot, n, d, n if n_total is None else n_total, _dptr(theta), flags,
            ctypes.byref(value), _dptr(grad)))
        return value.value, grad

    def fullrank_upload_stats(self):
        """How many blocking full-rank calls took the pipelined parameter upload (``vb_fullrank_upload_stats``)."""
        n = ctypes.c_uint64(0)
        self._check(self._lib.vb_fullrank_upload_stats(self._ctx, ctypes.byref(n)))
        return n.value

    def fullrank_set_theta(self, theta, d):
        theta = _f64(theta)
        self._check(self._lib.vb_fullrank_set_theta(self._ctx, _dptr(theta), d))

    def elbo_grad_fullrank_enqueue(self, slot, n, d, flags=0, n_total=None):
        self._check(self._lib.vb_elbo_grad_fullrank_enqueue(
            self._ctx, slot, n, d, n if n_total is None else n_total, flags))

    def fullrank_get(self, d):
        p = d + d * (d + 1) // 2
        value = ctypes.c_double(0.0)
        grad = pinned_array(p)
        self._check(self._lib.vb_fullrank_get(self._ctx, ctypes.byref(value), _dptr(grad), p))
        return value.value, grad

    # ------------------------------------------------------------------ small dense linear algebra
    def sym_sqrt(self, a, e=None):
        """Symmetric square root of an SPD matrix on the device (``vb_sym_sqrt``); with ``e`` also the solution
        ``x`` of ``root x + x root = e``.  Returns ``(root, x or None, info)`` with
        ``info = [iterations, final residual, ||root root - a|| / ||a||]``."""
        a = _f64(a)
        d = a.shape[0]
        root = np.empty((d, d), dtype=np.float64)
        info = np.zeros(3, dtype=np.float64)
        if e is None:
            self._check(self._lib.vb_sym_sqrt(self._ctx, _dptr(a), None, d, _dptr(root), None, _dptr(info)))
            return root, None, info
        e = _f64(e)
        x = np.empty((d, d), dtype=np.float64)
        self._check(self._lib.vb_sym_sqrt(self._ctx, _dptr(a), _dptr(e), d, _dptr(root), _dptr(x), _dptr(info)))
        return root, x, info

    def alpha_grad_mvt_chol(self, slot, n, d, df, theta, alpha, n_total=None):
        """AlphaDivergence of the multivariate t in throughput mode (Cholesky sampling, device chi-square draws): value
        and the gradient in the flat layout, nothing of order D^2 on the host."""
        theta = _f64(theta)
        value = ctypes.c_double(0.0)
        grad = np.empty(d + d * (d + 1) // 2, dtype=np.float64)
        self._check(self._lib.vb_alpha_grad_mvt_chol(self._ctx, slot, n, d, n if n_total is None else n_total, float(df),
                                                     _dptr(theta), float(alpha), ctypes.byref(value), _dptr(grad)))
        return value.value, grad

    def alpha_grad_fullrank(self, slot, n, d, theta, alpha, n_total=None):
        theta = _f64(theta)
        p = d + d * (d + 1) // 2
        value = ctypes.c_double(0.0)
        grad = np.empty(p, dtype=np.float64)
        self._check(self._lib.vb_alpha_grad_fullrank(self._ctx, slot, n, d, n if n_total is None else n_total,
                                                     _dptr(theta), float(alpha), ctypes.byref(value), _dptr(grad)))
        return value.value, grad

    def sym_sqrt_inv(self, a):
        """``(root, inverse root, info)`` of an SPD matrix (``vb_sym_sqrt_inv``)."""
        a = _f64(a)
        d = a.shape[0]
        root, inv_root = np.empty((d, d), dtype=np.float64), np.empty((d, d), dtype=np.float64)
        info = np.zeros(3, dtype=np.float64)
        self._check(self._lib.vb_sym_sqrt_inv(self._ctx, _dptr(a), None, d, _dptr(root), None, _dptr(info),
                                              _dptr(inv_root)))
        return root, inv_root, info

    def lowrank_path_terms(self, slot_eps, slot_z, n, d, k, sw, n_total=None):
        """Noise-only second moments of the low-rank family's path-derivative estimator (``vb_lowrank_path_terms``):
        ``(E'T (d, 2k), T'T (2k, 2k), sum eps (d), sum eps^2 (d), sum T (2k))`` with ``T = [z | eps sw]``."""
        sw = _f64(sw)
        out = np.empty(d * 2 * k + 4 * k * k + 2 * d + 2 * k, dtype=np.float64)
        self._check(self._lib.vb_lowrank_path_terms(self._ctx, slot_eps, slot_z, n, d, k,
                                                    n if n_total is None else n_total, _dptr(sw), _dptr(out)))
        a, b = d * 2 * k, d * 2 * k + 4 * k * k
        return (out[:a].reshape(d, 2 * k), out[a:b].reshape(2 * k, 2 * k), out[b:b + d], out[b + d:b + 2 * d],
                out[b + 2 * d:])

    def mvt_path_terms(self, slot, n, d, df, inv_s, n_total=None):
        """Noise-only sums of the t family's path-derivative estimator (``vb_mvt_path_terms``):
        ``(m_w, e_w, log1p_sum)``."""
        inv_s = _f64(inv_s)
        m_w = np.empty((d, d), dtype=np.float64)
        e_w = np.empty(d, dtype=np.float64)
        l1p = ctypes.c_double(0.0)
        self._check(self._lib.vb_mvt_path_terms(self._ctx, slot, n, d, n if n_total is None else n_total, float(df),
                                                _dptr(inv_s), _dptr(m_w), _dptr(e_w), ctypes.byref(l1p)))
        return m_w, e_w, l1p.value

    def alpha_sums_mvt(self, slot, n, d, df, alpha, mu, root, inv_s, sum_log_diag, n_total=None):
        """Weighted sample sums of AlphaDivergence over a multivariate t (``vb_alpha_sums_mvt``):
        ``(value, w_sum, g_sum, C)``."""
        mu, root = _f64(mu), _f64(root)
        inv_s = None if inv_s is None else _f64(inv_s)      # None: the device chi-square draws (chisq_generate)
        g_sum = np.empty(d, dtype=np.float64)
        C = np.empty((d, d), dtype=np.float64)
        value, w_sum = ctypes.c_double(0.0), ctypes.c_double(0.0)
        self._check(self._lib.vb_alpha_sums_mvt(self._ctx, slot, n, d, n if n_total is None else n_total, float(df),
                                                float(alpha), _dptr(mu), _dptr(root),
                                                None if inv_s is None else _dptr(inv_s),
                                                float(sum_log_diag), ctypes.byref(value), ctypes.byref(w_sum),
                                                _dptr(g_sum), _dptr(C)))
        return value.value, w_sum.value, g_sum, C

    # ------------------------------------------------------------------ device-resident fit
    def fit(self, slot, n, d, family, theta, n_iters, opt_kind, hyper, *, df=0.0, flags=0, cv_mode=0,
            n_total=None, row_offset=0, noise_kind=NOISE_NORMAL, noise_df=0.0, seed=1, first_stream=0,
            state=None, hist_len=0, log_directions=False, log_gradients=False, slot_aux=-1):
        """``n_iters`` iterations of {Philox noise -> objective -> optimiser step} enqueued back to back
        (``vb_fit``).  Returns (theta, values, history, state, directions or None, gradients or None)."""
        theta = _f64(theta).copy()
        p = theta.size
        hyper = _f64(np.asarray(hyper, dtype=np.float64))
        has_state = state is not None
        state = _f64(state).copy() if has_state else np.zeros(2 * p, dtype=np.float64)
        values = np.empty(n_iters, dtype=np.float64)
        history = np.empty((hist_len, p), dtype=np.float64)
        directions = np.empty((n_iters, p), dtype=np.float64) if log_directions else None
        gradients = np.empty((n_iters, p), dtype=np.float64) if log_gradients else None
        self._check(self._lib.vb_fit(
            self._ctx, slot, slot_aux, n, d, n if n_total is None else n_total, int(row_offset), family, float(df), flags,
            cv_mode, noise_kind, float(noise_df), int(seed), int(first_stream), opt_kind, _dptr(hyper), int(n_iters),
            _dptr(theta), p, _dptr(state), int(has_state), _dptr(values),
            _dptr(history) if hist_len else None, int(hist_len),
            _dptr(directions) if log_directions else None, _dptr(gradients) if log_gradients else None))
        return theta, values, history, state, directions, gradients

    def mvt_route_stats(self):
        """(refreshes that took log p / log prior from the sampling product's epilogue, steps whose chain-rule kernel stored the
        gradient into the mapped result buffer itself) -- ``vb_mvt_route_stats``."""
        a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self._lib.vb_mvt_route_stats(self._ctx, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def fit_history_mean(self, rows, p):
        """``np.mean(history[-rows:], axis=0)`` of the iterates the last ``fit`` kept, formed on the device from the rows still
        resident there (``vb_fit_history_mean``): the same additions in the same order, without the host's pass over them."""
        mean = pinned_array(p)
        self._check(self._lib.vb_fit_history_mean(self._ctx, int(rows), int(p), _dptr(mean)))
        return mean

    # ------------------------------------------------------------------ multi-GPU
    @staticmethod
    def comm_unique_id():
        lib = load()
        buf = ctypes.create_string_buffer(COMM_ID_BYTES)
        rc = lib.vb_comm_unique_id(buf)
        if rc != VB_OK:
            raise EngineError('vb_comm_unique_id failed: ' + lib.vb_last_error(None).decode())
        return buf.raw

    def comm_init(self, unique_id, n_ranks, rank):
        self._check(self._lib.vb_comm_init(self._ctx, unique_id, n_ranks, rank))
        self.n_ranks, self.rank = n_ranks, rank

    IPC_HANDLE_BYTES = 64

    def comm_ipc_window(self, cap_doubles=1 << 21):
        """Allocate this rank's window of the xGMI-native transport; returns its IPC handle (64 bytes)."""
        buf = ctypes.create_string_buffer(self.IPC_HANDLE_BYTES)
        self._check(self._lib.vb_comm_ipc_window(self._ctx, int(cap_doubles), buf))
        return buf.raw

    def comm_init_ipc(self, handles, n_ranks, rank):
        """``handles``: the ranks' window handles in rank order (``n_ranks * 64`` bytes)."""
        blob = b''.join(handles)
        if len(blob) != n_ranks * self.IPC_HANDLE_BYTES:
            raise ValueError('expected %d window handles of %d bytes' % (n_ranks, self.IPC_HANDLE_BYTES))
        self._check(self._lib.vb_comm_init_ipc(self._ctx, blob, n_ranks, rank))
        self.n_ranks, self.rank = n_ranks, rank
        self._ipc_on = True

    def comm_init_host(self, collective, n_ranks, rank):
        """Host-staged transport (``vb_comm_init_host``): ``collective(array, op)`` must overwrite the float64
        ``array`` in place with the ranks' elementwise sum (``op == 0``) or maximum (``op == 1``)."""
        def trampoline(_user, buf, count, op):
            try:
                collective(np.ctypeslib.as_array(buf, shape=(count,)), op)
                return 0
            except Exception:                # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        cb = HOST_COLLECTIVE_FN(trampoline)
        self._check(self._lib.vb_comm_init_host(self._ctx, ctypes.cast(cb, ctypes.c_void_p), None, n_ranks, rank))
        self._host_collective = cb           # the library keeps the pointer: keep the thunk alive with the engine
        self.n_ranks, self.rank = n_ranks, rank

    def comm_destroy(self):
        self._ipc_on = False
        self._check(self._lib.vb_comm_destroy(self._ctx))
        self._host_collective = None
        self.n_ranks, self.rank = 1, 0

    def comm_allreduce_time(self, count, warm=5, reps=50):
        """Microseconds per sum all-reduce of ``count`` doubles, the collective alone (``vb_comm_allreduce_time``; every
        rank calls it)."""
        us = ctypes.c_double(0.0)
        self._check(self._lib.vb_comm_allreduce_time(self._ctx, int(count), int(warm), int(reps), ctypes.byref(us)))
        return us.value

    def comm_info(self):
        """(ranks, rank) as the RCCL communicator reports them; (1, 0) without a communicator."""
        n, r = ctypes.c_int(0), ctypes.c_int(0)
        self._check(self._lib.vb_comm_info(self._ctx, ctypes.byref(n), ctypes.byref(r)))
        return n.value, r.value

    # ------------------------------------------------------------------ measurement
    def profile_enable(self, on=True):
        self._check(self._lib.vb_profile_enable(self._ctx, int(bool(on))))

    def profile_read(self, reset=True, kernel=PROF_MF_ACCUM):
        """(launches, evaluations, total ms) of one profiled kernel (``PROF_*``) since the last reset."""
        n = ctypes.c_int64(0)
        ev = ctypes.c_int64(0)
        ms = ctypes.c_double(0.0)
        self._check(self._lib.vb_profile_read_kernel(self._ctx, int(kernel), ctypes.byref(n), ctypes.byref(ev),
                                                     ctypes.byref(ms), int(reset)))
        return n.value, ev.value, ms.value


_default_engine = None


_blas_controller = None


def small_lapack(dim):
    """Context manager for the O(D^3) host factorisations that stay on the CPU (``eigh`` of the D x D scale for
    the symmetric root, ``approximations.py:348``): runs them on one BLAS thread when the matrix is small.
    OpenBLAS's threaded ``dsyevd`` is an order of magnitude SLOWER than the serial one at these sizes (256 x 256:
    94 vs 8 ms on 8 cores, 70-95 vs 5 ms on the 256-core GPU host), which made the whole MultivariateT objective
    call host-bound.  No-op when ``threadpoolctl`` is unavailable or ``dim`` > 1024."""
    global _blas_controller
    import contextlib
    if dim > 1024:
        return contextlib.nullcontext()
    if _blas_controller is None:
        try:
            from threadpoolctl import ThreadpoolController
            _blas_controller = ThreadpoolController()
        except Exception:           # pragma: no cover
            _blas_controller = False
    if not _blas_controller:
        return contextlib.nullcontext()
    return _blas_controller.limit(limits=1, user_api='blas')


_blas_sticky = None


def set_host_blas_threads(n=1):
    """Limit the BLAS thread pool of this process (sticky, via ``threadpoolctl``); returns True when applied.

    The host side of the dense-covariance objectives is a handful of D x D numpy products per call.  A threaded
    OpenBLAS leaves its worker threads spinning for ~30 ms after each of them (``OPENBLAS_THREAD_TIMEOUT``);
    inside a CPU-quota cgroup (the MI355X boxes of this pool: 256 visible cores, 64 BLAS threads) the spinning
    exhausts the quota and every few objective calls stall for 30-85 ms -- a 4.6 ms MultivariateT / DIS call
    then averages 25 ms.  One or a few threads are the right setting for matrices of this size;
    ``OPENBLAS_NUM_THREADS=1`` in the environment does the same."""
    global _blas_sticky
    try:
        from threadpoolctl import threadpool_limits
    except Exception:               # pragma: no cover
        return False
    _blas_sticky = threadpool_limits(limits=int(n), user_api='blas')
    return True


_blas_policy_done = False


def apply_host_blas_policy():
    """Called once by the dense-covariance objectives before their first D x D host product: limits the BLAS pool
    to ``VIABEL_AMD_HOST_BLAS_THREADS`` threads (default 1; ``0`` leaves the pool alone) -- see
    ``set_host_blas_threads`` for why."""
    global _blas_policy_done
    if _blas_policy_done:
        return
    _blas_policy_done = True
    try:
        n = int(os.environ.get('VIABEL_AMD_HOST_BLAS_THREADS', '1'))
    except ValueError:
        n = 1
    if n > 0 and _blas_sticky is None:
        set_host_blas_threads(n)


def default_engine():
    """Process-wide engine on device ``LOCAL_RANK`` (0 if unset); created on first use."""
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine()
    return _default_engine


def set_default_engine(engine):
    global _default_engine
    _default_engine = engine
