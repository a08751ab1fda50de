/*
 * viabel_hip.h -- C ABI of the MI355X (gfx950) BBVI gradient engine.
 *
 * Drop-in boundary for the hot path of jhuggins/viabel:
 *     objective(var_param) -> (value, grad)        viabel/objectives.py:32-44
 * The reference has no FFI; its seam is the duck-typed Python call above
 * (callers: viabel/optimization.py:95 and :539).  These entry points are what a
 * ctypes binding inside viabel's VariationalObjective.__call__ would bind
 * (INTEGRATION.md shows that binding).  Plain C: opaque handle, pointers, sizes,
 * int status codes.  No C++ or torch types cross this boundary.
 *
 * Conventions
 *   - every array is fp64, C order; `theta` follows viabel's flat parameter layout
 *     (viabel/approximations.py:185-189 [mu | log_sigma], :315-319 [mu | free-Cholesky]);
 *   - `value`/`grad` are caller-owned host buffers; the reference returns the NEGATIVE
 *     lower bound and its gradient (objectives.py:164, :271) and so do these calls;
 *   - one context per GPU; calls on one context are serialised by the caller; the
 *     synchronous entry points return after the context's HIP stream has drained;
 *   - return 0 on success, a VB_ERR_* code otherwise; vb_last_error() gives the text.
 */
#ifndef VIABEL_HIP_H
#define VIABEL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vb_ctx vb_ctx;

/* status codes */
#define VB_OK 0
#define VB_ERR_INVALID 1     /* bad argument (Python raises ValueError)            */
#define VB_ERR_HIP 2         /* HIP runtime failure (RuntimeError)                 */
#define VB_ERR_UNSUPPORTED 3 /* combination not implemented (NotImplementedError)  */
#define VB_ERR_STATE 4       /* call order / missing model or noise (RuntimeError) */
#define VB_ERR_NUMERIC 5     /* e.g. 'All weights zero!' objectives.py:326-328      */
#define VB_ERR_COMM 6        /* RCCL failure                                       */
#define VB_ERR_CALLBACK 7    /* a host model callback returned non-zero (Python re-raises its exception) */

/* approximation families (viabel/approximations.py) */
#define VB_FAMILY_MF_GAUSSIAN 0        /* :192-251 */
#define VB_FAMILY_MF_STUDENT_T 1       /* :254-312 */
#define VB_FAMILY_FULLRANK_GAUSSIAN 2  /* new family, layout of :315-319 */
#define VB_FAMILY_MULTIVARIATE_T 3     /* :322-382 */
#define VB_FAMILY_LOWRANK_GAUSSIAN 4   /* LRGaussian :610-731 */

/* device-resident target models (replace the Python callable of viabel/models.py:17-39) */
#define VB_MODEL_GAUSS_DIAG 0  /* sum_d norm.logpdf(x_d; mean_d, sd_d)   dparams=[mean(D)|sd(D)]           */
#define VB_MODEL_FUNNEL 1      /* quickstart funnel, D-dim               dparams=[tau], iparams=[scale_idx] */
#define VB_MODEL_GAUSS_FULL 2  /* N(mean, P^-1)                          dparams=[mean(D)|P(DxD)|logdetP]   */
#define VB_MODEL_LOGISTIC 3    /* Bayesian logistic regression, N(0, sd) prior (not in the reference)
                                  dparams=[X(n_data x D)|y(n_data)|prior_sd], iparams=[n_data]             */

#define VB_MODEL_SOURCE 4      /* a log density given as HIP source (vb_set_model_source): the adaptor for user models */

/* noise kinds for vb_noise_generate */
/* likelihoods of the regression target VB_MODEL_LOGISTIC (iparams = [n_data, link]; default Bernoulli-logit):
 * eta = x_i' b, prior b ~ N(0, prior_sd).  Gaussian: dparams carries noise_sd after prior_sd. */
#define VB_GLM_BERNOULLI_LOGIT 0   /* y_i ~ Bernoulli(sigmoid(eta_i)) */
#define VB_GLM_POISSON 1           /* y_i ~ Poisson(exp(eta_i)) */
#define VB_GLM_GAUSSIAN 2          /* y_i ~ N(eta_i, noise_sd) */

#define VB_NOISE_NORMAL 0
#define VB_NOISE_STUDENT_T 1

/* ExclusiveKL flags (viabel/objectives.py:123) */
#define VB_FLAG_PATH_DERIV 1u          /* use_path_deriv=True */
/* hessian_approx_method (objectives.py:144) */
#define VB_CV_NONE 0
#define VB_CV_FULL 1
#define VB_CV_MEAN_ONLY 2
#define VB_CV_LOO_DIAG 3
#define VB_CV_LOO_DIRECT 4

#define VB_MAX_SLOTS 64

/* ---- library / context ------------------------------------------------------------ */
const char* vb_version(void);
int vb_device_count(int* count);
int vb_create(int device_id, vb_ctx** out);
int vb_destroy(vb_ctx* ctx);
/* text of the last failure on `ctx` (ctx == NULL: last failure of vb_create / vb_device_count) */
const char* vb_last_error(vb_ctx* ctx);
/* name, CU count, HBM bytes of the context's device */
int vb_device_info(vb_ctx* ctx, char* name, size_t name_len, int* n_cu, uint64_t* hbm_bytes);
/* block until every enqueued call on the context's stream has finished */
int vb_sync(vb_ctx* ctx);

/* ---- noise slots: N x D fp64 matrices resident in HBM ------------------------------
 * Replaces the RandomState draws inside approx.sample (approximations.py:212-216,
 * :270-274).  Parity mode: the host draws RandomState(seed).randn(N, D) (the exact legacy
 * MT19937 stream) and uploads it.  Throughput mode: counter-based Philox4x32-10
 * on the device; element (row_offset + n, d) depends only on (seed, stream, global row, d),
 * so results do not depend on how the Monte-Carlo axis is sharded.                       */
int vb_noise_set_host(vb_ctx* ctx, int slot, const double* host, int64_t n, int64_t d);
int vb_noise_generate(vb_ctx* ctx, int slot, int kind, double df, uint64_t seed,
                      uint64_t stream, int64_t row_offset, int64_t n, int64_t d);
int vb_noise_get_host(vb_ctx* ctx, int slot, double* host, int64_t n, int64_t d);
/* First and second moments of the noise matrix in `slot` over its first n rows (this rank's rows; with a communicator
 * the sums are all-reduced): colsum[j] = sum_n eps_nj (d doubles) and, when `gram` is not NULL, gram[i * d + j] =
 * sum_n eps_ni eps_nj (d x d row-major, symmetric).  What the reduced forms of the RGE control variates
 * (objectives.py:200-268) need beside the plain estimator's own sums when the model's Hessian is not one of the
 * built-in closed forms: mean eps (every method) and M2 = E'E / N (`full`).  One column-sum pass, one split-K lower-
 * triangular MFMA Gram product.                                                                                   */
int vb_noise_moments(vb_ctx* ctx, int slot, int64_t n, int64_t d, double* colsum, double* gram);

/* Chi-square(df) draws on the device, df > 2: draw i is element (row_offset + i) of Philox stream (seed, stream) --
 * the per-sample radial scales s = sqrt(chi2 / df) of MultivariateT.sample (approximations.py:345-347) in throughput
 * mode.  The n draws stay in the context (one set at a time); vb_dis_refresh_mvt takes them when its `chi` is NULL. */
int vb_chisq_generate(vb_ctx* ctx, double df, uint64_t seed, uint64_t stream, int64_t row_offset, int64_t n);
int vb_chisq_get_host(vb_ctx* ctx, double* host, int64_t n);
/* A prediction, never a request: "the NEXT vb_noise_generate for each slot in `slot_mask` (bit i: slot i) -- and, when
 * with_chi != 0, the next vb_chisq_generate -- will repeat the last one with this seed".  AlphaDivergence seeds every call
 * from the host's global generator (objectives.py:455: seed = npr.randint(2**32)), so the engine's own look-ahead, which
 * follows a walking stream index, has nothing to follow; a host that can name its generator's next output lets the next
 * call's noise be generated behind this call's last kernel, while the host waits and turns around.  The shadow is adopted
 * only by a request that matches it exactly; after a wrong hint the request generates as usual.  The hint is consumed by
 * the next blocking call of this context.                                                                              */
int vb_noise_hint_seed(vb_ctx* ctx, uint64_t slot_mask /* bit s: noise slot s (VB_MAX_SLOTS = 64) */, int with_chi, uint64_t seed);
/* How many look-ahead buffers (noise matrices and chi-square vectors) this context has generated and how many of them a
 * request adopted: an observability counter -- a caller whose hints or stream walks stop matching sees the ratio drop
 * (results never depend on it).                                                                                       */
int vb_noise_ahead_stats(vb_ctx* ctx, uint64_t* generated, uint64_t* adopted);

/* ---- model ------------------------------------------------------------------------ */
int vb_set_model(vb_ctx* ctx, int model_id, int64_t dim, const double* dparams,
                 size_t n_dparams, const int64_t* iparams, size_t n_iparams);
/* A model outside the built-in set, as device code.  Replaces what the reference does with an arbitrary Python
 * callable and autograd (models.py:17-39; convenience.py:75 `bbvi(dim, log_density=...)`): `source` is HIP C++ that
 * defines
 *     __device__ double vb_log_density(const double* z, int d, const double* params, double* grad);
 * returning f(z) for one sample z[0..d) and writing grad f to grad[0..d) unless grad is NULL.  `params` (n_params
 * doubles: data, hyper-parameters) is uploaded with the model.  A model that is a sum over data may instead define
 *     #define VB_LOG_DENSITY_PARTS K            (a power of two, 2 <= K <= 64; dim <= 128)
 *     __device__ double vb_log_density_part(const double* z, int d, const double* params, double* grad,
 *                                           int part, int n_parts);
 * returning the share of f -- and ADDING the share of grad f to grad, which arrives zeroed -- that belongs to `part`
 * (e.g. observations part, part + K, ...; the prior in part 0): K threads then work on every sample and their shares
 * are added in a fixed order.  The source is compiled for this GPU with hiprtc;
 * VB_ERR_INVALID carries the compiler's log.  Supported by ExclusiveKL over all five families (entropy form and path
 * derivative, no control variates), by the alpha-divergence entry points (vb_alpha_grad_meanfield,
 * vb_alpha_grad_fullrank, vb_alpha_sums_mvt, vb_alpha_sums_lowrank), by the DIS refreshes (vb_dis_refresh_meanfield /
 * _mvt / _lowrank: log p of the state samples), by vb_log_weights_meanfield and by vb_model_logp.            */
int vb_set_model_source(vb_ctx* ctx, int64_t dim, const char* source, const double* params, size_t n_params);
/* A model that exists only as host code WITH its gradient -- the contract of the reference's StanModel
 * (models.py:80-104: log_prob / grad_log_prob behind a vjp) and, with a numerical gradient on the caller's side, of
 * Model(log_density) (models.py:17-39).  Wherever a source model's row kernel would run, the engine instead copies the
 * n x d samples (row-major, dense) to pinned host memory, waits for the stream, calls
 *     fn(user, z, n, d, f, grad)      -> 0 on success, anything else aborts the call with VB_ERR_CALLBACK
 * which fills f[0..n) and -- unless grad is NULL (value-only calls: vb_model_logp, DIS refreshes, diagnostics) --
 * grad[0..n*d), and copies both back.  Sampling, the variational log density, the weights and every reduction over the
 * Monte-Carlo axis stay on the device: this is an adaptor for targets nobody will write in HIP, not a CPU path of the
 * estimator.  Supported wherever vb_set_model_source is; every evaluation contains a host synchronisation, so the
 * *_enqueue entry points and vb_fit block once per evaluation.  `fn` and `user` must outlive the binding.          */
typedef int (*vb_model_callback)(void* user, const double* z, int64_t n, int64_t d, double* f, double* grad);
int vb_set_model_callback(vb_ctx* ctx, int64_t dim, vb_model_callback fn, void* user);
/* f(x_n), n < N, for host x (N x D): Model.__call__ (models.py:27-39) on the device */
int vb_model_logp(vb_ctx* ctx, const double* x_host, int64_t n, int64_t d, double* out_host);
/* f(x_n) and grad f(x_n), n < N, for host x (N x D): what autograd's grad of Model.__call__ returns in the
 * reference (models.py:17-39; tests/test_models.py:13-15 checks it with check_vjp).  g_host is N x D row-major;
 * f_host (N) may be NULL.  Every built-in target and source models.                                            */
int vb_model_grad(vb_ctx* ctx, const double* x_host, int64_t n, int64_t d, double* f_host, double* g_host);

/* ---- ExclusiveKL, mean-field families (objectives.py:150-273) ----------------------
 * theta = [mu(D) | log_sigma(D)] on the host; noise slot holds the N x D base draws
 * (normal for MF_GAUSSIAN, standard-t for MF_STUDENT_T).  n_total is the Monte-Carlo
 * sample count of the WHOLE job (== n on one GPU; sum over ranks when a communicator is
 * attached, in which case the partial sums are all-reduced before the epilogue).        */
int vb_elbo_grad_meanfield(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total,
                           int family, double df, const double* theta, unsigned flags,
                           int cv_mode, double* value, double* grad);
/* Same, enqueued without waiting: results land in result slot `rslot`; fetch them with
 * vb_result_get after vb_sync.  Lets a caller keep several evaluations in flight.       */
/* The same evaluation on fresh device noise without materialising it: element (row_offset + i, j) of Philox stream
 * (seed, stream) -- exactly what vb_noise_generate(kind = the family's base noise) would have written to `slot` --
 * is generated in registers by the streaming kernel (gauss_diag / funnel targets; `slot` only fixes the geometry). */
int vb_elbo_grad_meanfield_philox(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, int64_t row_offset,
                                  int family, double df, const double* theta, unsigned flags, int cv_mode,
                                  uint64_t seed, uint64_t stream, double* value, double* grad);
int vb_elbo_grad_meanfield_async(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total,
                                 int family, double df, const double* theta, unsigned flags,
                                 int cv_mode, int rslot);
/* `count` independent evaluations in one call: evaluation b streams noise slot slots[b] with the
 * parameter thetas[b * 2d ...] (row-major count x 2d) and lands in result slot rslots[b].  Up to 32
 * evaluations share one launch of each kernel (blockIdx.y), which amortises launch latency and
 * fills the chip; use it to evaluate the objective at many parameter vectors / noise draws at
 * once (multi-start fits, gradient-variance estimates, line searches).                      */
int vb_elbo_grad_meanfield_batch_async(vb_ctx* ctx, int count, const int* slots, int64_t n, int64_t d,
                                       int64_t n_total, int family, double df, const double* thetas,
                                       unsigned flags, int cv_mode, const int* rslots);
int vb_result_get(vb_ctx* ctx, int rslot, double* value, double* grad, int64_t p);

/* ---- AlphaDivergence, mean-field families (objectives.py:443-463) -----------------------
 * value = log(mean_n s_n)/alpha + max_n lw_n,  s_n = exp(lw_n - max)^alpha,
 * lw_n = f(z_n) - log q(z_n; theta);  grad = alpha/N sum_n s_n d lw_n/d theta (s not normalised,
 * objectives.py:460).  The noise slot holds the draws of RandomState(seed) for the seed the
 * caller took from the global numpy RNG (objectives.py:455).                                  */
int vb_alpha_grad_meanfield(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, int family, double df,
                            const double* theta, double alpha, double* value, double* grad);

/* ---- DISInclusiveKL, mean-field families (objectives.py:283-416) ------------------------
 * refresh (objectives.py:393-398, :338-368): the noise slot holds the base draws of the state
 * samples z_n = mu + sigma eps_n (kept on the device; z is never materialised).  Computes
 * log p(z_n), log q(z_n; theta), the tempering prior's log density (a diagonal Gaussian given as an
 * MFGaussian parameter [mu | log_sigma], tests/test_objectives.py:82-87), runs the 50-step ESS
 * bisection on the tempering parameter, and returns eps, ess and the unnormalised weights
 * w_n = exp(eps log prior + (1 - eps) log p - log q) (no max shift, :330).  log_p / log_q may be
 * NULL.  Returns VB_ERR_NUMERIC with 'All weights zero! ...' as objectives.py:326-328 does.
 * Sharded jobs (n < n_total, n_total = n x ranks): every rank all-gathers the three per-sample vectors and
 * runs the bisection redundantly; eps, w, log_p, log_q then cover all n_total samples on every rank.
 * Clipping (:370-386) and resampling (:408, global numpy RNG) stay with the caller, which hands
 * the per-sample weights (clipped w, or resampling counts) to
 * grad: value = -scale sum_n weights_n log q(z_n; theta), grad = d value / d theta (:405-414).  */
/* The tempering prior of DISInclusiveKL may be any approximation family (objectives.py:283-285, :317-319 call
 * temper_prior.log_density(temper_prior_params, z)).  The refresh calls below take the common case -- an MFGaussian
 * parameter (tests/test_objectives.py:82-87) -- as their `prior` argument; any other prior is installed here and then
 * REPLACES that argument in every vb_dis_refresh_* of the context until VB_PRIOR_DIAG_GAUSSIAN is set again:
 *   VB_PRIOR_DIAG_STUDENT_T  MFStudentT (approximations.py:281-286): loc (d), scale = log sigma (d), df > 0
 *   VB_PRIOR_DENSE           log pi0(x) = c - log_det_l - 1/2 |L^-1 (x - loc)|^2 (df = 0: FullRankGaussian, or any
 *                            Gaussian family through the Cholesky factor of its covariance) or the multivariate t
 *                            log pdf of _distributions.py:7-38 (df > 0): loc (d), scale = L^-1 (d x d row-major, lower
 *                            triangular), log_det_l = sum log L_ii.  One more N x d x d MFMA product per refresh.
 * The state samples are materialised for it where the refresh would not need them (mean-field families).            */
#define VB_PRIOR_DIAG_GAUSSIAN 0
#define VB_PRIOR_DIAG_STUDENT_T 1
#define VB_PRIOR_DENSE 2
int vb_dis_set_temper_prior(vb_ctx* ctx, int kind, int64_t d, double df, const double* loc, const double* scale,
                            double log_det_l);
int vb_dis_refresh_meanfield(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, int family, double df,
                             const double* theta, const double* prior_theta, double eps_prev,
                             double ess_target, int max_bisection_its, double* eps, double* ess,
                             double* w, double* log_p, double* log_q);
int vb_dis_grad_meanfield(vb_ctx* ctx, int slot, int64_t n, int64_t d, int family, double df,
                          const double* theta, const double* weights, double scale, double* value,
                          double* grad);

/* AlphaDivergence for the dense Gaussian family (theta = [mu | free Cholesky], z = mu + L eps): same value and
 * gradient convention as vb_alpha_grad_meanfield; grad has d + d (d + 1) / 2 entries. */
int vb_alpha_grad_fullrank(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, const double* theta,
                           double alpha, double* value, double* grad);
/* The same for the multivariate t in throughput mode (objectives.py:443-463 over approximations.py:322-382): samples
 * x = mu + (L z) / s through the Cholesky factor with the device's normals (slot) and chi-square draws
 * (vb_chisq_generate), weights, value and the gradient in the flat free-Cholesky layout -- tril(sum_n w_n g_n (z_n /
 * s_n)') with sum w on the log-diagonal -- all on the device: no matrix root, no D x D array on the host. */
int vb_alpha_grad_mvt_chol(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df, const double* theta,
                           double alpha, double* value, double* grad);

/* ---- DISInclusiveKL, MultivariateT family (approximations.py:322-382) -------------------
 * theta = [mu | free Cholesky of Sigma].  The O(D^3) factor algebra stays with the caller, as in the
 * reference (sqrtm at approximations.py:348, eigh at _distributions.py:26): refresh takes the symmetric
 * root Sigma^{1/2} and L^-1 (row-major D x D), the chi-square draws (N) and the normal draws in the noise
 * slot (drawn in that order, approximations.py:345-347); the device forms the samples
 * x = mu + (z Sigma^{1/2}) / sqrt(chi/df) with an MFMA GEMM, keeps them, evaluates log p, log q, the
 * tempering prior and runs the ESS bisection.  grad returns, for weights w_n and the CURRENT theta / L^-1,
 *   w_sum = sum w,  w_logq = sum w log q(x_n),  d_mu[D] = sum w c_n u_n,  gram[D x D] (lower triangle) =
 *   sum w c_n u_n u_n',  u_n = Sigma^-1 (x_n - mu),  c_n = (df + D)/(df + maha_n)
 * from which the caller assembles d/dtheta.
 * df = 0 selects the Gaussian limit (c_n = 1, no chi-square scaling, Gaussian log q): the dense Gaussian family
 * passes sqrt_sigma = L' (so that x = mu + z L') and its L^-1.                                    */
int vb_dis_refresh_mvt(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df, const double* theta,
                       const double* chi, const double* sqrt_sigma, const double* l_inv,
                       const double* prior_theta, double eps_prev, double ess_target, int max_bisection_its,
                       double* eps, double* ess, double* w, double* log_p, double* log_q);
/* ExclusiveKL (entropy form, objectives.py:154-164) of the multivariate t in the reference-identical mode, resident on the
 * device: noise slot and the context's chi-square draws hold numpy's streams (vb_legacy_rng_chisquare_device,
 * then vb_legacy_rng_randn_device -- approximations.py:345-347), the samples go through the SYMMETRIC root of Sigma = L L'
 * (:348) and the gradient through the root's Frechet derivative (the Sylvester equation R X + X R = sym(C) / N), both by
 * Newton-Schulz iterations on the device; value and the gradient in the flat [mu | free Cholesky] layout come back after
 * one copy.  info (4 doubles, may be NULL) = [root steps, root accuracy, derivative steps, its accuracy].
 * Sharded jobs (round 6): `n` = this rank's rows (the slot holds rows [shard begin, +n) of randn(n_total, d); the context's
 * chi-square buffer holds either those n draws or all n_total of numpy's -- the shard's block is taken), the sample sums
 * are all-reduced on the device and the O(D^3) chain rule runs redundantly on every rank: the same bits everywhere.
 * VB_ERR_UNSUPPORTED: an iteration did not resolve to 1e-12 -- use vb_elbo_sums_mvt with a host-side root.            */
int vb_elbo_grad_mvt_symroot(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df, const double* theta,
                             double* value, double* grad, double* info);
/* The same evaluation in the path-derivative form (ExclusiveKL(use_path_deriv=True), objectives.py:156-159): the value takes
 * the samples' mean log density instead of the entropy, and the score -d log q / dx (x_n) = c_n Sigma^(-1/2) z_n / s_n joins
 * the model gradient -- its part of the sums depends on the noise only (the sums of vb_mvt_path_terms, formed on the
 * device) and enters through Sigma^(-1/2), the second limit of the root's coupled iteration.                             */
int vb_elbo_grad_mvt_symroot_path(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df,
                                  const double* theta, double* value, double* grad, double* info);
/* AlphaDivergence (objectives.py:443-463) of the multivariate t in the reference-identical mode, resident on the device
 * (n / n_total as above): noise as for vb_elbo_grad_mvt_symroot (the caller's fresh RandomState(seed) of :455-456 drawn on the device), the
 * samples through the symmetric root, weights / value / weighted sums by the kernels of vb_alpha_sums_mvt, the chain rule
 * through the root's Frechet derivative with sum s on the free diagonal; value = the log-normaliser estimate, grad =
 * alpha / N times the weighted score.  VB_ERR_UNSUPPORTED: an iteration did not resolve -- vb_alpha_sums_mvt + host root. */
int vb_alpha_grad_mvt_symroot(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df, double alpha,
                              const double* theta, double* value, double* grad, double* info);

/* The reference-identical step resident on the device: as vb_dis_refresh_mvt with chi, sqrt_sigma, l_inv and
 * w NULL -- factors from theta on the device, chi-square draws from the context's buffer (vb_legacy_rng_chisquare_device
 * for numpy's stream: this rank's n draws, or all n_total of them -- the shard's block is taken), nothing copied back --
 * but the samples go through the SYMMETRIC root of Sigma = L L'
 * (approximations.py:348), formed on the device by a Newton-Schulz iteration scaled by the infinity norm; root_info (3
 * doubles, may be NULL) = [steps, last residual, ||R R - Sigma|| / ||Sigma||_inf].  vb_dis_step_mvt_packed follows.
 * VB_ERR_UNSUPPORTED: the iteration did not resolve the root to 1e-12 (no state was installed): use the host route. */
int vb_dis_refresh_mvt_symroot(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df, const double* theta,
                               const double* prior_theta, double eps_prev, double ess_target, int max_bisection_its,
                               double* root_info);
/* Throughput mode of the dense families: vb_dis_refresh_mvt with sqrt_sigma == NULL and l_inv == NULL forms mu, L and
 * L^-1 from `theta` on the device (blocked triangular inverse, GEMM levels) and draws the state samples through the
 * Cholesky factor, x = mu + (z L') / s -- same distribution as the symmetric root of approximations.py:348, which is
 * only needed to reproduce the reference's noise stream.  vb_dis_grad_mvt_packed is vb_dis_grad_mvt plus the chain
 * rule to the flat parameter on the device: value = -scale sum_n w_n log q(x_n; theta) and its gradient (d + d (d + 1) / 2). */
int vb_dis_grad_mvt_packed(vb_ctx* ctx, int64_t n, int64_t d, double df, const double* theta, const double* weights,
                           double scale, double* value, double* grad);
/* Device-resident step of the throughput mode: after vb_dis_refresh_mvt(..., w = NULL) -- which only enqueues
 * -- the tempered weights, eps, ess and the zero-weight status are still on the device.  Sharded jobs (round 6; SURVEY
 * 8(e), objectives.py:391-414): every rank samples and scores its `n` rows, [log p | log q | log prior] are all-gathered on
 * the device (3 n_total doubles), the bisection / smoothing / clipping / multinomial draw run REDUNDANTLY on the whole
 * vectors on every rank (the same kernels on the same bits: the same weights everywhere), each rank forms the weighted
 * sums over its own rows, ONE all-reduce of [sum w, sum w log q, sum a y, M (D x D)] follows and the D x D x D chain rule
 * is redundant again: every rank returns the same bits.  This call takes the gradient of
 * -scale sum_n w_n log q(x_n; theta) on those weights (resample_m == 0, objectives.py:405-406) or on the counts of
 * resample_m multinomial draws from them (objectives.py:408-414: np.random.choice; here Philox uniforms of
 * (seed, stream) inverted through the running sums of the weights; `scale` is then multiplied by sum_n w_n on the
 * device), and returns (value, grad) together with eps and ess of the refresh after ONE synchronisation: no weight
 * vector crosses PCIe, no host work of order N.  vb_dis_weights_get fetches the tempered weights for callers that
 * want to look at them.  The weight clipping of objectives.py:370-386 is the identity for thresholds >= 1 (the
 * default is 10); a smaller threshold is applied to the resident weights by vb_dis_clip_mvt before the step.      */
int vb_dis_step_mvt_packed(vb_ctx* ctx, int64_t n, int64_t d, double df, const double* theta, int64_t resample_m,
                           uint64_t seed, uint64_t stream, double scale, double* eps, double* ess,
                           double* khat /* NULL, or the tail shape of the last vb_dis_psis_mvt */, double* value, double* grad);
/* Between the two: Pareto smoothing of the device-resident tempered weights, w -> sum(w) exp(psislw(log w)) in place
 * (`DISInclusiveKL(psis_smooth=True)`: BASELINE configs[3] asks for "DISInclusiveKL with PSIS reweighting"; the smoothing
 * is viabel/_psis.py:113-209 as vb_psis_smooth does it).  Enqueues only; the step that follows uses the smoothed weights
 * and reports khat. */
int vb_dis_psis_mvt(vb_ctx* ctx, int64_t n_total, double reff);
/* ... and weight clipping (objectives.py:370-386, w_clip_threshold < 1) of the device-resident weights in place, after
 * the smoothing when both are on (the order of objectives.py:398-399 with psis_smooth in front).  The reference's own
 * line :385 cannot run; what is computed is the fixed point its recursion aims at -- weights at or above
 * threshold * sum(w_clipped) are set to exactly that value, the clipped set growing round by round until no unclipped
 * weight reaches it (oracle.objectives.DISInclusiveKL._clip).  Enqueues only (one workgroup, fixed summation order). */
int vb_dis_clip_mvt(vb_ctx* ctx, int64_t n_total, double threshold);
int vb_dis_weights_get(vb_ctx* ctx, double* w, int64_t n_total, int resampled /* 1: the counts of the last draw */);
/* [eps, ess, zero-weight status, khat] of the last device-resident refresh (what vb_dis_step_mvt_packed returns with its
 * gradient), for callers that weight the score themselves (vb_dis_grad_mvt_packed after a host resampling draw).   */
int vb_dis_scalars_get(vb_ctx* ctx, double out[4]);
/* Page-locked host memory for large result arrays (hipHostMalloc / hipHostFree; no context: any thread).  A gradient of
 * 525 824 doubles copied into PAGEABLE memory is staged by the runtime (118 us at D = 1024); into a block from here the
 * copy engine writes directly.  A binding may allocate the arrays it returns from such blocks (the Python one does:
 * viabel_amd/_lib.py, PinnedPool -- the reference returns freshly allocated numpy arrays, objectives.py:32-44).        */
int vb_host_alloc(size_t bytes, void** ptr);
int vb_host_free(void* ptr);      /* NULL: a no-op, as free(NULL) */
/* The DIS state samples live in the context, one set per family kind (0: mean-field, 1: MultivariateT / dense
 * Gaussian, 2: low-rank Gaussian); every refresh of a kind overwrites its set and bumps its generation counter.  A
 * caller that keeps weights for a later vb_dis_grad_* call compares the counter with the one it saw after its own
 * refresh: a mismatch means another objective refreshed in between and the weights no longer belong to the samples. */
int vb_dis_generation(vb_ctx* ctx, int kind, uint64_t* generation);
/* Parking a DIS state (round 6).  The reference keeps a DISInclusiveKL's state per OBJECT (objectives.py:391-403), so two
 * objectives with num_resampling_batches > 1 may take turns; here it lives in the context.  vb_dis_state_park DETACHES the
 * context's state of `kind` -- buffers (pointer moves, no copies), shapes, the parameter its residuals belong to, the
 * generation counter, and noise slot `slot` (the state's samples for the mean-field kind, the residuals of a
 * throughput-mode dense state; -1: none) -- into *handle; the context then has no state of that kind and allocates afresh
 * for whoever refreshes next.  vb_dis_state_unpark installs a parked state again (consuming the handle; what the context
 * held of that kind is released -- park it first if it is still needed).  vb_dis_state_drop frees a handle.  All three
 * synchronise the context's streams: they are for the rare hand-over, not for the hot path.                        */
int vb_dis_state_park(vb_ctx* ctx, int kind, int slot, void** handle);
int vb_dis_state_unpark(vb_ctx* ctx, void* handle);
int vb_dis_state_drop(void* handle);
/* log p / log q of the state samples of the last refresh (all n_total of them; either pointer may be NULL), for
 * callers that passed NULL to the refresh.  dense = 0: mean-field state, 1: MultivariateT / dense-Gaussian state. */
int vb_dis_state_get(vb_ctx* ctx, int dense, double* log_p, double* log_q, int64_t n_total);
int vb_dis_grad_mvt(vb_ctx* ctx, int64_t n, int64_t d, double df, const double* theta, const double* l_inv,
                    const double* weights, double* w_sum, double* w_logq, double* d_mu, double* gram);

/* ---- ExclusiveKL, low-rank-plus-diagonal Gaussian family ------------------------------
 * LRGaussian, viabel/approximations.py:610-731: theta = [mu (D) | log_sigma (D) | B (D x k, row-major)],
 * x = mu + z B' + sigma * eps with z (n x k, slot_z) drawn before eps (n x D, slot_eps) (:636-644);
 * entropy via the matrix determinant lemma (:559-573, :646-652); estimator objectives.py:154-164
 * (entropy form; flags must be 0).  grad has 2 D + D k entries in the theta layout.  1 <= k <= 16 (one streaming
 * pass with a lane's rows of B in registers); larger ranks: vb_elbo_sums_lowrank.  */
int vb_elbo_grad_lowrank(vb_ctx* ctx, int slot_eps, int slot_z, int64_t n, int64_t d, int64_t k,
                         int64_t n_total, const double* theta, unsigned flags, double* value,
                         double* grad);

/* The same estimator's SUMS for any rank k >= 1 (the reference's family has no rank limit, approximations.py:610-644):
 * out = [sum_n f(x_n) | sum_n g_n (D) | sum_n g_n * eps_n (D) | sum_n g_n z_n' (D x k, row-major)], 1 + 2 D + D k doubles,
 * all-reduced over the ranks of a sharded job.  Samples through an n x k x D MFMA product, the model's (f, G) of the
 * materialised samples (every built-in target and source models), G' Z split over the sample axis.  The entropy and its
 * gradient -- O(D k^2) through the k x k capacitance matrix -- are the caller's, as for the other low-rank entry points. */
int vb_elbo_sums_lowrank(vb_ctx* ctx, int slot_eps, int slot_z, int64_t n, int64_t d, int64_t k, const double* theta,
                         double* out);

/* ---- Importance weights and Pareto smoothing (diagnostics) ----------------------------
 * vb_log_weights_meanfield: log p(z_n) - log q(z_n; theta) for the samples z = mu + sigma * eps of
 * the noise staged in `slot` -- samples_and_log_weights, viabel/convenience.py:176-179.  The
 * weights stay resident on the device for vb_psis_smooth; `lw` (n doubles) may be NULL.
 * vb_psis_smooth: psislw of viabel/_psis.py:113-209 (tail size ceil(min(0.2 n, 3 sqrt(n / reff))),
 * Zhang-Stephens GPD fit _psis.py:212-332, smoothed weights normalised to log-sum-exp 0) --
 * psis_correction, viabel/convenience.py:166-169.  `lw_in` NULL smooths the device-resident
 * weights of the last vb_log_weights_* call; otherwise n host log weights are uploaded first.
 * Outputs: lw_out (n doubles, smoothed), khat (Pareto tail index; +inf when the tail has <= 4
 * samples).  VB_ERR_UNSUPPORTED when the tail exceeds the on-chip sort capacity (4096 values,
 * i.e. n > 1.86e6 at reff = 1).                                                              */
int vb_log_weights_meanfield(vb_ctx* ctx, int slot, int64_t n, int64_t d, int family, double df,
                             const double* theta, double* lw);
int vb_psis_smooth(vb_ctx* ctx, const double* lw_in, int64_t n, double reff, double* lw_out,
                   double* khat);

/* ---- ExclusiveKL, multivariate t family ------------------------------------------------
 * ExclusiveKL closure (objectives.py:154-164, entropy form) for MultivariateT
 * (approximations.py:322-382): x_n = mu + (z_n Sigma^{1/2}) / s_n with the normals z (n x D) in
 * `slot`, sqrt_sigma the symmetric root (D x D, row-major) and inv_s[n] = 1 / sqrt(chi2_n / df)
 * (:345-349).  Returns the sample sums f_sum = sum_n f(x_n), g_sum[D] = sum_n grad f(x_n) and
 * c_full[D x D] = sum_n grad f(x_n) (z_n / s_n)' (= d f_sum / d sqrt_sigma for an unconstrained
 * root); the O(D^3) chain rule root -> Sigma -> free Cholesky parameters stays with the caller,
 * like the root itself.  Sums cover all ranks when a communicator is attached (f_sum: add the
 * per-sample constant of the local rows only -- it is included).                              */
int vb_elbo_sums_mvt(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, const double* mu,
                     const double* sqrt_sigma, const double* inv_s, double* f_sum, double* g_sum,
                     double* c_full);

/* ---- ExclusiveKL, full-rank Gaussian family -------------------------------------------
 * New family (no reference class) behind the ApproximationFamily API with the flat
 * layout of viabel/approximations.py:315-319: theta = [mu (D) | free Cholesky (D(D+1)/2)],
 * z = mu + L eps.  Estimators: objectives.py:160-164 (entropy form) and, with VB_FLAG_PATH_DERIV, :156-159
 * (path derivative: the score L^-T eps enters through the noise Gram matrix and an explicit triangular inverse on
 * the device).  Targets: gauss_diag, funnel, gauss_full and the regression models.  fp64 MFMA GEMMs.
 * vb_elbo_grad_fullrank = set_theta + enqueue + get; the three-step form keeps theta and the
 * result resident on the device (P = D + D(D+1)/2 doubles is 4.2 MB at D = 1024).  Hand the blocking call a
 * gradient array from vb_host_alloc and the download is a direct DMA.  VB_FR_UPLOAD_PIPE=1 (round 6: bit-identical,
 * measured slower, off): the flat parameter crosses PCIe in three row chunks of L, last rows first, and the sampling
 * product of a chunk's column blocks starts behind its copy.          */
/* how many vb_elbo_grad_fullrank calls of this context took the pipelined upload (observability: tests) */
int vb_fullrank_upload_stats(vb_ctx* ctx, uint64_t* pipelined_calls);
int vb_elbo_grad_fullrank(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total,
                          const double* theta, unsigned flags, double* value, double* grad);
int vb_fullrank_set_theta(vb_ctx* ctx, const double* theta, int64_t d);
int vb_elbo_grad_fullrank_enqueue(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total,
                                  unsigned flags);
int vb_fullrank_get(vb_ctx* ctx, double* value, double* grad, int64_t p);

/* ---- ExclusiveKL for the multivariate t, throughput mode (approximations.py:342-354, objectives.py:160-164) --------
 * Samples through the CHOLESKY factor, x_n = mu + (L z_n) / s_n with s_n = sqrt(chi_n / df) from the device chi-square
 * draws of vb_chisq_generate (same distribution as the reference's symmetric root, :348, which parity mode keeps): with
 * the factor itself in the sampler, d/dL is tril(sum_n g_n (z_n / s_n)') directly -- no matrix square root, no
 * Sylvester solve, nothing of order D^2 or more on the host.  theta = [mu | free Cholesky of the scale matrix]; the
 * noise slot holds the n x d normals.  `value` is -(mean f + D/2 (1 + log 2 pi) + sum log L_ii): the caller swaps the
 * Gaussian entropy constant for the family's own (a function of df and D only).  Entropy form only.              */
int vb_elbo_grad_mvt_chol(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df, const double* theta,
                          double* value, double* grad);

/* ---- symmetric matrix square root on the device ------------------------------------------------------
 * root = a^(1/2) for a symmetric positive definite d x d host matrix -- scipy.linalg.sqrtm(Sigma) in
 * MultivariateT.sample (approximations.py:348) -- by coupled Newton-Schulz iterations (fp64 MFMA GEMMs only).
 * With e != NULL also x = the solution of  root x + x root = e  (the derivative of the root in direction e, what
 * autograd's sqrtm VJP computes for ExclusiveKL over a MultivariateT, objectives.py:154-164); e and x may be
 * NULL.  info (3 doubles, may be NULL) = [iterations, final ||I - Z Y||_F, ||root root - a||_F / ||a||_F]:
 * the caller decides from info[2] whether to keep the result or use its LAPACK path (the iteration resolves
 * condition numbers up to ~1e12). */
int vb_sym_sqrt(vb_ctx* ctx, const double* a, const double* e, int64_t d, double* root, double* x, double* info);
/* the same with the inverse root a^(-1/2) (d x d, may be NULL), which the iteration produces alongside */
int vb_sym_sqrt_inv(vb_ctx* ctx, const double* a, const double* e, int64_t d, double* root, double* x, double* info,
                    double* inv_root);

/* ---- ExclusiveKL with use_path_deriv=True over an LRGaussian (objectives.py:156-159, approximations.py:610-731)
 * The score Sigma^-1 (x - mu) of the low-rank Gaussian is linear in its two noise blocks (eps: n x d in slot_eps,
 * z: n x k in slot_z), so its contribution to the entropy-form sums follows from second moments of the noise.
 * With sw = sigma * (B / sigma^2) (d x k, row-major, host) and u_n = sw' eps_n, T = [z | u] (n x 2k), `out` receives
 *   [ E'T (d x 2k, row-major) | T'T (2k x 2k) | sum_n eps_n (d) | sum_n eps_n^2 (d, per column) | sum_n T_n (2k) ]
 * (d 2k + 4 k^2 + 2 d + 2k doubles); the O(d k^2) Woodbury algebra stays with the caller.  1 <= k <= 256. */
int vb_lowrank_path_terms(vb_ctx* ctx, int slot_eps, int slot_z, int64_t n, int64_t d, int64_t k, int64_t n_total,
                          const double* sw, double* out);

/* ---- ExclusiveKL with use_path_deriv=True over a MultivariateT (objectives.py:156-159) -----------------
 * The score of the t density at a sample, -dlog q/dx = c_n Sigma^(-1/2) z_n / s_n with
 * c_n = (df + D) / (df + |z_n|^2 / s_n^2), depends on the noise only.  Its contribution to the sums of
 * vb_elbo_sums_mvt is Sigma^(-1/2) m_w and Sigma^(-1/2) e_w with
 *   m_w[D x D] = sum_n (c_n / s_n^2) z_n z_n'   (symmetric; lower triangle computed, mirrored on return)
 *   e_w[D]     = sum_n (c_n / s_n) z_n
 * and the value needs log1p_sum = sum_n log(1 + |z_n|^2 / (s_n^2 df)).  inv_s[n] = 1 / s_n. */
int vb_mvt_path_terms(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df, const double* inv_s,
                      double* m_w, double* e_w, double* log1p_sum);

/* ---- AlphaDivergence over a MultivariateT (objectives.py:453-461 with approximations.py:342-357) --------
 * Samples x_n = mu + (z_n sqrt_sigma) / s_n as vb_elbo_sums_mvt.  The Mahalanobis distance of a sample is
 * |z_n|^2 / s_n^2 whatever the parameters are, so log q(x_n) = t-density constant - sum_log_diag
 * - (df + D)/2 log1p(|z_n|^2 / (s_n^2 df)) and the log weights, their maximum, w_n = exp(alpha (lw_n - max)) and
 * `value` = log(mean w) / alpha + max (:457-459) are formed on the device.  Returns w_sum = sum_n w_n,
 * g_sum[D] = sum_n w_n g_n and c_full[D x D] = sum_n w_n g_n (z_n / s_n)'; the caller applies alpha / N, the
 * chain rule through the symmetric root and adds w_sum to the free (log) diagonal.  Sums cover all ranks.
 * Throughput mode: inv_s == NULL takes s_n from the device chi-square draws of vb_chisq_generate, and `sqrt_sigma` may
 * be any root R with R' R = Sigma -- with R = L' (x = mu + L z / s) the chain rule is tril(c_full) itself. */
int vb_alpha_sums_mvt(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df, double alpha,
                      const double* mu, const double* sqrt_sigma, const double* inv_s, double sum_log_diag,
                      double* value, double* w_sum, double* g_sum, double* c_full);

/* ---- device-resident fit: the optimiser loop of optimization.py:83-127 without host round trips ----
 * Replaces  for k in range(n_iters): value, grad = objective(theta); theta -= lr * descent_direction(grad)
 * (StochasticGradientOptimizer.optimize, optimization.py:91-112) for an ExclusiveKL objective whose family
 * draws Philox noise: iteration k generates its noise matrix (stream first_stream + k, rows row_offset ..
 * row_offset + n of the global matrix) into `slot`, evaluates (value, grad) as vb_elbo_grad_meanfield /
 * vb_elbo_grad_fullrank would, and applies one optimiser step on the device; all n_iters iterations are
 * enqueued back to back and the call returns when the last one has finished.  The update arithmetic follows
 * numpy's operation order without fused multiply-adds, so the trajectory equals the host loop's bit for bit.
 *   family      VB_FAMILY_MF_GAUSSIAN / VB_FAMILY_MF_STUDENT_T (p = 2 d), VB_FAMILY_FULLRANK_GAUSSIAN
 *               (p = d + d (d + 1) / 2) or VB_FAMILY_LOWRANK_GAUSSIAN (p = 2 d + d k, 1 <= k <= 16: the n x k block
 *               of its noise goes to `slot_aux`; iteration c draws the n x d block from Philox stream
 *               2 (first_stream + c) and the n x k block from stream 2 (first_stream + c) + 1)
 *   opt_kind    VB_OPT_SGD (:129-130), VB_OPT_RMSPROP (:188-197), VB_OPT_ADAM (:308-326), VB_OPT_ADAGRAD (:430-433)
 *   hyper       [learning_rate, beta (RMSProp) or beta1 (Adam), beta2 (Adam), jitter]
 *   theta       in: start, out: parameter after n_iters steps (p doubles)
 *   state       in/out optimiser state [second moment (p) | momentum (p)]; read only if has_state != 0
 *   values      out: objective value of every iteration (n_iters)
 *   history     out: the last hist_len iterates (after their step), row-major hist_len x p; may be NULL
 *   directions  out: descent direction of every iteration, row-major n_iters x p (the optimiser's
 *               diagnostics log, optimization.py:108-109); may be NULL
 *   gradients   out: objective gradient of every iteration, row-major n_iters x p (FASO's grad_history,
 *               optimization.py:541); may be NULL
 * Sharded jobs (vb_comm_init): every rank passes its own n / row_offset and the same n_total; all ranks
 * apply the same step to the same all-reduced gradient. */
#define VB_OPT_SGD 0
#define VB_OPT_RMSPROP 1
#define VB_OPT_ADAM 2
#define VB_OPT_ADAGRAD 3
int vb_fit(vb_ctx* ctx, int slot, int slot_aux, int64_t n, int64_t d, int64_t n_total, int64_t row_offset, int family,
           double df, unsigned flags, int cv_mode, int noise_kind, double noise_df, uint64_t seed,
           uint64_t first_stream, int opt_kind, const double hyper[4], int64_t n_iters, double* theta, int64_t p,
           double* state, int has_state, double* values, double* history, int64_t hist_len, double* directions,
           double* gradients);
/* Round 6: the mean of the LAST `rows` iterates of the history the last vb_fit of this context kept (hist_len >= rows), each
 * component added in iteration order and divided by rows -- np.mean(history[-rows:], axis=0), the iterate average the
 * reference's optimisers return as opt_param (optimization.py:120-126), bit for bit, from the rows still resident on the device
 * (60 rows of 4.2 MB at the headline shape: 7 ms of numpy on the host, 23 us per iteration of a 300-iteration fit).
 * VB_ERR_STATE: no such history (no fit yet, fewer rows kept, another p, or the work buffer has been reused).        */
int vb_fit_history_mean(vb_ctx* ctx, int64_t rows, int64_t p, double* mean);
/* Observability (tests): how many DIS refreshes of the dense families took log p / log prior out of the sampling product's
 * epilogue (no pass over the samples: VB_MVT_EPI_ROWS), and how many steps had the chain-rule kernel store the gradient into the
 * mapped result buffer itself (VB_MVT_CHAIN_FETCH).                                                                         */
int vb_mvt_route_stats(vb_ctx* ctx, uint64_t* epilogue_rows, uint64_t* chain_fetch);

/* ---- LRGaussian (approximations.py:610-731) under DISInclusiveKL / AlphaDivergence (objectives.py:283-463) ----
 * theta = [mu (d) | log_sigma (d) | B (d x k, row-major)], x = mu + B z + sigma eps with the n x d block of the noise
 * in `slot_eps` and the n x k block in `slot_z` (1 <= k <= 64; beyond 16 the per-sample k x k products run one wave
 * per sample).  The caller passes the pieces of theta plus
 * m_inv = (I + B' diag(sigma^-2) B)^-1 (k x k) and log_q_const = -(d log 2 pi + log det Sigma) / 2 (the O(d k^2)
 * algebra through the capacitance matrix, approximations.py:559-607, stays on the host); per-sample work and every
 * contraction over the samples run on the device.  gauss_diag and funnel targets.
 *
 * vb_dis_refresh_lowrank: state refresh (objectives.py:393-401) as vb_dis_refresh_mvt; the samples stay on the device.
 * vb_dis_grad_lowrank: weighted sums of the state samples at a (new) parameter, with rho = (x - mu) / sigma,
 *   tau = m_inv Bs' rho (Bs = B / sigma):  out = [sum w rho tau' (d x k) | sum w tau tau' (k x k) | sum w rho (d) |
 *   sum w rho^2 (d) | sum w tau (k) | sum w | sum w log q]   (d k + k k + 2 d + k + 2 doubles).
 * vb_alpha_sums_lowrank: value (objectives.py:459), sum of the weights s_n and, with t = m_inv (z - Bs' eps),
 *   out = [sum s g z' (d x k) | sum s eps t' (d x k) | sum s t t' (k x k) | sum s g (d) | sum s g eps (d)]. */
int vb_dis_refresh_lowrank(vb_ctx* ctx, int slot_eps, int slot_z, int64_t n, int64_t d, int64_t k, int64_t n_total,
                           const double* mu, const double* log_sigma, const double* b, const double* m_inv,
                           double log_q_const, const double* prior_theta, double eps_prev, double ess_target,
                           int max_bisection_its, double* eps, double* ess, double* w, double* log_p, double* log_q);
int vb_dis_grad_lowrank(vb_ctx* ctx, int64_t n, int64_t d, int64_t k, const double* mu, const double* log_sigma,
                        const double* b, const double* m_inv, double log_q_const, const double* weights, double* out);
int vb_alpha_sums_lowrank(vb_ctx* ctx, int slot_eps, int slot_z, int64_t n, int64_t d, int64_t k, int64_t n_total,
                          double alpha, const double* mu, const double* log_sigma, const double* b, const double* m_inv,
                          double log_q_const, double* value, double* w_sum, double* out);

/* ---- multi-GPU: Monte-Carlo axis sharded, one RCCL all-reduce of the partial sums --- */
#define VB_COMM_ID_BYTES 128
int vb_comm_unique_id(char id[VB_COMM_ID_BYTES]);
int vb_comm_init(vb_ctx* ctx, const char id[VB_COMM_ID_BYTES], int n_ranks, int rank);
int vb_comm_destroy(vb_ctx* ctx);
/* what the attached communicator itself reports (RCCL: ncclCommCount / ncclCommUserRank); 1 / 0 without one */
int vb_comm_info(vb_ctx* ctx, int* n_ranks, int* rank);
/* Host-staged transport: the same sharded job with the caller's own collective in place of RCCL.  Every device
 * collective becomes: copy the vector to pinned host memory, wait for the stream, call `fn`, copy the result back.
 * `fn(user, buf, count, op)` must leave in `buf[0 .. count)` on every rank the elementwise sum (op VB_HOST_SUM) or
 * maximum (VB_HOST_MAX) of the ranks' vectors, combined in rank order, and return 0; it is called on the thread
 * that makes the vb_* call.  For ranks that RCCL cannot join -- two ranks on one GPU (RCCL refuses a duplicate
 * device: that is how the two-rank tests run on a one-GPU box), or a node without xGMI / RCCL -- at the price of a
 * stream synchronisation per collective; the arithmetic on the device is the sharded path's own.  */
#define VB_HOST_SUM 0
#define VB_HOST_MAX 1
typedef int (*vb_host_collective_fn)(void* user, double* buf, size_t count, int op);
int vb_comm_init_host(vb_ctx* ctx, vb_host_collective_fn fn, void* user, int n_ranks, int rank);
/* xGMI-native transport: no ring, no host.  Every rank allocates a window (vb_comm_ipc_window: room for collectives of
 * up to `cap_doubles` values; returns its hipIpcMemHandle), the caller exchanges the handles over its control plane
 * (all-gather of n_ranks x VB_IPC_HANDLE_BYTES bytes), vb_comm_init_ipc maps the peers' windows.  An all-reduce is then
 * three launches on the caller's stream: copy into the own window; reduce the own 1 / n_ranks slice reading the ranks'
 * windows IN RANK ORDER (one rank sums each element, in a fixed order: every rank gets the same bits, whatever the
 * timing -- which a ring does not promise); read the reduced slices back.  Device-side sequence flags in the windows
 * order the phases across ranks (system-scope atomics; bounded polls, a give-up poisons the result with NaN).
 * Functionally tested between two processes on one GPU; the 8-GPU timing is unmeasured.  At most 15 ranks. */
#define VB_IPC_HANDLE_BYTES 64
int vb_comm_ipc_window(vb_ctx* ctx, size_t cap_doubles, char handle[VB_IPC_HANDLE_BYTES]);
int vb_comm_init_ipc(vb_ctx* ctx, const char* handles, int n_ranks, int rank);
/* VB_ERR_COMM when a device-side wait of the IPC transport gave up (a peer did not arrive within VB_IPC_TIMEOUT_S
 * seconds of wall time, default 20; VB_IPC_POLL_LOG2 adds a poll-count bound) -- the affected results are NaN by construction; also checked by the next collective,
 * vb_sync and vb_fullrank_get.  VB_OK on every other transport.                                                     */
int vb_comm_check(vb_ctx* ctx);
/* Duration of the collective alone: `reps` back-to-back sum all-reduces of `count` doubles on the context's stream
 * between two HIP events (after `warm` untimed ones), microseconds per collective.  Collective: every rank calls it with
 * the same arguments.  Without a communicator the "collective" is nothing and the time is the events' own.           */
int vb_comm_allreduce_time(vb_ctx* ctx, size_t count, int warm, int reps, double* us_per_collective);

/* ---- numpy's legacy generator on the host (vb_legacy_rng.cpp) ------------------------------------
 * `numpy.random.RandomState(seed)` restated in C++: MT19937 seeded as numpy seeds it from an integer, the polar-method
 * normals with their one-value cache (`randn`, the noise of approximations.py:203), `standard_t` (:273-274) and
 * `chisquare` (:342) -- values AND generator state bit-identical to numpy's, call after call
 * (tests/test_legacy_rng_cpu.py).  Host code only, no vb_ctx, no GPU.  `randn` beyond 32 768 values evaluates the
 * attempts of the polar method on `threads` host threads (0: $VIABEL_AMD_RNG_THREADS, else up to 8): every attempt
 * consumes exactly four words of the stream, so the words are generated once, sequentially, and the log / sqrt of the
 * transform -- most of numpy's time -- runs in parallel with a prefix sum over the acceptance counts. */
typedef struct vb_legacy_rng vb_legacy_rng;
int vb_legacy_rng_create(uint32_t seed, vb_legacy_rng** out);
void vb_legacy_rng_destroy(vb_legacy_rng* rng);
int vb_legacy_rng_randn(vb_legacy_rng* rng, double* out, int64_t n, int threads);
int vb_legacy_rng_standard_t(vb_legacy_rng* rng, double df, double* out, int64_t n);
int vb_legacy_rng_chisquare(vb_legacy_rng* rng, double df, double* out, int64_t n);
int vb_legacy_rng_random_sample(vb_legacy_rng* rng, double* out, int64_t n);
/* (key[624], pos, has_gauss, cached_gaussian) as `RandomState.get_state()` / `set_state()` carry them */
int vb_legacy_rng_get_state(const vb_legacy_rng* rng, uint32_t key[624], int* pos, int* has_gauss, double* gauss);
int vb_legacy_rng_set_state(vb_legacy_rng* rng, const uint32_t key[624], int pos, int has_gauss, double gauss);
/* The same stream ON THE DEVICE: rows [row_begin, row_begin + rows) of rng.randn(n_total, d) -- what the reference's
 * families draw at approximations.py:203 / :213-216 / :343-347 -- written straight into noise slot `slot` (rows x d),
 * bit for bit numpy's values, and `rng` advanced exactly as randn(n_total, d) advances it (position, cached value of
 * an odd count): MT19937 in parallel streams through jump-ahead polynomials, the polar method's attempts, their
 * prefix sums and the scatter on the device; only the attempts whose logarithm the device cannot pin to the host C
 * library's rounding (about one in forty) are finished on the host.  VB_ERR_UNSUPPORTED (request beyond 1024 streams,
 * ~64 M values): nothing was changed, draw with vb_legacy_rng_randn and upload.                                    */
int vb_legacy_rng_randn_device(vb_ctx* ctx, vb_legacy_rng* rng, int slot, int64_t n_total, int64_t d, int64_t row_begin,
                               int64_t rows);
/* RandomState.standard_t(df, (n_total, d)) -- MFStudentT's noise, approximations.py:273-274 -- rows [row_begin,
 * row_begin + rows) into noise slot `slot`, and RandomState.chisquare(df, n) -- MultivariateT's scales, :345 -- into
 * the context's chi-square buffer (what vb_chisq_generate fills; also copied to host_out unless NULL): values and
 * generator state bit for bit numpy's.  The Marsaglia-Tsang rejection loop is evaluated for every possible state of
 * the stream in parallel and the true trajectory is stitched by composing per-chunk transition maps
 * (vb_legacy_gamma.hip); its logarithms are the host C library's own operation sequence (vb_legacy_rng_log_proven).
 * VB_ERR_UNSUPPORTED (df <= 2, log not proven on this host, request beyond the jump ladder): nothing was changed,
 * draw with vb_legacy_rng_standard_t / _chisquare.                                                                */
int vb_legacy_rng_standard_t_device(vb_ctx* ctx, vb_legacy_rng* rng, double df, int slot, int64_t n_total, int64_t d,
                                    int64_t row_begin, int64_t rows);
int vb_legacy_rng_chisquare_device(vb_ctx* ctx, vb_legacy_rng* rng, double df, int64_t n, double* host_out);
/* Look-ahead generation of the NEXT call's draws (round 6).  A family in the reference-identical mode makes the same device
 * draws call after call from one persistent generator (approximations.py:213-216, :270-274, :342-349).  The binding calls
 * vb_legacy_round_end(ctx, rng) when a call's draws are done: if the round just ended asked for what the round before it
 * asked for, the same requests are started from the generator's CURRENT state on a stream of their own, into shadow
 * buffers -- beside the objective's kernels that follow on the main stream.  The next vb_legacy_rng_*_device call that
 * finds the generator in exactly that state (all 624 words, position, cached normal) and asks for exactly that draw takes
 * the shadow by a pointer swap and sets the generator to the speculated end state; any other request, or a generator that
 * moved in between (a host draw, a reseed, set_state), discards the speculation and draws as before.  Values and generator
 * states are numpy's either way (tests/test_gpu_legacy_rng.py).  VB_LEGACY_AHEAD=0 turns it off.  A binding that does
 * not call vb_legacy_round_end gets no speculation.  vb_legacy_ahead_stats: jobs launched, requests adopted, jobs discarded. */
int vb_legacy_round_end(vb_ctx* ctx, vb_legacy_rng* rng);
int vb_legacy_ahead_stats(vb_ctx* ctx, uint64_t* launched, uint64_t* adopted, uint64_t* discarded);
/* 1 when this host's libm log() has been located and restated bit for bit (vb_glibc_log.h): the device draws above are
 * available and vb_legacy_rng_randn_device needs no host round trip.                                               */
int vb_legacy_rng_log_proven(void);
/* a number unique to this generator object for the life of the process (0 for NULL): a destroyed generator's address may be
 * handed out again, its uid never (the look-ahead draws key their per-generator history on it).                         */
uint64_t vb_legacy_rng_uid(const vb_legacy_rng* rng);

/* ---- measurement hooks (bench.py): HIP-event timing of the dominant kernels ---------
 * When enabled, every launch of a profiled kernel carries a start/stop event pair
 * (hipExtLaunchKernel: the kernel's own begin/end timestamps on the context's stream).
 * vb_profile_read returns launches, evaluations covered and total milliseconds of the mean-field
 * accumulation kernel (VB_PROF_MF_ACCUM) since the last reset; vb_profile_read_kernel does the same
 * for one kernel id and leaves the records of the other kernels alone.                   */
#define VB_PROF_MF_ACCUM 0        /* mean-field streaming kernel (mf_accum_kernel)                    */
#define VB_PROF_FR_SAMPLE_GEMM 1  /* dense family: Z = E L' + mu (triangular k range)                 */
#define VB_PROF_FR_MODEL_GEMM 2   /* correlated-Gaussian target: G = -(Z - m) P (dense)               */
#define VB_PROF_FR_GRAD_GEMM 3    /* dense family: C = G' E (lower-triangular tiles, split over rows) */
#define VB_PROF_NUM 4
int vb_profile_enable(vb_ctx* ctx, int on);
int vb_profile_read(vb_ctx* ctx, int64_t* launches, int64_t* evals, double* total_ms, int reset);
int vb_profile_read_kernel(vb_ctx* ctx, int kernel_id, int64_t* launches, int64_t* evals, double* total_ms,
                           int reset);

#ifdef __cplusplus
}
#endif
#endif /* VIABEL_HIP_H */
