// Per-sample (per-row) log weights for the mean-field families, and the reductions over them.
//
//   lw_n = f(z_n) - log q(z_n; theta),  z_n = mu + sigma * eps_n
// is what viabel/objectives.py:443-446 (AlphaDivergence.compute_log_weights) and
// objectives.py:393-395 (DISInclusiveKL state refresh: log q, log p) evaluate through
// approx.sample / approx.log_density / model; here one wave reduces one row of the noise matrix
// (coalesced 16-B loads, shuffle reduction), so the N x D samples are never materialised.
//
// Kernels
//   rs_cols_kernel      sigma = exp(log_sigma), sum(log_sigma)                          O(D)
//   rs_rowstats_kernel  f_n and the base log-density sum of row n                       streams eps
//   alpha_weights_kernel  max / exp / sum over the N log weights (one workgroup)        O(N)
#include "vb_common.h"

#include <vector>

namespace vb {

typedef double d2r __attribute__((ext_vector_type(2)));

constexpr double kLog2PiRs = 1.8378770664093454835606594728112;

__device__ __forceinline__ double rs_wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}
__device__ __forceinline__ double rs_wave_max(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x = fmax(x, __shfl_down(x, off, 64));
  return x;
}

// cols = [mu (ld) | sigma (ld)], pads zero; scal[0] = sum(log_sigma)
__global__ void __launch_bounds__(256) rs_cols_kernel(const double* __restrict__ theta_src, int d, int64_t ld,
                                                      double* __restrict__ cols, double* __restrict__ scal) {
  __shared__ double sh[4];
  double t = 0.0;
  for (int64_t i = threadIdx.x; i < ld; i += 256) {
    double mu = 0.0, sg = 0.0;
    if (i < d) {
      mu = theta_src[i];
      const double ls = theta_src[d + i];
      sg = exp(ls);
      t += ls;
    }
    cols[i] = mu;
    cols[ld + i] = sg;
  }
  t = rs_wave_sum(t);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) scal[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// One wave per row.  out_f[n] = f(z_n); out_b[n] = sum_d log base_pdf(eps_nd) (Gaussian or Student t).
// `m` is the model whose log density is wanted (the target, or a diagonal Gaussian tempering prior).
__global__ void __launch_bounds__(256) rs_rowstats_kernel(const double* __restrict__ noise, int64_t ld,
                                                          int64_t n, int d, const double* __restrict__ cols,
                                                          ModelDev m, int student, double df,
                                                          double* __restrict__ out_f,
                                                          double* __restrict__ out_b,
                                                          const double* __restrict__ q0 = nullptr,
                                                          const double* __restrict__ q1 = nullptr, double qc = 0.0,
                                                          double* __restrict__ out_q = nullptr) {
  // q0 != nullptr: the same pass also evaluates a diagonal Gaussian (mean q0, inverse variances q1, constant qc) at the
  // samples into out_q -- DISInclusiveKL's tempering prior next to the target, one read of the noise instead of two
  // (the sum runs in the order the second pass used: the same bits)
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const double* e_row = noise + row * ld;
  const double* mu = cols;
  const double* sg = cols + ld;
  const bool funnel = m.id == VB_MODEL_FUNNEL;
  double v = 0.0;
  if (funnel) v = fma(sg[m.k], e_row[m.k], mu[m.k]);
  double af = 0.0, ab = 0.0, aq = 0.0;
  for (int64_t c = 2 * lane; c < ld; c += 128) {   // columns [d, ld) are zero pads of noise and cols
    const d2r e = *reinterpret_cast<const d2r*>(e_row + c);
    const d2r mm = *reinterpret_cast<const d2r*>(mu + c);
    const d2r ss = *reinterpret_cast<const d2r*>(sg + c);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t col = c + h;
      const double eh = h ? e.y : e.x;
      const double z = fma(h ? ss.y : ss.x, eh, h ? mm.y : mm.x);
      if (col < d) {
        if (funnel) {
          if (col != m.k) af = fma(z, z, af);
        } else {
          const double dz = z - m.p0[col];
          af = fma(-0.5 * dz * dz, m.p1[col], af);
        }
        if (q0) {
          const double dq = z - q0[col];
          aq = fma(-0.5 * dq * dq, q1[col], aq);
        }
      }
      ab += student ? -0.5 * (df + 1.0) * log1p(eh * eh / df) : -0.5 * eh * eh;
    }
  }
  af = rs_wave_sum(af);
  ab = rs_wave_sum(ab);
  if (q0) aq = rs_wave_sum(aq);
  if (lane == 0) {
    if (q0) out_q[row] = aq + qc;
    double f;
    if (funnel) {
      const double it2 = 1.0 / (m.tau * m.tau), dm1 = (double)(d - 1);
      f = v * fma(-0.5 * v, it2, -dm1) - 0.5 * exp(-2.0 * v) * af + m.c0;
    } else {
      f = af + m.c0;
    }
    const double cb = student ? lgamma(0.5 * (df + 1.0)) - lgamma(0.5 * df) - 0.5 * log(df * M_PI)
                              : -0.5 * kLog2PiRs;
    out_f[row] = f;
    out_b[row] = ab + d * cb;
  }
}

// objectives.py:453-459: lw = f - log q, log_norm = max lw, s = exp(lw - log_norm)^alpha,
// value = log(mean s)/alpha + log_norm.  log q(z_n) = b_n - sum(log_sigma).  Three small one-workgroup
// kernels so that a sharded job can all-reduce the max and the sum in between.
__global__ void __launch_bounds__(1024) alpha_max_kernel(const double* __restrict__ f, const double* __restrict__ b,
                                                         const double* __restrict__ scal_in, int64_t n,
                                                         double* __restrict__ mx_out) {
  __shared__ double sh[16];
  const double sum_ls = scal_in[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double mx = -INFINITY;
  for (int64_t i = threadIdx.x; i < n; i += 1024) mx = fmax(mx, f[i] - b[i] + sum_ls);
  mx = rs_wave_max(mx);
  if (lane == 0) sh[wave] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = sh[0];
    for (int w = 1; w < 16; ++w) mx = fmax(mx, sh[w]);
    mx_out[0] = mx;
  }
}

__global__ void __launch_bounds__(1024) alpha_weights_kernel(const double* __restrict__ f,
                                                             const double* __restrict__ b,
                                                             const double* __restrict__ scal_in,
                                                             const double* __restrict__ mx_in, int64_t n,
                                                             double alpha, double* __restrict__ roww,
                                                             double* __restrict__ sum_out) {
  __shared__ double sh[16];
  const double sum_ls = scal_in[0], mx = mx_in[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double sv = exp(alpha * (f[i] - b[i] + sum_ls - mx));
    roww[i] = sv;
    s += sv;
  }
  s = rs_wave_sum(s);
  if (lane == 0) sh[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < 16; ++w) tot += sh[w];
    sum_out[0] = tot;
  }
}

__global__ void alpha_value_kernel(const double* mx, const double* sum_s, double n_total, double alpha,
                                   double* value) {
  value[0] = log(sum_s[0] / n_total) / alpha + mx[0];
}

// ---- host ---------------------------------------------------------------------------------------------
static int alpha_scalars_enqueue(vb_ctx* ctx, const double* f, const double* b, double* scal, int64_t n, int64_t n_total,
                                 double alpha, double* roww);
// Z = mu + exp(log_sigma) * E, materialised for a source model's row kernel (the built-in targets never need it)
__global__ void __launch_bounds__(256) rs_sample_kernel(const double* __restrict__ theta_src, const double* __restrict__ noise,
                                                        int64_t ld, double* __restrict__ Z, int64_t ldz, int64_t n, int d) {
  const int64_t row = blockIdx.x;
  const int col = blockIdx.y * 256 + threadIdx.x;
  if (col >= d) return;
  Z[row * ldz + col] = fma(exp(theta_src[d + col]), noise[row * ld + col], theta_src[col]);
}

int rowstats_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, const double* theta_src,
                     const ModelDev& model, int student, double df, double* cols, double* scal,
                     double* out_f, double* out_b, const ModelDev* prior, double* out_prior) {
  hipLaunchKernelGGL(rs_cols_kernel, dim3(1), dim3(256), 0, ctx->stream, theta_src, (int)d, ns.ld, cols, scal);
  VB_HIP(ctx, hipGetLastError());
  if (model.id == VB_MODEL_SOURCE) {
    // the row kernel forms the base log density b_n as for any target (its own f is thrown away: a diagonal
    // Gaussian over memory that exists); f_n comes from the user's kernel on the materialised samples
    ModelDev stand_in;
    stand_in.id = VB_MODEL_GAUSS_DIAG;
    stand_in.dim = model.dim;
    stand_in.p0 = cols;
    stand_in.p1 = cols;
    hipLaunchKernelGGL(rs_rowstats_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream,
                       (const double*)ns.buf.ptr, ns.ld, n, (int)d, (const double*)cols, stand_in, student, df,
                       out_f, out_b);
    VB_HIP(ctx, hipGetLastError());
    const int64_t ldz = round_up(d, 16);
    VB_TRY(ensure(ctx, ctx->lg_work, (size_t)n * ldz * sizeof(double)));
    double* Z = (double*)ctx->lg_work.ptr;
    hipLaunchKernelGGL(rs_sample_kernel, dim3((unsigned)n, (unsigned)((d + 255) / 256)), dim3(256), 0, ctx->stream,
                       theta_src, (const double*)ns.buf.ptr, ns.ld, Z, ldz, n, (int)d);
    VB_HIP(ctx, hipGetLastError());
    return user_rows_enqueue(ctx, ctx->stream, Z, ldz, n, (int)d, nullptr, 0, out_f);
  }
  hipLaunchKernelGGL(rs_rowstats_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream,
                     (const double*)ns.buf.ptr, ns.ld, n, (int)d, (const double*)cols, model, student, df,
                     out_f, out_b, prior ? prior->p0 : (const double*)nullptr, prior ? prior->p1 : (const double*)nullptr,
                     prior ? prior->c0 : 0.0, prior ? out_prior : (double*)nullptr);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// AlphaDivergence value and gradient for the mean-field families (objectives.py:453-461).  `n` rows are
// local; with a communicator the max and the sum of the weights are all-reduced (n_total = global N).
int alpha_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t n_total, int64_t d, int family, double df,
                  double alpha, const double* theta_src, double* out) {
  if (!(alpha != 0.0)) return fail(ctx, VB_ERR_INVALID, "alpha must be non-zero");
  if (ctx->model.id != VB_MODEL_GAUSS_DIAG && ctx->model.id != VB_MODEL_FUNNEL && ctx->model.id != VB_MODEL_SOURCE)
    return fail(ctx, VB_ERR_UNSUPPORTED, "mean-field alpha-divergence supports the gauss_diag, funnel and source models");
  if (ctx->model.dim != d) return fail(ctx, VB_ERR_INVALID, "model dimension != family dimension");
  if (n <= 0 || n > ns.n || d != ns.d || n_total < n) return fail(ctx, VB_ERR_INVALID, "noise slot shape mismatch");
  const int student = family == VB_FAMILY_MF_STUDENT_T;
  // scratch: [cols 2 ld | scal 16 | f n | b n | roww n];  scal: [0] sum log sigma, [8] max, [9] sum s, [10] value
  const int64_t o_scal = 2 * ns.ld, o_f = o_scal + 16, o_b = o_f + round_up(n, 16), o_w = o_b + round_up(n, 16);
  VB_TRY(ensure(ctx, ctx->rowvec, (size_t)(o_w + round_up(n, 16)) * sizeof(double)));
  double* base = (double*)ctx->rowvec.ptr;
  double* scal = base + o_scal;
  VB_TRY(rowstats_enqueue(ctx, ns, n, d, theta_src, ctx->model, student, df, base, scal, base + o_f, base + o_b));
  VB_TRY(alpha_scalars_enqueue(ctx, base + o_f, base + o_b, scal, n, n_total, alpha, base + o_w));      // max -> weights -> value
  MfCall c;
  c.count = 1;
  c.noise[0] = &ns;
  c.theta_src[0] = theta_src;
  c.out[0] = out;
  c.roww[0] = base + o_w;
  c.n = n;
  c.d = d;
  c.n_total = n_total;
  c.family = family;
  c.df = df;
  c.mode = 1;
  c.scale = alpha / (double)n_total;          // objectives.py:460: alpha * vjp / N
  c.value_src = scal + 10;
  return mf_enqueue(ctx, c);
}

// b_n = sum_d log N(eps_nd; 0, 1): the base log density of a row of standard-normal noise (one wave per row)
__global__ void __launch_bounds__(256) rs_normal_base_kernel(const double* __restrict__ E, int64_t ld, int64_t n, int d,
                                                             double* __restrict__ b) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const double* e = E + row * ld;
  double s = 0.0;
  for (int c = lane; c < d; c += 64) s = fma(e[c], e[c], s);
  s = rs_wave_sum(s);
  if (lane == 0) b[row] = -0.5 * s - 0.5 * d * 1.8378770664093454835606594728112;
}

__global__ void rs_set_scalar_kernel(double* dst, double v) { dst[0] = v; }
__global__ void rs_mvt_base_kernel(const double* __restrict__ E, int64_t ld, int64_t n, int d, double df, double lconst,
                                   const double* __restrict__ inv_s, double* __restrict__ b);

// objectives.py:453-459 in ONE workgroup when no communicator sits between the steps: max, weights and their sum, value
__global__ void __launch_bounds__(1024) alpha_one_rank_kernel(const double* __restrict__ f, const double* __restrict__ b,
                                                              const double* __restrict__ scal_in, int64_t n, double alpha,
                                                              double n_total, double* __restrict__ roww,
                                                              double* __restrict__ out /* [max, sum s, value] */) {
  __shared__ double sh[16];
  __shared__ double bc;
  const double sum_ls = scal_in[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double mx = -INFINITY;
  for (int64_t i = threadIdx.x; i < n; i += 1024) mx = fmax(mx, f[i] - b[i] + sum_ls);
  mx = rs_wave_max(mx);
  if (lane == 0) sh[wave] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m2 = sh[0];
    for (int w = 1; w < 16; ++w) m2 = fmax(m2, sh[w]);
    bc = m2;
  }
  __syncthreads();
  mx = bc;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double sv = exp(alpha * (f[i] - b[i] + sum_ls - mx));
    roww[i] = sv;
    s += sv;
  }
  s = rs_wave_sum(s);
  __syncthreads();
  if (lane == 0) sh[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < 16; ++w) tot += sh[w];      // the order alpha_weights_kernel adds them in
    out[0] = mx;
    out[1] = tot;
    out[2] = log(tot / n_total) / alpha + mx;
  }
}

// max -> weights -> value of the dense families' alpha pipelines (scal: [0] sum log scale, [8] max, [9] sum s, [10] value)
static int alpha_scalars_enqueue(vb_ctx* ctx, const double* f, const double* b, double* scal, int64_t n, int64_t n_total,
                                 double alpha, double* roww) {
  hipStream_t st = ctx->stream;
  if (!ctx->comm) {
    hipLaunchKernelGGL(alpha_one_rank_kernel, dim3(1), dim3(1024), 0, st, f, b, (const double*)scal, n, alpha,
                       (double)n_total, roww, scal + 8);
    VB_HIP(ctx, hipGetLastError());
    return VB_OK;
  }
  hipLaunchKernelGGL(alpha_max_kernel, dim3(1), dim3(1024), 0, st, f, b, (const double*)scal, n, scal + 8);
  VB_HIP(ctx, hipGetLastError());
  VB_TRY(comm_allreduce_max(ctx, st, scal + 8, 1));
  hipLaunchKernelGGL(alpha_weights_kernel, dim3(1), dim3(1024), 0, st, f, b, (const double*)scal,
                     (const double*)(scal + 8), n, alpha, roww, scal + 9);
  VB_HIP(ctx, hipGetLastError());
  VB_TRY(comm_allreduce_sum(ctx, st, scal + 9, 1));
  hipLaunchKernelGGL(alpha_value_kernel, dim3(1), dim3(1), 0, st, (const double*)(scal + 8), (const double*)(scal + 9),
                     (double)n_total, alpha, scal + 10);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// AlphaDivergence for the dense Gaussian family (objectives.py:453-461 with z = mu + L eps):
// log q(z_n) = b_n - sum log L_ii, so the weights follow from per-row f (model_logp_rows on the samples) exactly
// as for the mean-field families; the gradient is the entropy-form pipeline with the rows of G weighted.
// df > 0 with inv_s: the multivariate t sampled through its Cholesky factor, x = mu + (L eps) / s (throughput mode): the
// same pipeline with the rows scaled by 1 / s_n and the t family's base density -- the gradient leaves in the flat
// free-Cholesky layout, no matrix root and no D x D array on the host.
int alpha_fullrank_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t n_total, int64_t d, double alpha,
                           const double* theta_dev, double sum_log_diag, double* out, double df, const double* inv_s) {
  if (!(alpha != 0.0)) return fail(ctx, VB_ERR_INVALID, "alpha must be non-zero");
  if (n <= 0 || n > ns.n || d != ns.d || n_total < n) return fail(ctx, VB_ERR_INVALID, "noise slot shape mismatch");
  hipStream_t st = ctx->stream;
  const int64_t ldz = round_up(d, 16), nn = round_up(n, 16);
  // rowvec: [scal 16 | f n | b n | roww n];  scal: [0] sum log L_ii, [8] max, [9] sum s, [10] value
  VB_TRY(ensure(ctx, ctx->rowvec, (size_t)(16 + 3 * nn) * sizeof(double)));
  VB_TRY(ensure(ctx, ctx->scratch, (size_t)(n * ldz) * sizeof(double)));
  double* scal = (double*)ctx->rowvec.ptr;
  double *f = scal + 16, *b = f + nn, *roww = b + nn;
  double* Z = (double*)ctx->scratch.ptr;
  // (the pad columns of the sample matrix must hold zeros: the sampling product writes columns [0, d) only)
  if (ldz != d) VB_HIP(ctx, hipMemsetAsync(Z, 0, (size_t)(n * ldz) * sizeof(double), st));
  hipLaunchKernelGGL(rs_set_scalar_kernel, dim3(1), dim3(1), 0, st, scal, sum_log_diag);      // (no copy, no synchronisation)
  VB_TRY(fr_sample_enqueue(ctx, ns, n, d, theta_dev, Z, nullptr, nullptr, inv_s));
  const double* g_ready = nullptr;
  if (ctx->model.id == VB_MODEL_GAUSS_FULL && !inv_s) {
    // f of the correlated Gaussian costs the N x D x D product that also IS its gradient: keep G, the pipeline below
    // then needs neither its sampling product nor its model product
    VB_TRY(ensure(ctx, ctx->alpha_g, (size_t)(n * ldz) * sizeof(double)));
    VB_TRY(model_grad_rows(ctx, Z, ldz, n, d, (double*)ctx->alpha_g.ptr, f));
    g_ready = (const double*)ctx->alpha_g.ptr;
  } else {
    VB_TRY(model_logp_rows(ctx, Z, ldz, n, d, f));
  }
  if (inv_s) {
    const double lconst = lgamma(0.5 * (df + (double)d)) - lgamma(0.5 * df) - 0.5 * (double)d * log(M_PI * df);
    hipLaunchKernelGGL(rs_mvt_base_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, (const double*)ns.buf.ptr,
                       ns.ld, n, (int)d, df, lconst, inv_s, b);
  } else {
    hipLaunchKernelGGL(rs_normal_base_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, (const double*)ns.buf.ptr,
                       ns.ld, n, (int)d, b);
  }
  VB_HIP(ctx, hipGetLastError());
  VB_TRY(alpha_scalars_enqueue(ctx, f, b, scal, n, n_total, alpha, roww));
  FrWeighted wm;
  wm.roww = roww;
  wm.scale = alpha / (double)n_total;          // objectives.py:460: alpha * vjp / N
  wm.wsum = scal + 9;
  wm.value = scal + 10;
  wm.z_ready = Z;
  wm.g_ready = g_ready;
  if (inv_s)
    return fr_pipeline_enqueue(ctx, ns, n, d, n_total, theta_dev, out, nullptr, nullptr, inv_s, nullptr, 0, &wm);
  return fr_elbo_grad_enqueue(ctx, ns, n, d, n_total, theta_dev, out, 0, &wm);
}

// b_n = log t_df(x_n) + sum log L_ii for x_n = mu + (e_n R) / s_n: the Mahalanobis distance is |e_n|^2 / s_n^2
// whatever (mu, R) are (_distributions.py:7-38); one wave per row
__global__ void __launch_bounds__(256) rs_mvt_base_kernel(const double* __restrict__ E, int64_t ld, int64_t n, int d,
                                                          double df, double lconst,
                                                          const double* __restrict__ inv_s, double* __restrict__ b) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const double* e = E + row * ld;
  double s = 0.0;
  for (int c = lane; c < d; c += 64) s = fma(e[c], e[c], s);
  s = rs_wave_sum(s);
  if (lane == 0) {
    const double is = inv_s[row];
    b[row] = lconst - 0.5 * (df + d) * log1p(s * is * is / df);
  }
}

// AlphaDivergence for the multivariate t (objectives.py:453-461 with approximations.py:342-357): weights exactly
// as for the dense Gaussian, then the t family's pipeline with the rows of G weighted.  Returns the raw sums
// [. | sum w g | sum w g (e / s)'] and a pointer to [sum w, value]; the chain rule through the root is the caller's.
int alpha_mvt_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t n_total, int64_t d, double df, double alpha,
                      const double* mu_dev, const double* root_dev, const double* invs_dev, double sum_log_diag,
                      FrSums* sums, const double** value_wsum) {
  if (!(alpha != 0.0)) return fail(ctx, VB_ERR_INVALID, "alpha must be non-zero");
  if (n <= 0 || n > ns.n || d != ns.d || n_total < n) return fail(ctx, VB_ERR_INVALID, "noise slot shape mismatch");
  hipStream_t st = ctx->stream;
  const int64_t ldz = round_up(d, 16), nn = round_up(n, 16);
  VB_TRY(ensure(ctx, ctx->rowvec, (size_t)(16 + 3 * nn) * sizeof(double)));
  VB_TRY(ensure(ctx, ctx->scratch, (size_t)(n * ldz) * sizeof(double)));
  double* scal = (double*)ctx->rowvec.ptr;
  double *f = scal + 16, *b = f + nn, *roww = b + nn;
  double* Z = (double*)ctx->scratch.ptr;
  if (ldz != d) VB_HIP(ctx, hipMemsetAsync(Z, 0, (size_t)(n * ldz) * sizeof(double), st));
  hipLaunchKernelGGL(rs_set_scalar_kernel, dim3(1), dim3(1), 0, st, scal, sum_log_diag);
  VB_TRY(fr_sample_enqueue(ctx, ns, n, d, nullptr, Z, mu_dev, root_dev, invs_dev));
  VB_TRY(model_logp_rows(ctx, Z, ldz, n, d, f));
  const double lconst = lgamma(0.5 * (df + (double)d)) - lgamma(0.5 * df) - 0.5 * (double)d * log(M_PI * df);
  hipLaunchKernelGGL(rs_mvt_base_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, (const double*)ns.buf.ptr,
                     ns.ld, n, (int)d, df, lconst, invs_dev, b);
  VB_HIP(ctx, hipGetLastError());
  VB_TRY(alpha_scalars_enqueue(ctx, f, b, scal, n, n_total, alpha, roww));
  FrWeighted wm;
  wm.roww = roww;
  wm.scale = alpha / (double)n_total;
  wm.wsum = scal + 9;
  wm.value = scal + 10;
  wm.z_ready = Z;
  *value_wsum = scal + 9;                      // [sum w, value]
  return fr_pipeline_enqueue(ctx, ns, n, d, n_total, nullptr, nullptr, mu_dev, root_dev, invs_dev, sums, 0, &wm);
}

// ---- DISInclusiveKL state refresh (objectives.py:317-368) ---------------------------------------------
// Bisection on the tempering parameter for the effective sample size.
//   logw(e) = e * log prior + (1 - e) * log p - log q,   w = exp(logw)   (no max shift, :330)
//   ESS = (sum w)^2 / sum w^2   (:333-336);  50 bisection steps on [0, eps_prev], end-point snapping.
// The reference evaluates its 51 candidates one after the other; each needs all N weights, so a literal
// restatement is 51 dependent passes on one CU (424 us at N = 16 384, bound by 51 N fp64 `exp`).  Here the next
// kLook = 6 levels are looked ahead: one launch evaluates the ESS of all 2^6 - 1 = 63 midpoints the next six
// decisions can possibly visit (one workgroup per midpoint, the N weights strided over its 256 threads), and the
// following launch first replays those six decisions -- the same comparisons on the same numbers, in the same
// order -- to find its own interval.  Midpoints are formed by the reference's own expression (lower + upper) / 2
// from the same end points, so every candidate the reference visits is evaluated at the bit-identical eps, and the
// "all weights zero" error (:325-328) is raised only for candidates on the visited path.  9 launches + a final
// one that writes the weights: 51 x N exps on one CU become 63 x N / 63 per launch on 63 CUs.
// dis_state: [lower, upper, status] at the start of a round; res: [round][node] = {ess, max logw}.
#ifndef VB_LOOK
#define VB_LOOK 6
#define VB_LOOK_PARTS 4
#endif
constexpr int kLook = VB_LOOK;
constexpr int kLookNodes = 1 << kLook;       // heap order, node 1 = the interval's midpoint; index 0 unused
constexpr int kLookParts = VB_LOOK_PARTS;    // workgroups per candidate (the N samples in kLookParts blocks)
constexpr int kLookTab = kLookNodes * kLookParts * 3;   // [node][part]{sum w, sum w^2, max log w}

struct BisectWalk {
  double lower, upper;
  int status;
};
// replay `levels` decisions of a finished round (objectives.py:349-357); tab = that round's table
__device__ __forceinline__ BisectWalk bisect_replay(double lower, double upper, int status, int levels,
                                                    const double* tab, double ess_target) {
  int node = 1;
  for (int l = 0; l < levels; ++l) {
    const double guess = (lower + upper) / 2.0;
    double t1 = 0.0, t2 = 0.0, mx = -INFINITY;
#pragma unroll
    for (int p = 0; p < kLookParts; ++p) {          // fixed order: the ESS of a candidate is reproducible
      const double* e = tab + (node * kLookParts + p) * 3;
      t1 += e[0];
      t2 += e[1];
      mx = fmax(mx, e[2]);
    }
    if (mx == -INFINITY) status = 1;
    if (t1 * t1 / t2 > ess_target) {
      upper = guess;
      node = 2 * node;
    } else {
      lower = guess;
      node = 2 * node + 1;
    }
  }
  return BisectWalk{lower, upper, status};
}

// stage the previous round's table and interval in LDS with one coalesced fetch (replaying from global memory
// would be six dependent loads of freshly written lines)
__device__ __forceinline__ BisectWalk bisect_stage_replay(double* tab, const double* __restrict__ res_in,
                                                          const double* __restrict__ state_in, int prev_levels,
                                                          double ess_target, double eps_prev) {
  for (int e = threadIdx.x; e < kLookTab; e += blockDim.x) tab[e] = prev_levels > 0 ? res_in[e] : 0.0;
  // first round: the interval [0, eps_guess] (:346) comes as a kernel argument (an upload of it would put a host
  // synchronisation in the middle of the refresh)
  if (threadIdx.x < 3)
    tab[kLookTab + threadIdx.x] = prev_levels > 0 ? state_in[threadIdx.x] : (threadIdx.x == 1 ? eps_prev : 0.0);
  __syncthreads();
  return bisect_replay(tab[kLookTab], tab[kLookTab + 1], (int)tab[kLookTab + 2], prev_levels, tab, ess_target);
}

// sums of one candidate over the samples [i_begin, i_end): four independent elements per trip, so the dependent
// fp64 chains of `exp` overlap (with the 16 waves of a 1024-thread workgroup: four per SIMD)
template <bool STORE>
__device__ __forceinline__ void bisect_sums(const double* __restrict__ lp, const double* __restrict__ b,
                                            const double* __restrict__ lprior, double sum_ls, double guess,
                                            int64_t i_begin, int64_t i_end, double* __restrict__ w,
                                            double* __restrict__ lq_out, double* sh, double* out3) {
  double s1 = 0.0, s2 = 0.0, mx = -INFINITY;
  for (int64_t i0 = i_begin + threadIdx.x; i0 < i_end; i0 += 4 * 1024) {
    double lw[4], lq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * 1024;
      lq[u] = i < i_end ? b[i] - sum_ls : 0.0;
      lw[u] = i < i_end ? guess * lprior[i] + (1.0 - guess) * lp[i] - lq[u] : -INFINITY;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * 1024;
      if (i < i_end) {
        const double wv = exp(lw[u]);
        mx = fmax(mx, lw[u]);
        s1 += wv;
        s2 = fma(wv, wv, s2);
        if (STORE) {
          w[i] = wv;
          lq_out[i] = lq[u];
        }
      }
    }
  }
  s1 = rs_wave_sum(s1);
  s2 = rs_wave_sum(s2);
  mx = rs_wave_max(mx);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) {
    sh[wave] = s1;
    sh[16 + wave] = s2;
    sh[32 + wave] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t1 = 0.0, t2 = 0.0, tm = -INFINITY;
    for (int k = 0; k < 16; ++k) {
      t1 += sh[k];
      t2 += sh[16 + k];
      tm = fmax(tm, sh[32 + k]);
    }
    out3[0] = t1;
    out3[1] = t2;
    out3[2] = tm;
  }
}

// grid = candidates x kLookParts; workgroup (c, part) sums block `part` of the samples for heap node c + 1
__global__ void __launch_bounds__(1024) dis_bisect_round_kernel(const double* __restrict__ lp,
                                                                const double* __restrict__ b,
                                                                const double* __restrict__ lprior,
                                                                const double* __restrict__ scal_in, int64_t n,
                                                                double ess_target, int prev_levels, int levels,
                                                                const double* __restrict__ state_in,
                                                                const double* __restrict__ res_in,
                                                                double* __restrict__ state_out,
                                                                double* __restrict__ res_out, double eps_prev) {
  __shared__ double sh[48];
  __shared__ double tab[kLookTab + 4];
  const double sum_ls = scal_in[0];
  const BisectWalk wk = bisect_stage_replay(tab, res_in, state_in, prev_levels, ess_target, eps_prev);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    state_out[0] = wk.lower;
    state_out[1] = wk.upper;
    state_out[2] = (double)wk.status;
  }
  const int node = blockIdx.x / kLookParts + 1, part = blockIdx.x % kLookParts;
  int depth = 0;
  while ((node >> (depth + 1)) != 0) ++depth;       // node = 1 b_{depth-1} ... b_0
  double lower = wk.lower, upper = wk.upper;
  for (int l = depth - 1; l >= 0; --l) {
    const double guess = (lower + upper) / 2.0;
    if ((node >> l) & 1) lower = guess;
    else upper = guess;
  }
  const double guess = (lower + upper) / 2.0;
  const int64_t per = (n + kLookParts - 1) / kLookParts;
  const int64_t i_begin = part * per, i_end = i_begin + per < n ? i_begin + per : n;
  bisect_sums<false>(lp, b, lprior, sum_ls, guess, i_begin, i_end, nullptr, nullptr, sh,
                     res_out + (node * kLookParts + part) * 3);
}

// last step (:358-366): replay the final round, evaluate the weights at the final midpoint, snap eps to the ends.
// scal_out = [eps, ess, status]; status 1 = "all weights zero" (max logw == -inf, :325-328).
__global__ void __launch_bounds__(1024) dis_bisect_final_kernel(const double* __restrict__ lp,
                                                                const double* __restrict__ b,
                                                                const double* __restrict__ lprior,
                                                                const double* __restrict__ scal_in, int64_t n,
                                                                double ess_target, int prev_levels,
                                                                const double* __restrict__ state_in,
                                                                const double* __restrict__ res_in, double max_eps,
                                                                double* __restrict__ w, double* __restrict__ lq_out,
                                                                double* __restrict__ scal_out, double eps_prev) {
  __shared__ double sh[48];
  __shared__ double tab[kLookTab + 4];
  __shared__ double tot[3];
  const double sum_ls = scal_in[0];
  const BisectWalk wk = bisect_stage_replay(tab, res_in, state_in, prev_levels, ess_target, eps_prev);
  const double guess = (wk.lower + wk.upper) / 2.0;
  bisect_sums<true>(lp, b, lprior, sum_ls, guess, 0, n, w, lq_out, sh, tot);
  if (threadIdx.x == 0) {
    double eps = guess;
    if (wk.lower == 0.0) eps = 0.0;          // :363-366
    if (wk.upper == max_eps) eps = max_eps;
    scal_out[0] = eps;
    scal_out[1] = tot[0] * tot[0] / tot[1];
    scal_out[2] = (double)((wk.status != 0 || tot[2] == -INFINITY) ? 1 : 0);
  }
}

// (round 4: dis_bisect_enqueue is the speculative walk of vb_dis_bisect.hip; this one stays as its cross-check,
// VB_DIS_BISECT=0)
int dis_bisect_lookahead_enqueue(vb_ctx* ctx, const double* lp, const double* b, const double* lprior,
                                 const double* scal_in, int64_t n, double eps_prev, double ess_target, int max_its,
                                 double* w, double* lq_out, double* scal_out) {
  if (max_its < 0) return fail(ctx, VB_ERR_INVALID, "max_its must be >= 0");
  const int rounds = (max_its + kLook - 1) / kLook;
  // [state (rounds + 1) x 4 | res rounds x kLookTab]
  const size_t need = ((size_t)(rounds + 1) * 4 + (size_t)(rounds + 1) * kLookTab + 16) * sizeof(double);
  VB_TRY(ensure(ctx, ctx->bisect_work, need));
  double* state = (double*)ctx->bisect_work.ptr;
  double* res = state + (size_t)(rounds + 1) * 4;
  hipStream_t st = ctx->stream;
  int prev_levels = 0;
  for (int r = 0; r < rounds; ++r) {
    const int levels = max_its - r * kLook < kLook ? max_its - r * kLook : kLook;
    hipLaunchKernelGGL(dis_bisect_round_kernel, dim3((unsigned)(((1 << levels) - 1) * kLookParts)), dim3(1024), 0, st,
                       lp, b, lprior, scal_in, n, ess_target, prev_levels, levels, (const double*)(state + 4 * r),
                       (const double*)(res + (size_t)(r > 0 ? r - 1 : 0) * kLookTab), state + 4 * (r + 1),
                       res + (size_t)r * kLookTab, eps_prev);
    prev_levels = levels;
  }
  hipLaunchKernelGGL(dis_bisect_final_kernel, dim3(1), dim3(1024), 0, st, lp, b, lprior, scal_in, n, ess_target,
                     prev_levels, (const double*)(state + 4 * rounds),
                     (const double*)(res + (size_t)(rounds > 0 ? rounds - 1 : 0) * kLookTab), 1.0, w, lq_out,
                     scal_out, eps_prev);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// state layout (doubles): [cols_r 2 ld | scal 16 | out 16 | prior cols 2 ld | log p | b | log prior | w | lq]
struct DisLayout {
  int64_t o_cols, o_scal, o_out, o_prior, o_lp, o_b, o_lprior, o_w, o_lq, total;
};
static DisLayout dis_layout(int64_t n, int64_t ld) {
  DisLayout L;
  const int64_t nn = round_up(n, 16);
  L.o_cols = 0;
  L.o_scal = 2 * ld;
  L.o_out = L.o_scal + 16;
  L.o_prior = L.o_out + 16;
  L.o_lp = L.o_prior + 2 * ld;
  L.o_b = L.o_lp + nn;
  L.o_lprior = L.o_b + nn;
  L.o_w = L.o_lprior + nn;
  L.o_lq = L.o_w + nn;
  L.total = L.o_lq + nn;
  return L;
}

int dis_refresh_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t n_total, int64_t d, int family, double df,
                        const double* theta_src, const double* prior_host, double eps_prev, double ess_target,
                        int max_its, double* eps_out, double* ess_out, int* status_out, double* w_host,
                        double* logp_host, double* logq_host) {
  int64_t mine = 0;   // this rank's block inside the gathered per-sample vectors (shard_rows)
  VB_TRY(comm_shard_begin(ctx, n, n_total, &mine));
  if (ctx->model.id != VB_MODEL_GAUSS_DIAG && ctx->model.id != VB_MODEL_FUNNEL && ctx->model.id != VB_MODEL_SOURCE)
    return fail(ctx, VB_ERR_UNSUPPORTED, "mean-field DIS supports the gauss_diag, funnel and source models");
  if (ctx->model.dim != d) return fail(ctx, VB_ERR_INVALID, "model dimension != family dimension");
  if (n <= 0 || n > ns.n || d != ns.d) return fail(ctx, VB_ERR_INVALID, "noise slot shape mismatch");
  const int student = family == VB_FAMILY_MF_STUDENT_T;
  const int64_t ld = ns.ld;
  // per-sample vectors are sized for the WHOLE job: every rank gathers all log p / log q / log prior and
  // runs the (cheap, O(N)) bisection redundantly, so all ranks agree on eps and the weights
  const DisLayout L = dis_layout(n_total, ld);
  VB_TRY(ensure(ctx, ctx->dis_state, (size_t)L.total * sizeof(double)));
  double* base = (double*)ctx->dis_state.ptr;
  hipStream_t st = ctx->stream;

  // tempering prior: a diagonal Gaussian given as an MFGaussian parameter [mu | log_sigma]
  std::vector<double> pr((size_t)2 * ld, 0.0);
  double c0p = -0.5 * (double)d * kLog2PiRs;
  for (int64_t i = 0; i < d; ++i) {
    pr[i] = prior_host[i];
    pr[ld + i] = exp(-2.0 * prior_host[d + i]);
    c0p -= prior_host[d + i];
  }
  VB_TRY(push_small(ctx, st, pr.data(), pr.size() * sizeof(double), base + L.o_prior));   // (staged: `pr` may go)
  ModelDev prior;
  prior.id = VB_MODEL_GAUSS_DIAG;
  prior.dim = (int)d;
  prior.c0 = c0p;
  prior.p0 = base + L.o_prior;
  prior.p1 = base + L.o_prior + ld;

  // (built-in targets: the tempering prior's log density comes out of the same pass; a source model's pass is its own)
  const bool fused_prior = ctx->model.id != VB_MODEL_SOURCE;
  VB_TRY(rowstats_enqueue(ctx, ns, n, d, theta_src, ctx->model, student, df, base + L.o_cols, base + L.o_scal,
                          base + L.o_lp + mine, base + L.o_b + mine, fused_prior ? &prior : nullptr,
                          base + L.o_lprior + mine));
  if (!fused_prior) {
    // second pass: log prior(z_n); its base sums land in the (later overwritten) lq area
    hipLaunchKernelGGL(rs_rowstats_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st,
                       (const double*)ns.buf.ptr, ld, n, (int)d, (const double*)(base + L.o_cols), prior, student,
                       df, base + L.o_lprior + mine, base + L.o_lq);
    VB_HIP(ctx, hipGetLastError());
  }
  if (ctx->temper.kind != VB_PRIOR_DIAG_GAUSSIAN) {
    // a tempering prior that is not a diagonal Gaussian (vb_dis_set_temper_prior) needs the samples themselves
    const int64_t ldz = round_up(d, 16);
    VB_TRY(ensure(ctx, ctx->lg_work, (size_t)n * ldz * sizeof(double)));
    double* Z = (double*)ctx->lg_work.ptr;
    hipLaunchKernelGGL(rs_sample_kernel, dim3((unsigned)n, (unsigned)((d + 255) / 256)), dim3(256), 0, st, theta_src,
                       (const double*)ns.buf.ptr, ld, Z, ldz, n, (int)d);
    VB_HIP(ctx, hipGetLastError());
    VB_TRY(temper_prior_rows(ctx, Z, ldz, n, d, base + L.o_lprior + mine));
  }
  if (ctx->comm) {   // in-place all-gather: every rank contributed its own block
    VB_TRY(comm_gather_rows3(ctx, st, base + L.o_lp, base + L.o_b, base + L.o_lprior, mine, n, n_total));
  }
  VB_TRY(dis_bisect_enqueue(ctx, base + L.o_lp, base + L.o_b, base + L.o_lprior, base + L.o_scal, n_total, eps_prev,
                            ess_target, max_its, base + L.o_w, base + L.o_lq, base + L.o_out));
  double res[3];
  const size_t vec = (size_t)n_total * sizeof(double);
  const FetchSeg segs[4] = {{base + L.o_out, sizeof res, res}, {base + L.o_w, vec, w_host},
                            {base + L.o_lp, logp_host ? vec : 0, logp_host}, {base + L.o_lq, logq_host ? vec : 0, logq_host}};
  VB_TRY(fetch_blocking(ctx, st, segs, 4));
  *eps_out = res[0];
  *ess_out = res[1];
  *status_out = (int)res[2];
  ctx->dis_n = n;
  ctx->dis_n_total = n_total;
  ctx->dis_d = d;
  ++ctx->dis_gen[0];
  if ((int)(*status_out) == 3) return fail(ctx, VB_ERR_STATE, "tempering bisection: a workgroup of the resident kernel did not arrive at a grid barrier (results invalid); VB_DIS_RESIDENT=0 selects the launch chain");
  if (*status_out == 1)
    return fail(ctx, VB_ERR_NUMERIC, "All weights zero! Suggests overflow in importance density.");
  return VB_OK;
}

int dis_state_get(vb_ctx* ctx, double* logp_host, double* logq_host, int64_t n_total) {
  if (!ctx->dis_state.ptr || ctx->dis_n_total != n_total || n_total <= 0)
    return fail(ctx, VB_ERR_STATE, "no mean-field DIS state with %lld samples", (long long)n_total);
  const DisLayout L = dis_layout(ctx->dis_n_total, round_up(ctx->dis_d, 16) % 512 == 0 ? round_up(ctx->dis_d, 16) + 16
                                                                                       : round_up(ctx->dis_d, 16));
  double* base = (double*)ctx->dis_state.ptr;
  hipStream_t st = ctx->stream;
  if (logp_host)
    VB_HIP(ctx, hipMemcpyAsync(logp_host, base + L.o_lp, (size_t)n_total * sizeof(double), hipMemcpyDeviceToHost, st));
  if (logq_host)
    VB_HIP(ctx, hipMemcpyAsync(logq_host, base + L.o_lq, (size_t)n_total * sizeof(double), hipMemcpyDeviceToHost, st));
  VB_HIP(ctx, hipStreamSynchronize(st));
  return VB_OK;
}

int dis_grad_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int family, double df,
                     const double* theta_src, const double* w_host, double scale, double* out) {
  if (ctx->dis_n != n || ctx->dis_d != d || !ctx->dis_state.ptr)
    return fail(ctx, VB_ERR_STATE, "no DIS state of shape %lld x %lld (vb_dis_refresh_meanfield first)",
                (long long)n, (long long)d);
  if (n > ns.n || d != ns.d) return fail(ctx, VB_ERR_INVALID, "noise slot shape mismatch");
  const DisLayout L = dis_layout(ctx->dis_n_total, ns.ld);
  double* base = (double*)ctx->dis_state.ptr;
  VB_TRY(push_small(ctx, ctx->stream, w_host, (size_t)n * sizeof(double), base + L.o_w));   // caller keeps w_host
  ModelDev logq;
  logq.id = kModelLogQ;
  logq.dim = (int)d;
  logq.p0 = base + L.o_cols;            // mu_r
  logq.p1 = base + L.o_cols + ns.ld;    // sigma_r
  MfCall c;
  c.count = 1;
  c.noise[0] = &ns;
  c.theta_src[0] = theta_src;
  c.out[0] = out;
  c.roww[0] = base + L.o_w;
  c.n = n;
  c.d = d;
  c.n_total = n;
  c.family = family;
  c.df = df;
  c.mode = 2;
  c.scale = scale;
  c.model = &logq;
  return mf_enqueue(ctx, c);
}

}  // namespace vb
