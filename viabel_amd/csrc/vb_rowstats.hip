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
                                                          double* __restrict__ out_b) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const double* e_row = noise + row * ld;
  const double* mu = cols;
  const double* sg = cols + ld;
  const bool funnel = m.id == VB_MODEL_FUNNEL;
  double v = 0.0;
  if (funnel) v = fma(sg[m.k], e_row[m.k], mu[m.k]);
  double af = 0.0, ab = 0.0;
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
      }
      ab += student ? -0.5 * (df + 1.0) * log1p(eh * eh / df) : -0.5 * eh * eh;
    }
  }
  af = rs_wave_sum(af);
  ab = rs_wave_sum(ab);
  if (lane == 0) {
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
// value = log(mean s)/alpha + log_norm.  log q(z_n) = b_n - sum(log_sigma).
// One workgroup of 1024 threads; writes the per-row weights s_n and scal_out = [value, sum s].
__global__ void __launch_bounds__(1024) alpha_weights_kernel(const double* __restrict__ f,
                                                             const double* __restrict__ b,
                                                             const double* __restrict__ scal_in, int64_t n,
                                                             double alpha, double* __restrict__ roww,
                                                             double* __restrict__ scal_out) {
  __shared__ double sh[16];
  const double sum_ls = scal_in[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double mx = -INFINITY;
  for (int64_t i = threadIdx.x; i < n; i += 1024) mx = fmax(mx, f[i] - b[i] + sum_ls);
  mx = rs_wave_max(mx);
  if (lane == 0) sh[wave] = mx;
  __syncthreads();
  mx = sh[0];
  for (int w = 1; w < 16; ++w) mx = fmax(mx, sh[w]);
  __syncthreads();
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
    scal_out[0] = log(tot / (double)n) / alpha + mx;
    scal_out[1] = tot;
  }
}

// ---- host ---------------------------------------------------------------------------------------------
int rowstats_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, const double* theta_src,
                     const ModelDev& model, int student, double df, double* cols, double* scal,
                     double* out_f, double* out_b) {
  hipLaunchKernelGGL(rs_cols_kernel, dim3(1), dim3(256), 0, ctx->stream, theta_src, (int)d, ns.ld, cols, scal);
  VB_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(rs_rowstats_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream,
                     (const double*)ns.buf.ptr, ns.ld, n, (int)d, (const double*)cols, model, student, df,
                     out_f, out_b);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// AlphaDivergence value and gradient for the mean-field families (objectives.py:453-461).
int alpha_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int family, double df, double alpha,
                  const double* theta_src, double* out) {
  if (ctx->comm)
    return fail(ctx, VB_ERR_UNSUPPORTED, "AlphaDivergence is not sharded across GPUs yet (global max needed)");
  if (!(alpha != 0.0)) return fail(ctx, VB_ERR_INVALID, "alpha must be non-zero");
  if (ctx->model.id != VB_MODEL_GAUSS_DIAG && ctx->model.id != VB_MODEL_FUNNEL)
    return fail(ctx, VB_ERR_UNSUPPORTED, "mean-field path supports the gauss_diag and funnel models");
  if (ctx->model.dim != d) return fail(ctx, VB_ERR_INVALID, "model dimension != family dimension");
  if (n <= 0 || n > ns.n || d != ns.d) return fail(ctx, VB_ERR_INVALID, "noise slot shape mismatch");
  const int student = family == VB_FAMILY_MF_STUDENT_T;
  // scratch: [cols 2 ld | scal 16 | f n | b n | roww n]
  const int64_t o_scal = 2 * ns.ld, o_f = o_scal + 16, o_b = o_f + round_up(n, 16), o_w = o_b + round_up(n, 16);
  VB_TRY(ensure(ctx, ctx->rowvec, (size_t)(o_w + round_up(n, 16)) * sizeof(double)));
  double* base = (double*)ctx->rowvec.ptr;
  VB_TRY(rowstats_enqueue(ctx, ns, n, d, theta_src, ctx->model, student, df, base, base + o_scal, base + o_f,
                          base + o_b));
  hipLaunchKernelGGL(alpha_weights_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const double*)(base + o_f),
                     (const double*)(base + o_b), (const double*)(base + o_scal), n, alpha, base + o_w,
                     base + o_scal + 8);
  VB_HIP(ctx, hipGetLastError());
  MfCall c;
  c.count = 1;
  c.noise[0] = &ns;
  c.theta_src[0] = theta_src;
  c.out[0] = out;
  c.roww[0] = base + o_w;
  c.n = n;
  c.d = d;
  c.n_total = n;
  c.family = family;
  c.df = df;
  c.mode = 1;
  c.scale = alpha / (double)n;                // objectives.py:460: alpha * vjp / N
  c.value_src = base + o_scal + 8;
  return mf_enqueue(ctx, c);
}

}  // namespace vb
