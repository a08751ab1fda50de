// MultivariateT family on the device: DISInclusiveKL state refresh and weighted log-density gradient.
//
// Reference: viabel/approximations.py:322-382 (family), viabel/_distributions.py:7-38 (log pdf),
// viabel/objectives.py:391-414 (DIS).  theta = [mu | free Cholesky of Sigma = L L'].
//   sample   x_n = mu + (z_n Sigma^{1/2}) / s_n,  s_n = sqrt(chi2_n / df)   (symmetric root, :345-349)
//   log q(x) = c(df, D) - sum log L_ii - (df + D)/2 log(1 + maha/df),  maha = |L^-1 (x - mu)|^2
//   d log q / d mu = c_n u_n,  d log q / d Sigma = -1/2 Sigma^-1 + 1/2 c_n u_n u_n',
//                  u_n = Sigma^-1 (x_n - mu), c_n = (df + D)/(df + maha_n)        
// The O(D^3) factor algebra (sqrtm, L^-1, the chain rule to the free Cholesky parameters) stays on
// the host as in the reference; everything O(N D^2) / O(N D) runs here:
//   X  = mu + (Z R) / s              GEMM (MFMA), row-scaled epilogue           R = Sigma^{1/2}
//   E' = (X - mu) L^-T               GEMM (MFMA)            -> maha_n, log q_n  (row kernel)
//   U  = E' L^-1                     GEMM (MFMA)            = u_n as rows
//   S  = U' diag(w c) U              GEMM (MFMA, lower-triangular tiles, split over n)
// log p and the tempering prior are evaluated on X by the row kernel of vb_rows.hip; the ESS bisection
// is the kernel shared with the mean-field DIS path.
#include "vb_gemm_f64.h"
#include "vb_rng.h"

#include <vector>

namespace vb {

struct EpiSampleT {         // X = mu + acc / s_n
  double* X;
  int64_t ld;
  const double* mu;
  const double* inv_s;
  __device__ void operator()(int, int row, int col, double acc) const {
    X[(int64_t)row * ld + col] = fma(acc, inv_s[row], mu[col]);
  }
  __device__ d2v pair(int, int row, int col, double a0, double a1) const {
    const double is = inv_s[row];
    const d2v v = (d2v){fma(a0, is, mu[col]), fma(a1, is, mu[col + 1])};
    *reinterpret_cast<d2v*>(X + (int64_t)row * ld + col) = v;
    return v;
  }
};

// EpiSampleT that also leaves, per row and column block, the block's share of log p (diagonal-Gaussian target: mean m0,
// inverse variances iv) and of the tempering prior's log density (mean q0, inverse variances q1) -- EpiRowSums,
// vb_gemm_f64.h: the DIS refresh then needs no pass over the samples at all (round 6).  rowpart[(2 bn + q) n_stride + row].
struct EpiSampleTRows {
  double* X;
  int64_t ld;
  const double* mu;
  const double* inv_s;      // may be nullptr (the Gaussian member: no row scales)
  const double *m0, *iv, *q0, *q1;
  double* rowpart;
  int64_t n_stride;
  __device__ void operator()(int, int row, int col, double acc) const {      // (the register-staged kernel: stores only)
    X[(int64_t)row * ld + col] = inv_s ? fma(acc, inv_s[row], mu[col]) : acc + mu[col];
  }
  __device__ d2v rows_one(int row, int col, double acc) const {
    const double z = inv_s ? fma(acc, inv_s[row], mu[col]) : acc + mu[col];
    X[(int64_t)row * ld + col] = z;
    const double dz = z - m0[col], dq = z - q0[col];
    return (d2v){-0.5 * dz * dz * iv[col], -0.5 * dq * dq * q1[col]};
  }
  __device__ d2v rows_pair(int row, int col, double a0, double a1) const {
    const d2v m = *reinterpret_cast<const d2v*>(mu + col);
    d2v z;
    if (inv_s) {
      const double is = inv_s[row];
      z = (d2v){fma(a0, is, m.x), fma(a1, is, m.y)};
    } else {
      z = (d2v){a0 + m.x, a1 + m.y};
    }
    *reinterpret_cast<d2v*>(X + (int64_t)row * ld + col) = z;
    const d2v dz = z - *reinterpret_cast<const d2v*>(m0 + col), dq = z - *reinterpret_cast<const d2v*>(q0 + col);
    const d2v v = *reinterpret_cast<const d2v*>(iv + col), q = *reinterpret_cast<const d2v*>(q1 + col);
    return (d2v){fma(-0.5 * dz.x * dz.x, v.x, -0.5 * dz.y * dz.y * v.y), fma(-0.5 * dq.x * dq.x, q.x, -0.5 * dq.y * dq.y * q.y)};
  }
  __device__ void rows_out(int row, int bn, d2v tot) const {
    rowpart[(int64_t)(2 * bn) * n_stride + row] = tot.x;
    rowpart[(int64_t)(2 * bn + 1) * n_stride + row] = tot.y;
  }
};

// ... and the kernel that finishes the rows: the column blocks' shares in block order, the constants, and the t family's own
// row statistics from the noise rows' norms (rng_normal_kernel: maha_n = r_n^2 sum_c e_nc^2) -- what
// model_prior_maha_rows_kernel (vb_rows.hip) leaves, without reading either matrix
__global__ void __launch_bounds__(256) mvt_rows_combine_kernel(const double* __restrict__ rowpart, int64_t n_stride, int nparts,
                                                               int64_t n, int d, double c0, double qc,
                                                               const double* __restrict__ norms, const double* __restrict__ rs,
                                                               double df, double lq_const, double* __restrict__ lp,
                                                               double* __restrict__ lprior, double* __restrict__ maha,
                                                               double* __restrict__ lq, double* __restrict__ cn) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n) return;
  double a = 0.0, a2 = 0.0;
  for (int b = 0; b < nparts; ++b) {
    a += rowpart[(int64_t)(2 * b) * n_stride + row];
    a2 += rowpart[(int64_t)(2 * b + 1) * n_stride + row];
  }
  lp[row] = a + c0;
  lprior[row] = a2 + qc;
  const double r = rs ? rs[row] : 1.0;
  const double t = (r * r) * norms[row];
  maha[row] = t;
  lq[row] = df > 0.0 ? lq_const - 0.5 * (df + d) * log1p(t / df) : lq_const - 0.5 * t;
  cn[row] = df > 0.0 ? (df + d) / (df + t) : 1.0;
}

struct EpiSubVec {          // E = acc - c   (c = mu L^-T)
  double* E;
  int64_t ld;
  const double* c;
  __device__ void operator()(int, int row, int col, double acc) const {
    E[(int64_t)row * ld + col] = acc - c[col];
  }
  __device__ d2v pair(int, int row, int col, double a0, double a1) const {
    const d2v v = (d2v){a0 - c[col], a1 - c[col + 1]};
    *reinterpret_cast<d2v*>(E + (int64_t)row * ld + col) = v;
    return v;
  }
};

struct EpiStoreScaled {     // U = r_row * acc, UA = a_row * U (the two operands of the weighted Gram product in one epilogue)
  double* U;
  double* UA;
  int64_t ld;
  const double* w;          // a_row = w_row c_row: the weight and c_n = (df + D)/(df + maha_n) of the score
  const double* c;
  const double* r;          // nullptr: r_row = 1
  __device__ void operator()(int, int row, int col, double acc) const {
    const double u = r ? r[row] * acc : acc;
    U[(int64_t)row * ld + col] = u;
    UA[(int64_t)row * ld + col] = (w[row] * c[row]) * u;
  }
  __device__ d2v pair(int, int row, int col, double a0, double a1) const {
    const double ar = w[row] * c[row];
    if (r) {
      const double rr = r[row];
      a0 *= rr;
      a1 *= rr;
    }
    const d2v v = (d2v){a0, a1};
    *reinterpret_cast<d2v*>(U + (int64_t)row * ld + col) = v;
    *reinterpret_cast<d2v*>(UA + (int64_t)row * ld + col) = (d2v){ar * a0, ar * a1};
    return v;
  }
};

struct EpiStore {
  double* U;
  int64_t ld;
  __device__ void operator()(int, int row, int col, double acc) const { U[(int64_t)row * ld + col] = acc; }
  __device__ d2v pair(int, int row, int col, double a0, double a1) const {
    const d2v v = (d2v){a0, a1};
    *reinterpret_cast<d2v*>(U + (int64_t)row * ld + col) = v;
    return v;
  }
};

__device__ __forceinline__ double mvt_wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

// kMvtRowsPerWave rows per wave: maha_n = |E'_n|^2, log q_n; a lane adds its columns c = lane, lane + 64, ... of a row
// in that order.  (Four rows per wave -- more loads in flight, a quarter of the workgroups -- measured SLOWER here,
// 16.2 against 13.6 us at 16 384 x 256 beside the side stream's inverse; the model's row kernel, which shares its
// parameter loads between the rows, gains 2-3 us from the same change: vb_rows.hip.)
// (rs != nullptr: the rows of E are rs_n times what is stored -- the noise matrix itself, see mvt_residuals)
constexpr int kMvtRowsPerWave = 1;
__global__ void __launch_bounds__(256) mvt_rows_kernel(const double* __restrict__ E, int64_t ld, int64_t n, int d,
                                                       double df, double lq_const, double* __restrict__ maha,
                                                       double* __restrict__ lq, const double* __restrict__ rs,
                                                       double* __restrict__ cn) {
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * kMvtRowsPerWave;
  if (row0 >= n) return;
  const double* e[kMvtRowsPerWave];
  double r[kMvtRowsPerWave], s[kMvtRowsPerWave];
#pragma unroll
  for (int q = 0; q < kMvtRowsPerWave; ++q) {
    const int64_t row = row0 + q < n ? row0 + q : n - 1;      // (clamped: loads only)
    e[q] = E + row * ld;
    r[q] = rs ? rs[row] : 1.0;
    s[q] = 0.0;
  }
  for (int c = lane; c < d; c += 64) {
    double v[kMvtRowsPerWave];
#pragma unroll
    for (int q = 0; q < kMvtRowsPerWave; ++q) v[q] = e[q][c];
#pragma unroll
    for (int q = 0; q < kMvtRowsPerWave; ++q) {
      if (rs) v[q] *= r[q];               // the value the stored residual had: same rounding as before
      s[q] = fma(v[q], v[q], s[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < kMvtRowsPerWave; ++q) {
    const double t = mvt_wave_sum(s[q]);
    const int64_t row = row0 + q;
    if (lane == 0 && row < n) {
      maha[row] = t;
      lq[row] = df > 0.0 ? lq_const - 0.5 * (df + d) * log1p(t / df) : lq_const - 0.5 * t;   // df = 0: Gaussian limit
      cn[row] = df > 0.0 ? (df + d) / (df + t) : 1.0;      // c_n of the score: d log q / d mu = c_n u_n
    }
  }
}

// (sum w, sum w log q) in one workgroup, fixed order; eight loads of each vector in flight per thread (a plain strided
// loop pays one memory round trip per 1024 elements: 16 us at N = 16 384)
__global__ void __launch_bounds__(1024) mvt_wsums_kernel(const double* __restrict__ w, const double* __restrict__ lq,
                                                         int64_t n, double* __restrict__ out) {
  __shared__ double sh[2][16];
  double sw = 0.0, swl = 0.0;
  for (int64_t i0 = threadIdx.x; i0 < n; i0 += 8 * 1024) {
    double wv[8], lv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = i0 + u * 1024;
      wv[u] = i < n ? w[i] : 0.0;
      lv[u] = i < n ? lq[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      sw += wv[u];
      swl = fma(wv[u], lv[u], swl);
    }
  }
  sw = mvt_wave_sum(sw);
  swl = mvt_wave_sum(swl);
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = sw;
    sh[1][threadIdx.x >> 6] = swl;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t0 = 0.0, t1 = 0.0;
    for (int k = 0; k < 16; ++k) {
      t0 += sh[0][k];
      t1 += sh[1][k];
    }
    out[0] = t0;
    out[1] = t1;
  }
}

// Packed chain rule, one pass over the residual rows Y (N x D): with a_n = w_n c_n r_n (r_n = 1 / s_n when Y is the noise
// matrix itself, 1 when Y holds the residuals) it leaves
//   YA_n = a_n r_n Y_n      -- the scaled operand of the weighted Gram product M = sum_n a'_n y_n y_n',  y_n = r_n Y_n
//   colpart[rb][j] = sum over the 128 rows of block rb of a_n Y_nj      -- sum_n a'_n y_n, reduced by fr_reduce_kernel
// grid (ceil(D / 64), ceil(N / 128)): thread (c = t & 31, q = t >> 5) owns the column pair 2c, 2c + 1 of the rows
// r0 + q, q + 8, ... (eight 16-byte loads in flight); the eight row groups are combined through LDS in fixed order.
typedef double mvt_d2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) mvt_scale_colsum_kernel(const double* __restrict__ Y, int64_t ld, int64_t n, int d,
                                                               const double* __restrict__ w, const double* __restrict__ cn,
                                                               const double* __restrict__ r, double* __restrict__ YA,
                                                               double* __restrict__ colpart,
                                                               const double* __restrict__ lq, double* __restrict__ wpart) {
  __shared__ mvt_d2 cs[8][32];
  __shared__ double a_s[128], r_s[128], ws[2][2];
  const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int col = blockIdx.x * 64 + 2 * c;
  const int64_t r0 = (int64_t)blockIdx.y * 128;
  const bool ok0 = col < d, ok1 = col + 1 < d;
  if (threadIdx.x < 128) {      // the block's row weights once, through LDS
    const int64_t row = r0 + threadIdx.x;
    double av = 0.0, rv = 1.0;
    if (row < n) {
      rv = r ? r[row] : 1.0;
      av = (w[row] * cn[row]) * rv;
    }
    a_s[threadIdx.x] = av;
    r_s[threadIdx.x] = rv;
    if (blockIdx.x == 0) {
      // (sum w, sum w log q) of the block's rows -- wpart[rb], wpart[n_rb + rb]; fr_reduce_kernel adds the blocks in order
      // (they were a one-workgroup kernel of their own, 5-7 us on the step's critical path)
      const double wv = row < n ? w[row] : 0.0;
      const double s0 = mvt_wave_sum(wv), s1 = mvt_wave_sum(row < n ? wv * lq[row] : 0.0);
      if ((threadIdx.x & 63) == 0) ws[0][threadIdx.x >> 6] = s0, ws[1][threadIdx.x >> 6] = s1;
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < 2)
    wpart[(int64_t)threadIdx.x * gridDim.y + blockIdx.y] = ws[threadIdx.x][0] + ws[threadIdx.x][1];
  mvt_d2 s = (mvt_d2){0.0, 0.0};
  if (ok0) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      mvt_d2 g[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t row = r0 + q + 8 * (8 * half + i);
        g[i] = row < n ? *reinterpret_cast<const mvt_d2*>(Y + row * ld + col) : (mvt_d2){0.0, 0.0};
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int lr = q + 8 * (8 * half + i);
        if (!ok1) g[i].y = 0.0;
        g[i] *= a_s[lr];
        s += g[i];
        if (r0 + lr < n) *reinterpret_cast<mvt_d2*>(YA + (r0 + lr) * ld + col) = g[i] * r_s[lr];
      }
    }
  }
  cs[q][c] = s;
  __syncthreads();
  if (q == 0 && ok0) {
    const mvt_d2 tot = ((cs[0][c] + cs[1][c]) + (cs[2][c] + cs[3][c])) + ((cs[4][c] + cs[5][c]) + (cs[6][c] + cs[7][c]));
    colpart[(int64_t)blockIdx.y * ld + col] = tot.x;
    if (ok1) colpart[(int64_t)blockIdx.y * ld + col + 1] = tot.y;
  }
}

// ---- tempering priors other than a diagonal Gaussian (vb_dis_set_temper_prior; objectives.py:317-319) ----------------
// one wave per row: sum_d t.logpdf((x_d - loc_d) / sigma_d; df) - log sigma_d   (approximations.py:281-286)
__global__ void __launch_bounds__(256) prior_diag_t_rows_kernel(const double* __restrict__ X, int64_t ld, int64_t n, int d,
                                                                const double* __restrict__ loc,
                                                                const double* __restrict__ isig, double df, double c0,
                                                                double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const double* x = X + row * ld;
  double s = 0.0;
  for (int c = lane; c < d; c += 64) {
    const double r = (x[c] - loc[c]) * isig[c];
    s += log1p(r * r / df);
  }
  s = mvt_wave_sum(s);
  if (lane == 0) out[row] = c0 - 0.5 * (df + 1.0) * s;
}

int temper_prior_set(vb_ctx* ctx, int kind, int64_t d, double df, const double* loc, const double* scale,
                     double log_det_l) {
  vb_ctx::TemperPrior& T = ctx->temper;
  if (kind == VB_PRIOR_DIAG_GAUSSIAN) {
    T.kind = 0;
    return VB_OK;
  }
  if (kind != VB_PRIOR_DIAG_STUDENT_T && kind != VB_PRIOR_DENSE) return fail(ctx, VB_ERR_INVALID, "unknown prior kind %d", kind);
  if (d <= 0 || !loc || !scale) return fail(ctx, VB_ERR_INVALID, "prior needs d > 0, loc and scale");
  if (kind == VB_PRIOR_DIAG_STUDENT_T && !(df > 0.0)) return fail(ctx, VB_ERR_INVALID, "df must be positive");
  if (kind == VB_PRIOR_DENSE && !(df >= 0.0)) return fail(ctx, VB_ERR_INVALID, "df must be >= 0 (0: Gaussian)");
  const int64_t ld = round_up(d, 16);
  std::vector<double> host;
  double c0 = 0.0;
  if (kind == VB_PRIOR_DIAG_STUDENT_T) {
    host.assign((size_t)(2 * ld), 0.0);
    c0 = (double)d * (lgamma(0.5 * (df + 1.0)) - lgamma(0.5 * df) - 0.5 * log(df * M_PI));
    for (int64_t i = 0; i < d; ++i) {
      host[i] = loc[i];
      host[ld + i] = exp(-scale[i]);
      c0 -= scale[i];
    }
  } else {
    host.assign((size_t)(d * ld + ld), 0.0);      // W[k][j] = Linv[j][k]; c_j = sum_k Linv[j][k] loc_k
    for (int64_t j = 0; j < d; ++j) {
      double c = 0.0;
      for (int64_t k = 0; k <= j; ++k) {
        const double v = scale[j * d + k];
        host[(size_t)(k * ld + j)] = v;
        c += loc[k] * v;
      }
      host[(size_t)(d * ld + j)] = c;
    }
    c0 = df > 0.0 ? lgamma(0.5 * (df + d)) - lgamma(0.5 * df) - 0.5 * d * log(M_PI * df) - log_det_l
                  : -0.5 * d * log(2.0 * M_PI) - log_det_l;
  }
  VB_TRY(ensure(ctx, T.buf, host.size() * sizeof(double)));
  VB_HIP(ctx, hipMemcpyAsync(T.buf.ptr, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
  VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  legacy_poll(ctx);
  T.kind = kind, T.d = d, T.ld = ld, T.df = df, T.c0 = c0;
  return VB_OK;
}

int temper_prior_rows(vb_ctx* ctx, const double* X, int64_t ld, int64_t n, int64_t d, double* out) {
  vb_ctx::TemperPrior& T = ctx->temper;
  if (T.kind == 0) return fail(ctx, VB_ERR_STATE, "no tempering prior installed");
  if (T.d != d) return fail(ctx, VB_ERR_INVALID, "tempering prior has dimension %lld, the family %lld", (long long)T.d,
                            (long long)d);
  hipStream_t st = ctx->stream;
  const double* p = (const double*)T.buf.ptr;
  const unsigned grid = (unsigned)((n + 3) / 4);
  if (T.kind == VB_PRIOR_DIAG_STUDENT_T) {
    hipLaunchKernelGGL(prior_diag_t_rows_kernel, dim3(grid), dim3(256), 0, st, X, ld, n, (int)d, p, p + T.ld, T.df, T.c0, out);
    VB_HIP(ctx, hipGetLastError());
    return VB_OK;
  }
  // dense: U = (X - loc) L^-T = X W - c (MFMA), then |u_n|^2 and the Gaussian / t log pdf per row
  const int64_t nn = round_up(n, 16);
  VB_TRY(ensure(ctx, T.work, (size_t)(n * T.ld + 2 * nn) * sizeof(double)));
  double* U = (double*)T.work.ptr;
  GemmArgs g;
  g.A = X, g.lda = ld, g.B = p, g.ldb = T.ld;
  g.M = (int)n, g.N = (int)d, g.K = (int)d, g.tri_mode = 0;
  gemm_f64_launch<true>(st, g, 1, ctx->prop.multiProcessorCount, EpiSubVec{U, T.ld, p + d * T.ld});
  VB_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(mvt_rows_kernel, dim3((unsigned)((n + 4 * kMvtRowsPerWave - 1) / (4 * kMvtRowsPerWave))), dim3(256), 0, st,
                     (const double*)U, T.ld, n, (int)d, T.df, T.c0,
                     U + n * T.ld, out, (const double*)nullptr, U + n * T.ld + nn);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// ---- state layout ----------------------------------------------------------------------------------------
struct MvtLayout {
  int64_t ld, nn;
  int64_t o_x, o_e, o_u, o_ua, o_sl, o_root, o_wt, o_li, o_mu, o_c, o_invs, o_maha, o_lq, o_lp, o_lprior, o_w, o_wres, o_cdf, o_cnt, o_lqcopy,
      o_prior, o_scal, o_cpart, o_col, o_part, o_sums, o_theta, o_lt, o_lfull, o_tscr, o_grad, total;
  int splits, n_rb;
  FrSums S;
};

// n: local rows; n_total: whole-job rows (the gathered per-sample vectors log p / log q / log prior / w)
static MvtLayout mvt_layout(vb_ctx* ctx, int64_t n, int64_t n_total, int64_t d) {
  MvtLayout L;
  L.ld = round_up(d, 16);
  L.nn = round_up(n_total, 16);
  L.splits = gram_splits(ctx, (int)d, n);
  L.n_rb = (int)((n + 127) / 128);
  int64_t off = 0;
  auto carve = [&off](int64_t doubles) {
    const int64_t o = off;
    off += round_up(doubles, 16);
    return o;
  };
  const int64_t mat = n * L.ld, sq = d * L.ld;
  L.o_x = carve(mat);
  L.o_e = carve(mat);
  L.o_u = carve(mat);
  L.o_ua = carve(mat);
  L.o_root = carve(sq);
  L.o_wt = carve(sq);
  L.o_li = carve(sq);
  L.o_mu = carve(L.ld);
  L.o_c = carve(L.ld);
  L.o_invs = carve(L.nn);
  L.o_maha = carve(L.nn);
  L.o_lq = carve(L.nn);
  L.o_lp = carve(L.nn);
  L.o_lprior = carve(L.nn);
  L.o_w = carve(L.nn);
  L.o_wres = carve(L.nn);      // resampled weights (multinomial counts) of the device-resident step
  L.o_cdf = carve(L.nn + 32 + L.nn / 1024);  // running sums of the weights per chunk of 1024 | their total at [nn] | chunk totals at [nn + 16]
  L.o_cnt = carve(L.nn / 2 + 16);   // int counts
  L.o_lqcopy = carve(L.nn);
  L.o_prior = carve(2 * L.ld);
  L.o_scal = carve(32);
  L.o_cpart = carve((int64_t)L.splits * sq);
  L.o_col = carve((int64_t)L.n_rb * L.ld);
  L.o_part = carve(n + 2 * ((n + 3) / 4) + (int64_t)L.n_rb * ((d + 63) / 64));    // [a_n | ... | f partials]
  L.S.off_col = 16;
  L.S.off_c = 16 + L.ld;
  L.S.len = 16 + L.ld + sq;
  L.o_sums = carve(L.S.len);
  // device-side factor algebra (throughput mode): flat parameter, L', L, scratch, packed gradient
  L.o_theta = carve(d + d * (d + 1) / 2);
  L.o_lt = carve(sq);
  L.o_lfull = carve(sq);
  L.o_tscr = carve(sq);
  L.o_sl = carve(sq);      // S L of the packed chain rule (a carve of its own: n may be smaller than d)
  L.o_grad = carve(1 + d + d * (d + 1) / 2 + 8);      // value | packed gradient | [eps, ess, status, khat, value]
  L.total = off;
  return L;
}

static int upload_padded(vb_ctx* ctx, double* dst, int64_t ld, const double* src, int64_t rows, int64_t cols,
                         bool transpose) {
  std::vector<double> tmp((size_t)rows * ld, 0.0);
  for (int64_t i = 0; i < rows; ++i)
    for (int64_t j = 0; j < cols; ++j) tmp[(size_t)i * ld + j] = transpose ? src[j * cols + i] : src[i * cols + j];
  VB_HIP(ctx, hipMemcpyAsync(dst, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
  VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  legacy_poll(ctx);
  return VB_OK;
}

// ---- factor algebra on the device (throughput mode) ---------------------------------------------------------------
// out = in' (d x d, row stride ld), pads left alone
__global__ void __launch_bounds__(256) mvt_transpose_kernel(const double* __restrict__ in, double* __restrict__ out,
                                                            int d, int64_t ld) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int r = ty; r < 32; r += 8)
    tile[r][tx] = (r0 + r < d && c0 + tx < d) ? in[(int64_t)(r0 + r) * ld + c0 + tx] : 0.0;
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (c0 + r < d && r0 + tx < d) out[(int64_t)(c0 + r) * ld + r0 + tx] = tile[tx][r];
}

// c_j = sum_k mu_k Linv[j][k] (one wave per row of the lower-triangular inverse)
__global__ void __launch_bounds__(256) mvt_linv_mu_kernel(const double* __restrict__ Li, int64_t ld, int d,
                                                          const double* __restrict__ mu, double* __restrict__ c) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= d) return;
  double s = 0.0;
  for (int k = lane; k <= j; k += 64) s = fma(mu[k], Li[(int64_t)j * ld + k], s);
  s = mvt_wave_sum(s);
  if (lane == 0) c[j] = s;
}

// The O(D^2) leftovers of the factor algebra in ONE launch (they were four: two transposes, L^-1 mu, the scal memset),
// blockIdx.y = role: 0: Li = Wt' ; 1: Lfull = Lt' ; 2: c_j = sum_k mu_k Wt[k][j] (= (L^-1 mu)_j, read from the
// untransposed inverse so that it does not wait for role 0) ; role 2's first block also zeroes the 32 scalars the
// bisection accumulates into.
__global__ void __launch_bounds__(256) mvt_prep_kernel(const double* __restrict__ Wt, const double* __restrict__ Lt,
                                                       const double* __restrict__ mu, double* __restrict__ Li,
                                                       double* __restrict__ Lfull, double* __restrict__ c,
                                                       double* __restrict__ scal, int d, int64_t ld,
                                                       const double* __restrict__ chi, double df, int64_t n_inv,
                                                       double* __restrict__ inv_s, int roles) {
  // roles: bit r set = role r is part of this launch; blockIdx.y counts the set bits (the inverse may be formed on a
  // side stream: then roles 1 and 3 -- which need L' only -- go first and roles 0 and 2 follow the inverse)
  int role = 0;
  for (int seen = -1; role < 4; ++role)
    if ((roles >> role & 1) && ++seen == (int)blockIdx.y) break;
  const int tiles = (d + 31) / 32;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 32 && scal) scal[threadIdx.x] = 0.0;
  if (role == 3) {      // the t family's row scales 1 / s_n = 1 / sqrt(chi_n / df) (approximations.py:345): no launch of their own
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_inv; i += (int64_t)gridDim.x * 256)
      inv_s[i] = 1.0 / sqrt(chi[i] / df);
    return;
  }
  if (role < 2) {
    if ((int)blockIdx.x >= tiles * tiles) return;
    __shared__ double tile[32][33];
    const double* in = role == 0 ? Wt : Lt;
    double* out = role == 0 ? Li : Lfull;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = (blockIdx.x % tiles) * 32, r0 = (blockIdx.x / tiles) * 32;
    for (int r = ty; r < 32; r += 8)
      tile[r][tx] = (r0 + r < d && c0 + tx < d) ? in[(int64_t)(r0 + r) * ld + c0 + tx] : 0.0;
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
      if (c0 + r < d && r0 + tx < d) out[(int64_t)(c0 + r) * ld + r0 + tx] = tile[tx][r];
    return;
  }
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= d) return;
  double s = 0.0;
  for (int k = lane; k <= j; k += 64) s = fma(mu[k], Wt[(int64_t)k * ld + j], s);      // Wt = L^-T is upper triangular
  s = mvt_wave_sum(s);
  if (lane == 0) c[j] = s;
}

// packed gradient of -scale sum_n w_n log q(x_n; theta): d/dmu from the column sums, d/dL = tril(S L)
// - w_sum tril(L^-T) (only the diagonal 1 / L_ii of the upper-triangular L^-T survives), free diagonal x L_ii
__global__ void __launch_bounds__(256) mvt_pack_grad_kernel(const double* __restrict__ SL, const double* __restrict__ Lfull,
                                                            int64_t ld, int d, const double* __restrict__ sums,
                                                            int64_t off_col, double scale, double* __restrict__ out,
                                                            const double* __restrict__ scale_dev,
                                                            const double* __restrict__ res,
                                                            const double* __restrict__ Wt = nullptr) {
  if (scale_dev) scale *= scale_dev[0];      // (device-resident resampling: scale = sum w / (N M), sum w on the device)
  const int64_t n_flat = ((int64_t)d * d + 255) / 256;
  if ((int64_t)blockIdx.x >= n_flat) {
    // (Wt given) the column sums are v = sum_n a_n y_n of the residuals y = L^-1 (x - mu): d/dmu = L^-T v, one wave per
    // component over the upper-triangular Wt = L^-T
    const int lane = threadIdx.x & 63;
    const int j = (int)(blockIdx.x - n_flat) * 4 + (threadIdx.x >> 6);
    if (j >= d) return;
    double a = 0.0;
    for (int k = j + lane; k < d; k += 64) a = fma(Wt[(int64_t)j * ld + k], sums[off_col + k], a);
    a = mvt_wave_sum(a);
    if (lane == 0) out[1 + j] = -scale * a;
    return;
  }
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const double w_sum = sums[1], w_logq = sums[2];
  if (idx == 0) {
    out[0] = -scale * w_logq;
    // tail behind the gradient: [eps, ess, status, khat | value] -- the scalars of a step leave in one small copy while
    // the gradient goes straight into the caller's array
    double* tail = out + 1 + d + (int64_t)d * (d + 1) / 2;
    for (int q = 0; q < 4; ++q) tail[q] = res[q];
    tail[4] = -scale * w_logq;
  }
  if (idx < d && !Wt) out[1 + idx] = -scale * sums[off_col + idx];
  if (idx >= (int64_t)d * d) return;
  const int i = (int)(idx / d), j = (int)(idx % d);
  if (j > i) return;
  double g = SL[(int64_t)i * ld + j];
  if (i == j) g = g * Lfull[(int64_t)i * ld + i] - w_sum;
  out[1 + d + (int64_t)i * (i + 1) / 2 + j] = -scale * g;
}

// mvt_chain_kernel's optional second destination: the mapped host buffer of a blocking call (fetch_plan, vb_api.hip) -- the
// gradient's entries go there with the same stores that write them to device memory, the last workgroup to finish publishes
// the completion word: no gathering launch behind the producer (10.6 us at D = 256 for 265 KB).  grad == nullptr: off.
struct ChainHost {
  double* grad = nullptr;                 // plen doubles: the flat gradient (out + 1)
  double* tail = nullptr;                 // [eps, ess, status, khat | value]
  unsigned* ticket = nullptr;
  unsigned long long* done = nullptr;
  unsigned long long seq = 0;
};

// The packed chain rule of the direct route in ONE launch (round 6): dL = tril(L^-T M) and d/dmu = L^-T v straight into
// the flat gradient -- the D x D x D product used to be a full MFMA launch (one 64 x 64 tile per CU, 16 dependent slabs:
// 14.5 us at D = 256 for 33 MFLOP) followed by a pack kernel (4.9 us).  Here: one workgroup per LOWER 32 x 32 tile of the
// result (the upper triangle is never needed), k from the tile's first row (L^-T is upper triangular), plain fp64 FMAs on
// LDS tiles -- 36 workgroups at D = 256, at most 8 slabs of 32 deep.  M = sums + off_c (symmetric, both triangles: the
// mirrored reduction), v = sums + off_col, (sum w, sum w log q) = sums[1], sums[2]; diagonal tiles also form d/dmu for their
// rows.  Sums are formed in a fixed order: the same bits on every rank of a sharded job.
__global__ void __launch_bounds__(256) mvt_chain_kernel(const double* __restrict__ Wt, const double* __restrict__ Lfull,
                                                        int64_t ld, int d, const double* __restrict__ sums, int64_t off_col,
                                                        int64_t off_c, double scale, double* __restrict__ out,
                                                        const double* __restrict__ scale_dev, const double* __restrict__ res,
                                                        ChainHost H) {
  // four k groups of 64 threads; a group walks its quarter of every 64-deep k step (16 deep per group) with a 4 x 4 micro
  // tile per thread; the groups' partial tiles are added in group order at the end.  (One group walking the whole k range
  // in 32-deep steps -- the first version -- was a chain of eight load-wait-multiply rounds: 50 us at D = 256.)
  __shared__ double buf[4 * 2 * 16 * 33];      // per group: W tile k-major [16][33] | M tile [16][33]; later the partial tiles
  const int t = threadIdx.x, kg = t >> 6, l = t & 63;
  int ti = 0;
  while ((ti + 1) * (ti + 2) / 2 <= (int)blockIdx.x) ++ti;
  const int tj = (int)blockIdx.x - ti * (ti + 1) / 2;
  const int i0 = ti * 32, j0 = tj * 32;
  if (scale_dev) scale *= scale_dev[0];      // (device-resident resampling: scale = sum w / (N M), sum w on the device)
  const double* __restrict__ M = sums + off_c;
  double* Wk = buf + kg * (2 * 16 * 33);
  double* Mk = Wk + 16 * 33;
  const int r0 = (l >> 3) * 4, c0 = (l & 7) * 4;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (int k0 = i0; k0 < d; k0 += 64) {
    const int kb = k0 + 16 * kg;      // this group's 16 k values
    // W tile: rows i0 .. i0 + 31, k = kb .. kb + 15 (k contiguous in memory): 512 entries, 8 per thread; M tile: k = kb ..
    // kb + 15, columns j0 .. j0 + 31 (j contiguous).  All sixteen loads are issued UNCONDITIONALLY from clamped addresses and
    // masked afterwards: a load inside a condition compiles to a branch with its own wait -- sixteen memory round trips
    // per step instead of one (the first version of this loop: 29 us for the launch)
    double wv[8], mv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = l + 64 * q;
      const int i = i0 + (e >> 4), k = kb + (e & 15);
      wv[q] = Wt[(int64_t)(i < d ? i : d - 1) * ld + (k < d ? k : d - 1)];
      const int k2 = kb + (e >> 5), j = j0 + (e & 31);
      mv[q] = M[(int64_t)(k2 < d ? k2 : d - 1) * ld + (j < d ? j : d - 1)];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = l + 64 * q;
      const int r = e >> 4, kk = e & 15, i = i0 + r, k = kb + kk;
      Wk[kk * 33 + r] = (i < d && k < d && k >= i) ? wv[q] : 0.0;      // upper triangular: zero below the diagonal
      const int kk2 = e >> 5, c = e & 31, k2 = kb + kk2, j = j0 + c;
      Mk[kk2 * 33 + c] = (k2 < d && j < d) ? mv[q] : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < 16; ++kk) {
      double w[4], m[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) w[a] = Wk[kk * 33 + r0 + a], m[a] = Mk[kk * 33 + c0 + a];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = fma(w[a], m[b], acc[a][b]);
    }
    __syncthreads();
  }
  // partial tiles -> LDS (group-major), added in group order by the thread that owns the entry
  double* red = buf;      // [4][32][33]  (4 * 32 * 33 = 4224 doubles = the buffer's size)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) red[(kg * 32 + r0 + a) * 33 + c0 + b] = acc[a][b];
  __syncthreads();
  const double w_sum = sums[1], w_logq = sums[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = t + 256 * q, r = e >> 5, c = e & 31;
    const int i = i0 + r, j = j0 + c;
    if (i < d && j <= i) {
      double g = ((red[(0 * 32 + r) * 33 + c] + red[(1 * 32 + r) * 33 + c]) + red[(2 * 32 + r) * 33 + c]) + red[(3 * 32 + r) * 33 + c];
      if (i == j) g = g * Lfull[(int64_t)i * ld + i] - w_sum;
      out[1 + d + (int64_t)i * (i + 1) / 2 + j] = -scale * g;
      if (H.grad) H.grad[d + (int64_t)i * (i + 1) / 2 + j] = -scale * g;
    }
  }
  if (ti == tj) {      // d/dmu for the tile's rows: eight lanes per row over k >= i, combined by shuffles in a fixed order
    const int r = t >> 3, s8 = t & 7, i = i0 + r;
    double a = 0.0;
    if (i < d) {
      // (eight loads of each operand in flight per round: a plain loop waits for memory once per k -- 32 round trips)
      for (int kb = i + s8; kb < d; kb += 64) {
        double wv[8], vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = kb + 8 * u;
          wv[u] = k < d ? Wt[(int64_t)i * ld + k] : 0.0;
          vv[u] = k < d ? sums[off_col + k] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) a = fma(wv[u], vv[u], a);
      }
    }
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    a += __shfl_xor(a, 4, 64);
    if (s8 == 0 && i < d) {
      out[1 + i] = -scale * a;
      if (H.grad) H.grad[i] = -scale * a;
    }
  }
  if (blockIdx.x == 0 && t == 0) {
    out[0] = -scale * w_logq;
    double* tail = out + 1 + d + (int64_t)d * (d + 1) / 2;      // [eps, ess, status, khat | value]: see mvt_pack_grad_kernel
    for (int q = 0; q < 4; ++q) tail[q] = res[q];
    tail[4] = -scale * w_logq;
    if (H.grad) {
      for (int q = 0; q < 4; ++q) H.tail[q] = res[q];
      H.tail[4] = -scale * w_logq;
    }
  }
  if (H.grad) {      // what fetch_copy_kernel's workgroups do behind their stores (vb_api.hip): the last one publishes
    __threadfence_system();
    __syncthreads();
    if (t == 0) {
      const unsigned tk = __hip_atomic_fetch_add(H.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (tk == gridDim.x - 1) {
        __hip_atomic_store(H.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_system();
        __hip_atomic_store(H.done, H.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// The throughput mode's parameter upload (round 6): the flat parameter sits in MAPPED host memory and is read across the bus
// once, in 1-KiB wave-loads of consecutive entries (fr_unpack_kernel reads it as the 32 x 32 tiles of L it transposes: 256-byte
// runs that start anywhere in a line -- 11.3 us for 265 KB at D = 256, 23 GB/s); every entry goes to the device-resident copy
// (coalesced) and to its place in mu or L' (Lt[k][j] = L[j][k], exp on the diagonal: scattered 8-byte stores, 33 K of them
// at D = 256).  The same launch has the workgroups that zero L' below the diagonal and, when the caller asks for it, the
// two pieces of mvt_prep_kernel that do not depend on the unpacked factor -- the t family's row scales 1 / sqrt(chi_n / df) and the
// 32 scalars of the bisection -- so that the main stream needs no prep launch before its sampling product (the transposed copy
// L goes to the side stream's prep).  The values are fr_unpack_kernel's and mvt_prep_kernel's (same expressions): bit-identical.
// blocks [0, nb_t): 1024 entries each; [nb_t, nb_t + nb_z): the lower 32 x 32 tiles (diagonal ones included); the rest: chi.
__global__ void __launch_bounds__(256) mvt_unpack_kernel(const double* __restrict__ theta, int d, int64_t ld,
                                                         double* __restrict__ Lt, double* __restrict__ mu,
                                                         double* __restrict__ copy, const double* __restrict__ chi, double df,
                                                         int64_t n_inv, double* __restrict__ inv_s, double* __restrict__ scal,
                                                         int nb_t, int nb_z) {
  const int b = (int)blockIdx.x, t = (int)threadIdx.x;
  const int64_t p = (int64_t)d + (int64_t)d * (d + 1) / 2;
  if (b == 0 && scal && t < 32) scal[t] = 0.0;
  if (b < nb_t) {
    mvt_d2 v[2];
    int64_t at[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {      // (the staging slot is padded to whole 64-byte lines: a pair may reach past p, never past the slot)
      at[q] = (int64_t)b * 1024 + 2 * (t + 256 * q);
      v[q] = *reinterpret_cast<const mvt_d2*>(theta + (at[q] < p ? at[q] : 0));
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (at[q] >= p) continue;
      if (at[q] + 1 < p) *reinterpret_cast<mvt_d2*>(copy + at[q]) = v[q];
      else copy[at[q]] = v[q].x;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int64_t e = at[q] + h;
        if (e >= p) break;
        const double x = h ? v[q].y : v[q].x;
        if (e < d) {
          mu[e] = x;
          continue;
        }
        const int64_t tt = e - d;
        int64_t j = (int64_t)((sqrt(8.0 * (double)tt + 1.0) - 1.0) * 0.5);
        while ((j + 1) * (j + 2) / 2 <= tt) ++j;
        while (j * (j + 1) / 2 > tt) --j;
        const int64_t k = tt - j * (j + 1) / 2;
        Lt[k * ld + j] = k == j ? exp(x) : x;
      }
    }
    return;
  }
  if (b < nb_t + nb_z) {
    int kt = 0;
    const int z = b - nb_t;
    while ((kt + 1) * (kt + 2) / 2 <= z) ++kt;
    const int jt = z - kt * (kt + 1) / 2;
    const int tx = t & 31, ty = t >> 5;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const int k = kt * 32 + r, j = jt * 32 + tx;
      if (k < d && j < d && k > j) Lt[(int64_t)k * ld + j] = 0.0;
    }
    return;
  }
  const int nb_i = (int)gridDim.x - nb_t - nb_z;
  for (int64_t i = (int64_t)(b - nb_t - nb_z) * 256 + t; i < n_inv; i += (int64_t)nb_i * 256)
    inv_s[i] = 1.0 / sqrt(chi[i] / df);      // (mvt_prep_kernel, role 3)
}

// theta (host) -> device: mu, L' (o_lt), L (o_lfull), Wt = L^-T (o_wt), Li = L^-1 (o_li), c = L^-1 mu (o_c)
// zero_scal: the refresh's call also clears the 32 scalars the bisection accumulates into (a gradient at another
// parameter must leave them alone: eps, ess and the status of the refresh live there)
// chi != nullptr: the same launch also forms the n_inv row scales 1 / sqrt(chi / df) into L.o_invs
// defer_inverse: the inverse and what is read off it (Wt, Li, c) are formed on a SIDE stream behind the unpack while the
// main stream goes on with what needs L' only (the sampling product, the row kernels, the bisection); whoever reads
// Wt / Li / c calls mvt_join_inverse first.  Thirty-odd microseconds of small dependent launches (a 128 x 128 leaf
// inversion and two D/2-sized products at D = 256) leave the critical path of a throughput-mode step.
// (dev / test switches, read per call: VB_MVT_DIRECT, VB_MVT_SIDE_INVERSE, VB_MVT_FLAGSYNC -- "0" turns the round-5 route off)
static bool mvt_env_on(const char* name) {
  const char* e = getenv(name);
  return !(e && atoi(e) == 0);
}

// The side stream's launches are enqueued LATE (mvt_side_enqueue, called by the refresh once the main stream has its
// sampling product and row kernels queued): eight API calls ahead of them starve the main queue for ~30 us.
static int mvt_side_enqueue(vb_ctx* ctx) {
  if (!ctx->mvt_inv_pending || ctx->mvt_inv_queued) return VB_OK;
  ctx->mvt_inv_queued = true;
  const auto& a = ctx->mvt_inv_args;
  hipStream_t sd = ctx->mvt_side;
  VB_HIP(ctx, hipStreamWaitEvent(sd, ctx->mvt_ev_fork, 0));
  VB_TRY(fr_tri_inverse_enqueue(ctx, sd, a.base + a.o_theta, a.base + a.o_lt, a.d, a.ld, a.base + a.o_wt, a.base + a.o_tscr,
                                a.clean));
  const int tiles = (a.d + 31) / 32, gx = tiles * tiles > (a.d + 3) / 4 ? tiles * tiles : (a.d + 3) / 4;
  // (roles 0 and 2 -- Li and c, read off the inverse -- and, when the main stream did not form it, role 1: L = (L')')
  hipLaunchKernelGGL(mvt_prep_kernel, dim3((unsigned)gx, a.lfull_here ? 3 : 2), dim3(256), 0, sd, (const double*)(a.base + a.o_wt),
                     (const double*)(a.base + a.o_lt), (const double*)(a.base + a.o_mu), a.base + a.o_li, a.base + a.o_lfull,
                     a.base + a.o_c, (double*)nullptr, a.d, a.ld, (const double*)nullptr, 0.0, (int64_t)0, (double*)nullptr,
                     a.lfull_here ? 0x7 : 0x5);
  VB_HIP(ctx, hipGetLastError());
  VB_HIP(ctx, hipEventRecord(ctx->mvt_ev_join, sd));
  return VB_OK;
}

static int mvt_join_inverse(vb_ctx* ctx) {
  if (!ctx->mvt_inv_pending) return VB_OK;
  VB_TRY(mvt_side_enqueue(ctx));
  ctx->mvt_inv_pending = false;
  VB_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->mvt_ev_join, 0));
  return VB_OK;
}

static int mvt_factors_device(vb_ctx* ctx, const MvtLayout& L, double* base, int64_t d, const double* theta_host,
                              bool zero_scal = false, const double* chi = nullptr, double df = 0.0, int64_t n_inv = 0,
                              bool defer_inverse = false, bool lfull_main = true) {
  VB_TRY(mvt_join_inverse(ctx));      // (the unpack below overwrites what a pending inverse still reads)
  hipStream_t st = ctx->stream;
  const int D = (int)d;
  const size_t p = (size_t)(d + d * (d + 1) / 2);
  // the caller keeps ownership of theta_host: it goes through a pinned staging buffer of the context, so the copy is
  // asynchronous and nothing here waits for the stream
  if (ctx->mvt_pin_doubles < p) {
    if (ctx->mvt_pin) {
      legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
      VB_HIP(ctx, hipStreamSynchronize(st));
      legacy_poll(ctx);
      VB_HIP(ctx, hipHostFree(ctx->mvt_pin));
      ctx->mvt_pin = nullptr;
    }
    const size_t pr = (p + 7) / 8 * 8;      // (whole 64-byte lines per slot: mvt_unpack_kernel reads 16-byte pairs)
    VB_HIP(ctx, hipHostMalloc((void**)&ctx->mvt_pin, 2 * pr * sizeof(double), hipHostMallocMapped));
    VB_HIP(ctx, hipHostGetDevicePointer((void**)&ctx->mvt_pin_dev, ctx->mvt_pin, 0));
    ctx->mvt_pin_doubles = pr;
  }
  // two staging slots taken in turn; an event behind each slot's copy is waited for before the slot is rewritten (a
  // deferred refresh returns without a synchronisation, so back-to-back refreshes could otherwise overwrite a slot
  // whose copy is still queued behind kernels)
  ctx->mvt_pin_slot ^= 1;
  hipEvent_t& slot_ev = ctx->mvt_pin_ev[ctx->mvt_pin_slot];
  if (!slot_ev) VB_HIP(ctx, hipEventCreateWithFlags(&slot_ev, hipEventDisableTiming));
  else VB_HIP(ctx, hipEventSynchronize(slot_ev));
  double* stage = ctx->mvt_pin + (size_t)ctx->mvt_pin_slot * ctx->mvt_pin_doubles;
  memcpy(stage, theta_host, p * sizeof(double));
  if (L.ld != d)      // (pad columns of mu and c: the unpack and the prep kernel write columns [0, d) only)
    VB_HIP(ctx, hipMemsetAsync(base + L.o_mu, 0, (size_t)(2 * L.ld) * sizeof(double), st));
  // the unpack reads the staged parameter IN PLACE (mapped host memory, every entry once) and leaves the device copy
  // behind: no DMA-engine round trip between the host's memcpy and the first kernel
  const bool side_env = mvt_env_on("VB_MVT_SIDE_INVERSE");
  const bool unpack_env = mvt_env_on("VB_MVT_UNPACK");      // 0: fr_unpack_kernel's tiles + the main stream's prep launch
  // (the one-launch front: nothing but the unpack in front of the sampling product -- row scales and scalars ride along, L goes
  // to the side stream)
  const bool front = unpack_env && defer_inverse && side_env && !lfull_main;
  const double* theta_mapped = ctx->mvt_pin_dev + (size_t)ctx->mvt_pin_slot * ctx->mvt_pin_doubles;
  if (unpack_env) {
    const int nt = (D + 31) / 32, nb_t = (int)((p + 1023) / 1024), nb_z = nt * (nt + 1) / 2;
    int nb_i = front && chi ? (int)((n_inv + 1023) / 1024) : 0;
    nb_i = nb_i > 64 ? 64 : nb_i;
    hipLaunchKernelGGL(mvt_unpack_kernel, dim3((unsigned)(nb_t + nb_z + nb_i)), dim3(256), 0, st, theta_mapped, D, L.ld,
                       base + L.o_lt, base + L.o_mu, base + L.o_theta, front ? chi : (const double*)nullptr, df, n_inv,
                       base + L.o_invs, front && zero_scal ? base + L.o_scal : (double*)nullptr, nb_t, nb_z);
    VB_HIP(ctx, hipGetLastError());
  } else {
    VB_TRY(fr_unpack_enqueue(ctx, st, theta_mapped, D, L.ld, base + L.o_lt, base + L.o_mu, base + L.o_theta));
  }
  VB_HIP(ctx, hipEventRecord(slot_ev, st));
  // the inverse's strictly lower triangle (and the scratch) need zeroing only when the buffer or its layout changed:
  // nothing else writes o_wt in throughput mode
  const int64_t key[4] = {(int64_t)(uintptr_t)base, L.o_wt, L.o_tscr, d};
  const bool clean = memcmp(key, ctx->mvt_inv_key, sizeof key) == 0;
  memcpy(ctx->mvt_inv_key, key, sizeof key);
  // (the transposes write every entry of the d x d blocks; the pad columns hold the zeros of the allocation)
  const int tiles = (D + 31) / 32, gx = tiles * tiles > (D + 3) / 4 ? tiles * tiles : (D + 3) / 4;
  if (defer_inverse && side_env) {
    if (!ctx->mvt_side) {
      VB_HIP(ctx, hipStreamCreateWithFlags(&ctx->mvt_side, hipStreamNonBlocking));
      VB_HIP(ctx, hipEventCreateWithFlags(&ctx->mvt_ev_join, hipEventDisableTiming));
    }
    // (the fork: the staging slot's event, recorded right behind the unpack above -- no event of its own)
    ctx->mvt_ev_fork = slot_ev;
    auto& a = ctx->mvt_inv_args;
    a.base = base, a.o_theta = L.o_theta, a.o_lt = L.o_lt, a.o_wt = L.o_wt, a.o_tscr = L.o_tscr, a.o_mu = L.o_mu;
    a.o_li = L.o_li, a.o_lfull = L.o_lfull, a.o_c = L.o_c, a.ld = L.ld, a.d = D, a.clean = clean;
    a.lfull_here = front;
    ctx->mvt_inv_pending = true;
    ctx->mvt_inv_queued = false;
    if (front) return VB_OK;      // (row scales and scalars: the unpack's launch; L: the side stream's prep)
    hipLaunchKernelGGL(mvt_prep_kernel, dim3((unsigned)gx, chi ? 2 : 1), dim3(256), 0, st, (const double*)(base + L.o_wt),
                       (const double*)(base + L.o_lt), (const double*)(base + L.o_mu), base + L.o_li, base + L.o_lfull,
                       base + L.o_c, zero_scal ? base + L.o_scal : (double*)nullptr, D, L.ld, chi, df, n_inv, base + L.o_invs,
                       chi ? 0xA : 0x2);
    VB_HIP(ctx, hipGetLastError());
    return VB_OK;
  }
  VB_TRY(fr_tri_inverse_enqueue(ctx, st, base + L.o_theta, base + L.o_lt, D, L.ld, base + L.o_wt, base + L.o_tscr, clean));
  hipLaunchKernelGGL(mvt_prep_kernel, dim3((unsigned)gx, chi ? 4 : 3), dim3(256), 0, st, (const double*)(base + L.o_wt),
                     (const double*)(base + L.o_lt), (const double*)(base + L.o_mu), base + L.o_li, base + L.o_lfull,
                     base + L.o_c, zero_scal ? base + L.o_scal : (double*)nullptr, D, L.ld, chi, df, n_inv, base + L.o_invs,
                     chi ? 0xF : 0x7);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// E' = (X - mu) L^-T, maha, log q for the parameter `theta_host` with inverse factor `linv_host`
// (linv_host == nullptr: the factors come from theta on the device, mvt_factors_device -- unless the caller has just
// run it: factors_ready)
static int mvt_residuals(vb_ctx* ctx, const MvtLayout& L, double* base, int64_t n, int64_t d, double df,
                         const double* theta_host, const double* linv_host, int64_t lq_off, bool factors_ready = false,
                         const NoiseSlot* drawn_here = nullptr, double* defer_rows = nullptr) {
  const int n_cu = ctx->prop.multiProcessorCount;
  double logdet_half = 0.0;
  for (int64_t j = 0; j < d; ++j) logdet_half += theta_host[d + j * (j + 1) / 2 + j];
  if (linv_host || !drawn_here) VB_TRY(mvt_join_inverse(ctx));
  if (linv_host) {
    // Wt[k][j] = Linv[j][k] (B operand of dev L^-T), Li[k][j] = Linv[k][j] (B operand of E' L^-1), [mu | c]:
    // staged in one host buffer, three copies, ONE synchronisation
    const int64_t sq = d * L.ld;
    std::vector<double>& st = ctx->mvt_stage;
    st.assign((size_t)(2 * sq + 2 * L.ld), 0.0);
    double *wt = st.data(), *li = wt + sq, *vec = li + sq;
    for (int64_t j = 0; j < d; ++j) {
      double c = 0.0;                       // c_j = sum_k mu_k Linv[j][k]
      for (int64_t k = 0; k <= j; ++k) {
        const double v = linv_host[j * d + k];
        li[j * L.ld + k] = v;
        wt[k * L.ld + j] = v;
        c += theta_host[k] * v;
      }
      vec[j] = theta_host[j];
      vec[L.ld + j] = c;
    }
    ctx->mvt_inv_key[0] = 0;      // (the uploaded factor overwrites what the device inverse keeps clean)
    VB_HIP(ctx, hipMemcpyAsync(base + L.o_wt, wt, (size_t)sq * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    VB_HIP(ctx, hipMemcpyAsync(base + L.o_li, li, (size_t)sq * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    VB_HIP(ctx, hipMemcpyAsync(base + L.o_mu, vec, (size_t)(2 * L.ld) * sizeof(double), hipMemcpyHostToDevice,
                               ctx->stream));
    legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
    VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    legacy_poll(ctx);
  } else if (!factors_ready) {
    VB_TRY(mvt_factors_device(ctx, L, base, d, theta_host));
  }
  // throughput-mode refresh: the state samples were just drawn through this parameter's factor, x = mu + (z L') / s, so
  // their residuals (x - mu) L^-T ARE the scaled noise z / s -- no product, and no copy either: the consumers (the row
  // kernel below, the U = E' L^-1 product of the gradient) read the noise matrix and scale by 1 / s_n themselves
  ctx->mvt_e_noise = drawn_here ? (const double*)drawn_here->buf.ptr : nullptr;
  ctx->mvt_e_noise_ld = drawn_here ? drawn_here->ld : 0;
  if (!drawn_here) {
    GemmArgs g;
    g.A = base + L.o_x;
    g.lda = L.ld;
    g.B = base + L.o_wt;
    g.ldb = L.ld;
    g.M = (int)n;
    g.N = (int)d;
    g.K = (int)d;
    g.tri_mode = 0;
    gemm_f64_launch<true>(ctx->stream, g, 1, n_cu, EpiSubVec{base + L.o_e, L.ld, base + L.o_c});
  }
  VB_HIP(ctx, hipGetLastError());
  const double lq_const = df > 0.0
                              ? lgamma(0.5 * (df + d)) - lgamma(0.5 * df) - 0.5 * d * log(M_PI * df) - logdet_half
                              : -0.5 * d * log(2.0 * M_PI) - logdet_half;
  if (defer_rows && drawn_here) {      // (the caller's row pass over the samples takes maha / log q / c_n along: model_prior_maha_rows)
    *defer_rows = lq_const;
    return VB_OK;
  }
  hipLaunchKernelGGL(mvt_rows_kernel, dim3((unsigned)((n + 4 * kMvtRowsPerWave - 1) / (4 * kMvtRowsPerWave))), dim3(256), 0, ctx->stream,
                     drawn_here ? ctx->mvt_e_noise : (const double*)(base + L.o_e), drawn_here ? drawn_here->ld : L.ld, n,
                     (int)d, df, lq_const, base + L.o_maha, base + L.o_lq + lq_off,
                     drawn_here ? (const double*)(base + L.o_invs) : (const double*)nullptr, base + L.o_part);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

__global__ void __launch_bounds__(256) mvt_fill_kernel(double* __restrict__ dst, int64_t n, double v) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = v;
}

__global__ void __launch_bounds__(256) mvt_inv_scale_kernel(const double* __restrict__ chi, double df, int64_t n,
                                                            double* __restrict__ inv_s) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) inv_s[i] = 1.0 / sqrt(chi[i] / df);                                      // approximations.py:345
}

// ---- multinomial resampling on the device (objectives.py:408: np.random.choice(N, M, p = w / sum w)) -------------------
// rng = 'philox' reproduces no noise stream of the reference, so the draw does not have to be numpy's: M uniforms from
// the family's Philox stream, inverted through the running sums of the weights (the same searchsorted(side = 'right')
// the host path uses); the result is the vector of counts as doubles -- what the weighted-score kernels take.
// Running sums in two levels, every sum in a fixed order: workgroup c of mvt_cdf_kernel scans its chunk of 1024 weights
// (wave shuffles + one LDS hop over the 16 wave totals) and leaves the chunk's total behind; every workgroup of the draw
// kernel scans the chunk totals (the same numbers in the same order everywhere) and searches chunk, then element.  (A
// single workgroup forming all N running sums took 22 us at N = 16 384 -- one CU's dependent chains -- against 5.)
constexpr int kCdfChunk = 1024, kCdfMaxChunks = 2048;
__global__ void __launch_bounds__(kCdfChunk) mvt_cdf_kernel(const double* __restrict__ w, int64_t n, double* __restrict__ cdf,
                                                            double* __restrict__ chunk_total, int* __restrict__ counts) {
  __shared__ double wave_tot[16];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int64_t i = (int64_t)blockIdx.x * kCdfChunk + t;
  double v = i < n ? w[i] : 0.0;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double u = __shfl_up(v, off, 64);
    if (lane >= off) v += u;
  }
  if (lane == 63) wave_tot[wv] = v;
  __syncthreads();
  double before = 0.0;
  for (int q = 0; q < wv; ++q) before += wave_tot[q];
  v += before;
  if (i < n) {
    cdf[i] = v;
    counts[i] = 0;
  }
  if (t == kCdfChunk - 1) chunk_total[blockIdx.x] = v;
}

__global__ void __launch_bounds__(256) mvt_draw_kernel(const double* __restrict__ cdf, const double* __restrict__ chunk_total,
                                                       double* __restrict__ total, int64_t n, int64_t m, uint64_t seed,
                                                       uint64_t stream, int* __restrict__ counts) {
  // exclusive running sums of the chunk totals: eight consecutive chunks per thread, the threads' sums scanned by
  // wave shuffles and one LDS hop; pre[nc] = the sum of all weights
  __shared__ double pre[kCdfMaxChunks + 1];
  __shared__ double wave_tot[4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int nc = (int)((n + kCdfChunk - 1) / kCdfChunk);
  constexpr int kPer = kCdfMaxChunks / 256;
  double loc[kPer], run = 0.0;
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int c = t * kPer + k;
    loc[k] = run;                                   // exclusive inside the thread
    run += c < nc ? chunk_total[c] : 0.0;
  }
  double v = run;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double u = __shfl_up(v, off, 64);
    if (lane >= off) v += u;
  }
  const double up = __shfl_up(v, 1, 64);
  if (lane == 63) wave_tot[wv] = v;
  __syncthreads();
  double before = 0.0;
  for (int q = 0; q < wv; ++q) before += wave_tot[q];
  const double offset = before + (lane ? up : 0.0);
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int c = t * kPer + k;
    if (c <= nc) pre[c] = offset + loc[k];
  }
  if (t == 255) pre[kCdfMaxChunks] = offset + run;      // (nc == kCdfMaxChunks: the slot no thread's chunk index reaches)
  __syncthreads();
  const double sum = pre[nc];
  if (blockIdx.x == 0 && t == 0) total[0] = sum;
  const int64_t i = (int64_t)blockIdx.x * 256 + t;
  if (i >= m) return;
  Philox4 c;
  c.x = (uint32_t)i, c.y = (uint32_t)(i >> 32), c.z = (uint32_t)stream, c.w = 0x52534d50u;      // 'RSMP': its own sub-stream
  const Philox4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const double x = u01(r.x, r.y) * sum;
  int clo = 0, chi = nc;                   // the chunk: the last one whose running sum in front of it is <= x
  while (chi - clo > 1) {
    const int mid = (clo + chi) >> 1;
    if (pre[mid] <= x) clo = mid;
    else chi = mid;
  }
  const double base = pre[clo];
  const int64_t first = (int64_t)clo * kCdfChunk, last = first + kCdfChunk < n ? first + kCdfChunk : n;
  int64_t lo = first, hi = last;           // first index with running sum > x (searchsorted side = 'right')
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (base + cdf[mid] > x) hi = mid;
    else lo = mid + 1;
  }
  if (lo >= last) lo = last - 1;
  atomicAdd(&counts[lo], 1);               // integer: the counts do not depend on the order of arrival
}

__global__ void __launch_bounds__(256) mvt_counts_kernel(const int* __restrict__ counts, int64_t n, double* __restrict__ w) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) w[i] = (double)counts[i];
}

// The context's chi-square draws of this rank's rows: `n` of them (vb_chisq_generate with the shard's row offset) or all
// n_total of numpy's stream (vb_legacy_rng_chisquare_device draws the whole vector on every rank -- the generator must end
// where numpy's does), of which the shard's block is taken.  nullptr: neither.
static const double* mvt_chi_rows(vb_ctx* ctx, int64_t n, int64_t n_total, int64_t mine, double df) {
  if (!ctx->chi_dev.ptr || ctx->chi_df != df) return nullptr;
  if (ctx->chi_n == n) return (const double*)ctx->chi_dev.ptr;
  if (ctx->chi_n == n_total) return (const double*)ctx->chi_dev.ptr + mine;
  return nullptr;
}

int mvt_dis_refresh(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t n_total, int64_t d, double df,
                    const double* theta_host,
                    const double* chi_host, const double* root_host, const double* linv_host,
                    const double* prior_host, double eps_prev, double ess_target, int max_its, double* eps_out,
                    double* ess_out, double* w_host, double* logp_host, double* logq_host, bool sym_root,
                    double* root_info) {
  int64_t mine = 0;   // this rank's block inside the gathered per-sample vectors (shard_rows)
  VB_TRY(comm_shard_begin(ctx, n, n_total, &mine));
  if (ctx->model.id != VB_MODEL_GAUSS_DIAG && ctx->model.id != VB_MODEL_FUNNEL && ctx->model.id != VB_MODEL_SOURCE)
    return fail(ctx, VB_ERR_UNSUPPORTED, "row log-density implements gauss_diag, funnel and source models");
  if (ctx->model.dim != d) return fail(ctx, VB_ERR_INVALID, "model dimension != family dimension");
  if (n <= 0 || n > ns.n || d != ns.d) return fail(ctx, VB_ERR_INVALID, "noise slot shape mismatch");
  if (!(df > 2.0) && df != 0.0) return fail(ctx, VB_ERR_INVALID, "df must be greater than 2 (or 0: Gaussian limit)");
  const MvtLayout L = mvt_layout(ctx, n, n_total, d);
  VB_TRY(ensure(ctx, ctx->mvt_state, (size_t)L.total * sizeof(double)));
  double* base = (double*)ctx->mvt_state.ptr;
  hipStream_t st = ctx->stream;
  const int n_cu = ctx->prop.multiProcessorCount;

  ctx->mvt_theta.clear();   // until this refresh has succeeded the device residuals belong to no parameter
  const bool dev_factors = root_host == nullptr && linv_host == nullptr;
  if ((root_host == nullptr) != (linv_host == nullptr))
    return fail(ctx, VB_ERR_INVALID, "sqrt_sigma and l_inv are given together or not at all");
  if (sym_root && !dev_factors) return fail(ctx, VB_ERR_INVALID, "the device's symmetric root goes with the device's factors");
  // the context's chi-square draws (vb_chisq_generate / vb_legacy_rng_chisquare_device) become row scales in the factor
  // kernel's own launch
  const bool chi_dev_rows = !chi_host && df != 0.0;
  const double* chi_rows = chi_dev_rows ? mvt_chi_rows(ctx, n, n_total, mine, df) : nullptr;
  if (chi_dev_rows && !chi_rows)
    return fail(ctx, VB_ERR_STATE, "chi == NULL needs %lld (or all %lld) device chi-square(%g) draws (vb_chisq_generate)",
                (long long)n, (long long)n_total, df);
  if (dev_factors) {      // throughput mode: mu, L', L^-1 from theta on the device; the samples go through L' (see header)
    VB_TRY(mvt_factors_device(ctx, L, base, d, theta_host, true, chi_rows, df, n, true, /*lfull_main=*/sym_root));
    // reference-identical sampling (approximations.py:348): x = mu + (z Sigma^(1/2)) / s with the SYMMETRIC root, formed
    // on the device from the unpacked factor (VB_ERR_UNSUPPORTED: not resolved to 1e-12 -- the caller's LAPACK route)
    if (sym_root) {
      VB_TRY(mvt_side_enqueue(ctx));      // (the inverse runs beside the root's iteration)
      VB_TRY(sym_sqrt_dev(ctx, base + L.o_lfull, base + L.o_lt, d, L.ld, base + L.o_root, 1e-12, root_info));
    }
  } else {
    VB_TRY(upload_padded(ctx, base + L.o_root, L.ld, root_host, d, d, false));
  }
  ctx->mvt_dev_factors = dev_factors;
  std::vector<double> inv_s;
  if (!chi_host && df == 0.0 && dev_factors) {
    // the Gaussian member in throughput mode: s_n = 1, written on the device (no host vector, no synchronisation)
    hipLaunchKernelGGL(mvt_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, base + L.o_invs, n, 1.0);
    VB_HIP(ctx, hipGetLastError());
  } else if (chi_host || df == 0.0) {
    inv_s.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i)
      inv_s[i] = df > 0.0 ? 1.0 / sqrt(chi_host[i] / df) : 1.0;                       // approximations.py:345
    VB_HIP(ctx, hipMemcpyAsync(base + L.o_invs, inv_s.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
  } else if (!dev_factors) {   // the draws of vb_chisq_generate, already on the device (dev_factors: done by the factor kernel)
    hipLaunchKernelGGL(mvt_inv_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, chi_rows, df, n,
                       base + L.o_invs);
    VB_HIP(ctx, hipGetLastError());
  }
  std::vector<double> pr((size_t)2 * L.ld, 0.0);
  double c0p = -0.5 * (double)d * 1.8378770664093454835606594728112;
  for (int64_t i = 0; i < d; ++i) {
    pr[i] = prior_host[i];
    pr[L.ld + i] = exp(-2.0 * prior_host[d + i]);
    c0p -= prior_host[d + i];
  }
  // the tempering prior rarely changes between refreshes: its device copy is kept (keyed on the parameter values and
  // the state buffer), and in throughput mode mu is on the device already (the unpack) -- then nothing is uploaded
  // here and the refresh does not wait for the stream at all
  bool uploads = false;
  // (the key carries the copy's own offset and the layout's sizes: o_prior moves with n and n_total, and a smaller job
  // reuses the allocation -- base alone would let a second objective with another num_mc_samples read a stale copy)
  const double prior_key[5] = {(double)(uintptr_t)base, (double)L.o_prior, (double)n, (double)n_total, (double)L.ld};
  const bool prior_cached = ctx->mvt_prior.size() == (size_t)(2 * d + 5) &&
                            memcmp(ctx->mvt_prior.data() + 2 * d, prior_key, sizeof prior_key) == 0 &&
                            memcmp(ctx->mvt_prior.data(), prior_host, (size_t)(2 * d) * sizeof(double)) == 0;
  if (!prior_cached) {
    VB_HIP(ctx, hipMemcpyAsync(base + L.o_prior, pr.data(), pr.size() * sizeof(double), hipMemcpyHostToDevice, st));
    ctx->mvt_prior.assign(prior_host, prior_host + 2 * d);
    ctx->mvt_prior.insert(ctx->mvt_prior.end(), prior_key, prior_key + 5);
    uploads = true;
  }
  std::vector<double> mu((size_t)L.ld, 0.0);
  if (!dev_factors) {
    for (int64_t i = 0; i < d; ++i) mu[i] = theta_host[i];
    VB_HIP(ctx, hipMemcpyAsync(base + L.o_mu, mu.data(), mu.size() * sizeof(double), hipMemcpyHostToDevice, st));
    uploads = true;
  }
  if (uploads || !inv_s.empty()) VB_HIP(ctx, hipStreamSynchronize(st));      // stack-scoped staging buffers

  // X = mu + (Z R) / s
  GemmArgs g;
  g.A = (const double*)ns.buf.ptr;
  g.lda = ns.ld;
  const bool chol_samples = dev_factors && !sym_root;
  g.B = chol_samples ? base + L.o_lt : base + L.o_root;      // throughput mode: L' itself
  g.ldb = L.ld;
  g.M = (int)n;
  g.N = (int)d;
  g.K = (int)d;
  g.tri_mode = chol_samples ? 1 : 0;      // throughput mode: the root is L' (zero below the diagonal) -- half the product
  // (samples through the symmetric root: their residuals (x - mu) L^-T are a product, not the scaled noise)
  // throughput mode with a row-kernel target: ONE pass over samples and noise for log p, log prior, maha, log q, c_n
  const bool fuse_rows = chol_samples && mvt_env_on("VB_MVT_FUSED_ROWS") &&
                         (ctx->model.id == VB_MODEL_GAUSS_DIAG || ctx->model.id == VB_MODEL_FUNNEL);
  // ... or no pass over the samples at all (round 6; diagonal-Gaussian target and prior, at most 512 columns): log p and log
  // prior leave the sampling product's epilogue per column block, the noise rows' norms -- formed when Philox normals are
  // generated, else by a pass over the noise alone -- give the Mahalanobis terms, a kernel over the N rows puts them together.
  // Which route is taken depends on the shapes and the target only, never on what ran before.  VB_MVT_EPI_ROWS=0: the pass.
  const double* norms = nullptr;
  if (fuse_rows && ctx->model.id == VB_MODEL_GAUSS_DIAG && ctx->temper.kind == VB_PRIOR_DIAG_GAUSSIAN && d <= 512 &&
      mvt_env_on("VB_MVT_EPI_ROWS") && gemm_uses_dma(g))
    norms = noise_row_norms(ctx, const_cast<NoiseSlot&>(ns), st);      // (the slot is the context's own)
  constexpr int kEpiRowsBN = 64;      // column block of tile configuration 4 (64 x 64, two stages: this product's own choice)
  if (norms)
    gemm_f64_launch<true>(st, g, 1, n_cu,
                          EpiSampleTRows{base + L.o_x, L.ld, base + L.o_mu, base + L.o_invs,
                                         ctx->model.p0, ctx->model.p1, base + L.o_prior, base + L.o_prior + L.ld, base + L.o_u, n},
                          4);
  else
    gemm_f64_launch<true>(st, g, 1, n_cu, EpiSampleT{base + L.o_x, L.ld, base + L.o_mu, base + L.o_invs});
  VB_HIP(ctx, hipGetLastError());

  double lq_const = 0.0;
  VB_TRY(mvt_residuals(ctx, L, base, n, d, df, theta_host, linv_host, mine, dev_factors, chol_samples ? &ns : nullptr,
                       fuse_rows ? &lq_const : nullptr));
  if (norms) {
    ++ctx->mvt_epi_rows_calls;
    hipLaunchKernelGGL(mvt_rows_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const double*)(base + L.o_u), n,
                       (int)((d + kEpiRowsBN - 1) / kEpiRowsBN), n, (int)d, ctx->model.c0, c0p, norms,
                       (const double*)(base + L.o_invs), df, lq_const, base + L.o_lp + mine,
                       base + L.o_lprior + mine, base + L.o_maha, base + L.o_lq + mine, base + L.o_part);
    VB_HIP(ctx, hipGetLastError());
  } else if (fuse_rows) {
    VB_TRY(model_prior_maha_rows(ctx, base + L.o_x, L.ld, n, d, base + L.o_lp + mine, base + L.o_prior, base + L.o_prior + L.ld,
                                 c0p, base + L.o_lprior + mine, (const double*)ns.buf.ptr, ns.ld, base + L.o_invs, df, lq_const,
                                 base + L.o_maha, base + L.o_lq + mine, base + L.o_part));
  } else {
  // model and tempering prior (a diagonal Gaussian) in one pass over the samples
  VB_TRY(model_and_prior_logp_rows(ctx, base + L.o_x, L.ld, n, d, base + L.o_lp + mine, base + L.o_prior,
                                   base + L.o_prior + L.ld, c0p, base + L.o_lprior + mine));
  }
  if (ctx->temper.kind != VB_PRIOR_DIAG_GAUSSIAN)      // any other family as tempering prior: one more pass over X
    VB_TRY(temper_prior_rows(ctx, base + L.o_x, L.ld, n, d, base + L.o_lprior + mine));
  VB_TRY(mvt_side_enqueue(ctx));      // (a deferred inverse: its launches go out now that the main stream is fed)
  if (ctx->comm) {   // in-place all-gather: every rank contributed its own block
    VB_TRY(comm_gather_rows3(ctx, st, base + L.o_lq, base + L.o_lp, base + L.o_lprior, mine, n, n_total));
  }
  if (!dev_factors)      // (throughput mode: mvt_prep_kernel has zeroed them)
    VB_HIP(ctx, hipMemsetAsync(base + L.o_scal, 0, 32 * sizeof(double), st));   // scal[0] = 0: lq is used as is
  VB_TRY(dis_bisect_enqueue(ctx, base + L.o_lp, base + L.o_lq, base + L.o_lprior, base + L.o_scal, n_total, eps_prev,
                            ess_target, max_its, base + L.o_w, base + L.o_lqcopy, base + L.o_scal + 8));
  if (!w_host) {
    // device-resident step (vb_dis_step_mvt_packed): nothing comes back here -- eps, ess, the zero-weight status and the
    // weights stay on the device, the step call that follows reads them and synchronises once
    ctx->mvt_n = n;
    ctx->mvt_d = d;
    ctx->mvt_n_total = n_total;
    ctx->mvt_lq_off = mine;
    ++ctx->dis_gen[1];
    ctx->mvt_theta.assign(theta_host, theta_host + d + d * (d + 1) / 2);
    return VB_OK;
  }
  double res[3];
  const size_t vec = (size_t)n_total * sizeof(double);
  const FetchSeg segs[4] = {{base + L.o_scal + 8, sizeof res, res}, {base + L.o_w, vec, w_host},
                            {base + L.o_lp, logp_host ? vec : 0, logp_host}, {base + L.o_lq, logq_host ? vec : 0, logq_host}};
  VB_TRY(fetch_blocking(ctx, st, segs, 4));
  *eps_out = res[0];
  *ess_out = res[1];
  ctx->mvt_n = n;
  ctx->mvt_d = d;
  ctx->mvt_n_total = n_total;
  ctx->mvt_lq_off = mine;
  ++ctx->dis_gen[1];
  ctx->mvt_theta.assign(theta_host, theta_host + d + d * (d + 1) / 2);   // the residuals on the device belong to it
  if ((int)(res[2]) == 3) return fail(ctx, VB_ERR_STATE, "tempering bisection: a workgroup of the resident kernel did not arrive at a grid barrier (results invalid); VB_DIS_RESIDENT=0 selects the launch chain");
  if ((int)res[2] == 1)
    return fail(ctx, VB_ERR_NUMERIC, "All weights zero! Suggests overflow in importance density.");
  return VB_OK;
}

// ---- Pareto smoothing of the device-resident weights (DISInclusiveKL(psis_smooth=True) in throughput mode) -------------
// w -> sum(w) exp(psislw(log w)) in place (objectives.py has no such step: BASELINE configs[3] asks for "DISInclusiveKL
// with PSIS reweighting"; the smoothing itself is viabel/_psis.py:113-209, vb_psis.hip): logarithms and the total by one
// workgroup with the loads in flight, the one-workgroup PSIS kernel on them, exponentials back into the weight vector;
// khat lands next to (eps, ess, status) and comes back with the step's single synchronisation.
// (round 5: on up to kPsisPrepWg workgroups -- one workgroup took 15 us for 16 384 logarithms -- each leaving the sum of
// its slice in total_out[g]; the apply kernel adds the slices in order)
constexpr int kPsisPrepWg = 16;
__global__ void __launch_bounds__(1024) mvt_psis_prep_kernel(const double* __restrict__ w, int64_t n, double* __restrict__ lw,
                                                             double* __restrict__ total_out) {
  __shared__ double sh[16];
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t i_begin = blockIdx.x * per, i_end = i_begin + per < n ? i_begin + per : n;
  double sw = 0.0;
  for (int64_t i0 = i_begin + threadIdx.x; i0 < i_end; i0 += 8 * 1024) {
    double wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = i0 + u * 1024;
      wv[u] = i < i_end ? w[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = i0 + u * 1024;
      sw += wv[u];
      if (i < i_end) lw[i] = log(wv[u]);          // log 0 = -inf: a weight that stays zero
    }
  }
  sw = mvt_wave_sum(sw);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sw;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += sh[k];
    total_out[blockIdx.x] = t;
  }
}

__global__ void __launch_bounds__(256) mvt_psis_apply_kernel(const double* __restrict__ lw, const double* __restrict__ psis_out,
                                                             const double* __restrict__ total, int n_slices, int64_t n,
                                                             double* __restrict__ w, double* __restrict__ khat_out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double t = 0.0;
  for (int g = 0; g < n_slices; ++g) t += total[g];      // (uniform, out of the scalar cache; every thread the same order)
  if (i < n) w[i] = t * exp(lw[i]);
  if (i == 0) khat_out[0] = psis_out[0];
}

int mvt_dis_psis_enqueue(vb_ctx* ctx, int64_t n_total, double reff) {
  if (!ctx->mvt_state.ptr || ctx->mvt_n_total != n_total || n_total <= 1)
    return fail(ctx, VB_ERR_STATE, "no multivariate-t DIS state with %lld samples", (long long)n_total);
  // (sharded jobs: the weight vector covers all n_total samples on every rank -- the smoothing runs redundantly, the same
  // kernels on the same bits)
  const MvtLayout L = mvt_layout(ctx, ctx->mvt_n, ctx->mvt_n_total, ctx->mvt_d);
  double* base = (double*)ctx->mvt_state.ptr;
  hipStream_t st = ctx->stream;
  const int64_t nn = round_up(n_total, 16);
  VB_TRY(ensure(ctx, ctx->psis_lw, (size_t)(nn + 32) * sizeof(double)));
  double* lw = (double*)ctx->psis_lw.ptr;
  // (slice totals: 16 doubles behind the smoothed vector's 16-double result area)
  double* totals = lw + nn + 16;
  // (round 6: one launch when the multi-workgroup smoothing kernel applies -- it reads the weights and writes them back itself)
  bool fused = false;
  VB_TRY(psis_enqueue(ctx, n_total, reff, base + L.o_w, base + L.o_w, base + L.o_scal + 11, &fused));
  if (fused) {
    ctx->psis_n = 0;
    return VB_OK;
  }
  const int slices = (int)((n_total + 1023) / 1024) < kPsisPrepWg ? (int)((n_total + 1023) / 1024) : kPsisPrepWg;
  hipLaunchKernelGGL(mvt_psis_prep_kernel, dim3((unsigned)slices), dim3(1024), 0, st, (const double*)(base + L.o_w), n_total, lw,
                     totals);
  VB_HIP(ctx, hipGetLastError());
  VB_TRY(psis_enqueue(ctx, n_total, reff));
  hipLaunchKernelGGL(mvt_psis_apply_kernel, dim3((unsigned)((n_total + 255) / 256)), dim3(256), 0, st, (const double*)lw,
                     (const double*)(lw + nn), (const double*)totals, slices, n_total, base + L.o_w, base + L.o_scal + 11);
  VB_HIP(ctx, hipGetLastError());
  ctx->psis_n = 0;
  return VB_OK;
}

// ---- weight clipping on the device-resident weights (objectives.py:370-386) ------------------------------------------
// The fixed point the reference's recursion aims at (oracle.objectives.DISInclusiveKL._clip: the literal recursion does
// not terminate in floating point): the clipped set only grows; each round the unclipped weights are compared with
// thr * S, S = U / (1 - thr n); in the end the clipped ones are set to thr U / (1 - thr n).  One workgroup; every sum in
// a fixed order (strided per-thread partials, wave shuffles, sixteen wave totals added in order).
__device__ __forceinline__ double clip_block_sum(double v, double* sh) {
  v = mvt_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int k = 0; k < 16; ++k) t += sh[k];
  return t;
}

__global__ void __launch_bounds__(1024) dis_clip_kernel(double* __restrict__ w, int64_t n, double thr,
                                                        int* __restrict__ flag, double* __restrict__ info) {
  __shared__ double sh[16];
  const int t = threadIdx.x;
  double part = 0.0, any = 0.0;
  for (int64_t i = t; i < n; i += 1024) {
    part += w[i];
    flag[i] = 0;
  }
  double S = clip_block_sum(part, sh);
  for (int64_t i = t; i < n; i += 1024) any += w[i] > S * thr ? 1.0 : 0.0;
  any = clip_block_sum(any, sh);
  double n_clipped = 0.0, U_acc = 0.0;
  if (any > 0.0) {
    for (int64_t round = 0; round <= n; ++round) {
      double n_new = 0.0, u = 0.0;
      for (int64_t i = t; i < n; i += 1024) {
        if (flag[i]) continue;
        const double wi = w[i];
        if (wi >= S * thr) n_new += 1.0;
        else u += wi;
      }
      n_new = clip_block_sum(n_new, sh);
      if (n_new == 0.0) break;
      u = clip_block_sum(u, sh);
      const double n_trial = n_clipped + n_new;
      if (u == 0.0 || 1.0 - thr * n_trial <= 0.0) break;
      for (int64_t i = t; i < n; i += 1024)
        if (!flag[i] && w[i] >= S * thr) flag[i] = 1;
      n_clipped = n_trial;
      U_acc = u;
      S = u / (1.0 - thr * n_trial);
      __syncthreads();
    }
  }
  if (n_clipped > 0.0) {
    const double v = thr * U_acc / (1.0 - thr * n_clipped);
    for (int64_t i = t; i < n; i += 1024)
      if (flag[i]) w[i] = v;
  }
  if (t == 0) info[0] = n_clipped;
}

int mvt_dis_clip_enqueue(vb_ctx* ctx, int64_t n_total, double threshold) {
  if (!ctx->mvt_state.ptr || ctx->mvt_n_total != n_total || n_total <= 0)
    return fail(ctx, VB_ERR_STATE, "no multivariate-t DIS state with %lld samples", (long long)n_total);
  if (!(threshold > 0.0)) return fail(ctx, VB_ERR_INVALID, "clipping threshold must be positive");
  const MvtLayout L = mvt_layout(ctx, ctx->mvt_n, ctx->mvt_n_total, ctx->mvt_d);
  double* base = (double*)ctx->mvt_state.ptr;
  hipLaunchKernelGGL(dis_clip_kernel, dim3(1), dim3(1024), 0, ctx->stream, base + L.o_w, n_total, threshold,
                     (int*)(base + L.o_cnt), base + L.o_scal + 13);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// tempered weights of the last refresh (before any resampling), for callers that left them on the device
int mvt_dis_weights_get(vb_ctx* ctx, double* w_host, int64_t n_total, int resampled) {
  if (!ctx->mvt_state.ptr || ctx->mvt_n_total != n_total || n_total <= 0)
    return fail(ctx, VB_ERR_STATE, "no multivariate-t DIS state with %lld samples", (long long)n_total);
  const MvtLayout L = mvt_layout(ctx, ctx->mvt_n, ctx->mvt_n_total, ctx->mvt_d);
  VB_HIP(ctx, hipMemcpyAsync(w_host, (double*)ctx->mvt_state.ptr + (resampled ? L.o_wres : L.o_w), (size_t)n_total * sizeof(double),
                             hipMemcpyDeviceToHost, ctx->stream));
  legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
  VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  legacy_poll(ctx);
  return VB_OK;
}

// log p / log q of the state samples (all n_total of them), for callers that did not take them from the refresh
int mvt_dis_state_get(vb_ctx* ctx, double* logp_host, double* logq_host, int64_t n_total) {
  if (!ctx->mvt_state.ptr || ctx->mvt_n_total != n_total || n_total <= 0)
    return fail(ctx, VB_ERR_STATE, "no multivariate-t DIS state with %lld samples", (long long)n_total);
  const MvtLayout L = mvt_layout(ctx, ctx->mvt_n, ctx->mvt_n_total, ctx->mvt_d);
  double* base = (double*)ctx->mvt_state.ptr;
  hipStream_t st = ctx->stream;
  if (logp_host)
    VB_HIP(ctx, hipMemcpyAsync(logp_host, base + L.o_lp, (size_t)n_total * sizeof(double), hipMemcpyDeviceToHost, st));
  if (logq_host)
    VB_HIP(ctx, hipMemcpyAsync(logq_host, base + L.o_lqcopy, (size_t)n_total * sizeof(double), hipMemcpyDeviceToHost, st));
  legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
  VB_HIP(ctx, hipStreamSynchronize(st));
  legacy_poll(ctx);
  return VB_OK;
}

// [eps, ess, zero-weight status, khat] of the last device-resident refresh
int mvt_dis_scalars_get(vb_ctx* ctx, double out[4]) {
  if (!ctx->mvt_state.ptr || ctx->mvt_n <= 0) return fail(ctx, VB_ERR_STATE, "no multivariate-t DIS state");
  const MvtLayout L = mvt_layout(ctx, ctx->mvt_n, ctx->mvt_n_total, ctx->mvt_d);
  double* base = (double*)ctx->mvt_state.ptr;
  VB_HIP(ctx, hipMemcpyAsync(out, base + L.o_scal + 8, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
  VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  legacy_poll(ctx);
  return VB_OK;
}

// sum_n w_n [log q_n, d log q_n / d mu, u_n u_n' c_n] for the state samples at parameter theta_host
// packed_out != nullptr (throughput mode): the chain rule to the flat parameter runs on the device too and
// packed_out = [value | grad] of -scale sum_n w_n log q(x_n; theta) comes back instead of the raw sums
// w_host == nullptr (packed_out only): the weights are on the device already -- the refresh's own (resample_m == 0) or
// multinomial counts drawn from them here (resample_m > 0, scale then multiplies sum w on the device); res_out receives
// [eps, ess, zero-weight status] of the last refresh in the same synchronisation
int mvt_dis_grad(vb_ctx* ctx, int64_t n, int64_t d, double df, const double* theta_host, const double* linv_host,
                 const double* w_host, double* wsum_out, double* wlogq_out, double* dmu_out, double* gram_out,
                 double scale, double* packed_out, int64_t resample_m, uint64_t seed, uint64_t stream, double* res_out,
                 double* grad_direct) {
  if (ctx->mvt_n != n || ctx->mvt_d != d || !ctx->mvt_state.ptr)
    return fail(ctx, VB_ERR_STATE, "no multivariate-t DIS state of shape %lld x %lld", (long long)n, (long long)d);
  const MvtLayout L = mvt_layout(ctx, n, ctx->mvt_n_total, d);
  double* base = (double*)ctx->mvt_state.ptr;
  hipStream_t st = ctx->stream;
  const int n_cu = ctx->prop.multiProcessorCount;
  // this rank's rows inside the whole-job vectors (weights, counts): shard_rows
  int64_t mine = 0;
  VB_TRY(comm_shard_begin(ctx, n, ctx->mvt_n_total, &mine));
  const int64_t n_all = ctx->mvt_n_total;
  const double* wdev = base + L.o_w + mine;
  const double* scale_dev = nullptr;
  if (w_host) {
    // the caller's weights of THIS rank's rows (resampling counts of a host draw, or weights it smoothed / clipped itself)
    // go into the count area: the resident tempered weights of the refresh stay what they are
    VB_HIP(ctx, hipMemcpyAsync(base + L.o_wres, w_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
    VB_HIP(ctx, hipStreamSynchronize(st));
    legacy_poll(ctx);
    wdev = base + L.o_wres;
  } else {
    if (!packed_out) return fail(ctx, VB_ERR_UNSUPPORTED, "device-resident weights: packed gradient only");
    if (resample_m > 0) {
      // the multinomial draw over ALL n_total weights, redundantly on every rank of a sharded job (counter-based
      // uniforms, integer counts: the same counts everywhere); each rank then weights its own rows with its block
      int* counts = (int*)(base + L.o_cnt);
      if ((n_all + kCdfChunk - 1) / kCdfChunk > kCdfMaxChunks)
        return fail(ctx, VB_ERR_UNSUPPORTED, "device multinomial draw: at most %d weights", kCdfChunk * kCdfMaxChunks);
      hipLaunchKernelGGL(mvt_cdf_kernel, dim3((unsigned)((n_all + kCdfChunk - 1) / kCdfChunk)), dim3(kCdfChunk), 0, st,
                         (const double*)(base + L.o_w), n_all, base + L.o_cdf, base + L.o_cdf + L.nn + 16, counts);
      hipLaunchKernelGGL(mvt_draw_kernel, dim3((unsigned)((resample_m + 255) / 256)), dim3(256), 0, st,
                         (const double*)(base + L.o_cdf), (const double*)(base + L.o_cdf + L.nn + 16), base + L.o_cdf + L.nn, n_all,
                         resample_m, seed, stream, counts);
      hipLaunchKernelGGL(mvt_counts_kernel, dim3((unsigned)((n_all + 255) / 256)), dim3(256), 0, st, (const int*)counts, n_all,
                         base + L.o_wres);
      VB_HIP(ctx, hipGetLastError());
      wdev = base + L.o_wres + mine;
      scale_dev = base + L.o_cdf + L.nn;
    }
  }
  {
    // the residuals E' = (X - mu) L^-T, the Mahalanobis distances and log q on the device belong to the parameter of
    // the last residual pass: a gradient at that same parameter (the call that follows a refresh when
    // num_resampling_batches = 1) reuses them instead of repeating the N x D x D product and its uploads
    const size_t p = (size_t)(d + d * (d + 1) / 2);
    const bool same = ctx->mvt_theta.size() == p && memcmp(ctx->mvt_theta.data(), theta_host, p * sizeof(double)) == 0;
    if (!same) {
      VB_TRY(mvt_residuals(ctx, L, base, n, d, df, theta_host, linv_host, 0));
      ctx->mvt_lq_off = 0;      // (this rank's log q now start the vector; the gathered copy of the refresh is in o_lqcopy)
      ctx->mvt_theta.assign(theta_host, theta_host + p);
      ctx->mvt_dev_factors = linv_host == nullptr;
    } else if (packed_out && !ctx->mvt_dev_factors) {
      VB_TRY(mvt_factors_device(ctx, L, base, d, theta_host));      // the chain rule below needs L on the device
      ctx->mvt_dev_factors = true;
    }
  }
  const double* lq_rows = base + L.o_lq + ctx->mvt_lq_off;      // log q of this rank's rows at theta_host
  FrSums S = L.S;
  S.sums = base + L.o_sums;
  const int64_t n_part = (n + 3) / 4;
  const double* cn = base + L.o_part;       // n doubles, written by mvt_rows_kernel
  double* fpart = base + L.o_part + n + 2 * n_part;
  const int64_t slab = d * L.ld;
  // The residuals y_n = L^-1 (x_n - mu) as rows: the noise matrix (scaled by 1 / s_n on the fly) or E'.
  const bool noise_rows = ctx->mvt_e_noise != nullptr;
  const double* Y = noise_rows ? ctx->mvt_e_noise : (const double*)(base + L.o_e);
  const int64_t ldy = noise_rows ? ctx->mvt_e_noise_ld : L.ld;
  // Packed chain rule (round 5): sum_n a_n u_n y_n' = L^-T M with M = sum_n a_n y_n y_n' and d/dmu = L^-T sum_n a_n y_n
  // (u_n = L^-T y_n), so the N x D x D product U = E' L^-1 is not formed at all: one pass over Y leaves the scaled
  // operand a_n y_n and the weighted column sums, the Gram product runs on Y itself, and L^-T enters once, in the
  // D x D x D product of the chain rule (where L used to).  VB_MVT_DIRECT=0: the route through U.
  const bool direct_env = mvt_env_on("VB_MVT_DIRECT");
  const bool direct = packed_out != nullptr && ldy == L.ld && direct_env;
  if (!direct) VB_TRY(mvt_join_inverse(ctx));
  if (direct) {
    const double* rs = noise_rows ? (const double*)(base + L.o_invs) : nullptr;
    double* wpart = base + L.o_part + n;      // 2 n_rb doubles: the scale pass's (sum w, sum w log q) per row block
    hipLaunchKernelGGL(mvt_scale_colsum_kernel, dim3((unsigned)((d + 63) / 64), (unsigned)L.n_rb), dim3(256), 0, st, Y, L.ld, n,
                       (int)d, wdev, cn, rs, base + L.o_ua, base + L.o_col, lq_rows, wpart);
    VB_HIP(ctx, hipGetLastError());
    VB_TRY(gram_lower_enqueue(ctx, base + L.o_ua, Y, L.ld, (int)d, n, L.splits, base + L.o_cpart, L.ld, slab));
    VB_TRY(fr_reduce_enqueue(ctx, base + L.o_cpart, L.splits, slab, (int)d, L.ld, base + L.o_col, L.n_rb, L.ld, fpart, 0, S,
                             true, wpart));
  } else {
  // U = E' L^-1
  GemmArgs g;
  g.A = Y;
  g.lda = ldy;
  g.B = base + L.o_li;
  g.ldb = L.ld;
  g.M = (int)n;
  g.N = (int)d;
  g.K = (int)d;
  g.tri_mode = 3;          // L^-1 is lower triangular: B[k][j] == 0 for k < j -- half the product
  // (sum w, sum w log q) ride in slots 1 and 2 of the sum vector so that one all-reduce covers everything; U and its
  // row-scaled copy a_n U (a_n = w_n c_n, c_n left behind by the residual pass) leave the GEMM together
  hipLaunchKernelGGL(mvt_wsums_kernel, dim3(1), dim3(1024), 0, st, wdev, lq_rows, n, S.sums + 1);
  gemm_f64_launch<true>(st, g, 1, n_cu, EpiStoreScaled{base + L.o_u, base + L.o_ua, L.ld, wdev, cn,
                                                       noise_rows ? (const double*)(base + L.o_invs) : nullptr});
  VB_HIP(ctx, hipGetLastError());
  // weighted Gram product U' diag(a) U (lower tiles) with the column sums of a U out of the same kernel
  bool cs_fused = false;
  VB_TRY(gram_lower_colsum_enqueue(ctx, base + L.o_ua, base + L.o_u, L.ld, (int)d, n, L.splits, base + L.o_cpart, L.ld,
                                   slab, base + L.o_col, L.ld, L.n_rb, &cs_fused));
  if (!cs_fused)
    VB_TRY(fr_colsum_enqueue(ctx, base + L.o_ua, nullptr, L.ld, n, (int)d, 0, nullptr, base + L.o_col, fpart));
  // (packed chain rule: the Gram matrix leaves the reduction symmetric -- both triangles -- and S L reads it where it is)
  VB_TRY(fr_reduce_enqueue(ctx, base + L.o_cpart, L.splits, slab, (int)d, L.ld, base + L.o_col,
                           cs_fused ? L.splits : L.n_rb, L.ld, fpart,
                           cs_fused ? 0 : L.n_rb * (int)((d + 127) / 128), S, packed_out != nullptr));
  }
  if (ctx->comm) VB_TRY(comm_allreduce_sum(ctx, st, S.sums, (size_t)S.len));
  if (packed_out) {
    // S = sym(gram), dL = tril(S L) - w_sum diag(1 / L_ii), free diagonal x L_ii: one D x D x D product and a pack kernel
    const int D = (int)d;
    VB_TRY(mvt_join_inverse(ctx));
    const size_t plen = (size_t)(d + d * (d + 1) / 2);
    if (direct && d <= 512 && mvt_env_on("VB_MVT_CHAIN")) {
      // (direct route, moderate D: product and pack as ONE launch over the lower 32 x 32 tiles -- mvt_chain_kernel)
      const int nt = (D + 31) / 32;
      // a blocking call's results leave with the kernel's own stores (ChainHost; VB_MVT_CHAIN_FETCH=0: the gathering launch)
      double tail[5];
      const FetchSeg segs[2] = {{base + L.o_grad + 1, plen * sizeof(double), grad_direct},
                                {base + L.o_grad + 1 + plen, sizeof tail, tail}};
      FetchPlan plan;
      ChainHost H;
      if (grad_direct && mvt_env_on("VB_MVT_FLAGSYNC") && mvt_env_on("VB_MVT_CHAIN_FETCH")) {
        VB_TRY(fetch_plan(ctx, segs, 2, &plan));
        if (plan.ok) {
          H.grad = (double*)(plan.dev + plan.first[0]);
          H.tail = (double*)(plan.dev + plan.first[1]);
          H.ticket = plan.ticket, H.done = plan.done_dev, H.seq = plan.seq;
        }
      }
      hipLaunchKernelGGL(mvt_chain_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 0, st, (const double*)(base + L.o_wt),
                         (const double*)(base + L.o_lfull), L.ld, D, (const double*)S.sums, S.off_col, S.off_c, scale,
                         base + L.o_grad, scale_dev, (const double*)(base + L.o_scal + 8), H);
      VB_HIP(ctx, hipGetLastError());
      if (H.grad) {
        ++ctx->mvt_chain_fetch_calls;
        VB_TRY(fetch_wait(ctx, st, plan, segs));
        packed_out[0] = tail[4];
        if (res_out)
          for (int q = 0; q < 4; ++q) res_out[q] = tail[q];
        return VB_OK;
      }
    } else {
    GemmArgs gs;
    gs.A = direct ? base + L.o_wt : S.sums + S.off_c;      // direct: L^-T M instead of S L
    gs.lda = L.ld;
    gs.B = direct ? S.sums + S.off_c : base + L.o_lfull;
    gs.ldb = L.ld;
    gs.M = D;
    gs.N = D;
    gs.K = D;
    gs.tri_mode = 0;
    gemm_f64_launch<true>(st, gs, 1, n_cu, EpiStore{base + L.o_sl, L.ld});
    hipLaunchKernelGGL(mvt_pack_grad_kernel, dim3((unsigned)(((int64_t)D * D + 255) / 256 + (direct ? (D + 3) / 4 : 0))), dim3(256), 0, st,
                       (const double*)(base + L.o_sl), (const double*)(base + L.o_lfull), L.ld, D,
                       (const double*)S.sums, S.off_col, scale, base + L.o_grad, scale_dev,
                       (const double*)(base + L.o_scal + 8), direct ? (const double*)(base + L.o_wt) : (const double*)nullptr);
    VB_HIP(ctx, hipGetLastError());
    }
    if (grad_direct && mvt_env_on("VB_MVT_FLAGSYNC")) {
      // gradient and the five scalars through mapped memory behind a polled completion word (fetch_blocking)
      double tail[5];
      const FetchSeg segs[2] = {{base + L.o_grad + 1, plen * sizeof(double), grad_direct},
                                {base + L.o_grad + 1 + plen, sizeof tail, tail}};
      VB_TRY(fetch_blocking(ctx, st, segs, 2));
      packed_out[0] = tail[4];
      if (res_out)
        for (int q = 0; q < 4; ++q) res_out[q] = tail[q];
      return VB_OK;
    }
    if (grad_direct) {       // gradient into the caller's array, the five scalars in one small copy (no host staging)
      double tail[5];
      VB_HIP(ctx, hipMemcpyAsync(grad_direct, base + L.o_grad + 1, plen * sizeof(double), hipMemcpyDeviceToHost, st));
      VB_HIP(ctx, hipMemcpyAsync(tail, base + L.o_grad + 1 + plen, sizeof tail, hipMemcpyDeviceToHost, st));
      legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
      VB_HIP(ctx, hipStreamSynchronize(st));
      legacy_poll(ctx);
      packed_out[0] = tail[4];
      if (res_out)
        for (int q = 0; q < 4; ++q) res_out[q] = tail[q];
      return VB_OK;
    }
    VB_HIP(ctx, hipMemcpyAsync(packed_out, base + L.o_grad, (1 + plen) * sizeof(double), hipMemcpyDeviceToHost, st));
    if (res_out) VB_HIP(ctx, hipMemcpyAsync(res_out, base + L.o_scal + 8, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
    legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
    VB_HIP(ctx, hipStreamSynchronize(st));
    legacy_poll(ctx);
    return VB_OK;
  }
  double sc[2];
  VB_HIP(ctx, hipMemcpyAsync(sc, S.sums + 1, sizeof sc, hipMemcpyDeviceToHost, st));
  VB_HIP(ctx, hipMemcpyAsync(dmu_out, S.sums + S.off_col, (size_t)d * sizeof(double), hipMemcpyDeviceToHost, st));
  VB_HIP(ctx, hipMemcpy2DAsync(gram_out, (size_t)d * sizeof(double), S.sums + S.off_c, (size_t)L.ld * sizeof(double),
                               (size_t)d * sizeof(double), (size_t)d, hipMemcpyDeviceToHost, st));
  legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
  VB_HIP(ctx, hipStreamSynchronize(st));
  legacy_poll(ctx);
  *wsum_out = sc[0];
  *wlogq_out = sc[1];
  return VB_OK;
}

// ---- ExclusiveKL (entropy form, objectives.py:154-164) for the multivariate t --------------------------------
// x_n = mu + (z_n R) / s_n with R = Sigma^{1/2} (symmetric).  The device returns the sample sums
//   F = sum_n f(x_n),  sum_n g_n,  C = sum_n g_n (z_n / s_n)'   (full D x D: dF / dR for an unconstrained R);
// the O(D^3) chain rule R -> Sigma -> free Cholesky parameters is the caller's (host), like the symmetric root.
// ---- path derivative over a multivariate t: noise-only sums (see vb_mvt_path_terms in the header) --------
// one wave per row: maha_n = |z_n|^2 / s_n^2, c_n = (df + D) / (df + maha_n);  Es[n] = sqrt(c_n) z_n / s_n (so that
// Es' Es = m_w), a_n = c_n / s_n, and per-block partial sums of log(1 + maha_n / df)
__global__ void __launch_bounds__(256) mvt_path_rows_kernel(const double* __restrict__ E, int64_t ld, int64_t n, int d,
                                                            double df, const double* __restrict__ inv_s,
                                                            double* __restrict__ Es, int64_t ldw,
                                                            double* __restrict__ a, double* __restrict__ part) {
  __shared__ double sh[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  double l1p = 0.0;
  if (row < n) {
    const double* e = E + row * ld;
    double ss = 0.0;
    for (int c = lane; c < d; c += 64) ss = fma(e[c], e[c], ss);
    ss = mvt_wave_sum(ss);
    ss = __shfl(ss, 0, 64);
    const double is = inv_s[row];
    const double maha = ss * is * is;
    const double cn = (df + d) / (df + maha);
    const double sc = sqrt(cn) * is;
    double* w = Es + row * ldw;
    for (int c = lane; c < d; c += 64) w[c] = sc * e[c];
    if (lane == 0) {
      a[row] = cn * is;
      l1p = log1p(maha / df);
    }
  }
  if (lane == 0) sh[wave] = l1p;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// The noise-only sums of the path derivative on the device: S->sums = [sum log1p(maha / df) | e_w (from off_col) | m_w
// (from off_c, row stride round_up(d, 16); mirror: both triangles)] in ctx->mvt_elbo.  inv_s_dev: n row scales on the
// device; inv_s_host != nullptr: uploaded first.
static int mvt_path_terms_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, double df,
                                  const double* inv_s_host, const double* inv_s_dev, bool mirror, FrSums* S_out) {
  const int64_t ldw = round_up(d, 16), slab = d * ldw;
  const int splits = gram_splits(ctx, (int)d, n);
  const int n_rb = (int)((n + 127) / 128);
  const int n_part = (int)((n + 3) / 4);
  int64_t off = 0;
  auto carve = [&off](int64_t doubles) {
    const int64_t o = off;
    off += round_up(doubles, 16);
    return o;
  };
  FrSums S;
  S.off_col = 16;
  S.off_c = 16 + ns.ld;
  S.len = 16 + ns.ld + slab;
  const int64_t o_invs = carve(n), o_a = carve(n), o_es = carve(n * ldw), o_cpart = carve((int64_t)splits * slab),
                o_col = carve((int64_t)n_rb * ns.ld), o_part = carve(n_part), o_sums = carve(S.len),
                o_fdummy = carve((int64_t)n_rb * ((d + 127) / 128));   // the column-sum kernel's (zero) f partials
  VB_TRY(ensure(ctx, ctx->mvt_elbo, (size_t)off * sizeof(double)));
  double* base = (double*)ctx->mvt_elbo.ptr;
  S.sums = base + o_sums;
  hipStream_t st = ctx->stream;
  if (inv_s_host) {
    VB_HIP(ctx, hipMemcpyAsync(base + o_invs, inv_s_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    inv_s_dev = base + o_invs;
  }
  VB_HIP(ctx, hipMemsetAsync(base + o_es, 0, (size_t)(n * ldw) * sizeof(double), st));   // pad columns
  const double* E = (const double*)ns.buf.ptr;
  hipLaunchKernelGGL(mvt_path_rows_kernel, dim3((unsigned)n_part), dim3(256), 0, st, E, ns.ld, n, (int)d, df,
                     inv_s_dev, base + o_es, ldw, base + o_a, base + o_part);
  VB_HIP(ctx, hipGetLastError());
  VB_TRY(gram_lower_enqueue(ctx, base + o_es, base + o_es, ldw, (int)d, n, splits, base + o_cpart, ldw, slab));
  VB_TRY(fr_colsum_enqueue(ctx, E, nullptr, ns.ld, n, (int)d, 0, nullptr, base + o_col, base + o_fdummy,
                           base + o_a));
  VB_TRY(fr_reduce_enqueue(ctx, base + o_cpart, splits, slab, (int)d, ldw, base + o_col, n_rb, ns.ld, base + o_part,
                           n_part, S, mirror));
  if (ctx->comm) VB_TRY(comm_allreduce_sum(ctx, st, S.sums, (size_t)S.len));
  *S_out = S;
  return VB_OK;
}

int mvt_path_terms(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int64_t n_total, double df,
                   const double* inv_s_host, double* m_w, double* e_w, double* log1p_sum) {
  if (n <= 0 || n > ns.n || d != ns.d) return fail(ctx, VB_ERR_INVALID, "noise slot shape mismatch");
  const int64_t ldw = round_up(d, 16);
  FrSums S;
  VB_TRY(mvt_path_terms_enqueue(ctx, ns, n, d, df, inv_s_host, nullptr, false, &S));
  hipStream_t st = ctx->stream;
  std::vector<double> low((size_t)d * d);
  VB_HIP(ctx, hipMemcpyAsync(log1p_sum, S.sums, sizeof(double), hipMemcpyDeviceToHost, st));
  VB_HIP(ctx, hipMemcpyAsync(e_w, S.sums + S.off_col, (size_t)d * sizeof(double), hipMemcpyDeviceToHost, st));
  VB_HIP(ctx, hipMemcpy2DAsync(low.data(), (size_t)d * sizeof(double), S.sums + S.off_c, (size_t)ldw * sizeof(double),
                               (size_t)d * sizeof(double), (size_t)d, hipMemcpyDeviceToHost, st));
  legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
  VB_HIP(ctx, hipStreamSynchronize(st));
  legacy_poll(ctx);
  for (int64_t i = 0; i < d; ++i)
    for (int64_t j = 0; j <= i; ++j) m_w[i * d + j] = m_w[j * d + i] = low[(size_t)(i * d + j)];
  (void)n_total;
  return VB_OK;
}

int mvt_elbo_sums(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int64_t n_total, const double* mu_host,
                  const double* root_host, const double* inv_s_host, double* f_sum, double* g_sum, double* c_full) {
  if (n <= 0 || n > ns.n || d != ns.d) return fail(ctx, VB_ERR_INVALID, "noise slot shape mismatch");
  const int64_t ld = round_up(d, 16);
  int64_t off = 0;
  auto carve = [&off](int64_t doubles) {
    const int64_t o = off;
    off += round_up(doubles, 16);
    return o;
  };
  const int64_t o_root = carve(d * ld), o_mu = carve(ld), o_invs = carve(n);
  VB_TRY(ensure(ctx, ctx->mvt_elbo, (size_t)off * sizeof(double)));
  double* base = (double*)ctx->mvt_elbo.ptr;
  VB_TRY(upload_padded(ctx, base + o_root, ld, root_host, d, d, false));
  VB_HIP(ctx, hipMemcpyAsync(base + o_mu, mu_host, (size_t)d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  VB_HIP(ctx, hipMemcpyAsync(base + o_invs, inv_s_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  FrSums S;
  VB_TRY(fr_pipeline_enqueue(ctx, ns, n, d, n_total, nullptr, nullptr, base + o_mu, base + o_root, base + o_invs, &S));
  hipStream_t st = ctx->stream;
  VB_HIP(ctx, hipMemcpyAsync(f_sum, S.sums, sizeof(double), hipMemcpyDeviceToHost, st));
  VB_HIP(ctx, hipMemcpyAsync(g_sum, S.sums + S.off_col, (size_t)d * sizeof(double), hipMemcpyDeviceToHost, st));
  VB_HIP(ctx, hipMemcpy2DAsync(c_full, (size_t)d * sizeof(double), S.sums + S.off_c, (size_t)ld * sizeof(double),
                               (size_t)d * sizeof(double), (size_t)d, hipMemcpyDeviceToHost, st));
  legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
  VB_HIP(ctx, hipStreamSynchronize(st));
  legacy_poll(ctx);
  // the per-sample additive constant of f; the all-reduced sums cover every rank's rows
  *f_sum += (double)(ctx->comm ? n_total : n) * ctx->model.c0;
  return VB_OK;
}

// ---- the reference-identical ExclusiveKL of the multivariate t, resident on the device ------------------------------------
// (objectives.py:154-164 over approximations.py:342-349, rng='numpy'.)  The noise slot and the context's
// chi-square draws hold numpy's streams (vb_legacy_rng_chisquare_device, _randn_device); mu, L, L' come from theta on the
// device; the symmetric root R of Sigma = L L' by sym_sqrt_dev; the sample sums F, sum g, C = sum g (z / s)' by the
// dense-family pipeline; then the chain rule the host used to run in numpy:
//     Gs = (C + C') / (2 N);   R X + X R = Gs (sym_sqrt_frechet_dev);   dL = tril(2 X L), free (log) diagonal x L_ii, + 1
//     value = -(F / N + c0 + sum log L_ii),   grad = -[sum g / N | dL]          (the family's entropy drops the df-only terms)
namespace {

// Gs = (C + C') / (2 N) on the d x ld layout
__global__ void __launch_bounds__(256) mvt_gs_kernel(const double* __restrict__ C, double* __restrict__ Gs, int d, int64_t ld,
                                                     double inv_2n) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)d * ld) return;
  const int i = (int)(idx / ld), j = (int)(idx % ld);
  Gs[idx] = j < d ? (C[idx] + C[(int64_t)j * ld + i]) * inv_2n : 0.0;
}

// v_i += sum_k A[i][k] x_k, one wave per component (the path derivative's sum g += Sigma^(-1/2) e_w)
__global__ void __launch_bounds__(256) mvt_matvec_add_kernel(const double* __restrict__ A, int64_t ld, int d,
                                                             const double* __restrict__ x, double* __restrict__ v) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= d) return;
  double a = 0.0;
  for (int k = lane; k < d; k += 64) a = fma(A[(int64_t)i * ld + k], x[k], a);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
  if (lane == 0) v[i] += a;
}

// out = [value | grad]: XL = X L (d x ld), Lfull = L, theta's log-diagonal for the entropy
// l1p != nullptr: the path-derivative form (objectives.py:156-159) -- the value takes the samples' mean log density
// lqc - sum log L_ii - half_dfd l1p[0] / N instead of the entropy, and the free diagonal has no entropy term
__global__ void __launch_bounds__(256) mvt_ekl_pack_kernel(const double* __restrict__ XL, const double* __restrict__ Lfull,
                                                           const double* __restrict__ theta, int64_t ld, int d,
                                                           const double* __restrict__ sums, int64_t off_col, double inv_n,
                                                           double c0, double* __restrict__ out,
                                                           const double* __restrict__ l1p = nullptr, double lqc = 0.0,
                                                           double half_dfd = 0.0) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0) {      // value: one workgroup adds the log-diagonal in a fixed order
    __shared__ double sh[4];
    double s = 0.0;
    for (int j = threadIdx.x; j < d; j += 256) s += theta[d + (int64_t)j * (j + 1) / 2 + j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      double v = (sums[0] * inv_n + c0) + ((sh[0] + sh[1]) + (sh[2] + sh[3]));
      if (l1p) v += half_dfd * l1p[0] * inv_n - lqc;
      out[0] = -v;
    }
  }
  if (idx < d) out[1 + idx] = -(sums[off_col + idx] * inv_n);
  if (idx >= (int64_t)d * d) return;
  const int i = (int)(idx / d), j = (int)(idx % d);
  if (j > i) return;
  double g = 2.0 * XL[(int64_t)i * ld + j];
  if (i == j) g = g * Lfull[(int64_t)i * ld + i] + (l1p ? 0.0 : 1.0);
  out[1 + d + (int64_t)i * (i + 1) / 2 + j] = -g;
}

}  // namespace

// value_grad_host: 1 + d + d (d + 1) / 2 doubles.  VB_ERR_UNSUPPORTED: a root iteration did not resolve (nothing returned).
// path_deriv: objectives.py:156-159 -- the model gradient g_n gives way to g_n - d log q / dx (x_n) = g_n + c_n
// Sigma^(-1/2) z_n / s_n, whose part of the sums depends on the noise only: C += Sigma^(-1/2) m_w, sum g += Sigma^(-1/2) e_w
// (mvt_path_terms_enqueue); Sigma^(-1/2) is the coupled iteration's second limit.
struct EpiAddTo {           // C += acc
  double* C;
  int64_t ld;
  __device__ void operator()(int, int row, int col, double acc) const { C[(int64_t)row * ld + col] += acc; }
  __device__ d2v pair(int, int row, int col, double a0, double a1) const {
    d2v* p = reinterpret_cast<d2v*>(C + (int64_t)row * ld + col);
    const d2v v = (d2v){p->x + a0, p->y + a1};
    *p = v;
    return v;
  }
};

int mvt_elbo_symroot(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t n_total, int64_t d, double df,
                     const double* theta_host, double* value_grad_host, double* info, bool path_deriv) {
  // Sharded jobs (round 6): this rank's rows of numpy's streams, the sample sums all-reduced inside fr_pipeline_enqueue /
  // mvt_path_terms_enqueue, everything of order D^3 -- both Newton-Schulz iterations, the chain rule -- redundantly on
  // every rank from the same bits.
  int64_t mine = 0;
  VB_TRY(comm_shard_begin(ctx, n, n_total, &mine));
  if (n <= 0 || n > ns.n || d != ns.d) return fail(ctx, VB_ERR_INVALID, "noise slot shape mismatch");
  const double* chi_rows = mvt_chi_rows(ctx, n, n_total, mine, df);
  if (!chi_rows)
    return fail(ctx, VB_ERR_STATE, "needs %lld (or all %lld) device chi-square(%g) draws (vb_legacy_rng_chisquare_device)",
                (long long)n, (long long)n_total, df);
  // (a buffer of its own: the DIS state of ctx->mvt_state -- an interleaved DISInclusiveKL's samples -- is left alone)
  const MvtLayout L = mvt_layout(ctx, n, n_total, d);
  VB_TRY(ensure(ctx, ctx->mvt_ekl_state, (size_t)L.total * sizeof(double)));
  double* base = (double*)ctx->mvt_ekl_state.ptr;
  hipStream_t st = ctx->stream;
  const int n_cu = ctx->prop.multiProcessorCount;
  const int D = (int)d;
  const int64_t sq = d * L.ld;
  VB_TRY(mvt_factors_device(ctx, L, base, d, theta_host));
  double rinfo[3] = {0.0, 0.0, 0.0};
  // (the inverse root, when asked for, lands in the slot of L^-1, which this path does not read)
  VB_TRY(sym_sqrt_dev(ctx, base + L.o_lfull, base + L.o_lt, d, L.ld, base + L.o_root, 1e-12, rinfo,
                      path_deriv ? base + L.o_li : (double*)nullptr));
  hipLaunchKernelGGL(mvt_inv_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, chi_rows, df, n, base + L.o_invs);
  FrSums S;
  VB_TRY(fr_pipeline_enqueue(ctx, ns, n, d, n_total, nullptr, nullptr, base + L.o_mu, base + L.o_root, base + L.o_invs, &S));
  FrSums P;
  P.sums = nullptr;
  if (path_deriv) {
    VB_TRY(mvt_path_terms_enqueue(ctx, ns, n, d, df, nullptr, base + L.o_invs, true, &P));
    GemmArgs gp;      // C += Sigma^(-1/2) m_w
    gp.A = base + L.o_li, gp.lda = L.ld, gp.B = P.sums + P.off_c, gp.ldb = L.ld;
    gp.M = D, gp.N = D, gp.K = D, gp.tri_mode = 0;
    gemm_f64_launch<true>(st, gp, 1, n_cu, EpiAddTo{S.sums + S.off_c, L.ld});
    hipLaunchKernelGGL(mvt_matvec_add_kernel, dim3((unsigned)((D + 3) / 4)), dim3(256), 0, st, (const double*)(base + L.o_li),
                       L.ld, D, (const double*)(P.sums + P.off_col), S.sums + S.off_col);
    VB_HIP(ctx, hipGetLastError());
  }
  hipLaunchKernelGGL(mvt_gs_kernel, dim3((unsigned)((sq + 255) / 256)), dim3(256), 0, st, (const double*)(S.sums + S.off_c),
                     base + L.o_tscr, D, L.ld, 0.5 / (double)n_total);
  VB_HIP(ctx, hipGetLastError());
  double xinfo[3] = {0.0, 0.0, 0.0};
  VB_TRY(sym_sqrt_frechet_dev(ctx, base + L.o_lfull, base + L.o_lt, base + L.o_tscr, d, L.ld, base + L.o_sl, 1e-12, xinfo));
  GemmArgs g;      // X L
  g.A = base + L.o_sl, g.lda = L.ld, g.B = base + L.o_lfull, g.ldb = L.ld;
  g.M = D, g.N = D, g.K = D, g.tri_mode = 0;
  gemm_f64_launch<true>(st, g, 1, n_cu, EpiStore{base + L.o_tscr, L.ld});
  hipLaunchKernelGGL(mvt_ekl_pack_kernel, dim3((unsigned)(((int64_t)D * D + 255) / 256)), dim3(256), 0, st,
                     (const double*)(base + L.o_tscr), (const double*)(base + L.o_lfull), (const double*)(base + L.o_theta), L.ld, D,
                     (const double*)S.sums, S.off_col, 1.0 / (double)n_total, ctx->model.c0, base + L.o_grad,
                     path_deriv ? (const double*)P.sums : (const double*)nullptr,
                     lgamma(0.5 * (df + (double)d)) - lgamma(0.5 * df) - 0.5 * (double)d * log(M_PI * df), 0.5 * (df + (double)d));
  VB_HIP(ctx, hipGetLastError());
  const size_t plen = (size_t)(1 + d + d * (d + 1) / 2);
  const FetchSeg seg{base + L.o_grad, plen * sizeof(double), value_grad_host};
  VB_TRY(fetch_blocking(ctx, st, &seg, 1));
  if (info) info[0] = rinfo[0], info[1] = rinfo[2], info[2] = xinfo[0], info[3] = xinfo[2];
  return VB_OK;
}

// ---- AlphaDivergence of the multivariate t in the reference-identical mode, resident on the device (round 5, late) -----
// objectives.py:443-463 over approximations.py:342-349: the weighted sums of alpha_mvt_enqueue (weights, value, sum s g,
// C = sum s g (z / s)') with the samples through the device's symmetric root, then the chain rule of mvt_elbo_symroot --
// Gs = sym(C), the root's Frechet derivative, X L -- with sum s on the free diagonal (of log q(x(theta); theta) only
// -sum log L_ii moves) and alpha / N in front.  value_grad_host: 1 + d + d (d + 1) / 2 doubles.
namespace {

__global__ void __launch_bounds__(256) mvt_alpha_pack_kernel(const double* __restrict__ XL, const double* __restrict__ Lfull,
                                                             int64_t ld, int d, const double* __restrict__ sums,
                                                             int64_t off_col, const double* __restrict__ vw, double scale,
                                                             double* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx == 0) out[0] = vw[1];
  if (idx < d) out[1 + idx] = scale * sums[off_col + idx];
  if (idx >= (int64_t)d * d) return;
  const int i = (int)(idx / d), j = (int)(idx % d);
  if (j > i) return;
  double g = 2.0 * XL[(int64_t)i * ld + j];
  if (i == j) g = g * Lfull[(int64_t)i * ld + i] + vw[0];
  out[1 + d + (int64_t)i * (i + 1) / 2 + j] = scale * g;
}

}  // namespace

int mvt_alpha_symroot(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t n_total, int64_t d, double df, double alpha,
                      const double* theta_host, double* value_grad_host, double* info) {
  int64_t mine = 0;      // (sharded jobs as mvt_elbo_symroot: alpha_mvt_enqueue all-reduces the maximum and the weighted sums)
  VB_TRY(comm_shard_begin(ctx, n, n_total, &mine));
  if (n <= 0 || n > ns.n || d != ns.d) return fail(ctx, VB_ERR_INVALID, "noise slot shape mismatch");
  const double* chi_rows = mvt_chi_rows(ctx, n, n_total, mine, df);
  if (!chi_rows)
    return fail(ctx, VB_ERR_STATE, "needs %lld (or all %lld) device chi-square(%g) draws (vb_legacy_rng_chisquare_device)",
                (long long)n, (long long)n_total, df);
  const MvtLayout L = mvt_layout(ctx, n, n_total, d);
  VB_TRY(ensure(ctx, ctx->mvt_ekl_state, (size_t)L.total * sizeof(double)));
  double* base = (double*)ctx->mvt_ekl_state.ptr;
  hipStream_t st = ctx->stream;
  const int n_cu = ctx->prop.multiProcessorCount;
  const int D = (int)d;
  const int64_t sq = d * L.ld;
  VB_TRY(mvt_factors_device(ctx, L, base, d, theta_host));
  double rinfo[3] = {0.0, 0.0, 0.0};
  VB_TRY(sym_sqrt_dev(ctx, base + L.o_lfull, base + L.o_lt, d, L.ld, base + L.o_root, 1e-12, rinfo));
  hipLaunchKernelGGL(mvt_inv_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, chi_rows, df, n, base + L.o_invs);
  double sum_log_diag = 0.0;
  for (int64_t j = 0; j < d; ++j) sum_log_diag += theta_host[d + j * (j + 1) / 2 + j];
  FrSums S;
  const double* vw = nullptr;
  VB_TRY(alpha_mvt_enqueue(ctx, ns, n, n_total, d, df, alpha, base + L.o_mu, base + L.o_root, base + L.o_invs, sum_log_diag, &S, &vw));
  hipLaunchKernelGGL(mvt_gs_kernel, dim3((unsigned)((sq + 255) / 256)), dim3(256), 0, st, (const double*)(S.sums + S.off_c),
                     base + L.o_tscr, D, L.ld, 0.5);
  VB_HIP(ctx, hipGetLastError());
  double xinfo[3] = {0.0, 0.0, 0.0};
  VB_TRY(sym_sqrt_frechet_dev(ctx, base + L.o_lfull, base + L.o_lt, base + L.o_tscr, d, L.ld, base + L.o_sl, 1e-12, xinfo));
  GemmArgs g;      // X L
  g.A = base + L.o_sl, g.lda = L.ld, g.B = base + L.o_lfull, g.ldb = L.ld;
  g.M = D, g.N = D, g.K = D, g.tri_mode = 0;
  gemm_f64_launch<true>(st, g, 1, n_cu, EpiStore{base + L.o_tscr, L.ld});
  hipLaunchKernelGGL(mvt_alpha_pack_kernel, dim3((unsigned)(((int64_t)D * D + 255) / 256)), dim3(256), 0, st,
                     (const double*)(base + L.o_tscr), (const double*)(base + L.o_lfull), L.ld, D, (const double*)S.sums,
                     S.off_col, vw, alpha / (double)n_total, base + L.o_grad);
  VB_HIP(ctx, hipGetLastError());
  const size_t plen = (size_t)(1 + d + d * (d + 1) / 2);
  const FetchSeg seg{base + L.o_grad, plen * sizeof(double), value_grad_host};
  VB_TRY(fetch_blocking(ctx, st, &seg, 1));
  if (info) info[0] = rinfo[0], info[1] = rinfo[2], info[2] = xinfo[0], info[3] = xinfo[2];
  return VB_OK;
}

// AlphaDivergence sums for the multivariate t (see vb_alpha_sums_mvt in the header)
int mvt_alpha_sums(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int64_t n_total, double df, double alpha,
                   const double* mu_host, const double* root_host, const double* inv_s_host, double sum_log_diag,
                   double* value, double* w_sum, double* g_sum, double* c_full) {
  if (n <= 0 || n > ns.n || d != ns.d) return fail(ctx, VB_ERR_INVALID, "noise slot shape mismatch");
  const int64_t ld = round_up(d, 16);
  int64_t off = 0;
  auto carve = [&off](int64_t doubles) {
    const int64_t o = off;
    off += round_up(doubles, 16);
    return o;
  };
  const int64_t o_root = carve(d * ld), o_mu = carve(ld), o_invs = carve(n);
  VB_TRY(ensure(ctx, ctx->mvt_elbo, (size_t)off * sizeof(double)));
  double* base = (double*)ctx->mvt_elbo.ptr;
  hipStream_t st = ctx->stream;
  VB_TRY(upload_padded(ctx, base + o_root, ld, root_host, d, d, false));
  VB_HIP(ctx, hipMemsetAsync(base + o_mu, 0, (size_t)ld * sizeof(double), st));
  VB_HIP(ctx, hipMemcpyAsync(base + o_mu, mu_host, (size_t)d * sizeof(double), hipMemcpyHostToDevice, st));
  if (inv_s_host) {
    VB_HIP(ctx, hipMemcpyAsync(base + o_invs, inv_s_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
  } else {      // throughput mode: the chi-square draws of vb_chisq_generate, already on the device
    if (!(df > 2.0)) return fail(ctx, VB_ERR_INVALID, "df must be greater than 2");
    if (ctx->chi_n != n || ctx->chi_df != df || !ctx->chi_dev.ptr)
      return fail(ctx, VB_ERR_STATE, "inv_s == NULL needs %lld device chi-square(%g) draws (vb_chisq_generate)",
                  (long long)n, df);
    hipLaunchKernelGGL(mvt_inv_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       (const double*)ctx->chi_dev.ptr, df, n, base + o_invs);
    VB_HIP(ctx, hipGetLastError());
  }
  FrSums S;
  const double* vw = nullptr;
  VB_TRY(alpha_mvt_enqueue(ctx, ns, n, n_total, d, df, alpha, base + o_mu, base + o_root, base + o_invs, sum_log_diag,
                           &S, &vw));
  double two[2];
  VB_HIP(ctx, hipMemcpyAsync(two, vw, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
  VB_HIP(ctx, hipMemcpyAsync(g_sum, S.sums + S.off_col, (size_t)d * sizeof(double), hipMemcpyDeviceToHost, st));
  VB_HIP(ctx, hipMemcpy2DAsync(c_full, (size_t)d * sizeof(double), S.sums + S.off_c, (size_t)ld * sizeof(double),
                               (size_t)d * sizeof(double), (size_t)d, hipMemcpyDeviceToHost, st));
  legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
  VB_HIP(ctx, hipStreamSynchronize(st));
  legacy_poll(ctx);
  *w_sum = two[0];
  *value = two[1];
  return VB_OK;
}

// ExclusiveKL, throughput mode: the dense-Gaussian pipeline (triangular sampling GEMM, model, lower-triangular gradient
// GEMM, packed reduction) with the rows scaled by 1 / s_n -- x = mu + (L z) / s -- so the gradient comes out in the free-
// Cholesky layout with no root and no Sylvester equation (vb_elbo_grad_mvt_chol)
int mvt_elbo_chol_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int64_t n_total, double df,
                          const double* theta_dev, double* out_dev) {
  if (!(df > 2.0)) return fail(ctx, VB_ERR_INVALID, "df must be greater than 2");
  if (ctx->chi_n != n || ctx->chi_df != df || !ctx->chi_dev.ptr)
    return fail(ctx, VB_ERR_STATE, "needs %lld device chi-square(%g) draws (vb_chisq_generate)", (long long)n, df);
  VB_TRY(ensure(ctx, ctx->mvt_invs, (size_t)n * sizeof(double)));
  hipLaunchKernelGGL(mvt_inv_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const double*)ctx->chi_dev.ptr, df, n, (double*)ctx->mvt_invs.ptr);
  VB_HIP(ctx, hipGetLastError());
  return fr_pipeline_enqueue(ctx, ns, n, d, n_total, theta_dev, out_dev, nullptr, nullptr, (const double*)ctx->mvt_invs.ptr,
                             nullptr, 0, nullptr);
}

// AlphaDivergence, throughput mode: the dense family's weighted pipeline with the rows scaled by 1 / s_n and the t
// family's base density (alpha_fullrank_enqueue) -- weights, value and the gradient in the flat layout on the device
int mvt_alpha_chol_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int64_t n_total, double df, double alpha,
                           const double* theta_dev, double sum_log_diag, double* out_dev) {
  VB_TRY(ensure(ctx, ctx->mvt_invs, (size_t)n * sizeof(double)));
  hipLaunchKernelGGL(mvt_inv_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const double*)ctx->chi_dev.ptr, df, n, (double*)ctx->mvt_invs.ptr);
  VB_HIP(ctx, hipGetLastError());
  return alpha_fullrank_enqueue(ctx, ns, n, n_total, d, alpha, theta_dev, sum_log_diag, out_dev, df,
                                (const double*)ctx->mvt_invs.ptr);
}

}  // namespace vb
