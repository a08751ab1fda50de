// Per-row (per-sample) quantities: f(x_n) for explicit x.
//
// Model.__call__ (viabel/models.py:27-39) for the device-resident targets; used by the
// diagnostics-side callers (convenience.py:176-179) and by the DIS / alpha objectives for the
// per-sample log weights (objectives.py:394-395, :445).  One wave per row, lanes stride the
// columns (coalesced), DPP/shuffle reduction over the wave.
#include "vb_common.h"
#include "vb_gemm_f64.h"

namespace vb {

__device__ __forceinline__ double wave_sum_rows(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

// PRIOR: the same pass also evaluates a diagonal Gaussian (mean q0, inverse variances q1, constant qc) into out2 --
// the tempering prior of DISInclusiveKL (objectives.py:316-318) next to the model, one read of the samples instead of two
// Round 5: a wave owns kRowsPerWave consecutive rows (one until then): the model's and the prior's parameters are loaded
// once per column step and used for all of them, four rows' loads are in flight together, and a quarter of the
// workgroups are dispatched (13.6-14.3 -> 10.9-12.4 us at 16 384 x 256).  A lane still adds its columns c = lane, lane + 64, ... of a
// row in that order: the same bits as before.
constexpr int kRowsPerWave = 4;
template <bool PRIOR>
__global__ void __launch_bounds__(256) model_logp_rows_kernel(const double* __restrict__ x,
                                                              int64_t ld, int64_t n, int d,
                                                              ModelDev m, double* __restrict__ out,
                                                              const double* __restrict__ q0,
                                                              const double* __restrict__ q1, double qc,
                                                              double* __restrict__ out2) {
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * kRowsPerWave;
  if (row0 >= n) return;
  const double* xr[kRowsPerWave];
#pragma unroll
  for (int r = 0; r < kRowsPerWave; ++r) xr[r] = x + (row0 + r < n ? row0 + r : n - 1) * ld;      // (clamped: loads only)
  double acc[kRowsPerWave], acc2[kRowsPerWave];
#pragma unroll
  for (int r = 0; r < kRowsPerWave; ++r) acc[r] = 0.0, acc2[r] = 0.0;
  if (m.id == VB_MODEL_GAUSS_DIAG) {
    for (int c = lane; c < d; c += 64) {
      const double mc = m.p0[c], iv = m.p1[c];
      const double qm = PRIOR ? q0[c] : 0.0, qi = PRIOR ? q1[c] : 0.0;
      double z[kRowsPerWave];
#pragma unroll
      for (int r = 0; r < kRowsPerWave; ++r) z[r] = xr[r][c];
#pragma unroll
      for (int r = 0; r < kRowsPerWave; ++r) {
        const double dz = z[r] - mc;
        acc[r] -= 0.5 * dz * dz * iv;
        if (PRIOR) {
          const double dq = z[r] - qm;
          acc2[r] -= 0.5 * dq * dq * qi;
        }
      }
    }
  } else {   // funnel
    double w[kRowsPerWave];
#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) w[r] = exp(-2.0 * xr[r][m.k]);
    for (int c = lane; c < d; c += 64) {
      const double qm = PRIOR ? q0[c] : 0.0, qi = PRIOR ? q1[c] : 0.0;
      double z[kRowsPerWave];
#pragma unroll
      for (int r = 0; r < kRowsPerWave; ++r) z[r] = xr[r][c];
#pragma unroll
      for (int r = 0; r < kRowsPerWave; ++r) {
        if (c == m.k)
          acc[r] += -0.5 * z[r] * z[r] / (m.tau * m.tau) - (double)(d - 1) * z[r];
        else
          acc[r] -= 0.5 * z[r] * z[r] * w[r];
        if (PRIOR) {
          const double dq = z[r] - qm;
          acc2[r] -= 0.5 * dq * dq * qi;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < kRowsPerWave; ++r) {
    const double a = wave_sum_rows(acc[r]);
    const double a2 = PRIOR ? wave_sum_rows(acc2[r]) : 0.0;
    if (lane == 0 && row0 + r < n) {
      out[row0 + r] = a + m.c0;
      if (PRIOR) out2[row0 + r] = a2 + qc;
    }
  }
}

// The same pass with the multivariate t's own row statistics riding along (round 6): the device-resident DIS refresh in
// throughput mode reads TWO N x D matrices row by row -- the samples X for log p / log prior (above) and the noise E, whose
// rows scaled by 1 / s_n ARE the residuals L^-1 (x_n - mu), for maha_n, log q_n and c_n (mvt_rows_kernel, vb_mvt.hip) -- in
// two one-wave-per-row launches of ~13 and ~10 us at 16 384 x 256 (dispatch-bound: 2.5-3 TB/s).  One launch, four rows per
// wave, both matrices' loads in flight together.  Every sum is formed as the two kernels form it (a lane adds its columns
// c = lane, lane + 64, ... of a row in that order; v = e * r first, then fma(v, v, s)): the same bits.
__global__ void __launch_bounds__(256) model_prior_maha_rows_kernel(const double* __restrict__ x, int64_t ld, int64_t n, int d,
                                                                    ModelDev m, double* __restrict__ out,
                                                                    const double* __restrict__ q0, const double* __restrict__ q1,
                                                                    double qc, double* __restrict__ out2,
                                                                    const double* __restrict__ E, int64_t lde,
                                                                    const double* __restrict__ rs, double df, double lq_const,
                                                                    double* __restrict__ maha, double* __restrict__ lq,
                                                                    double* __restrict__ cn) {
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * kRowsPerWave;
  if (row0 >= n) return;
  const double *xr[kRowsPerWave], *er[kRowsPerWave];
  double rr[kRowsPerWave], acc[kRowsPerWave], acc2[kRowsPerWave], ss[kRowsPerWave], w[kRowsPerWave];
#pragma unroll
  for (int r = 0; r < kRowsPerWave; ++r) {
    const int64_t row = row0 + r < n ? row0 + r : n - 1;      // (clamped: loads only)
    xr[r] = x + row * ld;
    er[r] = E + row * lde;
    rr[r] = rs ? rs[row] : 1.0;
    acc[r] = 0.0, acc2[r] = 0.0, ss[r] = 0.0;
    w[r] = m.id == VB_MODEL_FUNNEL ? exp(-2.0 * xr[r][m.k]) : 0.0;
  }
  for (int c = lane; c < d; c += 64) {
    const double mc = m.id == VB_MODEL_GAUSS_DIAG ? m.p0[c] : 0.0, iv = m.id == VB_MODEL_GAUSS_DIAG ? m.p1[c] : 0.0;
    const double qm = q0[c], qi = q1[c];
    double z[kRowsPerWave], e[kRowsPerWave];
#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) z[r] = xr[r][c], e[r] = er[r][c];
#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) {
      if (m.id == VB_MODEL_GAUSS_DIAG) {
        const double dz = z[r] - mc;
        acc[r] -= 0.5 * dz * dz * iv;
      } else if (c == m.k) {
        acc[r] += -0.5 * z[r] * z[r] / (m.tau * m.tau) - (double)(d - 1) * z[r];
      } else {
        acc[r] -= 0.5 * z[r] * z[r] * w[r];
      }
      const double dq = z[r] - qm;
      acc2[r] -= 0.5 * dq * dq * qi;
      double v = e[r];
      if (rs) v *= rr[r];
      ss[r] = fma(v, v, ss[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < kRowsPerWave; ++r) {
    const double a = wave_sum_rows(acc[r]), a2 = wave_sum_rows(acc2[r]), t = wave_sum_rows(ss[r]);
    if (lane == 0 && row0 + r < n) {
      const int64_t row = row0 + r;
      out[row] = a + m.c0;
      out2[row] = a2 + qc;
      maha[row] = t;
      lq[row] = df > 0.0 ? lq_const - 0.5 * (df + d) * log1p(t / df) : lq_const - 0.5 * t;
      cn[row] = df > 0.0 ? (df + d) / (df + t) : 1.0;
    }
  }
}

int model_prior_maha_rows(vb_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, int64_t d, double* out_dev,
                          const double* prior_mean, const double* prior_ivar, double prior_c0, double* prior_out,
                          const double* E, int64_t lde, const double* rs, double df, double lq_const, double* maha, double* lq,
                          double* cn) {
  if (ctx->model.id != VB_MODEL_GAUSS_DIAG && ctx->model.id != VB_MODEL_FUNNEL) return VB_ERR_UNSUPPORTED;
  const unsigned grid = (unsigned)((n + 4 * kRowsPerWave - 1) / (4 * kRowsPerWave));
  hipLaunchKernelGGL(model_prior_maha_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream, x_dev, ld, n, (int)d, ctx->model, out_dev,
                     prior_mean, prior_ivar, prior_c0, prior_out, E, lde, rs, df, lq_const, maha, lq, cn);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// ---- dense targets: f needs a GEMM -----------------------------------------------------------------
struct EpiStoreRows {        // Y = acc
  double* Y;
  int64_t ldy;
  __device__ void operator()(int, int row, int col, double acc) const { Y[(int64_t)row * ldy + col] = acc; }
};

struct EpiGlmGradDirect {   // G = acc - z / sd^2
  double* G;
  int64_t ldz;
  const double* Z;
  double ivp;
  __device__ void operator()(int, int row, int col, double acc) const {
    const int64_t i = (int64_t)row * ldz + col;
    G[i] = fma(-ivp, Z[i], acc);
  }
  __device__ d2v pair(int, int row, int col, double a0, double a1) const {
    const int64_t i = (int64_t)row * ldz + col;
    const d2v z = *reinterpret_cast<const d2v*>(Z + i);
    const d2v v = (d2v){fma(-ivp, z.x, a0), fma(-ivp, z.y, a1)};
    *reinterpret_cast<d2v*>(G + i) = v;
    return v;
  }
};

struct EpiGlmGradSlab {     // slab_split = acc
  double* W;
  int64_t ldz, slab;
  __device__ void operator()(int split, int row, int col, double acc) const {
    W[split * slab + (int64_t)row * ldz + col] = acc;
  }
};

// G = sum of the split slabs (fixed order) - z / sd^2
__global__ void __launch_bounds__(256) glm_grad_reduce_kernel(const double* __restrict__ W, int splits, int64_t slab,
                                                              const double* __restrict__ Z, double ivp,
                                                              double* __restrict__ G, int64_t ldz, int64_t n, int d) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * ldz) return;
  if ((int)(idx % ldz) >= d) return;
  double s = 0.0;
  for (int k = 0; k < splits; ++k) s += W[k * slab + idx];
  G[idx] = fma(-ivp, Z[idx], s);
}

struct EpiLogLikTerm {       // T = log-likelihood term of observation `col` at eta = acc
  double* T;
  int64_t ldt;
  const double* y;
  int link;
  double aux;
  __device__ void operator()(int, int row, int col, double eta) const {
    double dl;
    T[(int64_t)row * ldt + col] = glm_term(link, aux, y[col], eta, &dl);
  }
};

__global__ void __launch_bounds__(256) rows_center_kernel(const double* __restrict__ x, int64_t ld, int64_t n,
                                                          int d, const double* __restrict__ mean,
                                                          double* __restrict__ xc) {
  const int64_t row = blockIdx.x;            // rows on x: gridDim.y stops at 65 535
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c < ld) xc[row * ld + c] = c < d ? x[row * ld + c] - mean[c] : 0.0;
}

// out[row] = scale * sum_c a[row][c] * (b ? b[row][c] : 1) + add_sq * sum_c sq[row][c]^2 + c0
__global__ void __launch_bounds__(256) rows_dot_kernel(const double* __restrict__ a, int64_t lda,
                                                       const double* __restrict__ b, int64_t ldb, int width,
                                                       double scale, const double* __restrict__ sq, int64_t ldsq,
                                                       int sq_width, double add_sq, double c0, int64_t n,
                                                       double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  double acc = 0.0, s2 = 0.0;
  for (int c = lane; c < width; c += 64) acc += a[row * lda + c] * (b ? b[row * ldb + c] : 1.0);
  if (sq)
    for (int c = lane; c < sq_width; c += 64) s2 = fma(sq[row * ldsq + c], sq[row * ldsq + c], s2);
  acc = wave_sum_rows(acc);
  s2 = wave_sum_rows(s2);
  if (lane == 0) out[row] = fma(scale, acc, fma(add_sq, s2, c0));
}

// gauss_full: f = -1/2 (x - m)' P (x - m) + c0
static int gauss_full_rows(vb_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, int64_t d, double* out_dev) {
  const ModelDev& m = ctx->model;
  hipStream_t st = ctx->stream;
  VB_TRY(ensure(ctx, ctx->rows_work, (size_t)2 * n * ld * sizeof(double)));
  double* Xc = (double*)ctx->rows_work.ptr;
  double* Y = Xc + n * ld;
  hipLaunchKernelGGL(rows_center_kernel, dim3((unsigned)n, (unsigned)((ld + 255) / 256)), dim3(256), 0, st, x_dev, ld,
                     n, (int)d, m.p0, Xc);
  VB_HIP(ctx, hipGetLastError());
  GemmArgs g;
  g.A = Xc, g.lda = ld, g.B = m.p1, g.ldb = m.ldp;
  g.M = (int)n, g.N = (int)d, g.K = (int)d, g.tri_mode = 0;
  gemm_f64_launch<true>(st, g, 1, ctx->prop.multiProcessorCount, EpiStoreRows{Y, ld});
  VB_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(rows_dot_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, (const double*)Xc, ld,
                     (const double*)Y, ld, (int)d, -0.5, (const double*)nullptr, (int64_t)0, 0, 0.0, m.c0, n, out_dev);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// logistic: f(b) = sum_i [y_i eta_i - log(1 + exp(eta_i))] - |b|^2 / (2 sd^2) + c0, eta = X b; row chunks bound
// the n x n_data term matrix to 512 MB
static int logistic_rows(vb_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, int64_t d, double* out_dev) {
  const ModelDev& m = ctx->model;
  hipStream_t st = ctx->stream;
  const int64_t ldt = m.ldq;
  int64_t chunk = ((int64_t)64 << 20) / ldt;
  chunk = chunk < 128 ? 128 : (chunk > n ? n : chunk);
  VB_TRY(ensure(ctx, ctx->rows_work, (size_t)chunk * ldt * sizeof(double)));
  double* T = (double*)ctx->rows_work.ptr;
  for (int64_t r0 = 0; r0 < n; r0 += chunk) {
    const int64_t rows = n - r0 < chunk ? n - r0 : chunk;
    GemmArgs g;
    g.A = x_dev + r0 * ld, g.lda = ld, g.B = m.p1, g.ldb = m.ldq;
    g.M = (int)rows, g.N = (int)m.n_data, g.K = (int)d, g.tri_mode = 0;
    gemm_f64_launch<true>(st, g, 1, ctx->prop.multiProcessorCount, EpiLogLikTerm{T, ldt, m.p2, m.link, m.aux});
    VB_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(rows_dot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const double*)T, ldt,
                       (const double*)nullptr, (int64_t)0, (int)m.n_data, 1.0, x_dev + r0 * ld, ld, (int)d,
                       -0.5 / (m.tau * m.tau), m.c0, rows, out_dev + r0);
    VB_HIP(ctx, hipGetLastError());
  }
  return VB_OK;
}

int glm_grad_enqueue(vb_ctx* ctx, hipStream_t st, const ModelDev& m, const double* R, int64_t ldr, const double* Z,
                     double* G, int64_t ldz, int64_t n, int d) {
  const int n_cu = ctx->prop.multiProcessorCount;
  GemmArgs gg;                       // [n x d x n_data]
  gg.A = R;
  gg.lda = ldr;
  gg.B = m.p0;
  gg.ldb = m.ldp;
  gg.M = (int)n;
  gg.N = d;
  gg.K = (int)m.n_data;
  gg.tri_mode = 0;
  const double ivp = 1.0 / (m.tau * m.tau);
  // few output tiles and a long contraction: split the observations so that every CU gets a workgroup or two
  const int64_t tiles = gemm_max_blocks(n, d);
  int splits = 1;
  if (tiles < n_cu) {
    splits = (int)(2 * n_cu / tiles);
    const int max_splits = (int)(m.n_data / 128);
    if (splits > max_splits) splits = max_splits;
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
  }
  if (splits == 1) {
    gemm_f64_launch<true>(st, gg, 1, n_cu, EpiGlmGradDirect{G, ldz, Z, ivp});
    VB_HIP(ctx, hipGetLastError());
    return VB_OK;
  }
  const int64_t slab = n * ldz;
  VB_TRY(ensure(ctx, ctx->glm_work, (size_t)(splits * slab) * sizeof(double)));
  double* W = (double*)ctx->glm_work.ptr;
  gemm_f64_launch<true>(st, gg, splits, n_cu, EpiGlmGradSlab{W, ldz, slab});
  VB_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(glm_grad_reduce_kernel, dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, st, (const double*)W,
                     splits, slab, Z, ivp, G, ldz, n, d);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// ---- per-sample gradients (vb_model_grad): f and grad f of given points ----------------------------------------
// The reference gets them from autograd (models.py:17-39; tests/test_models.py:13-15 checks the vjp); here every
// target has its own device gradient, and this is the entry that shows it to the caller -- a hand-written
// vb_log_density can be checked against differences of its own f (SourceModel.check_gradient).
__global__ void __launch_bounds__(256) model_grad_rows_kernel(const double* __restrict__ X, int64_t ld, int64_t n, int d,
                                                              ModelDev m, double* __restrict__ G) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const double* x = X + row * ld;
  double* g = G + row * ld;
  if (m.id == VB_MODEL_GAUSS_DIAG) {
    for (int c = lane; c < d; c += 64) g[c] = -(x[c] - m.p0[c]) * m.p1[c];
    return;
  }
  // funnel: d/dx_c = -x_c e^{-2v} (c != k);  d/dv = -v / tau^2 - (d - 1) + e^{-2v} sum_{c != k} x_c^2
  const double v = x[m.k], w = exp(-2.0 * v);
  double ss = 0.0;
  for (int c = lane; c < d; c += 64)
    if (c != m.k) {
      g[c] = -x[c] * w;
      ss = fma(x[c], x[c], ss);
    }
  ss = wave_sum_rows(ss);
  if (lane == 0) g[m.k] = fma(-v, 1.0 / (m.tau * m.tau), -(double)(d - 1)) + w * ss;
}

struct EpiStoreNegRows {     // Y = -acc
  double* Y;
  int64_t ldy;
  __device__ void operator()(int, int row, int col, double acc) const { Y[(int64_t)row * ldy + col] = -acc; }
};

struct EpiTermResidual {     // T = log-likelihood term of observation `col` at eta = acc, R = its derivative in eta
  double* T;
  double* R;
  int64_t ldt;
  const double* y;
  int link;
  double aux;
  __device__ void operator()(int, int row, int col, double eta) const {
    double dl;
    T[(int64_t)row * ldt + col] = glm_term(link, aux, y[col], eta, &dl);
    R[(int64_t)row * ldt + col] = dl;
  }
};

int model_grad_rows(vb_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, int64_t d, double* g_dev, double* f_dev) {
  const ModelDev& m = ctx->model;
  hipStream_t st = ctx->stream;
  const int n_cu = ctx->prop.multiProcessorCount;
  VB_HIP(ctx, hipMemsetAsync(g_dev, 0, (size_t)n * ld * sizeof(double), st));
  if (m.id == VB_MODEL_SOURCE) return user_rows_enqueue(ctx, st, x_dev, ld, n, (int)d, g_dev, ld, f_dev);
  if (m.id == VB_MODEL_GAUSS_DIAG || m.id == VB_MODEL_FUNNEL) {
    VB_TRY(model_logp_rows(ctx, x_dev, ld, n, d, f_dev));
    hipLaunchKernelGGL(model_grad_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, x_dev, ld, n, (int)d, m,
                       g_dev);
    VB_HIP(ctx, hipGetLastError());
    return VB_OK;
  }
  if (m.id == VB_MODEL_GAUSS_FULL) {      // G = -(x - m) P (P symmetric), f = 1/2 (x - m) . G + c0
    VB_TRY(ensure(ctx, ctx->rows_work, (size_t)n * ld * sizeof(double)));
    double* Xc = (double*)ctx->rows_work.ptr;
    hipLaunchKernelGGL(rows_center_kernel, dim3((unsigned)n, (unsigned)((ld + 255) / 256)), dim3(256), 0, st, x_dev, ld,
                       n, (int)d, m.p0, Xc);
    VB_HIP(ctx, hipGetLastError());
    GemmArgs g;
    g.A = Xc, g.lda = ld, g.B = m.p1, g.ldb = m.ldp;
    g.M = (int)n, g.N = (int)d, g.K = (int)d, g.tri_mode = 0;
    gemm_f64_launch<true>(st, g, 1, n_cu, EpiStoreNegRows{g_dev, ld});
    VB_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(rows_dot_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, (const double*)Xc, ld,
                       (const double*)g_dev, ld, (int)d, 0.5, (const double*)nullptr, (int64_t)0, 0, 0.0, m.c0, n, f_dev);
    VB_HIP(ctx, hipGetLastError());
    return VB_OK;
  }
  if (m.id == VB_MODEL_LOGISTIC) {        // eta = x X' -> terms and residuals -> f, G = R X - x / sd^2; row chunks
    const int64_t ldt = m.ldq;
    int64_t chunk = ((int64_t)32 << 20) / ldt;
    chunk = chunk < 128 ? 128 : (chunk > n ? n : chunk);
    VB_TRY(ensure(ctx, ctx->rows_work, (size_t)2 * chunk * ldt * sizeof(double)));
    double* T = (double*)ctx->rows_work.ptr;
    double* R = T + chunk * ldt;
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
      const int64_t rows = n - r0 < chunk ? n - r0 : chunk;
      GemmArgs g;
      g.A = x_dev + r0 * ld, g.lda = ld, g.B = m.p1, g.ldb = m.ldq;
      g.M = (int)rows, g.N = (int)m.n_data, g.K = (int)d, g.tri_mode = 0;
      gemm_f64_launch<true>(st, g, 1, n_cu, EpiTermResidual{T, R, ldt, m.p2, m.link, m.aux});
      VB_HIP(ctx, hipGetLastError());
      hipLaunchKernelGGL(rows_dot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const double*)T, ldt,
                         (const double*)nullptr, (int64_t)0, (int)m.n_data, 1.0, x_dev + r0 * ld, ld, (int)d,
                         -0.5 / (m.tau * m.tau), m.c0, rows, f_dev + r0);
      VB_HIP(ctx, hipGetLastError());
      VB_TRY(glm_grad_enqueue(ctx, st, m, R, ldt, x_dev + r0 * ld, g_dev + r0 * ld, ld, rows, (int)d));
    }
    return VB_OK;
  }
  return fail(ctx, VB_ERR_UNSUPPORTED, "row gradient: unknown model id %d", m.id);
}

int model_logp_rows(vb_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, int64_t d,
                    double* out_dev) {
  if (ctx->model.id == VB_MODEL_GAUSS_FULL) return gauss_full_rows(ctx, x_dev, ld, n, d, out_dev);
  if (ctx->model.id == VB_MODEL_LOGISTIC) return logistic_rows(ctx, x_dev, ld, n, d, out_dev);
  if (ctx->model.id == VB_MODEL_SOURCE) return user_rows_enqueue(ctx, ctx->stream, x_dev, ld, n, (int)d, nullptr, 0, out_dev);
  if (ctx->model.id != VB_MODEL_GAUSS_DIAG && ctx->model.id != VB_MODEL_FUNNEL)
    return fail(ctx, VB_ERR_UNSUPPORTED, "row log-density: unknown model id %d", ctx->model.id);
  const unsigned grid = (unsigned)((n + 4 * kRowsPerWave - 1) / (4 * kRowsPerWave));
  hipLaunchKernelGGL(model_logp_rows_kernel<false>, dim3(grid), dim3(256), 0, ctx->stream, x_dev, ld, n,
                     (int)d, ctx->model, out_dev, (const double*)nullptr, (const double*)nullptr, 0.0, (double*)nullptr);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// log density of the model AND of a diagonal Gaussian (mean, inverse variances, additive constant) at the same rows:
// one pass when the model is evaluated by the row kernel, otherwise the model's own path followed by the row kernel
int model_and_prior_logp_rows(vb_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, int64_t d, double* out_dev,
                              const double* prior_mean, const double* prior_ivar, double prior_c0, double* prior_out) {
  if (ctx->model.id == VB_MODEL_GAUSS_DIAG || ctx->model.id == VB_MODEL_FUNNEL) {
    const unsigned grid = (unsigned)((n + 4 * kRowsPerWave - 1) / (4 * kRowsPerWave));
    hipLaunchKernelGGL(model_logp_rows_kernel<true>, dim3(grid), dim3(256), 0, ctx->stream, x_dev, ld, n, (int)d,
                       ctx->model, out_dev, prior_mean, prior_ivar, prior_c0, prior_out);
    VB_HIP(ctx, hipGetLastError());
    return VB_OK;
  }
  VB_TRY(model_logp_rows(ctx, x_dev, ld, n, d, out_dev));
  const ModelDev saved = ctx->model;
  ModelDev prior;
  prior.id = VB_MODEL_GAUSS_DIAG;
  prior.dim = (int)d;
  prior.c0 = prior_c0;
  prior.p0 = prior_mean;
  prior.p1 = prior_ivar;
  ctx->model = prior;
  const int rc = model_logp_rows(ctx, x_dev, ld, n, d, prior_out);
  ctx->model = saved;
  return rc;
}

}  // namespace vb
