// LRGaussian (viabel/approximations.py:610-731) under DISInclusiveKL and AlphaDivergence
// (viabel/objectives.py:283-416, :419-463): the reference's objectives are family-generic and differentiate
// `approx.log_density` / `approx.sample` with autograd.  Here the per-sample work runs on the device and the
// O(D k^2) algebra through the k x k capacitance matrix stays with the caller (as for the family's ExclusiveKL).
//
// Sigma = B B' + diag(sigma^2), Bs = B / sigma (rows scaled), M = I + Bs' Bs.  For a sample x and rho = (x - mu) / sigma:
//     v = Bs' rho,   tau = M^-1 v,   Q = (x - mu)' Sigma^-1 (x - mu) = |rho|^2 - v' tau          (Woodbury, :692-700)
//     a = Sigma^-1 (x - mu) = (rho - Bs tau) / sigma,            B' a = tau
// DIS (samples fixed):   d log q / d mu = a,  d/d log_sigma = -sigma^2 diag(Sigma^-1) + sigma^2 a^2,
//                        d/dB = -Sigma^-1 B + a tau'
//     -> weighted sums of rho, rho^2, rho tau', tau tau', tau, log q  (two skinny GEMMs against T = [tau | 1 | log q]).
// Alpha (x = mu + B z + sigma eps moves with theta; rho = Bs z + eps, t = z - tau = M^-1 (z - Bs' eps)):
//     d lw / d mu = g,  d/d log_sigma = g sigma eps + sigma^2 diag(Sigma^-1) - c^2 - c eps  (c = Bs t),
//     d/dB = g z' + ((c + eps) / sigma) t' + Sigma^-1 B
//     -> weighted sums of g, g eps, g z', eps t', t t'.
// Per row the device does O(D k) work plus the model; every D x k / k x k contraction over the samples is an fp64
// MFMA GEMM (vb_gemm_f64.h) split over the sample axis and reduced in fixed order.
#include "vb_common.h"
#include "vb_gemm_f64.h"

#include <cmath>
#include <vector>

namespace vb {
namespace {

// Row strides (doubles): ldk = round_up(k, 16) for the D x k matrices, the k x k inverse capacitance matrix and the n x k
// block V; ldt = ldk + 16 for T = [tau or t (ldk) | 1 | log q | 0 ...] and the padded z block Zp: the "ones" column sits at
// ldk, log q at ldk + 1.  k <= 16 gives the 16 / 32 / 16 / 17 of rounds 1-2; round 3 lifts the rank to 64.
constexpr int kMaxRank = 64;

__device__ __forceinline__ double lro_wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}
__device__ __forceinline__ double lro_wave_max(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x = fmax(x, __shfl_down(x, off, 64));
  return x;
}

// X = mu + Z B' + sigma E (approximations.py:636-644); Zp = [z | 0 ... | 1 at column 16] (row stride kLdt)
__global__ void __launch_bounds__(256) lro_sample_kernel(const double* __restrict__ E, int64_t lde,
                                                         const double* __restrict__ Z, int64_t ldz, int64_t n, int d,
                                                         int k, const double* __restrict__ mu,
                                                         const double* __restrict__ sigma,
                                                         const double* __restrict__ B, double* __restrict__ X,
                                                         int64_t ldx, double* __restrict__ Zp, int ldk, int ldt) {
  const int64_t row = blockIdx.x;            // rows on x: gridDim.y stops at 65 535
  const int c = blockIdx.y * 256 + threadIdx.x;
  const double* z = Z + row * ldz;
  if (c < ldx) {
    double x = 0.0;
    if (c < d) {
      x = fma(sigma[c], E[row * lde + c], mu[c]);
      for (int j = 0; j < k; ++j) x = fma(B[(int64_t)c * ldk + j], z[j], x);
    }
    X[row * ldx + c] = x;
  }
  if (blockIdx.y == 0 && (int)threadIdx.x < ldt)
    Zp[row * ldt + threadIdx.x] = (int)threadIdx.x < k ? z[threadIdx.x] : ((int)threadIdx.x == ldk ? 1.0 : 0.0);
}

// R = (X - mu) / sigma, rr[n] = |rho_n|^2; one wave per row
__global__ void __launch_bounds__(256) lro_resid_kernel(const double* __restrict__ X, int64_t ld, int64_t n, int d,
                                                        const double* __restrict__ mu,
                                                        const double* __restrict__ isig, double* __restrict__ R,
                                                        double* __restrict__ rr) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  double s = 0.0;
  for (int c = lane; c < (int)ld; c += 64) {
    double r = 0.0;
    if (c < d) r = (X[row * ld + c] - mu[c]) * isig[c];
    R[row * ld + c] = r;
    s = fma(r, r, s);
  }
  s = lro_wave_sum(s);
  if (lane == 0) rr[row] = s;
}

// per row: tau = Minv v, Q = |rho|^2 - v' tau, log q = cq - Q / 2; T = [tau (or z - tau) | 1 | log q]
__global__ void __launch_bounds__(256) lro_tau_kernel(const double* __restrict__ V, const double* __restrict__ rr,
                                                      const double* __restrict__ Minv, int k, double cq, int64_t n,
                                                      const double* __restrict__ Zp, int t_mode,
                                                      double* __restrict__ T, double* __restrict__ logq) {
  constexpr int kLdk = 16, kLdt = 32, kColOne = 16, kColLq = 17;      // (this kernel: k <= 16)
  __shared__ double mi[kLdk * kLdk];
  if (threadIdx.x < kLdk * kLdk) mi[threadIdx.x] = Minv[threadIdx.x];
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n) return;
  double v[kLdk], tau[kLdk];
#pragma unroll
  for (int j = 0; j < kLdk; ++j) v[j] = j < k ? V[row * kLdk + j] : 0.0;
  double vt = 0.0;
#pragma unroll
  for (int i = 0; i < kLdk; ++i) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < kLdk; ++j) s = fma(mi[i * kLdk + j], v[j], s);
    tau[i] = i < k ? s : 0.0;
    vt = fma(v[i], tau[i], vt);
  }
  const double lq = cq - 0.5 * (rr[row] - vt);
  logq[row] = lq;
  double* t = T + row * kLdt;
#pragma unroll
  for (int j = 0; j < kLdk; ++j) t[j] = t_mode ? (j < k ? Zp[row * kLdt + j] - tau[j] : 0.0) : tau[j];
#pragma unroll
  for (int j = kLdk; j < kLdt; ++j) t[j] = j == kColOne ? 1.0 : (j == kColLq ? lq : 0.0);
}

// the same for 16 < k <= 64: one WAVE per row, lane i holds v_i and forms tau_i = sum_j Minv[i][j] v_j with v_j broadcast
// lane by lane (Minv in LDS, 32 KB at k = 64); the row of T is written by the lanes in two passes
__global__ void __launch_bounds__(256) lro_tau_wave_kernel(const double* __restrict__ V, const double* __restrict__ rr,
                                                           const double* __restrict__ Minv, int k, int ldk, int ldt,
                                                           double cq, int64_t n, const double* __restrict__ Zp, int t_mode,
                                                           double* __restrict__ T, double* __restrict__ logq) {
  extern __shared__ double mi_dyn[];
  for (int e = threadIdx.x; e < k * ldk; e += 256) mi_dyn[e] = Minv[e];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const double v = lane < k ? V[row * ldk + lane] : 0.0;
  double tau = 0.0;
  for (int j = 0; j < k; ++j) {
    const double vj = __shfl(v, j, 64);
    if (lane < k) tau = fma(mi_dyn[lane * ldk + j], vj, tau);
  }
  const double vt = lro_wave_sum(v * tau);          // (lanes >= k contribute zero)
  const double lq = cq - 0.5 * (rr[row] - __shfl(vt, 0, 64));
  if (lane == 0) logq[row] = lq;
  double* t = T + row * ldt;
  for (int c = lane; c < ldt; c += 64) {      // column c < k <= 64 is written by the lane that holds tau_c
    double val = 0.0;
    if (c < k) val = t_mode ? Zp[row * ldt + c] - tau : tau;
    else if (c == ldk) val = 1.0;
    else if (c == ldk + 1) val = lq;
    t[c] = val;
  }
}

// Tw = w (.) T rows
__global__ void __launch_bounds__(256) lro_scale_rows_kernel(const double* __restrict__ T, const double* __restrict__ w,
                                                             int64_t n, double* __restrict__ Tw, int ldt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * ldt) return;
  Tw[i] = T[i] * w[i / ldt];
}

// per 128-row block and column: sum_n w_n A_nc B_nc  -> part[rb][c]
__global__ void __launch_bounds__(256) lro_colsum_prod_kernel(const double* __restrict__ A, int64_t lda,
                                                              const double* __restrict__ B, int64_t ldb, int64_t ld,
                                                              const double* __restrict__ w, int64_t n, int d,
                                                              double* __restrict__ part) {
  __shared__ double sh[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * 128, r1 = r0 + 128 < n ? r0 + 128 : n;
  double s = 0.0;
  if (c < d)
    for (int64_t r = r0 + q; r < r1; r += 4) s = fma(w[r] * A[r * lda + c], B[r * ldb + c], s);
  sh[q][threadIdx.x & 63] = s;
  __syncthreads();
  if (q == 0 && c < d) part[(int64_t)blockIdx.y * ld + c] = (sh[0][c & 63] + sh[1][c & 63]) + (sh[2][c & 63] + sh[3][c & 63]);
}

__global__ void __launch_bounds__(256) lro_slab_sum_kernel(const double* __restrict__ W, int slabs, int64_t slab,
                                                           double* __restrict__ out, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  double s = 0.0;
  for (int k = 0; k < slabs; ++k) s += W[k * slab + i];   // fixed order
  out[i] = s;
}

// model gradient rows G = grad f(X) for the targets without cross-sample structure (one wave per row)
__global__ void __launch_bounds__(256) lro_model_grad_kernel(const double* __restrict__ X, int64_t ld, int64_t n, int d,
                                                             ModelDev m, double* __restrict__ G) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const double* x = X + row * ld;
  double* g = G + row * ld;
  if (m.id == VB_MODEL_GAUSS_DIAG) {
    for (int c = lane; c < (int)ld; c += 64) g[c] = c < d ? -(x[c] - m.p0[c]) * m.p1[c] : 0.0;
    return;
  }
  const double v = x[m.k], w = exp(-2.0 * v);       // funnel (see fr_funnel_kernel)
  double ss = 0.0;
  for (int c = lane; c < (int)ld; c += 64) {
    double gc = 0.0;
    if (c < d && c != m.k) {
      gc = -x[c] * w;
      ss = fma(x[c], x[c], ss);
    }
    if (c != m.k) g[c] = gc;
  }
  ss = lro_wave_sum(ss);
  if (lane == 0) g[m.k] = fma(-v, 1.0 / (m.tau * m.tau), -(double)(d - 1)) + w * ss;
}

// AlphaDivergence weights (objectives.py:457-459): m = max lw, s = exp(alpha (lw - m)), S = sum s
__global__ void __launch_bounds__(1024) lro_lw_max_kernel(const double* __restrict__ f, const double* __restrict__ lq,
                                                          int64_t n, double* __restrict__ out) {
  __shared__ double sh[16];
  double mx = -INFINITY;
  for (int64_t i = threadIdx.x; i < n; i += 1024) mx = fmax(mx, f[i] - lq[i]);
  mx = lro_wave_max(mx);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) mx = fmax(mx, sh[w]);
    out[0] = fmax(mx, sh[0]);
  }
}
__global__ void __launch_bounds__(1024) lro_weights_kernel(const double* __restrict__ f, const double* __restrict__ lq,
                                                           const double* __restrict__ mx_in, int64_t n, double alpha,
                                                           double* __restrict__ w, double* __restrict__ sum_out) {
  __shared__ double sh[16];
  const double mx = mx_in[0];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double sv = exp(alpha * (f[i] - lq[i] - mx));
    w[i] = sv;
    s += sv;
  }
  s = lro_wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int k = 0; k < 16; ++k) tot += sh[k];
    sum_out[0] = tot;
  }
}

struct EpiStoreV {           // V = acc
  double* V;
  int64_t ld;
  __device__ void operator()(int, int row, int col, double acc) const { V[(int64_t)row * ld + col] = acc; }
};
struct EpiSlabW {            // slab_split = acc
  double* W;
  int64_t ld, slab;
  __device__ void operator()(int split, int row, int col, double acc) const {
    W[split * slab + (int64_t)row * ld + col] = acc;
  }
};

struct LroLayout {
  int64_t ld, nn;
  int splits, n_rb;
  int ldk, ldt, col1;      // strides of the k-wide and T matrices, column of the ones (log q sits at col1 + 1)
  int64_t o_x, o_r, o_g, o_v, o_t, o_tw, o_zp, o_rr, o_f, o_lq, o_lpr, o_w, o_lqc, o_scal, o_mu, o_isig, o_bs, o_minv,
      o_sig, o_b, o_prior, o_w1, o_w2, o_et, o_et2, o_tt, o_col, o_cs1, o_cs2, o_pack, total;
};

// n: local rows, n_total: whole-job rows (gathered per-sample vectors)
LroLayout lro_layout(int64_t n, int64_t n_total, int64_t d, int64_t k) {
  LroLayout L;
  L.ldk = (int)round_up(k, 16);
  L.ldt = L.ldk + 16;
  L.col1 = L.ldk;
  L.ld = round_up(d, 16);
  L.nn = round_up(n_total, 16);
  L.n_rb = (int)((n + 127) / 128);
  static const int rows_per_split = [] {
    const char* e = getenv("VB_LR_SPLIT_ROWS");
    const int v = e ? atoi(e) : 0;
    return v >= 16 ? v : 64;
  }();
  int splits = (int)(n / rows_per_split);
  L.splits = splits > 32 ? 32 : (splits < 1 ? 1 : splits);
  int64_t off = 0;
  auto carve = [&off](int64_t doubles) {
    const int64_t o = off;
    off += round_up(doubles, 16);
    return o;
  };
  const int64_t mat = n * L.ld;
  L.o_x = carve(mat);
  L.o_r = carve(mat);
  L.o_g = carve(mat);
  L.o_v = carve(n * L.ldk);
  L.o_t = carve(n * L.ldt);
  L.o_tw = carve(n * L.ldt);
  L.o_zp = carve(n * L.ldt);
  L.o_rr = carve(n);
  L.o_f = carve(L.nn);
  L.o_lq = carve(L.nn);
  L.o_lpr = carve(L.nn);
  L.o_w = carve(L.nn);
  L.o_lqc = carve(L.nn);
  L.o_scal = carve(64);
  L.o_mu = carve(L.ld);
  L.o_isig = carve(L.ld);
  L.o_bs = carve(d * L.ldk);
  L.o_minv = carve(L.ldk * L.ldk);
  L.o_sig = carve(L.ld);
  L.o_b = carve(d * L.ldk);
  L.o_prior = carve(2 * L.ld);
  L.o_w1 = carve((int64_t)L.splits * d * L.ldt);
  L.o_w2 = carve((int64_t)L.splits * L.ldt * L.ldt);
  L.o_et = carve(d * L.ldt);
  L.o_et2 = carve(d * L.ldt);
  L.o_tt = carve(L.ldt * L.ldt);
  L.o_col = carve((int64_t)L.n_rb * L.ld);
  L.o_cs1 = carve(L.ld);
  L.o_cs2 = carve(L.ld);
  L.o_pack = carve(2 * d * L.ldk + L.ldk * L.ldk + 4 * d + 64);
  L.total = off;
  return L;
}

// the parameter at which log q is evaluated: mu, 1 / sigma, Bs = B / sigma, M^-1
struct LroParam {
  const double *mu, *log_sigma, *B, *minv;
  double cq;     // -(D log 2 pi + log det Sigma) / 2
};

// One staged upload (push_small: no stream wait): the device layout carves mu | 1/sigma | Bs | M^-1 | sigma | B in this
// order without gaps (every piece a multiple of 16 doubles), so the host image is built in the same order and the
// evaluation-only case sends its prefix.
int lro_upload_param(vb_ctx* ctx, const LroLayout& L, double* base, int64_t d, int64_t k, const LroParam& p,
                     bool sampling) {
  const int64_t eval_len = L.o_sig - L.o_mu, all_len = L.o_prior - L.o_mu;
  if (L.o_isig - L.o_mu != L.ld || L.o_bs - L.o_isig != L.ld || L.o_minv - L.o_bs != d * L.ldk ||
      L.o_sig - L.o_minv != (int64_t)L.ldk * L.ldk || L.o_b - L.o_sig != L.ld || L.o_prior - L.o_b != d * L.ldk)
    return fail(ctx, VB_ERR_STATE, "low-rank parameter block is not contiguous");
  std::vector<double> h((size_t)(sampling ? all_len : eval_len), 0.0);
  double *mu = h.data(), *isig = mu + L.ld, *bs = isig + L.ld, *mi = bs + d * L.ldk;
  double *sig = sampling ? mi + (int64_t)L.ldk * L.ldk : nullptr, *b = sampling ? sig + L.ld : nullptr;
  for (int64_t i = 0; i < d; ++i) {
    const double s = exp(p.log_sigma[i]);
    mu[i] = p.mu[i];
    isig[i] = 1.0 / s;
    if (sampling) sig[i] = s;
    for (int64_t j = 0; j < k; ++j) {
      if (sampling) b[i * L.ldk + j] = p.B[i * k + j];
      bs[i * L.ldk + j] = p.B[i * k + j] / s;
    }
  }
  for (int64_t i = 0; i < k; ++i)
    for (int64_t j = 0; j < k; ++j) mi[i * L.ldk + j] = p.minv[i * k + j];
  return push_small(ctx, ctx->stream, h.data(), h.size() * sizeof(double), base + L.o_mu);   // copies `h` before returning
}

// The packed result of the weighted-sum calls: up to 8 strided blocks gathered into the contiguous `pack` in one launch
// (six 2-D device copies of ~5 us each before).
struct LroPackSeg {
  const double* src;
  double* dst;
  int rows, cols;
  int64_t src_ld, dst_ld;
};
struct LroPackArgs {
  LroPackSeg s[8];
};

__global__ __launch_bounds__(256) void lro_pack_kernel(const LroPackArgs a) {
  const LroPackSeg g = a.s[blockIdx.y];
  const int64_t total = (int64_t)g.rows * g.cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / g.cols, c = i - r * g.cols;
    g.dst[r * g.dst_ld + c] = g.src[r * g.src_ld + c];
  }
}

int lro_pack_enqueue(vb_ctx* ctx, hipStream_t st, const LroPackSeg* segs, int n_seg) {
  LroPackArgs a{};
  int64_t most = 1;
  for (int i = 0; i < n_seg; ++i) {
    a.s[i] = segs[i];
    const int64_t t = (int64_t)segs[i].rows * segs[i].cols;
    if (t > most) most = t;
  }
  int64_t bx = (most + 1023) / 1024;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(lro_pack_kernel, dim3((unsigned)bx, (unsigned)n_seg), dim3(256), 0, st, a);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// R, V, T (and log q at `lq_out`) of the stored samples X at the uploaded parameter
int lro_rows(vb_ctx* ctx, const LroLayout& L, double* base, int64_t n, int64_t d, int64_t k, double cq, int t_mode,
             double* lq_out) {
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(lro_resid_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, (const double*)(base + L.o_x),
                     L.ld, n, (int)d, (const double*)(base + L.o_mu), (const double*)(base + L.o_isig), base + L.o_r,
                     base + L.o_rr);
  VB_HIP(ctx, hipGetLastError());
  GemmArgs g;                                   // V = R Bs  (n x k)
  g.A = base + L.o_r, g.lda = L.ld, g.B = base + L.o_bs, g.ldb = L.ldk;
  g.M = (int)n, g.N = (int)k, g.K = (int)d, g.tri_mode = 0;
  gemm_f64_launch<true>(st, g, 1, ctx->prop.multiProcessorCount, EpiStoreV{base + L.o_v, L.ldk});
  VB_HIP(ctx, hipGetLastError());
  if (k <= 16)
    hipLaunchKernelGGL(lro_tau_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const double*)(base + L.o_v),
                       (const double*)(base + L.o_rr), (const double*)(base + L.o_minv), (int)k, cq, n,
                       (const double*)(base + L.o_zp), t_mode, base + L.o_t, lq_out);
  else
    hipLaunchKernelGGL(lro_tau_wave_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), (size_t)(k * L.ldk) * sizeof(double), st,
                       (const double*)(base + L.o_v), (const double*)(base + L.o_rr), (const double*)(base + L.o_minv),
                       (int)k, L.ldk, L.ldt, cq, n, (const double*)(base + L.o_zp), t_mode, base + L.o_t, lq_out);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// out (d x L.ldt) = A' Tw, contraction over the n samples (A: n x ld, Tw: n x L.ldt)
int lro_atb(vb_ctx* ctx, const LroLayout& L, double* base, const double* A, int64_t lda, int m_rows, const double* Tw,
            int64_t n, double* W, double* out) {
  hipStream_t st = ctx->stream;
  GemmArgs g;
  g.A = A, g.lda = lda, g.B = Tw, g.ldb = L.ldt;
  g.M = m_rows, g.N = L.ldt, g.K = (int)n, g.tri_mode = 0;
  const int64_t slab = (int64_t)m_rows * L.ldt;
  gemm_f64_launch<false>(st, g, L.splits, ctx->prop.multiProcessorCount, EpiSlabW{W, L.ldt, slab});
  VB_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(lro_slab_sum_kernel, dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, st, (const double*)W,
                     L.splits, slab, out, slab);
  VB_HIP(ctx, hipGetLastError());
  (void)base;
  return VB_OK;
}

// out[c] = sum_n w_n A_nc B_nc
int lro_colsum_prod(vb_ctx* ctx, const LroLayout& L, double* base, const double* A, int64_t lda, const double* B,
                    int64_t ldb, const double* w, int64_t n, int64_t d, double* out) {
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(lro_colsum_prod_kernel, dim3((unsigned)((d + 63) / 64), (unsigned)L.n_rb), dim3(256), 0, st, A, lda,
                     B, ldb, L.ld, w, n, (int)d, base + L.o_col);
  hipLaunchKernelGGL(lro_slab_sum_kernel, dim3((unsigned)((L.ld + 255) / 256)), dim3(256), 0, st,
                     (const double*)(base + L.o_col), L.n_rb, L.ld, out, d);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// (every caller takes a source model: DIS needs f of the samples only, the alpha sums load the user kernel's G)
int lro_check(vb_ctx* ctx, const NoiseSlot& ns, const NoiseSlot& nz, int64_t n, int64_t d, int64_t k) {
  if (n <= 0 || n > ns.n || d != ns.d || n > nz.n || k != nz.d || k < 1 || k > kMaxRank)
    return fail(ctx, VB_ERR_INVALID, "noise slots must hold n x d and n x k (1 <= k <= 64) matrices");
  if (ctx->model.id != VB_MODEL_GAUSS_DIAG && ctx->model.id != VB_MODEL_FUNNEL &&
      ctx->model.id != VB_MODEL_SOURCE)
    return fail(ctx, VB_ERR_UNSUPPORTED, "low-rank DIS / alpha objectives implement the gauss_diag, funnel and source "
                "models");
  if (ctx->model.dim != d) return fail(ctx, VB_ERR_INVALID, "model dimension != family dimension");
  return VB_OK;
}

}  // namespace

// ---- DISInclusiveKL state refresh (objectives.py:393-401) --------------------------------------------------
int lr_dis_refresh(vb_ctx* ctx, const NoiseSlot& ns, const NoiseSlot& nz, int64_t n, int64_t n_total, int64_t d,
                   int64_t k, const double* mu, const double* log_sigma, const double* B, const double* minv, double cq,
                   const double* prior_host, double eps_prev, double ess_target, int max_its, double* eps_out,
                   double* ess_out, double* w_host, double* logp_host, double* logq_host) {
  VB_TRY(lro_check(ctx, ns, nz, n, d, k));
  int64_t mine = 0;   // this rank's block inside the gathered per-sample vectors (shard_rows)
  VB_TRY(comm_shard_begin(ctx, n, n_total, &mine));
  const LroLayout L = lro_layout(n, n_total, d, k);
  VB_TRY(ensure(ctx, ctx->lr_obj, (size_t)L.total * sizeof(double)));
  double* base = (double*)ctx->lr_obj.ptr;
  hipStream_t st = ctx->stream;
  VB_TRY(lro_upload_param(ctx, L, base, d, k, LroParam{mu, log_sigma, B, minv, cq}, true));
  std::vector<double> pr((size_t)2 * L.ld, 0.0);
  double c0p = -0.5 * (double)d * 1.8378770664093454835606594728112;
  for (int64_t i = 0; i < d; ++i) {
    pr[i] = prior_host[i];
    pr[L.ld + i] = exp(-2.0 * prior_host[d + i]);
    c0p -= prior_host[d + i];
  }
  VB_TRY(push_small(ctx, st, pr.data(), pr.size() * sizeof(double), base + L.o_prior));
  hipLaunchKernelGGL(lro_sample_kernel, dim3((unsigned)n, (unsigned)((L.ld + 255) / 256)), dim3(256), 0, st,
                     (const double*)ns.buf.ptr, ns.ld, (const double*)nz.buf.ptr, nz.ld, n, (int)d, (int)k,
                     (const double*)(base + L.o_mu), (const double*)(base + L.o_sig), (const double*)(base + L.o_b),
                     base + L.o_x, L.ld, base + L.o_zp, L.ldk, L.ldt);
  VB_HIP(ctx, hipGetLastError());
  VB_TRY(lro_rows(ctx, L, base, n, d, k, cq, 0, base + L.o_lq + mine));
  VB_TRY(model_and_prior_logp_rows(ctx, base + L.o_x, L.ld, n, d, base + L.o_f + mine, base + L.o_prior,
                                   base + L.o_prior + L.ld, c0p, base + L.o_lpr + mine));
  if (ctx->temper.kind != VB_PRIOR_DIAG_GAUSSIAN)      // any other family as tempering prior (vb_dis_set_temper_prior)
    VB_TRY(temper_prior_rows(ctx, base + L.o_x, L.ld, n, d, base + L.o_lpr + mine));
  if (ctx->comm) {
    VB_TRY(comm_gather_rows3(ctx, st, base + L.o_f, base + L.o_lq, base + L.o_lpr, mine, n, n_total));
  }
  VB_HIP(ctx, hipMemsetAsync(base + L.o_scal, 0, 32 * sizeof(double), st));   // scal[0] = 0: lq is used as is
  VB_TRY(dis_bisect_enqueue(ctx, base + L.o_f, base + L.o_lq, base + L.o_lpr, base + L.o_scal, n_total, eps_prev,
                            ess_target, max_its, base + L.o_w, base + L.o_lqc, base + L.o_scal + 8));
  double res[3];
  const size_t vec = (size_t)n_total * sizeof(double);
  const FetchSeg segs[4] = {{base + L.o_scal + 8, sizeof res, res}, {base + L.o_w, vec, w_host},
                            {base + L.o_f, logp_host ? vec : 0, logp_host}, {base + L.o_lqc, logq_host ? vec : 0, logq_host}};
  VB_TRY(fetch_blocking(ctx, st, segs, 4));
  *eps_out = res[0];
  *ess_out = res[1];
  ctx->lr_n = n;
  ctx->lr_d = d;
  ctx->lr_k = k;
  ctx->lr_n_total = n_total;
  ++ctx->dis_gen[2];
  if ((int)(res[2]) == 3) return fail(ctx, VB_ERR_STATE, "tempering bisection: a workgroup of the resident kernel did not arrive at a grid barrier (results invalid); VB_DIS_RESIDENT=0 selects the launch chain");
  if ((int)res[2] == 1)
    return fail(ctx, VB_ERR_NUMERIC, "All weights zero! Suggests overflow in importance density.");
  return VB_OK;
}

// weighted sums of the state samples at a (new) parameter: out = [sum w rho tau' (d x k) | sum w tau tau' (k x k) |
// sum w rho (d) | sum w rho^2 (d) | sum w tau (k) | sum w | sum w log q]
int lr_dis_grad(vb_ctx* ctx, int64_t n, int64_t d, int64_t k, const double* mu, const double* log_sigma, const double* B,
                const double* minv, double cq, const double* w_host, double* out_host) {
  if (ctx->lr_n != n || ctx->lr_d != d || ctx->lr_k != k || !ctx->lr_obj.ptr)
    return fail(ctx, VB_ERR_STATE, "no low-rank DIS state of shape %lld x %lld (k = %lld)", (long long)n, (long long)d,
                (long long)k);
  const LroLayout L = lro_layout(n, ctx->lr_n_total, d, k);
  double* base = (double*)ctx->lr_obj.ptr;
  hipStream_t st = ctx->stream;
  VB_TRY(push_small(ctx, st, w_host, (size_t)n * sizeof(double), base + L.o_w));
  VB_TRY(lro_upload_param(ctx, L, base, d, k, LroParam{mu, log_sigma, B, minv, cq}, false));
  VB_TRY(lro_rows(ctx, L, base, n, d, k, cq, 0, base + L.o_lq));
  hipLaunchKernelGGL(lro_scale_rows_kernel, dim3((unsigned)((n * L.ldt + 255) / 256)), dim3(256), 0, st,
                     (const double*)(base + L.o_t), (const double*)(base + L.o_w), n, base + L.o_tw, L.ldt);
  VB_HIP(ctx, hipGetLastError());
  VB_TRY(lro_atb(ctx, L, base, base + L.o_r, L.ld, (int)d, base + L.o_tw, n, base + L.o_w1, base + L.o_et));
  VB_TRY(lro_atb(ctx, L, base, base + L.o_t, L.ldt, L.ldt, base + L.o_tw, n, base + L.o_w2, base + L.o_tt));
  VB_TRY(lro_colsum_prod(ctx, L, base, base + L.o_r, L.ld, base + L.o_r, L.ld, base + L.o_w, n, d, base + L.o_cs1));
  // pack
  const int64_t out_len = d * k + k * k + 2 * d + k + 2;
  double* pack = base + L.o_pack;
  double* tail = pack + d * k + k * k;
  const double* ones_row = base + L.o_tt + (int64_t)L.col1 * L.ldt;
  const LroPackSeg ps[6] = {{base + L.o_et, pack, (int)d, (int)k, L.ldt, k},
                            {base + L.o_tt, pack + d * k, (int)k, (int)k, L.ldt, k},
                            {base + L.o_et + L.col1, tail, (int)d, 1, L.ldt, 1},            // sum w rho
                            {base + L.o_cs1, tail + d, 1, (int)d, 0, 0},                    // sum w rho^2
                            {ones_row, tail + 2 * d, 1, (int)k, 0, 0},                      // sum w tau
                            {ones_row + L.col1, tail + 2 * d + k, 1, 2, 0, 0}};             // sum w, sum w log q
  VB_TRY(lro_pack_enqueue(ctx, st, ps, 6));
  if (ctx->comm) VB_TRY(comm_allreduce_sum(ctx, st, pack, (size_t)out_len));
  const FetchSeg seg{pack, (size_t)out_len * sizeof(double), out_host};
  return fetch_blocking(ctx, st, &seg, 1);
}

// ---- AlphaDivergence (objectives.py:453-461) ------------------------------------------------------------------
// out = [sum s g z' (d x k) | sum s eps t' (d x k) | sum s t t' (k x k) | sum s g (d) | sum s g eps (d)]
int lr_alpha_sums(vb_ctx* ctx, const NoiseSlot& ns, const NoiseSlot& nz, int64_t n, int64_t n_total, int64_t d,
                  int64_t k, double alpha, const double* mu, const double* log_sigma, const double* B,
                  const double* minv, double cq, double* value_out, double* wsum_out, double* out_host) {
  VB_TRY(lro_check(ctx, ns, nz, n, d, k));
  if (!(alpha != 0.0)) return fail(ctx, VB_ERR_INVALID, "alpha must be non-zero");
  const LroLayout L = lro_layout(n, n_total, d, k);
  VB_TRY(ensure(ctx, ctx->lr_obj, (size_t)L.total * sizeof(double)));
  ctx->lr_n = 0;                       // the buffer no longer holds a DIS state
  ++ctx->dis_gen[2];
  double* base = (double*)ctx->lr_obj.ptr;
  hipStream_t st = ctx->stream;
  const double* E = (const double*)ns.buf.ptr;
  VB_TRY(lro_upload_param(ctx, L, base, d, k, LroParam{mu, log_sigma, B, minv, cq}, true));
  hipLaunchKernelGGL(lro_sample_kernel, dim3((unsigned)n, (unsigned)((L.ld + 255) / 256)), dim3(256), 0, st, E, ns.ld,
                     (const double*)nz.buf.ptr, nz.ld, n, (int)d, (int)k, (const double*)(base + L.o_mu),
                     (const double*)(base + L.o_sig), (const double*)(base + L.o_b), base + L.o_x, L.ld, base + L.o_zp, L.ldk, L.ldt);
  VB_HIP(ctx, hipGetLastError());
  VB_TRY(lro_rows(ctx, L, base, n, d, k, cq, 1, base + L.o_lq));          // T = [t | 1 | log q], t = z - tau
  const bool source = ctx->model.id == VB_MODEL_SOURCE;
  if (source) {      // the user's row kernel gives f and G in one pass (pad columns of G stay zero)
    VB_HIP(ctx, hipMemsetAsync(base + L.o_g, 0, (size_t)n * L.ld * sizeof(double), st));
    VB_TRY(user_rows_enqueue(ctx, st, base + L.o_x, L.ld, n, (int)d, base + L.o_g, L.ld, base + L.o_f));
  } else {
    VB_TRY(model_logp_rows(ctx, base + L.o_x, L.ld, n, d, base + L.o_f));
  }
  double* scal = base + L.o_scal;
  hipLaunchKernelGGL(lro_lw_max_kernel, dim3(1), dim3(1024), 0, st, (const double*)(base + L.o_f),
                     (const double*)(base + L.o_lq), n, scal + 8);
  VB_HIP(ctx, hipGetLastError());
  VB_TRY(comm_allreduce_max(ctx, st, scal + 8, 1));
  hipLaunchKernelGGL(lro_weights_kernel, dim3(1), dim3(1024), 0, st, (const double*)(base + L.o_f),
                     (const double*)(base + L.o_lq), (const double*)(scal + 8), n, alpha, base + L.o_w, scal + 9);
  VB_HIP(ctx, hipGetLastError());
  VB_TRY(comm_allreduce_sum(ctx, st, scal + 9, 1));
  if (!source)
    hipLaunchKernelGGL(lro_model_grad_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st,
                       (const double*)(base + L.o_x), L.ld, n, (int)d, ctx->model, base + L.o_g);
  VB_HIP(ctx, hipGetLastError());
  const dim3 sgrid((unsigned)((n * L.ldt + 255) / 256));
  // E' (s T): sum s eps t';  T' (s T): sum s t t'
  hipLaunchKernelGGL(lro_scale_rows_kernel, sgrid, dim3(256), 0, st, (const double*)(base + L.o_t),
                     (const double*)(base + L.o_w), n, base + L.o_tw, L.ldt);
  VB_TRY(lro_atb(ctx, L, base, E, ns.ld, (int)d, base + L.o_tw, n, base + L.o_w1, base + L.o_et));
  VB_TRY(lro_atb(ctx, L, base, base + L.o_t, L.ldt, L.ldt, base + L.o_tw, n, base + L.o_w2, base + L.o_tt));
  // G' (s [z | 1]): sum s g z', sum s g
  hipLaunchKernelGGL(lro_scale_rows_kernel, sgrid, dim3(256), 0, st, (const double*)(base + L.o_zp),
                     (const double*)(base + L.o_w), n, base + L.o_tw, L.ldt);
  VB_TRY(lro_atb(ctx, L, base, base + L.o_g, L.ld, (int)d, base + L.o_tw, n, base + L.o_w1, base + L.o_et2));
  VB_TRY(lro_colsum_prod(ctx, L, base, base + L.o_g, L.ld, E, ns.ld, base + L.o_w, n, d, base + L.o_cs1));
  const int64_t out_len = 2 * d * k + k * k + 2 * d;
  double* pack = base + L.o_pack;
  double* tail = pack + 2 * d * k + k * k;
  const LroPackSeg ps[5] = {{base + L.o_et2, pack, (int)d, (int)k, L.ldt, k},
                            {base + L.o_et, pack + d * k, (int)d, (int)k, L.ldt, k},
                            {base + L.o_tt, pack + 2 * d * k, (int)k, (int)k, L.ldt, k},
                            {base + L.o_et2 + L.col1, tail, (int)d, 1, L.ldt, 1},           // sum s g
                            {base + L.o_cs1, tail + d, 1, (int)d, 0, 0}};                   // sum s g eps
  VB_TRY(lro_pack_enqueue(ctx, st, ps, 5));
  if (ctx->comm) VB_TRY(comm_allreduce_sum(ctx, st, pack, (size_t)out_len));
  double sc[2];
  const FetchSeg segs[2] = {{scal + 8, sizeof sc, sc}, {pack, (size_t)out_len * sizeof(double), out_host}};
  VB_TRY(fetch_blocking(ctx, st, segs, 2));
  *wsum_out = sc[1];
  *value_out = log(sc[1] / (double)n_total) / alpha + sc[0];                     // objectives.py:459
  return VB_OK;
}

// ---- ExclusiveKL sums for ANY rank k (entropy form; the streaming kernel of vb_lowrank.hip keeps a lane's rows of B in
// registers and stops at k = 16) ------------------------------------------------------------------------------------------
// The reference's LRGaussian has no rank limit (approximations.py:610-644, :685-707).  For k > 16 the evaluation is
// assembled from GEMMs: X = mu + sigma E + Z B' (n x kp x d product with the sampling epilogue), the model's (f, G) of
// the materialised samples (model_grad_rows: every built-in target and source models), sum_n g and sum_n g eps
// (column passes) and sum_n g z' = G' Z (d x kp, contraction over the samples split into slabs and summed in fixed order).
// out = [sum f | sum g (d) | sum g eps (d) | sum g z' (d x k row-major)]; the entropy and its gradient are O(D k^2) host
// algebra through the capacitance matrix, as for the k <= 16 objectives.
namespace {

struct EpiLrSample {         // X = acc + mu + sigma E
  double* X;
  int64_t ldx;
  const double* mu;
  const double* sigma;
  const double* E;
  int64_t lde;
  __device__ void operator()(int, int row, int col, double acc) const {
    X[(int64_t)row * ldx + col] = acc + fma(sigma[col], E[(int64_t)row * lde + col], mu[col]);
  }
};

struct EpiSlabAny {          // slab_split[i][j] = acc, row stride ldc
  double* C;
  int64_t ldc, slab;
  __device__ void operator()(int split, int row, int col, double acc) const {
    C[split * slab + (int64_t)row * ldc + col] = acc;
  }
};

// theta = [mu | log sigma | B (d x k, row-major)] on the device -> mu, sigma = exp(log sigma) (pads zero) and
// B' (kp x ld, zero padded): the transposition and the exponentials the host loop used to do
__global__ void __launch_bounds__(256) lrs_prep_kernel(const double* __restrict__ theta, int d, int k, int64_t ld, int kp,
                                                       double* __restrict__ mu, double* __restrict__ sigma,
                                                       double* __restrict__ bt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // over (kp + 1) x ld
  if (i >= (int64_t)(kp + 1) * ld) return;
  const int row = (int)(i / ld), c = (int)(i % ld);
  if (row == kp) {
    mu[c] = c < d ? theta[c] : 0.0;
    sigma[c] = c < d ? exp(theta[d + c]) : 0.0;
  } else {
    bt[i] = (c < d && row < k) ? theta[2 * (int64_t)d + (int64_t)c * k + row] : 0.0;
  }
}

// per 128-row block and column: sum_n G_nc and sum_n G_nc E_nc (the two column passes of the gradient in one read of G)
__global__ void __launch_bounds__(256) lrs_colsums_kernel(const double* __restrict__ G, int64_t ldg,
                                                          const double* __restrict__ E, int64_t lde, int64_t ld, int64_t n,
                                                          int d, double* __restrict__ part /* [n_rb][2][ld] */) {
  __shared__ double sh[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * 128, r1 = r0 + 128 < n ? r0 + 128 : n;
  double sg = 0.0, se = 0.0;
  if (c < d)
    for (int64_t r = r0 + q; r < r1; r += 4) {
      const double g = G[r * ldg + c];
      sg += g;
      se = fma(g, E[r * lde + c], se);
    }
  sh[0][q][threadIdx.x & 63] = sg;
  sh[1][q][threadIdx.x & 63] = se;
  __syncthreads();
  if (q < 2 && c < d) {
    const int l = c & 63;
    part[((int64_t)blockIdx.y * 2 + q) * ld + c] = (sh[q][0][l] + sh[q][1][l]) + (sh[q][2][l] + sh[q][3][l]);
  }
}

// out = [sum f | sum g (d) | sum g eps (d) | sum g z' (d x k row-major)], every sum in a fixed order: element i of the
// vector is one thread's; sum f is the first workgroup's
__global__ void __launch_bounds__(256) lrs_finish_kernel(const double* __restrict__ f, int64_t n,
                                                         const double* __restrict__ colpart, int n_rb, int64_t ld,
                                                         const double* __restrict__ slabs, int splits, int64_t slab, int kp,
                                                         int d, int k, double* __restrict__ out) {
  __shared__ double sh[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = 2 * (int64_t)d + (int64_t)d * k;
  if (i < total) {
    double s = 0.0;
    if (i < 2 * (int64_t)d) {
      const int which = (int)(i / d), c = (int)(i % d);
      for (int rb = 0; rb < n_rb; ++rb) s += colpart[((int64_t)rb * 2 + which) * ld + c];
    } else {
      const int64_t e = i - 2 * (int64_t)d;
      const int64_t src = (e / k) * kp + e % k;
      for (int q = 0; q < splits; ++q) s += slabs[q * slab + src];
    }
    out[1 + i] = s;
  }
  if (blockIdx.x == 0) {
    double s = 0.0;
    for (int64_t j = threadIdx.x; j < n; j += 256) s += f[j];
    s = lro_wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  }
}

}  // namespace

int lr_elbo_sums_any_rank(vb_ctx* ctx, const NoiseSlot& ns, const NoiseSlot& nz, int64_t n, int64_t d, int64_t k,
                          const double* theta_host, double* out_host) {
  if (n <= 0 || n > ns.n || d != ns.d || n > nz.n || k != nz.d || k < 1)
    return fail(ctx, VB_ERR_INVALID, "noise slots must hold n x d and n x k matrices");
  if (ctx->model.dim != d) return fail(ctx, VB_ERR_INVALID, "model dimension != family dimension");
  const int64_t ld = round_up(d, 16), kp = round_up(k, 16);
  if (nz.ld < kp) return fail(ctx, VB_ERR_STATE, "low-rank noise slot row stride %lld < %lld", (long long)nz.ld, (long long)kp);
  const int n_rb = (int)((n + 127) / 128);
  int splits = (int)(n / 256);
  splits = splits > 32 ? 32 : (splits < 1 ? 1 : splits);
  const int64_t p = 2 * d + d * k, n_out = 1 + p;
  int64_t off = 0;
  auto carve = [&off](int64_t doubles) {
    const int64_t o = off;
    off += round_up(doubles, 16);
    return o;
  };
  const int64_t o_x = carve(n * ld), o_g = carve(n * ld), o_f = carve(n), o_theta = carve(p), o_mu = carve(ld),
                o_sig = carve(ld), o_bt = carve(kp * ld), o_col = carve((int64_t)n_rb * 2 * ld),
                o_slabs = carve((int64_t)splits * d * kp), o_out = carve(n_out);
  VB_TRY(ensure(ctx, ctx->lr_obj, (size_t)off * sizeof(double)));
  double* base = (double*)ctx->lr_obj.ptr;
  hipStream_t st = ctx->stream;
  // the parameter goes up as it is, through the pinned staging area (no synchronisation: the call's last statement
  // waits for the stream, so the area is free again when the next call writes it)
  VB_TRY(ensure_pinned(ctx, (size_t)(p + n_out) * sizeof(double)));
  memcpy(ctx->pin_host, theta_host, (size_t)p * sizeof(double));
  VB_HIP(ctx, hipMemcpyAsync(base + o_theta, ctx->pin_host, (size_t)p * sizeof(double), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(lrs_prep_kernel, dim3((unsigned)(((kp + 1) * ld + 255) / 256)), dim3(256), 0, st,
                     (const double*)(base + o_theta), (int)d, (int)k, ld, (int)kp, base + o_mu, base + o_sig, base + o_bt);
  VB_HIP(ctx, hipGetLastError());
  const double* E = (const double*)ns.buf.ptr;
  const double* Z = (const double*)nz.buf.ptr;
  const int n_cu = ctx->prop.multiProcessorCount;
  GemmArgs g1;                                  // X = Z B' (+ mu + sigma E): n x d, contraction over the kp columns of Z
  g1.A = Z, g1.lda = nz.ld, g1.B = base + o_bt, g1.ldb = ld;
  g1.M = (int)n, g1.N = (int)d, g1.K = (int)kp, g1.tri_mode = 0;
  gemm_f64_launch<true>(st, g1, 1, n_cu, EpiLrSample{base + o_x, ld, base + o_mu, base + o_sig, E, ns.ld});
  VB_HIP(ctx, hipGetLastError());
  // (pad columns of G are never read: the column pass and the product below stop at d)
  VB_TRY(model_grad_rows(ctx, base + o_x, ld, n, d, base + o_g, base + o_f));
  hipLaunchKernelGGL(lrs_colsums_kernel, dim3((unsigned)((d + 63) / 64), (unsigned)n_rb), dim3(256), 0, st,
                     (const double*)(base + o_g), ld, E, ns.ld, ld, n, (int)d, base + o_col);
  VB_HIP(ctx, hipGetLastError());
  GemmArgs g2;                                  // G' Z: d x kp, contraction over the n samples
  g2.A = base + o_g, g2.lda = ld, g2.B = Z, g2.ldb = nz.ld;
  g2.M = (int)d, g2.N = (int)kp, g2.K = (int)n, g2.tri_mode = 0;
  const int64_t slab = d * kp;
  gemm_f64_launch<false>(st, g2, splits, n_cu, EpiSlabAny{base + o_slabs, kp, slab});
  VB_HIP(ctx, hipGetLastError());
  double* outd = base + o_out;
  hipLaunchKernelGGL(lrs_finish_kernel, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, st, (const double*)(base + o_f), n,
                     (const double*)(base + o_col), n_rb, ld, (const double*)(base + o_slabs), splits, slab, (int)kp, (int)d,
                     (int)k, outd);
  VB_HIP(ctx, hipGetLastError());
  if (ctx->comm) VB_TRY(comm_allreduce_sum(ctx, st, outd, (size_t)n_out));
  const FetchSeg seg{outd, (size_t)n_out * sizeof(double), out_host};
  return fetch_blocking(ctx, st, &seg, 1);
}

}  // namespace vb

using namespace vb;

extern "C" {

static int lr_args_ok(vb_ctx* ctx, int slot_eps, int slot_z) {
  if (slot_eps < 0 || slot_eps >= VB_MAX_SLOTS || slot_z < 0 || slot_z >= VB_MAX_SLOTS)
    return fail(ctx, VB_ERR_INVALID, "slot out of range [0, %d)", VB_MAX_SLOTS);
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (!ctx->noise[slot_eps].buf.ptr || !ctx->noise[slot_z].buf.ptr) return fail(ctx, VB_ERR_STATE, "noise slot is empty");
  return VB_OK;
}

int vb_dis_refresh_lowrank(vb_ctx* ctx, int slot_eps, int slot_z, int64_t n, int64_t d, int64_t k, int64_t n_total,
                           const double* mu, const double* log_sigma, const double* b, const double* m_inv,
                           double log_q_const, const double* prior_theta, double eps_prev, double ess_target,
                           int max_bisection_its, double* eps, double* ess, double* w, double* log_p, double* log_q) {
  if (!ctx || !mu || !log_sigma || !b || !m_inv || !prior_theta || !eps || !ess || !w)
    return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(lr_args_ok(ctx, slot_eps, slot_z));
  VB_HIP(ctx, hipSetDevice(ctx->device));
  return lr_dis_refresh(ctx, ctx->noise[slot_eps], ctx->noise[slot_z], n, n_total, d, k, mu, log_sigma, b, m_inv,
                        log_q_const, prior_theta, eps_prev, ess_target, max_bisection_its, eps, ess, w, log_p, log_q);
}

int vb_dis_grad_lowrank(vb_ctx* ctx, int64_t n, int64_t d, int64_t k, const double* mu, const double* log_sigma,
                        const double* b, const double* m_inv, double log_q_const, const double* weights, double* out) {
  if (!ctx || !mu || !log_sigma || !b || !m_inv || !weights || !out) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  return lr_dis_grad(ctx, n, d, k, mu, log_sigma, b, m_inv, log_q_const, weights, out);
}

int vb_alpha_sums_lowrank(vb_ctx* ctx, int slot_eps, int slot_z, int64_t n, int64_t d, int64_t k, int64_t n_total,
                          double alpha, const double* mu, const double* log_sigma, const double* b, const double* m_inv,
                          double log_q_const, double* value, double* w_sum, double* out) {
  if (!ctx || !mu || !log_sigma || !b || !m_inv || !value || !w_sum || !out)
    return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(lr_args_ok(ctx, slot_eps, slot_z));
  if (n_total < n) return fail(ctx, VB_ERR_INVALID, "n_total must be >= n");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  return lr_alpha_sums(ctx, ctx->noise[slot_eps], ctx->noise[slot_z], n, n_total, d, k, alpha, mu, log_sigma, b, m_inv,
                       log_q_const, value, w_sum, out);
}

}  // extern "C"
