// ExclusiveKL for the low-rank-plus-diagonal Gaussian family.
// Reference: viabel/approximations.py:610-731 (LRGaussian): theta = [mu (D) | log_sigma (D) | B (D x k)],
//   sample  x = mu + z B' + sigma * eps,  z ~ N(0, I_k) drawn before eps ~ N(0, I_D)      (:636-644)
//   entropy H = D/2 (log 2 pi + 1) + 1/2 log det(B B' + diag(sigma^2))                      (:646-652, :559-573)
// and viabel/objectives.py:154-164 (entropy form of the ELBO).  With g_n = grad f(x_n):
//   d/dmu = -mean g,  d/dlog_sigma = -mean(g * eps) sigma - dH/dlog_sigma,  d/dB = -mean g z' - dH/dB.
// The matrix determinant lemma gives log det Sigma = 2 sum log sigma + log det M, M = I_k + B' D^-1 B, hence
//   dH/dB = D^-1 B M^-1,   dH/dlog_sigma_i = 1 - b_i' M^-1 b_i / sigma_i^2        (O(D k^2), one workgroup).
//
// One streaming pass over eps (HBM-bound for small k; 5k + 11 fp64 FMA-class ops per 16 B): lane l of a wave
// owns columns 2l, 2l+1 of a 128-column block with its two rows of B in registers; the row's z_n (k doubles) is
// wave-uniform and comes through the scalar cache.  Per-column sums G = sum g, GE = sum g eps, GZ_j = sum g z_j
// stay in registers; fixed-order reductions, no atomics.  k <= 16.
#include "vb_common.h"

#include <cstdlib>

namespace vb {

typedef double lr_d2 __attribute__((ext_vector_type(2)));

constexpr int kLrCols = 128, kLrWaves = 4;
constexpr int kLrMaxRows = 128;           // rows per workgroup (their z block lives in LDS: 128 x 18 doubles)
constexpr int kLrScal = 32;                 // scalar slots at the head of the sum vector
constexpr double kLog2PiLr = 1.8378770664093454835606594728112;

struct LrArgs {
  const double* eps;      // N x D noise, row stride ld
  const double* z;        // N x k noise, row stride ldk
  int64_t ld, ldk, n;
  int d, k, Dp, n_rb, n_cb, rows_per_wg, n_prep;
  // workspace
  double* theta_dev;      // [mu | log_sigma | B] device copy
  double* colp;           // [3][Dp]
  double* Bp;             // [Dp][KP]
  double* rowscal;        // [N][2]  (funnel)
  double* prepscal;       // [3 + KP][n_prep]
  double* partials;       // [n_rb][2 + KP][Dp]
  double* pscal;          // [3 + KP][n_rb * n_cb]
  double* sums;           // [kLrScal | (2 + KP) x Dp]
  double* mpart;          // [n_mblk][KP x KP] partial B' D^-1 B of 256-row blocks (prep)
  double* minv;           // [KP x KP] inverse capacitance matrix | [KP x KP] = log det M  (capacitance kernel)
  int n_mblk;
  // source model (the user's row kernel ran on X = the samples): its gradient matrix and values
  const double* G;        // N x d, row stride ld (the noise's), pad columns zero
  const double* frow;     // [N]
};

__device__ __forceinline__ double lr_wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

// ---- prep: theta -> column constants, padded B, funnel row scalars ------------------------------------
template <int KP>
__global__ void __launch_bounds__(256) lr_prep_kernel(const double* __restrict__ theta_src, const LrArgs a,
                                                      const ModelDev m) {
  __shared__ double sh[4][3 + KP];
  __shared__ double thc[2 + KP];     // coupling column's (mu, log sigma, B row): one PCIe read each per block
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int d = a.d, k = a.k;
  const bool funnel = m.id == VB_MODEL_FUNNEL;
  if (funnel) {
    if (threadIdx.x < 2) thc[threadIdx.x] = theta_src[threadIdx.x * d + m.k];
    else if (threadIdx.x < 2 + KP)
      thc[threadIdx.x] = (int)threadIdx.x - 2 < k ? theta_src[2 * (int64_t)d + (int64_t)m.k * k + (threadIdx.x - 2)] : 0.0;
    __syncthreads();
  }
  // theta lives in pinned host memory: this block's 256 rows of B (contiguous, 256 k doubles) cross PCIe once,
  // coalesced, into LDS; every later use (device copy, padded B, capacitance products) reads the LDS copy
  __shared__ double Bl[256 * KP];
  const int64_t row0 = (int64_t)blockIdx.x * 256;
  const bool has_rows = row0 < d;
  double my_ls = 0.0;
  if (has_rows) {
    const int64_t nel = (d - row0 < 256 ? d - row0 : 256) * k;
    double tmp[KP];                       // all of a thread's PCIe reads in flight at once
#pragma unroll
    for (int u = 0; u < KP; ++u) {
      const int64_t e = (int64_t)u * 256 + threadIdx.x;
      tmp[u] = e < nel ? theta_src[2 * (int64_t)d + row0 * k + e] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < KP; ++u) {
      const int64_t e = (int64_t)u * 256 + threadIdx.x;
      if (e < nel) Bl[e] = tmp[u];
    }
    __syncthreads();
  }
  if (i < a.Dp) {
    double c0 = 0.0, c1 = 0.0, c2 = 0.0;
    const bool live = i < d && !(funnel && i == m.k);
    if (i < d) {
      const double mu = theta_src[i], ls = theta_src[d + i];
      my_ls = ls;
      a.theta_dev[i] = mu;
      a.theta_dev[d + i] = ls;
      for (int j = 0; j < k; ++j) a.theta_dev[2 * (int64_t)d + i * k + j] = Bl[threadIdx.x * k + j];
      if (live) {
        const bool centred = m.id == VB_MODEL_GAUSS_DIAG;     // funnel, source model: x itself
        c0 = centred ? mu - m.p0[i] : mu;
        c1 = exp(ls);
        c2 = centred ? m.p1[i] : 0.0;
      }
    }
    a.colp[i] = c0;
    a.colp[a.Dp + i] = c1;
    a.colp[2 * (int64_t)a.Dp + i] = c2;
#pragma unroll
    for (int j = 0; j < KP; ++j) a.Bp[i * KP + j] = (live && j < k) ? Bl[threadIdx.x * k + j] : 0.0;
  }
  double acc[3 + KP];
#pragma unroll
  for (int q = 0; q < 3 + KP; ++q) acc[q] = 0.0;
  if (funnel && i < a.n) {
    const int c = m.k;
    const double ek = a.eps[i * a.ld + c];
    double v = fma(exp(thc[1]), ek, thc[0]);
    double zj[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) {
      zj[j] = j < k ? a.z[i * a.ldk + j] : 0.0;
      v = fma(zj[j], thc[2 + j], v);
    }
    const double it2 = 1.0 / (m.tau * m.tau), dm1 = (double)(d - 1);
    a.rowscal[2 * i] = exp(-2.0 * v);
    a.rowscal[2 * i + 1] = ek;
    const double gk = fma(-v, it2, -dm1);
    acc[0] = v * fma(-0.5 * v, it2, -dm1);
    acc[1] = gk;
    acc[2] = gk * ek;
#pragma unroll
    for (int j = 0; j < KP; ++j) acc[3 + j] = gk * zj[j];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 3 + KP; ++q) {
    const double s = lr_wave_sum(acc[q]);
    if (lane == 0) sh[wave][q] = s;
  }
  __syncthreads();
  if (threadIdx.x < 3 + KP)
    a.prepscal[(int64_t)threadIdx.x * a.n_prep + blockIdx.x] =
        (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
  // this block's share of B' D^-1 B (the k x k capacitance matrix of the determinant lemma): its 256 rows, scaled
  // by 1 / sigma, go through LDS; the 256 threads form G groups of KP^2 (p, q) pairs, group g sums rows g, g + G, ...
  if ((int)blockIdx.x < a.n_mblk) {
    constexpr int G = 256 / (KP * KP);
    __shared__ double wt[256][KP + 1];
    __shared__ double mg[G][KP * KP];
    const int t = threadIdx.x;
    const double iv = i < d ? exp(-my_ls) : 0.0;
#pragma unroll
    for (int j = 0; j < KP; ++j) wt[t][j] = (i < d && j < k) ? Bl[t * k + j] * iv : 0.0;
    __syncthreads();
    const int pq = t % (KP * KP), grp = t / (KP * KP);
    const int p = pq / KP, q = pq % KP;
    double acc4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int r = grp; r < 256; r += 4 * G) {
      acc4[0] = fma(wt[r][p], wt[r][q], acc4[0]);
      acc4[1] = fma(wt[r + G][p], wt[r + G][q], acc4[1]);
      acc4[2] = fma(wt[r + 2 * G][p], wt[r + 2 * G][q], acc4[2]);
      acc4[3] = fma(wt[r + 3 * G][p], wt[r + 3 * G][q], acc4[3]);
    }
    mg[grp][pq] = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
    __syncthreads();
    if (t < KP * KP) {
      double tot = 0.0;
#pragma unroll
      for (int g = 0; g < G; ++g) tot += mg[g][t];
      a.mpart[(int64_t)blockIdx.x * (KP * KP) + t] = tot;
    }
  }
}

// ---- capacitance matrix M = I + B' D^-1 B: Cholesky, inverse, log det (one small workgroup) --------------------
// Right-looking Cholesky in LDS with one thread per matrix entry (3 barriers per column); the inverse by one
// thread per column (forward / back substitution with the factor in LDS).
template <int KP>
__global__ void __launch_bounds__(KP * KP) lr_capacitance_kernel(const LrArgs a) {
  __shared__ double M[KP][KP + 1];
  const int t = threadIdx.x, p = t / KP, q = t % KP;
  double s = (p == q) ? 1.0 : 0.0;
  for (int b = 0; b < a.n_mblk; ++b) s += a.mpart[(int64_t)b * (KP * KP) + t];
  M[p][q] = s;
  __syncthreads();
  for (int j = 0; j < KP; ++j) {
    if (t == 0) M[j][j] = sqrt(M[j][j]);
    __syncthreads();
    if (q == j && p > j) M[p][j] /= M[j][j];
    __syncthreads();
    if (p > j && q > j && q <= p) M[p][q] -= M[p][j] * M[q][j];
    __syncthreads();
  }
  if (t < KP) {   // column t of M^-1: C y = e_t, C' x = y
    double y[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) {
      double v = (i == t) ? 1.0 : 0.0;
#pragma unroll
      for (int r = 0; r < KP; ++r)
        if (r < i) v -= M[i][r] * y[r];
      y[i] = v / M[i][i];
    }
#pragma unroll
    for (int ii = 0; ii < KP; ++ii) {
      const int i = KP - 1 - ii;
      double v = y[i];
#pragma unroll
      for (int r = 0; r < KP; ++r)
        if (r > i) v -= M[r][i] * y[r];
      y[i] = v / M[i][i];
    }
#pragma unroll
    for (int i = 0; i < KP; ++i) a.minv[i * KP + t] = y[i];
  }
  if (t == 0) {
    double ld = 0.0;
    for (int j = 0; j < KP; ++j) ld += log(M[j][j]);
    a.minv[KP * KP] = 2.0 * ld;
  }
}

// ---- source model: the samples X = mu + sigma * eps + z B' as a matrix for the user's row kernel -----------------
template <int KP>
__global__ void __launch_bounds__(256) lr_sample_kernel(const LrArgs a, double* __restrict__ X) {
  __shared__ double zr[KP];
  const int64_t row = blockIdx.x;            // rows on x: gridDim.y stops at 65 535
  const int col = blockIdx.y * 256 + threadIdx.x;
  if (threadIdx.x < KP) zr[threadIdx.x] = (int)threadIdx.x < a.k ? a.z[row * a.ldk + threadIdx.x] : 0.0;
  __syncthreads();
  if (col >= a.d) return;
  double x = fma(a.colp[a.Dp + col], a.eps[row * a.ld + col], a.colp[col]);
#pragma unroll
  for (int j = 0; j < KP; ++j) x = fma(zr[j], a.Bp[(int64_t)col * KP + j], x);
  X[row * a.ld + col] = x;
}

// ---- the streaming pass --------------------------------------------------------------------------------
template <int MODEL, int KP>
__global__ void __launch_bounds__(256) lr_accum_kernel(const LrArgs a) {
  constexpr bool SRC = MODEL == VB_MODEL_SOURCE;           // g is loaded (the user's kernel made it), not computed
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rb = blockIdx.x / a.n_cb, cb = blockIdx.x % a.n_cb;
  const int c0i = cb * kLrCols + 2 * lane;
  const bool lane_ok = c0i < a.ld;
  const int64_t r0 = (int64_t)rb * a.rows_per_wg;
  const int64_t r1 = r0 + a.rows_per_wg < a.n ? r0 + a.rows_per_wg : a.n;
  const lr_d2 cp0 = *reinterpret_cast<const lr_d2*>(a.colp + c0i);
  const lr_d2 cp1 = *reinterpret_cast<const lr_d2*>(a.colp + a.Dp + c0i);
  lr_d2 cp2 = (lr_d2){0.0, 0.0};
  if (MODEL == VB_MODEL_GAUSS_DIAG) cp2 = *reinterpret_cast<const lr_d2*>(a.colp + 2 * (int64_t)a.Dp + c0i);
  double b0[KP], b1[KP];
#pragma unroll
  for (int j = 0; j < KP; ++j) {
    b0[j] = SRC ? 0.0 : a.Bp[(int64_t)c0i * KP + j];
    b1[j] = SRC ? 0.0 : a.Bp[(int64_t)(c0i + 1) * KP + j];
  }
  lr_d2 aG = (lr_d2){0.0, 0.0}, aGE = (lr_d2){0.0, 0.0};
  double gz0[KP], gz1[KP], qz[KP];
#pragma unroll
  for (int j = 0; j < KP; ++j) gz0[j] = gz1[j] = qz[j] = 0.0;
  double F = 0.0, QG = 0.0, QGE = 0.0;

  // one row: x = c0 + sigma e + z B', model gradient, accumulate
  auto row = [&](const lr_d2 ev, const lr_d2 gv, const double* zj, double w, double ek) __attribute__((always_inline)) {
    double x0 = fma(cp1.x, ev.x, cp0.x), x1 = fma(cp1.y, ev.y, cp0.y);
    if (!SRC) {
#pragma unroll
      for (int j = 0; j < KP; ++j) {
        x0 = fma(zj[j], b0[j], x0);
        x1 = fma(zj[j], b1[j], x1);
      }
    }
    double g0, g1;
    if (SRC) {
      g0 = gv.x;
      g1 = gv.y;
    } else if (MODEL == VB_MODEL_GAUSS_DIAG) {            // x holds z - m
      g0 = -x0 * cp2.x;
      g1 = -x1 * cp2.y;
      F = fma(0.5 * x0, g0, fma(0.5 * x1, g1, F));
    } else {                                              // funnel, non-coupling columns
      g0 = -x0 * w;
      g1 = -x1 * w;
      const double t = w * fma(x0, x0, x1 * x1);
      QG += t;
      QGE = fma(t, ek, QGE);
#pragma unroll
      for (int j = 0; j < KP; ++j) qz[j] = fma(t, zj[j], qz[j]);
    }
    aG += (lr_d2){g0, g1};
    aGE += (lr_d2){g0 * ev.x, g1 * ev.y};
#pragma unroll
    for (int j = 0; j < KP; ++j) {
      gz0[j] = fma(g0, zj[j], gz0[j]);
      gz1[j] = fma(g1, zj[j], gz1[j]);
    }
  };

  // the workgroup's rows of z (k doubles each) and the funnel's row scalars go to LDS once; every wave then
  // reads "its" row with broadcast ds_reads (all lanes, same address) instead of a scalar-load round trip per row
  __shared__ double zs[kLrMaxRows][KP + 2];
  {
    // all loads of the staging are issued before the first LDS store (kLrMaxRows x (KP + 2) / 256 per thread)
    constexpr int kStage = (kLrMaxRows * (KP + 2) + 255) / 256;
    const int nel = (int)(r1 - r0) * (KP + 2);
    double sv[kStage];
#pragma unroll
    for (int q = 0; q < kStage; ++q) {
      const int idx = q * 256 + threadIdx.x;
      const int rr = idx / (KP + 2), j = idx % (KP + 2);
      sv[q] = 0.0;
      if (idx < nel) {
        if (j < KP) sv[q] = a.z[(r0 + rr) * a.ldk + j];
        else if (MODEL == VB_MODEL_FUNNEL) sv[q] = a.rowscal[2 * (r0 + rr) + (j - KP)];
      }
    }
#pragma unroll
    for (int q = 0; q < kStage; ++q) {
      const int idx = q * 256 + threadIdx.x;
      if (idx < nel) (&zs[0][0])[idx] = sv[q];
    }
  }
  __syncthreads();

  constexpr int RIF = 8;                                   // eps rows in flight per wave (8 KiB)
  const double* __restrict__ eps = a.eps + c0i;
  const double* __restrict__ gsrc = SRC ? a.G + c0i : nullptr;
  const bool cols_full = (cb + 1) * kLrCols <= a.ld;      // every lane's 16-B load stays inside the row
  int64_t base = r0 + wave;
  // full steps of RIF rows, software-pipelined: the loads of step s + 1 are issued before the arithmetic of step s
  auto full = [&](int64_t bs) { return cols_full && bs + (int64_t)kLrWaves * (RIF - 1) < r1; };
  auto load_step = [&](int64_t bs, lr_d2* e, lr_d2* gq) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < RIF; ++u) {
      e[u] = __builtin_nontemporal_load(reinterpret_cast<const lr_d2*>(eps + (bs + (int64_t)kLrWaves * u) * a.ld));
      if constexpr (SRC)
        gq[u] = __builtin_nontemporal_load(reinterpret_cast<const lr_d2*>(gsrc + (bs + (int64_t)kLrWaves * u) * a.ld));
    }
  };
  if (full(base)) {
    lr_d2 e[RIF], en[RIF], gq[SRC ? RIF : 1], gqn[SRC ? RIF : 1];
    load_step(base, e, gq);
    for (;;) {
      const int64_t next = base + (int64_t)kLrWaves * RIF;
      const bool more = full(next);                       // wave-uniform
      if (more) load_step(next, en, gqn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < RIF; ++u) {
        const double* zr = zs[(int)(base - r0) + kLrWaves * u];
        double zj[KP];
#pragma unroll
        for (int j = 0; j < KP; ++j) zj[j] = zr[j];
        row(e[u], SRC ? gq[SRC ? u : 0] : (lr_d2){0.0, 0.0}, zj, zr[KP], zr[KP + 1]);
      }
      base = next;
      if (!more) break;
#pragma unroll
      for (int u = 0; u < RIF; ++u) {
        e[u] = en[u];
        if constexpr (SRC) gq[u] = gqn[u];
      }
    }
  }
  // ragged tail (last rows / last column block): one row at a time, predicated
  for (; base < r1; base += kLrWaves) {
    lr_d2 ev = (lr_d2){0.0, 0.0}, gv = (lr_d2){0.0, 0.0};
    if (lane_ok) ev = __builtin_nontemporal_load(reinterpret_cast<const lr_d2*>(eps + base * a.ld));
    if (SRC && lane_ok) gv = __builtin_nontemporal_load(reinterpret_cast<const lr_d2*>(gsrc + base * a.ld));
    const double* zr = zs[(int)(base - r0)];
    double zj[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) zj[j] = zr[j];
    row(ev, gv, zj, zr[KP], zr[KP + 1]);
  }
  if (SRC && cb == 0)                                      // sum f of this row block, once per row
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) F += a.frow[r];

  // combine the 4 waves in fixed order, up to 6 fields per barrier pair; one partial per column per workgroup
  constexpr int kChunk = 6;
  __shared__ lr_d2 red[kChunk][kLrWaves][64];
  __shared__ double reds[kLrWaves][3 + KP];
  double* part = a.partials + (int64_t)rb * (2 + KP) * a.Dp;
#pragma unroll
  for (int f0 = 0; f0 < 2 + KP; f0 += kChunk) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kChunk; ++c) {
      const int f = f0 + c;
      if (f < 2 + KP) red[c][wave][lane] = f == 0 ? aG : (f == 1 ? aGE : (lr_d2){gz0[f >= 2 ? f - 2 : 0], gz1[f >= 2 ? f - 2 : 0]});
    }
    __syncthreads();
    for (int c = wave; c < kChunk; c += kLrWaves) {
      const int f = f0 + c;
      if (f < 2 + KP && c0i < a.Dp)
        *reinterpret_cast<lr_d2*>(part + (int64_t)f * a.Dp + c0i) =
            (red[c][0][lane] + red[c][1][lane]) + (red[c][2][lane] + red[c][3][lane]);
    }
  }
  {
    double s[3 + KP];
    s[0] = F, s[1] = QG, s[2] = QGE;
#pragma unroll
    for (int j = 0; j < KP; ++j) s[3 + j] = qz[j];
#pragma unroll
    for (int q = 0; q < 3 + KP; ++q) {
      const double t = lr_wave_sum(s[q]);
      if (lane == 0) reds[wave][q] = t;
    }
  }
  __syncthreads();
  if (threadIdx.x < 3 + KP)
    a.pscal[(int64_t)threadIdx.x * (a.n_rb * a.n_cb) + blockIdx.x] =
        (reds[0][threadIdx.x] + reds[1][threadIdx.x]) + (reds[2][threadIdx.x] + reds[3][threadIdx.x]);
}

// ---- finalize: row-block partials -> sum vector ------------------------------------------------------------
// sums = [F, G_c, GE_c, GZ_c[KP] (funnel coupling column), ... | G (Dp) | GE (Dp) | GZ_0 (Dp) | ...]
template <int KP>
__global__ void __launch_bounds__(256) lr_finalize_kernel(const LrArgs a, int funnel) {
  __shared__ double sh[4];
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)(2 + KP) * a.Dp;
  if (idx < total) {
    const int64_t stride = (int64_t)(2 + KP) * a.Dp;
    double s = 0.0;
    for (int rb0 = 0; rb0 < a.n_rb; rb0 += 32) {      // 32 loads in flight, summed in row-block order
      double v[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) v[u] = rb0 + u < a.n_rb ? a.partials[(int64_t)(rb0 + u) * stride + idx] : 0.0;
#pragma unroll
      for (int u = 0; u < 32; ++u) s += v[u];
    }
    a.sums[kLrScal + idx] = s;
  }
  if (blockIdx.x == 0) {
    // scalars: wave q sums slot q, q + 4, ... over the workgroups (accumulate kernel) and prep blocks
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_wg = a.n_rb * a.n_cb;
    for (int q = wave; q < 3 + KP; q += 4) {
      double s = 0.0, p = 0.0;
      for (int i = lane; i < n_wg; i += 64) s += a.pscal[(int64_t)q * n_wg + i];
      for (int i = lane; i < a.n_prep; i += 64) p += a.prepscal[(int64_t)q * a.n_prep + i];
      s = lr_wave_sum(s);
      p = lr_wave_sum(p);
      if (lane == 0) {
        if (!funnel) {
          if (q == 0) a.sums[0] = s;
          else a.sums[q] = 0.0;
        } else if (q == 0) {
          sh[0] = p;                       // FK; F = FK - QG / 2 is formed below
        } else {
          a.sums[q] = s + p;               // G_c, GE_c, GZ_c[j] of the coupling column
          if (q == 1) sh[1] = s;           // QG
        }
      }
    }
    __syncthreads();
    if (funnel && threadIdx.x == 0) a.sums[0] = sh[0] - 0.5 * sh[1];
    if (threadIdx.x >= 3 + KP && threadIdx.x < kLrScal) a.sums[threadIdx.x] = 0.0;
  }
}

// ---- epilogue: sum vector + entropy gradient -> (value, grad) -------------------------------------------------
template <int KP>
__global__ void __launch_bounds__(256) lr_epilogue_kernel(const LrArgs a, double n_total, double model_c0,
                                                          int coupling, double* __restrict__ out) {
  __shared__ double Minv[KP][KP + 1];
  __shared__ double sh[4];
  const int d = a.d, k = a.k, t = threadIdx.x;
  const double* mu = a.theta_dev;
  const double* ls = a.theta_dev + d;
  const double* B = a.theta_dev + 2 * (int64_t)d;
  if (t < KP * KP) Minv[t / KP][t % KP] = a.minv[t];
  const double logdet_m = a.minv[KP * KP];
  __syncthreads();
  const double invN = 1.0 / n_total;
  const double* S = a.sums;
  {
    const int i = blockIdx.x * 256 + t;       // one row of the gradient per thread
    if (i < d) {
    const double lsi = ls[i], iv = exp(-2.0 * lsi);
    double bi[KP], u[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) bi[j] = j < k ? B[(int64_t)i * k + j] : 0.0;
    double quad = 0.0;
#pragma unroll
    for (int j = 0; j < KP; ++j) {
      double s = 0.0;
#pragma unroll
      for (int p = 0; p < KP; ++p) s = fma(Minv[j][p], bi[p], s);
      u[j] = s;
      quad = fma(bi[j], s, quad);
    }
    const bool cpl = i == coupling;
    const double G = cpl ? S[1] : S[kLrScal + i];
    const double GE = cpl ? S[2] : S[kLrScal + a.Dp + i];
    out[1 + i] = -G * invN;
    out[1 + d + i] = -(GE * exp(lsi) * invN) - (1.0 - quad * iv);
    for (int j = 0; j < k; ++j) {
      const double GZ = cpl ? S[3 + j] : S[kLrScal + (int64_t)(2 + j) * a.Dp + i];
      out[1 + 2 * (int64_t)d + (int64_t)i * k + j] = -GZ * invN - u[j] * iv;
    }
    }
  }
  (void)mu;
  // value = -(mean f + H), by the first workgroup
  if (blockIdx.x == 0) {
    double sum_ls = 0.0;
    for (int i = t; i < d; i += 256) sum_ls += ls[i];
    sum_ls = lr_wave_sum(sum_ls);
    if ((t & 63) == 0) sh[t >> 6] = sum_ls;
    __syncthreads();
    if (t == 0) {
      const double H = 0.5 * d * (kLog2PiLr + 1.0) + ((sh[0] + sh[1]) + (sh[2] + sh[3])) + 0.5 * logdet_m;
      out[0] = -(S[0] * invN + model_c0 + H);
    }
  }
}

template <int KP>
static int lr_run(vb_ctx* ctx, const NoiseSlot& ns, const NoiseSlot& nz, int64_t n, int64_t d, int k, int64_t n_total,
                  const double* theta_src, double* out) {
  const ModelDev& m = ctx->model;
  hipStream_t st = ctx->stream;
  const bool funnel = m.id == VB_MODEL_FUNNEL;
  LrArgs a;
  a.eps = (const double*)ns.buf.ptr, a.z = (const double*)nz.buf.ptr;
  a.ld = ns.ld, a.ldk = nz.ld, a.n = n, a.d = (int)d, a.k = k;
  a.G = nullptr, a.frow = nullptr;
  a.n_cb = (int)((d + kLrCols - 1) / kLrCols);
  a.Dp = a.n_cb * kLrCols;
  const char* wg_env = getenv("VB_LR_WG_PER_CU");
  int n_rb_target = (wg_env && *wg_env ? atoi(wg_env) : 2) * ctx->prop.multiProcessorCount / a.n_cb;
  if (n_rb_target < 8) n_rb_target = 8;
  int rows_per_wg = (int)((n + n_rb_target - 1) / n_rb_target);
  if (rows_per_wg > kLrMaxRows) rows_per_wg = kLrMaxRows;
  a.rows_per_wg = (int)round_up(rows_per_wg < kLrWaves ? kLrWaves : rows_per_wg, kLrWaves);
  a.n_rb = (int)((n + a.rows_per_wg - 1) / a.rows_per_wg);
  const int64_t prep_items = funnel ? (n > a.Dp ? n : a.Dp) : a.Dp;
  a.n_prep = (int)((prep_items + 255) / 256);
  int64_t off = 0;
  auto carve = [&off](int64_t doubles) {
    const int64_t o = off;
    off += round_up(doubles, 16);
    return o;
  };
  const int64_t o_theta = carve(2 * d + d * k), o_colp = carve(3 * (int64_t)a.Dp), o_bp = carve((int64_t)(a.Dp + 2) * KP),
                o_rows = carve(funnel ? 2 * n : 0), o_prep = carve((int64_t)(3 + KP) * a.n_prep),
                o_part = carve((int64_t)a.n_rb * (2 + KP) * a.Dp), o_pscal = carve((int64_t)(3 + KP) * a.n_rb * a.n_cb),
                o_sums = carve(kLrScal + (int64_t)(2 + KP) * a.Dp);
  a.n_mblk = (int)((d + 255) / 256);
  const int64_t o_mpart = carve((int64_t)a.n_mblk * KP * KP), o_minv = carve(KP * KP + 1);
  VB_TRY(ensure(ctx, ctx->lr_work, (size_t)off * sizeof(double)));
  double* base = (double*)ctx->lr_work.ptr;
  a.theta_dev = base + o_theta, a.colp = base + o_colp, a.Bp = base + o_bp, a.rowscal = base + o_rows;
  a.prepscal = base + o_prep, a.partials = base + o_part, a.pscal = base + o_pscal, a.sums = base + o_sums;
  a.mpart = base + o_mpart, a.minv = base + o_minv;
  hipLaunchKernelGGL((lr_prep_kernel<KP>), dim3((unsigned)a.n_prep), dim3(256), 0, st, theta_src, a, m);
  VB_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL((lr_capacitance_kernel<KP>), dim3(1), dim3(KP * KP), 0, st, a);
  VB_HIP(ctx, hipGetLastError());
  const dim3 grid((unsigned)(a.n_rb * a.n_cb));
  if (m.id == VB_MODEL_SOURCE) {
    // samples as a matrix -> the user's row kernel -> (f, G); the streaming pass then loads G beside the noise
    const int64_t o_x = 0, o_g = round_up(n * a.ld, 16), o_f = o_g + round_up(n * a.ld, 16);
    VB_TRY(ensure(ctx, ctx->lg_work, (size_t)(o_f + round_up(n, 16)) * sizeof(double)));
    double* wb = (double*)ctx->lg_work.ptr;
    double *X = wb + o_x, *G = wb + o_g, *frow = wb + o_f;
    VB_HIP(ctx, hipMemsetAsync(G, 0, (size_t)n * a.ld * sizeof(double), st));     // pad columns are streamed too
    hipLaunchKernelGGL((lr_sample_kernel<KP>), dim3((unsigned)n, (unsigned)((d + 255) / 256)), dim3(256), 0, st, a, X);
    VB_HIP(ctx, hipGetLastError());
    VB_TRY(user_rows_enqueue(ctx, st, X, a.ld, n, (int)d, G, a.ld, frow));
    a.G = G, a.frow = frow;
    hipLaunchKernelGGL((lr_accum_kernel<VB_MODEL_SOURCE, KP>), grid, dim3(256), 0, st, a);
  } else if (funnel)
    hipLaunchKernelGGL((lr_accum_kernel<VB_MODEL_FUNNEL, KP>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((lr_accum_kernel<VB_MODEL_GAUSS_DIAG, KP>), grid, dim3(256), 0, st, a);
  VB_HIP(ctx, hipGetLastError());
  const int64_t fin_items = (int64_t)(2 + KP) * a.Dp;
  hipLaunchKernelGGL((lr_finalize_kernel<KP>), dim3((unsigned)((fin_items + 255) / 256)), dim3(256), 0, st, a,
                     funnel ? 1 : 0);
  VB_HIP(ctx, hipGetLastError());
  if (ctx->comm) VB_TRY(comm_allreduce_sum(ctx, st, a.sums, (size_t)(kLrScal + fin_items)));
  hipLaunchKernelGGL((lr_epilogue_kernel<KP>), dim3((unsigned)a.n_mblk), dim3(256), 0, st, a, (double)n_total, m.c0,
                     funnel ? m.k : -1, out);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

int lr_elbo_grad_enqueue(vb_ctx* ctx, const NoiseSlot& ns, const NoiseSlot& nz, int64_t n, int64_t d, int64_t k,
                         int64_t n_total, const double* theta_src, double* out) {
  const ModelDev& m = ctx->model;
  if (m.id != VB_MODEL_GAUSS_DIAG && m.id != VB_MODEL_FUNNEL && m.id != VB_MODEL_SOURCE)
    return fail(ctx, VB_ERR_UNSUPPORTED, "low-rank path supports the gauss_diag, funnel and source models (model id %d "
                "bound)", m.id);
  if (m.dim != d) return fail(ctx, VB_ERR_INVALID, "model dimension %d != family dimension %lld", m.dim, (long long)d);
  if (k < 1 || k > 16) return fail(ctx, VB_ERR_UNSUPPORTED, "low-rank path supports 1 <= k <= 16 (got %lld)", (long long)k);
  if (n <= 0 || n > ns.n || d != ns.d || n > nz.n || k != nz.d)
    return fail(ctx, VB_ERR_INVALID, "noise slots must hold n x d and n x k matrices");
  if (n_total < n) return fail(ctx, VB_ERR_INVALID, "n_total must be >= n");
  if (k <= 4) return lr_run<4>(ctx, ns, nz, n, d, (int)k, n_total, theta_src, out);
  if (k <= 8) return lr_run<8>(ctx, ns, nz, n, d, (int)k, n_total, theta_src, out);
  return lr_run<16>(ctx, ns, nz, n, d, (int)k, n_total, theta_src, out);
}

}  // namespace vb
