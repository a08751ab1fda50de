// Full-rank Gaussian family: ExclusiveKL value and gradient on gfx950 fp64 matrix cores.
//
// The reference has no dense Gaussian family; this follows its ApproximationFamily
// contract (viabel/approximations.py:26-182) with MultivariateT's parameter layout
// (approximations.py:315-319): theta = [mu (D) | free Cholesky of Sigma (D(D+1)/2)], L = chol with
// exp on the diagonal, z_n = mu + L eps_n.  The estimator is objectives.py:154-164 (entropy form):
//   value  = -(mean_n f(z_n) + 1/2 D (1 + log 2 pi) + sum_i log L_ii)
//   d/dmu  = -mean_n g_n                       g_n = grad f(z_n)
//   d/dL   = -tril(mean_n g_n eps_n')          free diagonal: dL_ii * L_ii - 1
//
// Pipeline (one HIP stream):
//   fr_unpack          theta -> mu, L^T (dense, zeros below the diagonal of L^T)
//   GEMM 1 (MFMA)      Z = E L^T + mu      [N x D x D, triangular k-range]  fused model epilogue
//   model              gauss_diag: G in the GEMM-1 epilogue; funnel: row kernel Z -> G;
//                      gauss_full: GEMM 2 (MFMA)  G = -(Z - m) P
//   fr_colsum          column sums of G (-> d/dmu) and sum_n f(z_n), per 128-row block
//   GEMM 3 (MFMA)      C = G^T E           [D x D x N, lower-triangular tiles, split-K]
//   fr_reduce          fixed-order sum of the split-K slabs / row-block partials -> sum vector
//   [RCCL all-reduce of the sum vector when the Monte-Carlo axis is sharded]
//   fr_epilogue        sum vector -> (value, grad) in the flat free-Cholesky layout
//
// Bound: fp64 MFMA (4 N D^2 flop dense convention vs N D 8 bytes: AI ~ D/2 flop/B >> ridge).
#include "vb_gemm_f64.h"
#include "vb_fit.h"

namespace vb {

constexpr double kLog2PiFr = 1.8378770664093454835606594728112;

__device__ __forceinline__ double fr_wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

// block-wide sum (256 threads), total returned to every thread
__device__ __forceinline__ double fr_block_sum(double x, double* sh) {
  x = fr_wave_sum(x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = x;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ---- unpack ---------------------------------------------------------------------------------------
// Lt[k][j] = L[j][k] (so Lt is upper triangular), row stride ldl; also copies mu.  A transpose of the packed
// triangle: row j of L is contiguous in theta (offset d + j (j + 1) / 2), column j of Lt is strided.  32 x 32 tiles go
// through LDS so that both the reads (along k within a packed row) and the writes (along j within a row of Lt) are
// contiguous 256-B runs; a thread-per-element gather took 10.6 us at D = 1024, this takes a third of it.  grid =
// (tiles over k, tiles over j); block (32, 8).
// copy != nullptr: theta is read ONCE -- it may be mapped host memory -- and every entry is also written to `copy`
// (the device-resident parameter the triangular inverse and the fit kernels read)
__global__ void __launch_bounds__(256) fr_unpack_kernel(const double* __restrict__ theta, int d,
                                                        int64_t ldl, double* __restrict__ Lt,
                                                        double* __restrict__ mu, double* __restrict__ copy = nullptr) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int k0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
  if (blockIdx.y == 0) {
    const int i = k0 + (int)threadIdx.x;
    if (threadIdx.x < 32 && i < d) {
      const double v = theta[i];
      mu[i] = v;
      if (copy) copy[i] = v;
    }
  }
  if (k0 > j0 + 31) {            // the whole tile lies below the diagonal of Lt (k > j): zeros
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const int k = k0 + r, j = j0 + tx;
      if (k < d && j < d) Lt[(int64_t)k * ldl + j] = 0.0;
    }
    return;
  }
#pragma unroll
  for (int r = ty; r < 32; r += 8) {       // read L[j0 + r][k0 + tx]
    const int j = j0 + r, k = k0 + tx;
    double v = 0.0;
    if (j < d && k <= j) {
      v = theta[d + (int64_t)j * (j + 1) / 2 + k];
      if (copy) copy[d + (int64_t)j * (j + 1) / 2 + k] = v;
      if (k == j) v = exp(v);
    }
    tile[r][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {       // write Lt[k0 + r][j0 + tx]
    const int k = k0 + r, j = j0 + tx;
    if (k < d && j < d) Lt[(int64_t)k * ldl + j] = tile[tx][r];
  }
}

// ---- GEMM epilogues ---------------------------------------------------------------------------------
struct EpiStoreZ {          // Z = acc [* rs_n] + mu - shift   (shift = 0, or the target mean for gauss_full)
  double* Z;
  int64_t ldz;
  const double* mu;
  const double* shift;      // may be nullptr
  const double* rs;         // per-row scale (multivariate t: 1 / s_n), may be nullptr
  __device__ void operator()(int, int row, int col, double acc) const {
    if (rs) acc *= rs[row];
    double z = acc + mu[col];
    if (shift) z -= shift[col];
    Z[(int64_t)row * ldz + col] = z;
  }
  __device__ d2v pair(int, int row, int col, double a0, double a1) const {
    if (rs) a0 *= rs[row], a1 *= rs[row];
    const d2v m = *reinterpret_cast<const d2v*>(mu + col);
    d2v z = (d2v){a0 + m.x, a1 + m.y};
    if (shift) {
      const d2v sh = *reinterpret_cast<const d2v*>(shift + col);
      z.x -= sh.x, z.y -= sh.y;
    }
    *reinterpret_cast<d2v*>(Z + (int64_t)row * ldz + col) = z;
    return z;
  }
};

// The sampling product with the k range of its HEAVY column blocks cut in two (round 6; VB_FR_HSPLIT=1):
// split 0 (k < k_half, the whole product for the columns whose k range ends there) stores Z as EpiStoreZ does; split 1
// (k >= k_half: only the column blocks to the right of k_half have any) stores its partial products into `slab`, and
// fr_zfix_kernel adds them to Z's right half.  Deterministic (fixed order: split 0 + split 1).
struct EpiStoreZHeavy {
  double* Z;
  int64_t ldz;
  const double* mu;
  const double* shift;      // may be nullptr
  double* slab;
  int k_half;
  __device__ void operator()(int split, int row, int col, double acc) const {
    if (split == 0) {
      double z = acc + mu[col];
      if (shift) z -= shift[col];
      Z[(int64_t)row * ldz + col] = z;
    } else if (col >= k_half) {
      slab[(int64_t)row * ldz + col] = acc;
    }
  }
  __device__ d2v pair(int split, int row, int col, double a0, double a1) const {
    if (split == 0) {
      const d2v m = *reinterpret_cast<const d2v*>(mu + col);
      d2v z = (d2v){a0 + m.x, a1 + m.y};
      if (shift) {
        const d2v sh = *reinterpret_cast<const d2v*>(shift + col);
        z.x -= sh.x, z.y -= sh.y;
      }
      *reinterpret_cast<d2v*>(Z + (int64_t)row * ldz + col) = z;
      return z;
    }
    if (col >= k_half) *reinterpret_cast<d2v*>(slab + (int64_t)row * ldz + col) = (d2v){a0, a1};
    return (d2v){a0, a1};
  }
};

__global__ void __launch_bounds__(256) fr_zfix_kernel(const double* __restrict__ slab, int64_t n, int d, int64_t ldz, int k_half,
                                                      double* __restrict__ Z) {
  const int half_pairs = (d - k_half + 1) / 2;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * half_pairs) return;
  const int64_t row = t / half_pairs;
  const int c = k_half + 2 * (int)(t % half_pairs);
  const int64_t idx = row * ldz + c;
  d2v z = *reinterpret_cast<const d2v*>(Z + idx);
  const d2v p = *reinterpret_cast<const d2v*>(slab + idx);
  z.x += p.x;
  if (c + 1 < d) z.y += p.y;
  *reinterpret_cast<d2v*>(Z + idx) = z;
}

// regression targets (VB_MODEL_LOGISTIC with a VB_GLM_* likelihood): eta = Z X' -> R = dloglik / deta and the
// log-likelihood sum, then G = R X - Z / prior_sd^2 by glm_grad_enqueue (the two GEMMs of vb_logistic.h behind the
// sampling GEMM)
struct EpiGlm {
  double* R;
  int64_t ldr;
  const double* y;
  double* part;
  int link;
  double aux;
  __device__ double operator()(int, int row, int col, double eta) const {
    double dl;
    const double ll = glm_term(link, aux, y[col], eta, &dl);
    R[(int64_t)row * ldr + col] = dl;
    return ll;
  }
};

struct EpiGaussDiag {       // G = -(z - m) / sd^2  straight from the GEMM-1 accumulators
  double* G;
  int64_t ldz;
  const double* mu;
  const double* mean;
  const double* ivar;
  const double* rs;         // per-row scale, may be nullptr
  __device__ void operator()(int, int row, int col, double acc) const {
    if (rs) acc *= rs[row];
    const double dz = acc + mu[col] - mean[col];
    G[(int64_t)row * ldz + col] = -dz * ivar[col];
  }
};

// ... and, per workgroup, the sum of f = -1/2 (z - m)^2 / sd^2 over the tile (the diagonal Gaussian's log density of the
// samples this tile has just formed): with the column sums of G out of the gradient product no pass over G is left
struct EpiGaussDiagF {
  double* G;
  int64_t ldz;
  const double* mu;
  const double* mean;
  const double* ivar;
  double* part;
  __device__ double operator()(int, int row, int col, double acc) const {
    const double dz = acc + mu[col] - mean[col], iv = ivar[col];
    G[(int64_t)row * ldz + col] = -dz * iv;
    return -0.5 * dz * dz * iv;
  }
  __device__ d2v pair(int, int row, int col, double a0, double a1) const {
    const double d0 = a0 + mu[col] - mean[col], d1 = a1 + mu[col + 1] - mean[col + 1];
    const double i0 = ivar[col], i1 = ivar[col + 1];
    *reinterpret_cast<d2v*>(G + (int64_t)row * ldz + col) = (d2v){-d0 * i0, -d1 * i1};
    return (d2v){-0.5 * d0 * d0 * i0, -0.5 * d1 * d1 * i1};
  }
};

struct EpiNegate {          // G = -acc   (gauss_full: G = -(Z - m) P)
  double* G;
  int64_t ldz;
  __device__ void operator()(int, int row, int col, double acc) const {
    G[(int64_t)row * ldz + col] = -acc;
  }
  __device__ d2v pair(int, int row, int col, double a0, double a1) const {
    const d2v v = (d2v){-a0, -a1};
    *reinterpret_cast<d2v*>(G + (int64_t)row * ldz + col) = v;
    return v;
  }
};

struct EpiAccumulate {      // G += acc   (path derivative: the rows of G take the score's L^-T eps)
  double* G;
  int64_t ldz;
  __device__ void operator()(int, int row, int col, double acc) const { G[(int64_t)row * ldz + col] += acc; }
  __device__ d2v pair(int, int row, int col, double a0, double a1) const {
    d2v* p = reinterpret_cast<d2v*>(G + (int64_t)row * ldz + col);
    const d2v v = *p + (d2v){a0, a1};
    *p = v;
    return v;
  }
};

// G = -acc and, per workgroup, the sum of f = 1/2 (z - m)' g over the tile (gauss_full: log p(z) = c0 - 1/2 (z - m)' P
// (z - m) = c0 + 1/2 (z - m)' g): Zc holds z - m, the very rows this workgroup has just multiplied (L2-resident), read
// back with the store's own 16-byte pattern.  The model's log density of the materialised samples, not an identity of
// the variational family.
struct EpiNegateF {
  double* G;
  int64_t ldz;
  const double* Zc;
  double* part;
  __device__ double operator()(int, int row, int col, double acc) const {
    G[(int64_t)row * ldz + col] = -acc;
    return -0.5 * acc * Zc[(int64_t)row * ldz + col];
  }
  __device__ d2v pair(int, int row, int col, double a0, double a1) const {
    const d2v z = *reinterpret_cast<const d2v*>(Zc + (int64_t)row * ldz + col);
    *reinterpret_cast<d2v*>(G + (int64_t)row * ldz + col) = (d2v){-a0, -a1};
    return (d2v){-0.5 * a0 * z.x, -0.5 * a1 * z.y};
  }
};

// C_split[i][j] = acc, and the column sums of G per split through the kernel's EpiColsum hook (the waves of the diagonal
// tiles that have nothing to multiply add up the A tiles in LDS)
struct EpiSplitSlabCs {
  double* C;
  int64_t ldc, slab;
  double* colsum;
  int64_t colsum_ld;
  __device__ void operator()(int split, int row, int col, double acc) const {
    C[split * slab + (int64_t)row * ldc + col] = acc;
  }
  __device__ d2v pair(int split, int row, int col, double a0, double a1) const {
    const d2v v = (d2v){a0, a1};
    *reinterpret_cast<d2v*>(C + split * slab + (int64_t)row * ldc + col) = v;
    return v;
  }
};

struct EpiSplitSlab {       // C_split[i][j] = acc
  double* C;
  int64_t ldc, slab;
  __device__ void operator()(int split, int row, int col, double acc) const {
    C[split * slab + (int64_t)row * ldc + col] = acc;
  }
  __device__ d2v pair(int split, int row, int col, double a0, double a1) const {
    const d2v v = (d2v){a0, a1};
    *reinterpret_cast<d2v*>(C + split * slab + (int64_t)row * ldc + col) = v;
    return v;
  }
};

}  // namespace vb
#include "vb_fullrank_fused.h"
namespace vb {

// ---- funnel: row kernel Z -> G, f ---------------------------------------------------------------
// one wave per row; G[n][j] = -z_j w (j != k), G[n][k] = -v/tau^2 - (D-1) + w sum_{j != k} z_j^2
// roww != nullptr: the rows of G leave scaled by the sample's weight (AlphaDivergence: no separate pass over G)
__global__ void __launch_bounds__(256) fr_funnel_kernel(const double* __restrict__ Z, double* __restrict__ G,
                                                        int64_t ldz, int64_t n, int d, ModelDev m,
                                                        double* __restrict__ fpart, const double* __restrict__ roww) {
  __shared__ double sh[4];
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  double f = 0.0;
  if (row < n) {
    const double* z = Z + row * ldz;
    double* g = G + row * ldz;
    const double v = z[m.k];
    const double w = exp(-2.0 * v);
    const double rw = roww ? roww[row] : 1.0;
    double ss = 0.0;
    for (int c = lane; c < d; c += 64) {
      if (c == m.k) continue;
      const double zc = z[c];
      const double gc = -zc * w;
      g[c] = roww ? gc * rw : gc;
      ss = fma(zc, zc, ss);
    }
    ss = fr_wave_sum(ss);
    if (lane == 0) {
      const double it2 = 1.0 / (m.tau * m.tau), dm1 = (double)(d - 1);
      const double gk = fma(-v, it2, -dm1) + w * ss;
      g[m.k] = roww ? gk * rw : gk;
      f = v * fma(-0.5 * v, it2, -dm1) - 0.5 * w * ss;
    }
  }
  f = fr_block_sum(f, sh);
  if (threadIdx.x == 0) fpart[blockIdx.x] = f;
}

// ---- column sums of G and sum of f per 128-row block ------------------------------------------------
// grid (ceil(D / 128), ceil(N / 128)); thread (c = t & 63, q = t >> 6) sums rows r0 + q, q + 4, ... of
// the column pair 2c, 2c + 1 (16-B loads, 16 rows in flight); the 4 row groups are combined through LDS in
// fixed order.  Rows are padded to 16 doubles and pad columns of G / Zc are never written with non-finite
// values by the producers, but they are masked anyway.
// fmode 0: no f here, 1: gauss_diag f = -1/2 g^2 / ivar, 2: gauss_full f = 1/2 zc g,
// 3: regression prior f = -1/2 scal zc^2 (the likelihood part comes from the GEMM epilogue)
typedef double fr_d2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) fr_colsum_kernel(const double* __restrict__ G,
                                                        const double* __restrict__ Zc, int64_t ldz,
                                                        int64_t n, int d, int fmode,
                                                        const double* __restrict__ ivar,
                                                        double* __restrict__ colpart,
                                                        double* __restrict__ fpart, double scal = 0.0,
                                                        const double* __restrict__ roww = nullptr,
                                                        int square = 0, double* __restrict__ Gscaled = nullptr,
                                                        const double* __restrict__ rs_out = nullptr) {
  __shared__ double sh[4];
  __shared__ fr_d2 cs[4][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int col = blockIdx.x * 128 + 2 * c;
  const int64_t r0 = (int64_t)blockIdx.y * 128;
  const int64_t r1 = r0 + 128 < n ? r0 + 128 : n;
  const bool ok0 = col < d, ok1 = col + 1 < d;
  fr_d2 s = (fr_d2){0.0, 0.0};
  double f = 0.0;
  if (ok0) {
    fr_d2 hiv = (fr_d2){0.0, 0.0};
    if (fmode == 1) hiv = (fr_d2){-0.5 / ivar[col], ok1 ? -0.5 / ivar[col + 1] : 0.0};
    for (int64_t rb = r0 + q; rb < r1; rb += 64) {      // sixteen rows in flight, summed in row order
      // loads first, all of them, from addresses clamped into the block (a load inside `if (r < r1)` next to the
      // write-back below is waited for before the next one is issued: 21 us instead of 12 at 16 384 x 256)
      fr_d2 g[16], z[16];
      double rw[16], rs[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t r = rb + 4 * i < r1 ? rb + 4 * i : r1 - 1;
        g[i] = *reinterpret_cast<const fr_d2*>(G + r * ldz + col);
        z[i] = fmode >= 2 ? *reinterpret_cast<const fr_d2*>(Zc + r * ldz + col) : (fr_d2){0.0, 0.0};
        rw[i] = roww ? roww[r] : 1.0;
        rs[i] = Gscaled ? rs_out[r] : 1.0;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t r = rb + 4 * i;
        const bool in = r < r1;
        if (roww) g[i] *= rw[i];
        if (square) g[i] *= g[i];
        if (!in) g[i] = (fr_d2){0.0, 0.0}, z[i] = (fr_d2){0.0, 0.0};
        if (!ok1) g[i].y = 0.0, z[i].y = 0.0;
        // the t family's chain rule wants the rows of G scaled by 1 / s_n AFTER these (unscaled) sums: written back
        // from here instead of by a pass of its own
        if (Gscaled && in) *reinterpret_cast<fr_d2*>(Gscaled + r * ldz + col) = g[i] * rs[i];
        s += g[i];
        if (fmode == 1) f = fma(hiv.x * g[i].x, g[i].x, fma(hiv.y * g[i].y, g[i].y, f));
        if (fmode == 2) f = fma(0.5 * z[i].x, g[i].x, fma(0.5 * z[i].y, g[i].y, f));
        if (fmode == 3) f = fma(-0.5 * scal * z[i].x, z[i].x, fma(-0.5 * scal * z[i].y, z[i].y, f));
      }
    }
  }
  cs[q][c] = s;
  __syncthreads();
  if (q == 0 && ok0) {
    const fr_d2 tot = (cs[0][c] + cs[1][c]) + (cs[2][c] + cs[3][c]);
    colpart[(int64_t)blockIdx.y * ldz + col] = tot.x;
    if (ok1) colpart[(int64_t)blockIdx.y * ldz + col + 1] = tot.y;
  }
  f = fr_block_sum(f, sh);
  if (threadIdx.x == 0) fpart[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = f;
}

// ---- reduce: split-K slabs, row-block partials -> sum vector ------------------------------------------
// sum vector layout: [F | colsum (ldz) | C (d x ldl)], F at index 0, colsum from 16, C from 16 + ldz

// One thread per pair of adjacent C entries (16-B loads, CHUNK split slabs in flight, summed in slab order); entries above
// the diagonal are not computed by the GEMM and are written as zero.
// Round 6: the three reductions have workgroups of their own -- blocks [0, nb_c) the C pairs, the next ceil(ldz / 256) the
// column sums, the last one the scalars -- where block 0 used to do all three one after the other (at D = 256 with 64 slabs
// and 128 row blocks: ten dependent round trips in one workgroup, 10.6 us for 21 MB); CHUNK = 64 for launches that leave
// the SIMDs a wave or two each anyway (the registers cost nothing there: every slab's load in flight at once).  The sums are
// formed in the same order as before: the same bits.
template <int CHUNK>
__global__ void __launch_bounds__(256) fr_reduce_kernel(const double* __restrict__ Cpart, int splits,
                                                        int64_t slab, int d, int64_t ldl,
                                                        const double* __restrict__ colpart, int n_rb,
                                                        int64_t ldz, const double* __restrict__ fpart,
                                                        int n_fpart, FrSums S, int full,
                                                        const double* __restrict__ wpart, int nb_c) {
  __shared__ double sh[4];
  const int nb_col = (int)((ldz + 255) / 256);
  if ((int)blockIdx.x < nb_c) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t idx = 2 * tid;
    const int64_t nC = (int64_t)d * ldl;
    if (idx >= nC) return;
    const int i = (int)(idx / ldl), j = (int)(idx % ldl);
    fr_d2 s = (fr_d2){0.0, 0.0};
    if (full == 1 || j <= i) {
      for (int k0 = 0; k0 < splits; k0 += CHUNK) {
        fr_d2 v[CHUNK];
#pragma unroll
        for (int u = 0; u < CHUNK; ++u)      // (clamped, not predicated: no load behind a branch; the surplus is not added)
          v[u] = *reinterpret_cast<const fr_d2*>(Cpart + (k0 + u < splits ? k0 + u : splits - 1) * slab + idx);
#pragma unroll
        for (int u = 0; u < CHUNK; ++u)
          if (k0 + u < splits) s += v[u];
      }
      if (full != 1 && j + 1 > i) s.y = 0.0;
    }
    if (full == 2) {
      // mirrored: the symmetric matrix given by its lower triangle, both halves written here (entry (i, j <= i) and its
      // image (j, i) by the thread that owns the lower one; the upper entries' own threads write nothing) -- the caller
      // multiplies by it next and needs no symmetrising pass
      double* C = S.sums + S.off_c;
      if (j < d && j <= i) {
        C[idx] = s.x;
        if (j < i) C[(int64_t)j * ldl + i] = s.x;
      } else if (j >= d) {
        C[idx] = 0.0;
      }
      if (j + 1 < d && j + 1 <= i) {
        C[idx + 1] = s.y;
        if (j + 1 < i) C[(int64_t)(j + 1) * ldl + i] = s.y;
      } else if (j + 1 >= d) {
        C[idx + 1] = 0.0;
      }
    } else {
      *reinterpret_cast<fr_d2*>(S.sums + S.off_c + idx) = s;
    }
    return;
  }
  if ((int)blockIdx.x < nb_c + nb_col) {
    const int64_t tid = (int64_t)((int)blockIdx.x - nb_c) * 256 + threadIdx.x;
    if (tid >= ldz) return;
    double s = 0.0;
    if (tid < d) {
      for (int rb0 = 0; rb0 < n_rb; rb0 += 32) {      // 32 loads in flight (16 until round 5), summed in row-block order
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = rb0 + u < n_rb ? colpart[(int64_t)(rb0 + u) * ldz + tid] : 0.0;
#pragma unroll
        for (int u = 0; u < 32; ++u) s += v[u];
      }
    }
    S.sums[S.off_col + tid] = s;
    return;
  }
  double f = 0.0;
  for (int e = threadIdx.x; e < n_fpart; e += 256) f += fpart[e];
  f = fr_block_sum(f, sh);
  if (threadIdx.x == 0) S.sums[0] = f;
  if (wpart) {      // two more scalars given as per-row-block partials (wpart[q n_rb + rb]) -> sums[1], sums[2]
    for (int q = 0; q < 2; ++q) {
      double t = 0.0;
      for (int e = threadIdx.x; e < n_rb; e += 256) t += wpart[(int64_t)q * n_rb + e];
      t = fr_block_sum(t, sh);
      if (threadIdx.x == 0) S.sums[1 + q] = t;
    }
  }
}

// (both launch sites: the C pairs' blocks, the column sums' blocks, one block of scalars)
static void fr_reduce_launch(vb_ctx* ctx, hipStream_t st, const double* Cpart, int splits, int64_t slab, int d, int64_t ldl,
                             const double* colpart, int n_rb, int64_t ldz, const double* fpart, int n_fpart, FrSums S, int full,
                             const double* wpart) {
  const int nb_c = (int)((slab / 2 + 255) / 256), nb_col = (int)((ldz + 255) / 256);
  const dim3 grid((unsigned)(nb_c + nb_col + 1));
  if (splits > 16 && nb_c <= 2 * ctx->prop.multiProcessorCount)
    hipLaunchKernelGGL(fr_reduce_kernel<64>, grid, dim3(256), 0, st, Cpart, splits, slab, d, ldl, colpart, n_rb, ldz, fpart, n_fpart,
                       S, full, wpart, nb_c);
  else
    hipLaunchKernelGGL(fr_reduce_kernel<16>, grid, dim3(256), 0, st, Cpart, splits, slab, d, ldl, colpart, n_rb, ldz, fpart, n_fpart,
                       S, full, wpart, nb_c);
}

#ifdef VB_DBG_IDLE
// experiment (tools/build_variant.sh idle -DVB_DBG_IDLE): one wave that sleeps for `ticks` of the 100 MHz clock
__global__ void dbg_idle_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
#endif

// ---- full-rank Gaussian: split reduction straight into the flat (paragami) layout -----------------------
// Same pair-per-thread reduction as fr_reduce_kernel (same summation order), but entry (i, j <= i) lands at the
// packed position i (i + 1) / 2 + j -- the order of the free-Cholesky block of theta -- so a sharded job
// all-reduces D (D + 1) / 2 doubles instead of D x ldl, and entries above the diagonal are never written.
// FUSE (no communicator): the O(P) epilogue arithmetic is applied on the spot and `out` = [value | grad] is
// written directly; otherwise the raw sums go to S for the all-reduce and fr_epilogue_packed_kernel finishes.
template <bool FUSE>
__global__ void __launch_bounds__(256) fr_reduce_packed_kernel(
    const double* __restrict__ Cpart, int splits, int64_t slab, int d, int64_t ldl,
    const double* __restrict__ colpart, int n_rb, int64_t ldz, const double* __restrict__ fpart, int n_fpart,
    FrSums S, const double* __restrict__ theta, double n_local_w, double n_total, double c0,
    double* __restrict__ out, int pd, FrWeighted wm) {
  const double ent = pd ? 0.0 : 1.0;      // the entropy's -1 on the free diagonal (absent with the path derivative)
  // weighted mode (AlphaDivergence, objectives.py:458-460): the rows of G carried the weights s_n, the result is
  // scale * [sum s g | tril(sum s g eps') with the free diagonal x L_ii + sum s], the value comes from wm.value
  const bool weighted = wm.scale != 0.0;
  const double wsum = weighted ? wm.wsum[0] : 0.0;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t idx = 2 * tid;
  const int64_t nC = (int64_t)d * ldl;
  const double invN = 1.0 / n_total;
  if (idx < nC) {
    const int i = (int)(idx / ldl), j = (int)(idx % ldl);
    if (j <= i) {
      fr_d2 s = (fr_d2){0.0, 0.0};
      for (int k0 = 0; k0 < splits; k0 += 16) {      // sixteen slabs in flight (eight until round 5), added in slab order
        fr_d2 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
          v[u] = k0 + u < splits ? *reinterpret_cast<const fr_d2*>(Cpart + (k0 + u) * slab + idx) : (fr_d2){0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
      }
      const int64_t p = (int64_t)i * (i + 1) / 2 + j;
      if (FUSE && weighted) {
        double g0 = s.x;
        if (j == i) g0 = g0 * exp(theta[d + p]) + wsum;
        out[1 + d + p] = wm.scale * g0;
        if (j + 1 <= i) {
          double g1 = s.y;
          if (j + 1 == i) g1 = g1 * exp(theta[d + p + 1]) + wsum;
          out[1 + d + p + 1] = wm.scale * g1;
        }
      } else if (FUSE) {
        double g0 = -s.x * invN;
        if (j == i) g0 = g0 * exp(theta[d + p]) - ent;              // free (log) diagonal + entropy
        out[1 + d + p] = g0;
        if (j + 1 <= i) {
          double g1 = -s.y * invN;
          if (j + 1 == i) g1 = g1 * exp(theta[d + p + 1]) - ent;
          out[1 + d + p + 1] = g1;
        }
      } else {
        S.sums[S.off_c + p] = s.x;
        if (j + 1 <= i) S.sums[S.off_c + p + 1] = s.y;
      }
    }
  }
  if (tid < ldz) {
    double s = 0.0;
    if (tid < d) {
      for (int rb0 = 0; rb0 < n_rb; rb0 += 32) {      // 32 loads in flight (16 until round 5), summed in row-block order
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = rb0 + u < n_rb ? colpart[(int64_t)(rb0 + u) * ldz + tid] : 0.0;
#pragma unroll
        for (int u = 0; u < 32; ++u) s += v[u];
      }
    }
    if (FUSE) {
      if (tid < d) out[1 + tid] = weighted ? wm.scale * s : -s * invN;
    } else {
      S.sums[S.off_col + tid] = s;
    }
  }
  if (blockIdx.x == 0) {
    // The scalar tail: sum of the f partials and sum of the log-diagonal.  One workgroup, so it is written for
    // latency: every load of a pass is requested before anything is summed, and the block sums share one pair of
    // barriers.
    const int tx = threadIdx.x;
    double f = 0.0, t = 0.0;
    for (int e0 = 0; e0 < n_fpart; e0 += 4 * 256) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * 256 + tx;
        v[u] = e < n_fpart ? fpart[e] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) f += v[u];
    }
    for (int c0 = 0; c0 < d; c0 += 4 * 256) {
      double dg[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * 256 + tx, cc = c < d ? c : 0;
        dg[u] = FUSE ? theta[d + (int64_t)cc * (cc + 1) / 2 + cc] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c0 + u * 256 + tx < d) t += dg[u];
    }
    __shared__ double sh3[8];
    f = fr_wave_sum(f), t = fr_wave_sum(t);
    if ((tx & 63) == 0) sh3[tx >> 6] = f, sh3[4 + (tx >> 6)] = t;
    __syncthreads();
    if (tx == 0) {
      f = (sh3[0] + sh3[1]) + (sh3[2] + sh3[3]);
      const double sum_logdiag = (sh3[4] + sh3[5]) + (sh3[6] + sh3[7]);
      if (FUSE) {
        const double F = f + n_local_w * c0;
        const double half_sq = pd ? 0.5 * S.sums[1] * invN : 0.5 * d;    // 1/2 mean ||eps||^2 or its expectation
        const double H = half_sq + 0.5 * d * kLog2PiFr + sum_logdiag;
        out[0] = weighted ? wm.value[0] : -(F * invN + H);
      } else {
        S.sums[0] = f;
      }
    }
  }
}

// epilogue of the sharded job: all-reduced packed sums -> (value, grad)
__global__ void __launch_bounds__(256) fr_epilogue_packed_kernel(FrSums S, const double* __restrict__ theta,
                                                                 int d, double n_local_w, double n_total,
                                                                 double c0, double* __restrict__ out, int pd,
                                                                 FrWeighted wm) {
  __shared__ double sh[4];
  const int64_t np = (int64_t)d * (d + 1) / 2;
  const double invN = 1.0 / n_total;
  // grid-stride: under the overlapped schedule the kernel runs beside the next evaluation's sampling product and is
  // launched with a few dozen workgroups, so that it takes a few slots for a little longer instead of streaming two
  // thousand workgroups through the dispatcher between the product's tiles (which cost that product 18 us)
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < np; p += (int64_t)gridDim.x * 256) {
    int i = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while ((int64_t)(i + 1) * (i + 2) / 2 <= p) ++i;
    while ((int64_t)i * (i + 1) / 2 > p) --i;
    const bool diag = p == (int64_t)i * (i + 1) / 2 + i;
    if (wm.scale != 0.0) {
      double g = S.sums[S.off_c + p];
      if (diag) g = g * exp(theta[d + p]) + wm.wsum[0];
      out[1 + d + p] = wm.scale * g;
    } else {
      double g = -S.sums[S.off_c + p] * invN;                       // d value / d L_ij
      if (diag) g = g * exp(theta[d + p]) - (pd ? 0.0 : 1.0);
      out[1 + d + p] = g;
    }
  }
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < d; p += (int64_t)gridDim.x * 256)
    out[1 + p] = wm.scale != 0.0 ? wm.scale * S.sums[S.off_col + p] : -S.sums[S.off_col + p] * invN;
  if (blockIdx.x == 0) {
    double t = 0.0;
    for (int i = threadIdx.x; i < d; i += 256) t += theta[d + (int64_t)i * (i + 1) / 2 + i];
    const double sum_logdiag = fr_block_sum(t, sh);
    if (threadIdx.x == 0) {
      const double F = S.sums[0] + n_local_w * c0;
      const double half_sq = pd ? 0.5 * S.sums[1] * invN : 0.5 * d;
      const double H = half_sq + 0.5 * d * kLog2PiFr + sum_logdiag;
      out[0] = wm.scale != 0.0 ? wm.value[0] : -(F * invN + H);
    }
  }
}

// G[n][:] *= rs[n]  (multivariate t: the Gram GEMM then yields sum_n g_n (z_n / s_n)')
__global__ void __launch_bounds__(256) fr_rowscale_kernel(double* __restrict__ G, int64_t ldz, int64_t n, int d,
                                                          const double* __restrict__ rs) {
  const int64_t row = blockIdx.x;            // rows on x: gridDim.y stops at 65 535
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c < d) G[row * ldz + c] *= rs[row];
}

// ---- path derivative ("sticking the landing", objectives.py:156-159) for the dense Gaussian ----------------
// value = -mean(f(z) - log q(z; stop(theta))): the score -dlog q/dz = L^-T eps is added to the model gradient row by
// row, G~ = G + E L^-1 (one N x D x D / 2 product, fr_pipeline_enqueue), and the sums of G~ are the entropy form's sums
// without the entropy term; value: 1/2 sum ||eps_n||^2 / N replaces D / 2.  L^-1 = ((L')^-1)' is formed explicitly
// (blocked recursive inversion, see fr_triinv_leaf_kernel and the host loop, then a transposition).  Rounds 2-3 went
// through the noise Gram matrix (C' = C + L^-T sum eps eps'): the same N x D^2 flops for the Gram product plus a
// D x D x D product and a second column pass.
struct EpiStoreD {          // C_z = sign * acc   (z: product index of a batched launch)
  double* C;
  int64_t ld;
  double sign;
  int64_t batch_c;
  __device__ void operator()(int z, int row, int col, double acc) const {
    C[z * batch_c + (int64_t)row * ld + col] = sign * acc;
  }
};

// Inverse of the diagonal blocks of U = L' (rows / columns [s, s + kTriLeaf)): one WAVE per column j of a block's
// inverse, column-oriented back substitution.  Lane r keeps the running sum p_r = sum_{k > i} U[r][k] x_k of "its"
// rows r and r + 64 in registers; step i reads p_i with v_readlane (i is wave-uniform), forms
// x_i = (delta_ij - p_i) / U_ii and adds U[r][i] x_i to every p_r, r < i.  U[.][i] is row i of L -- contiguous in
// the packed parameter; the four waves of a workgroup work on four columns of the same block and first stage the
// block's rows in LDS (128 KB, all loads in flight at once), so the dependent chain per step is a readlane, a
// subtract, a multiply and an FMA fed from LDS.  The D columns of all blocks run in parallel.
constexpr int kTriLeaf = 128;
__device__ __forceinline__ double fr_readlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__global__ void __launch_bounds__(256) fr_triinv_leaf_kernel(const double* __restrict__ theta,
                                                             const double* __restrict__ U, int d, int64_t ld,
                                                             double* __restrict__ X) {
  extern __shared__ double ls[];                            // ls[i * kTriLeaf + r] = L[s + i][s + r], r < i, else 0
  const int lane = threadIdx.x & 63;
  const int c0 = blockIdx.x * 4;                            // first of the workgroup's four columns
  const int s = c0 / kTriLeaf * kTriLeaf;
  const int jmax = (c0 + 3 < d ? c0 + 3 : d - 1) - s;       // last row any of the four waves needs
  // staging: batches of 16 unconditional loads per thread (addresses clamped into the row, the value selected
  // afterwards) -- written as `r < i ? theta[..] : 0` the compiler branched around each load and waited for it
  // before the next one: 64 dependent round trips, 20 of the kernel's 27 us at D = 256
  const int total = (jmax + 1) * kTriLeaf;
  for (int e0 = threadIdx.x; e0 < total; e0 += 256 * 16) {
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int e = e0 + 256 * k < total ? e0 + 256 * k : total - 1;
      const int i = e / kTriLeaf, r = e % kTriLeaf;
      const int64_t gi = s + i;
      v[k] = theta[d + gi * (gi + 1) / 2 + s + (r < i ? r : 0)];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int e = e0 + 256 * k;
      if (e < total) ls[e] = (e % kTriLeaf) < (e / kTriLeaf) ? v[k] : 0.0;
    }
  }
  __syncthreads();
  const int c = __builtin_amdgcn_readfirstlane(c0 + (threadIdx.x >> 6));   // this wave's column
  if (c >= d) return;
  const int j = c - s;
  double p0 = 0.0, p1 = 0.0, x0 = 0.0, x1 = 0.0;            // rows lane, lane + 64 of the block
  // 1 / U_ii of "its" rows, computed by all lanes at once and read with v_readlane in the loop
  double q0 = 1.0, q1 = 1.0;
  if (lane <= j) q0 = 1.0 / U[(int64_t)(s + lane) * ld + s + lane];
  if (lane + 64 <= j) q1 = 1.0 / U[(int64_t)(s + lane + 64) * ld + s + lane + 64];
  // two straight-line phases (no per-step selects): pivots in rows >= 64 touch both register slots, pivots in
  // rows < 64 only the first (rows >= 64 lie below them: L[s + i][s + r] = 0 for r >= i).  The row of the NEXT step
  // is read from LDS while this step's chain (readlane, subtract, multiply, FMA) runs.
  int i = j;
  double l0 = ls[i * kTriLeaf + lane], l1 = ls[i * kTriLeaf + lane + 64];
  for (; i >= 64; --i) {
    const int in = i > 0 ? i - 1 : 0;
    const double l0n = ls[in * kTriLeaf + lane], l1n = ls[in * kTriLeaf + lane + 64];
    const double xi = ((i == j ? 1.0 : 0.0) - fr_readlane(p1, i - 64)) * fr_readlane(q1, i - 64);
    if (lane + 64 == i) x1 = xi;
    p0 = fma(l0, xi, p0);
    p1 = fma(l1, xi, p1);
    l0 = l0n;
    l1 = l1n;
  }
  for (; i >= 0; --i) {
    const int in = i > 0 ? i - 1 : 0;
    const double l0n = ls[in * kTriLeaf + lane];
    const double xi = ((i == j ? 1.0 : 0.0) - fr_readlane(p0, i)) * fr_readlane(q0, i);
    if (lane == i) x0 = xi;
    p0 = fma(l0, xi, p0);
    l0 = l0n;
  }
  double* Xb = X + (int64_t)s * ld + s;
  if (lane <= j) Xb[(int64_t)lane * ld + j] = x0;
  if (lane + 64 <= j) Xb[(int64_t)(lane + 64) * ld + j] = x1;
}

// out[j][i] = in[i][j] for a d x d matrix (row stride ld both sides), 32 x 32 tiles through LDS
__global__ void __launch_bounds__(256) fr_transpose_kernel(const double* __restrict__ in, double* __restrict__ out, int d,
                                                           int64_t ld) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + r, j = j0 + tx;
    tile[r][tx] = (i < d && j < d) ? in[(int64_t)i * ld + j] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int j = j0 + r, i = i0 + tx;
    if (i < d && j < d) out[(int64_t)j * ld + i] = tile[tx][r];
  }
}

// sum of squares of the n x d noise matrix: one partial per workgroup (rows strided over the grid), summed in a fixed
// order by fr_sumsq_final_kernel
__global__ void __launch_bounds__(256) fr_sumsq_kernel(const double* __restrict__ E, int64_t ld, int64_t n, int d,
                                                       double* __restrict__ part) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int64_t r = blockIdx.x; r < n; r += gridDim.x) {
    const double* e = E + r * ld;
    for (int c = 2 * threadIdx.x; c < d; c += 512) {
      const fr_d2 v = *reinterpret_cast<const fr_d2*>(e + c);
      s = fma(v.x, v.x, s);
      if (c + 1 < d) s = fma(v.y, v.y, s);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ void __launch_bounds__(256) fr_sumsq_final_kernel(const double* __restrict__ part, int count,
                                                             double* __restrict__ out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += 256) s += part[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// theta -> mu, L' (dense, row stride ldl) on stream st
int fr_unpack_enqueue(vb_ctx* ctx, hipStream_t st, const double* theta_dev, int D, int64_t ldl, double* Lt, double* mu,
                      double* theta_copy) {
  hipLaunchKernelGGL(fr_unpack_kernel, dim3((unsigned)((D + 31) / 32), (unsigned)((D + 31) / 32)), dim3(256), 0, st,
                     theta_dev, D, ldl, Lt, mu, theta_copy);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// ---- the blocking call's pipelined parameter upload (FrUpload, vb_common.h) -------------------------------------------
// L' columns [32 jt0, 32 (jt0 + gridDim.y)) from the flat parameter: fr_unpack_kernel's tiles with the column tile index
// offset by jt0; mu is written by the launch that has with_mu set.
__global__ void __launch_bounds__(256) fr_unpack_cols_kernel(const double* __restrict__ theta, int d, int64_t ldl,
                                                             double* __restrict__ Lt, double* __restrict__ mu, int jt0,
                                                             int with_mu) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int k0 = blockIdx.x * 32, j0 = (jt0 + (int)blockIdx.y) * 32;
  if (with_mu && blockIdx.y == 0) {
    const int i = k0 + (int)threadIdx.x;
    if (threadIdx.x < 32 && i < d) mu[i] = theta[i];
  }
  if (k0 > j0 + 31) {            // below the diagonal of L' (k > j): zeros
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const int k = k0 + r, j = j0 + tx;
      if (k < d && j < d) Lt[(int64_t)k * ldl + j] = 0.0;
    }
    return;
  }
#pragma unroll
  for (int r = ty; r < 32; r += 8) {       // read L[j0 + r][k0 + tx]
    const int j = j0 + r, k = k0 + tx;
    double v = 0.0;
    if (j < d && k <= j) {
      v = theta[d + (int64_t)j * (j + 1) / 2 + k];
      if (k == j) v = exp(v);
    }
    tile[r][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, j = j0 + tx;
    if (k < d && j < d) Lt[(int64_t)k * ldl + j] = tile[tx][r];
  }
}

// Start the upload of `theta_host` (caller's pageable array) into ctx->fr_theta and its unpacked copy (ctx->fr_lt): mu and
// the LAST rows of L first.  The sampling product Z = E L' + mu cuts its k range per column block (tri_mode 1), so column
// block b needs rows [64 b, 64 b + 64) of L -- contiguous in the flat parameter -- and the heaviest blocks need the last
// rows: they go first, their product starts behind the first chunk's event, and the light blocks' rows arrive while the
// heavy tiles run.  Three chunks of about equal bytes (boundaries at multiples of 64 rows).  Everything else that reads the
// parameter is ordered behind the last chunk's event by fr_pipeline_enqueue.
int fr_upload_begin(vb_ctx* ctx, const double* theta_host, int64_t d) {
  vb_ctx::FrUpload& U = ctx->fr_up;
  const int D = (int)d;
  const int64_t ldl = round_up(d, 16);
  const int tn = gemm_tiles(D, 64);
  if (!ctx->up_stream) {
    VB_HIP(ctx, hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      VB_HIP(ctx, hipStreamCreateWithFlags(&ctx->up_side[i], hipStreamNonBlocking));
      VB_HIP(ctx, hipEventCreateWithFlags(&ctx->up_ev_join[i], hipEventDisableTiming));
    }
    VB_HIP(ctx, hipEventCreateWithFlags(&ctx->up_ev_main, hipEventDisableTiming));
    for (int c = 0; c < 4; ++c) VB_HIP(ctx, hipEventCreateWithFlags(&U.ev[c], hipEventDisableTiming));
  }
  VB_TRY(ensure(ctx, ctx->fr_lt, (size_t)(ldl + d * ldl) * sizeof(double)));
  double* mu = (double*)ctx->fr_lt.ptr;
  double* Lt = mu + ldl;
  double* dev = (double*)ctx->fr_theta.ptr;
  // chunk boundaries in column blocks of 64: equal areas of the triangle, from the bottom
  const char* ce = getenv("VB_FR_UPLOAD_CHUNKS");      // (experiments; read per call)
  const int want = ce ? atoi(ce) : 3;
  int nc = want < 1 ? 1 : want > 4 ? 4 : want;
  if (nc > tn) nc = tn;
  int edge[5];
  edge[0] = tn;
  for (int c = 1; c < nc; ++c) {
    int e = (int)(tn * sqrt((double)(nc - c) / nc) + 0.5);
    if (e >= edge[c - 1]) e = edge[c - 1] - 1;
    if (e < nc - c) e = nc - c;
    edge[c] = e;
  }
  edge[nc] = 0;
  U.n_chunks = nc;
  U.consumed = false;
  hipStream_t us = ctx->up_stream;
  // what is in flight on the main stream may still read the previous parameter
  VB_HIP(ctx, hipEventRecord(ctx->up_ev_main, ctx->stream));
  if (ctx->fr_busy) VB_HIP(ctx, hipStreamWaitEvent(us, ctx->up_ev_main, 0));
  ctx->fr_busy = false;
  VB_HIP(ctx, hipMemcpyAsync(dev, theta_host, (size_t)d * sizeof(double), hipMemcpyHostToDevice, us));      // mu
  for (int c = 0; c < nc; ++c) {
    const int b0 = edge[c + 1], b1 = edge[c];
    const int64_t r0 = (int64_t)b0 * 64, r1 = (int64_t)b1 * 64 < d ? (int64_t)b1 * 64 : d;
    const int64_t o0 = d + r0 * (r0 + 1) / 2, o1 = d + r1 * (r1 + 1) / 2;
    U.bn_begin[c] = b0;
    U.bn_count[c] = b1 - b0;
    VB_HIP(ctx, hipMemcpyAsync(dev + o0, theta_host + o0, (size_t)(o1 - o0) * sizeof(double), hipMemcpyHostToDevice, us));
    const int jt0 = (int)(r0 / 32), jt1 = (int)((r1 + 31) / 32);
    hipLaunchKernelGGL(fr_unpack_cols_kernel, dim3((unsigned)((D + 31) / 32), (unsigned)(jt1 - jt0)), dim3(256), 0, us,
                       (const double*)dev, D, ldl, Lt, mu, jt0, c == 0 ? 1 : 0);
    VB_HIP(ctx, hipGetLastError());
    VB_HIP(ctx, hipEventRecord(U.ev[c], us));
  }
  ctx->fr_up_active = true;
  ctx->fr_lt_d = d;                 // the unpacked copy is (being made) current: the pipeline must not unpack again
  ctx->fr_lt_owner = nullptr;
  return VB_OK;
}

// ---- optimiser step + unpack in one kernel (vb_fit, dense family) --------------------------------------------------------
// The loop used to run fit_step_kernel (5.9 us at D = 1024) and, at the top of the next evaluation, fr_unpack_kernel
// (6.6 us) on the parameter it had just written.  Same tiling as the unpack: a 32 x 32 tile of the packed triangle is
// read along its rows -- here together with the gradient entry and the optimiser state of each element -- stepped
// (fit_step_apply: numpy's operation order, no contraction, so the trajectory stays the host loop's bit for bit),
// written back, and transposed through LDS into L'.  Tiles below the diagonal of L' hold zeros from the first
// evaluation's full unpack and are not touched again.
__global__ void __launch_bounds__(256) fr_step_unpack_kernel(FitStep a, int d, int64_t ldl, double* __restrict__ Lt,
                                                             double* __restrict__ mu) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int k0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
  if (blockIdx.y == 0) {
    const int i = k0 + (int)threadIdx.x;
    if (threadIdx.x < 32 && i < d) {
      double s1, s2, th;
      fit_step_load(a, i, &s1, &s2, &th);
      mu[i] = fit_step_apply_vals(a, i, a.out[1 + i], s1, s2, th);
    }
    if (blockIdx.x == 0 && threadIdx.x == 32) a.values[a.k] = a.out[0];
  }
  if (k0 > j0 + 31) return;            // below the diagonal of L' (k > j): no parameter lives here
#pragma unroll
  for (int r = ty; r < 32; r += 8) {       // L[j0 + r][k0 + tx]
    const int j = j0 + r, k = k0 + tx;
    double v = 0.0;
    if (j < d && k <= j) {
      const int64_t p = d + (int64_t)j * (j + 1) / 2 + k;
      double s1, s2, th;
      fit_step_load(a, p, &s1, &s2, &th);
      v = fit_step_apply_vals(a, p, a.out[1 + p], s1, s2, th);
      if (k == j) v = exp(v);
    }
    tile[r][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {       // L'[k0 + r][j0 + tx]
    const int k = k0 + r, j = j0 + tx;
    if (k < d && j < d && k <= j) Lt[(int64_t)k * ldl + j] = tile[tx][r];
  }
}

int fr_step_unpack_enqueue(vb_ctx* ctx, const FitStep& a, int64_t d) {
  const int64_t ldl = round_up(d, 16);
  if (!ctx->fr_lt.ptr || ctx->fr_lt.bytes < (size_t)(ldl + d * ldl) * sizeof(double))
    return fail(ctx, VB_ERR_STATE, "no unpacked parameter buffer (the first evaluation makes it)");
  double* mu = (double*)ctx->fr_lt.ptr;
  hipLaunchKernelGGL(fr_step_unpack_kernel, dim3((unsigned)((d + 31) / 32), (unsigned)((d + 31) / 32)), dim3(256), 0,
                     ctx->stream, a, (int)d, ldl, mu + ldl, mu);
  VB_HIP(ctx, hipGetLastError());
  ctx->fr_lt_owner = a.theta;
  ctx->fr_lt_d = d;
  return VB_OK;
}

// Xa = (L')^-1 = U^-1 (upper triangular, row stride ldl) by recursive doubling: diagonal blocks of kTriLeaf rows are
// inverted by back substitution (one wave per column), then [[A, B], [0, C]]^-1 = [[A^-1, -A^-1 B C^-1], [0, C^-1]]
// level by level -- two batched GEMMs per level, D^3 / 3 flops in all instead of a triangular solve.  T: D x ldl scratch.
// clean == true: the caller vouches that Xa's strictly lower triangle still holds the zeros of an earlier call on the same
// buffer and layout (nothing here ever writes below the diagonal, and every block of T is written before it is read), so
// the two D x ld memsets are skipped
int fr_tri_inverse_enqueue(vb_ctx* ctx, hipStream_t st, const double* theta_dev, const double* Lt, int D, int64_t ldl,
                           double* Xa, double* T, bool clean) {
  const int n_cu = ctx->prop.multiProcessorCount;
  const int64_t slab = (int64_t)D * ldl;
  if (!clean) {
    VB_HIP(ctx, hipMemsetAsync(Xa, 0, (size_t)slab * sizeof(double), st));
    VB_HIP(ctx, hipMemsetAsync(T, 0, (size_t)slab * sizeof(double), st));
  }
  static const hipError_t leaf_attr = hipFuncSetAttribute(
      reinterpret_cast<const void*>(fr_triinv_leaf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
      kTriLeaf * kTriLeaf * (int)sizeof(double));         // 128 KB of LDS per workgroup
  VB_HIP(ctx, leaf_attr);
  hipLaunchKernelGGL(fr_triinv_leaf_kernel, dim3((unsigned)((D + 3) / 4)), dim3(256),
                     kTriLeaf * kTriLeaf * sizeof(double), st, theta_dev, Lt, D, ldl, Xa);
  VB_HIP(ctx, hipGetLastError());
  GemmArgs gn;
  gn.lda = ldl;
  gn.ldb = ldl;
  gn.tri_mode = 0;
  for (int b = kTriLeaf; b < D; b *= 2) {
    // the pairs of a level are independent products of one shape: one batched launch (blockIdx.z = pair) for
    // the full pairs, one more for a ragged last pair
    const int full = D / (2 * b);
    const int64_t pair_stride = (int64_t)2 * b * ldl + 2 * b;
    for (int pass = 0; pass < 2; ++pass) {
      const int s0 = pass == 0 ? 0 : full * 2 * b;
      const int count = pass == 0 ? full : (s0 + b < D ? 1 : 0);
      if (count == 0) continue;
      const int b2 = pass == 0 ? b : D - s0 - b;
      const int64_t oa = (int64_t)s0 * ldl + s0, ob = (int64_t)s0 * ldl + s0 + b,
                    oc = (int64_t)(s0 + b) * ldl + s0 + b;
      gn.batch = 1;
      gn.batch_a = pair_stride;
      gn.batch_b = pair_stride;
      gn.A = Lt + ob;      // B block (b x b2)
      gn.B = Xa + oc;      // C^-1 (b2 x b2)
      gn.M = b;
      gn.N = b2;
      gn.K = b2;
      gemm_f64_launch<true>(st, gn, count, n_cu, EpiStoreD{T + ob, ldl, 1.0, pair_stride});
      gn.A = Xa + oa;      // A^-1 (b x b)
      gn.B = T + ob;
      gn.K = b;
      gemm_f64_launch<true>(st, gn, count, n_cu, EpiStoreD{Xa + ob, ldl, -1.0, pair_stride});
    }
  }
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// ---- wrappers shared with the multivariate-t path (vb_mvt.hip) ---------------------------------------
int fr_colsum_enqueue(vb_ctx* ctx, const double* G, const double* Zc, int64_t ldz, int64_t n, int d, int fmode,
                      const double* ivar, double* colpart, double* fpart, const double* roww, int square) {
  const int n_rb = (int)((n + 127) / 128);
  hipLaunchKernelGGL(fr_colsum_kernel, dim3((unsigned)((d + 127) / 128), (unsigned)n_rb), dim3(256), 0, ctx->stream,
                     G, Zc, ldz, n, d, fmode, ivar, colpart, fpart, 0.0, roww, square);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

int fr_reduce_enqueue(vb_ctx* ctx, const double* Cpart, int splits, int64_t slab, int d, int64_t ldl,
                      const double* colpart, int n_rb, int64_t ldz, const double* fpart, int n_fpart, FrSums S, bool mirror,
                      const double* wpart) {
  fr_reduce_launch(ctx, ctx->stream, Cpart, splits, slab, d, ldl, colpart, n_rb, ldz, fpart, n_fpart, S, mirror ? 2 : 0, wpart);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

int gram_lower_enqueue(vb_ctx* ctx, const double* A, const double* B, int64_t ld, int d, int64_t n, int splits,
                       double* Cpart, int64_t ldc, int64_t slab) {
  GemmArgs g3;
  g3.A = A;
  g3.lda = ld;
  g3.B = B;
  g3.ldb = ld;
  g3.M = d;
  g3.N = d;
  g3.K = (int)n;
  g3.tri_mode = 2;
  const char* cfg_env = getenv("VB_GRAM_CFG");      // experiments: the tile configuration of this product alone (vb_gemm_f64.h)
  const char* xcd_env = getenv("VB_GRAM_XCD");
  g3.xcd_group = xcd_env ? atoi(xcd_env) : 1;
  gemm_f64_launch<false>(ctx->stream, g3, splits, ctx->prop.multiProcessorCount, EpiSplitSlab{Cpart, ldc, slab},
                         cfg_env ? atoi(cfg_env) : 0);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// The same product with the column sums of A formed out of the gradient GEMM's LDS tiles (EpiSplitSlabCs: one row of
// `colsum` per split, row stride colsum_ld) -- no separate pass over A.  *fused = false: the shape does not go through
// the LDS-DMA kernel (or there are more splits than rows in the caller's column-sum buffer) and the caller needs
// fr_colsum_enqueue as before.
int gram_lower_colsum_enqueue(vb_ctx* ctx, const double* A, const double* B, int64_t ld, int d, int64_t n, int splits,
                              double* Cpart, int64_t ldc, int64_t slab, double* colsum, int64_t colsum_ld,
                              int colsum_rows, bool* fused) {
  GemmArgs g3;
  g3.A = A;
  g3.lda = ld;
  g3.B = B;
  g3.ldb = ld;
  g3.M = d;
  g3.N = d;
  g3.K = (int)n;
  g3.tri_mode = 2;
  *fused = n % kGemmBK == 0 && gemm_uses_dma(g3) && splits <= colsum_rows;
  if (*fused)
    gemm_f64_launch<false>(ctx->stream, g3, splits, ctx->prop.multiProcessorCount,
                           EpiSplitSlabCs{Cpart, ldc, slab, colsum, colsum_ld});
  else
    gemm_f64_launch<false>(ctx->stream, g3, splits, ctx->prop.multiProcessorCount, EpiSplitSlab{Cpart, ldc, slab});
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// vb_noise_moments: column sums and (optionally) the Gram matrix of a noise slot -> host
int noise_moments(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, double* colsum_host, double* gram_host) {
  if (n <= 0 || n > ns.n || d != ns.d) return fail(ctx, VB_ERR_INVALID, "noise slot shape mismatch");
  const int D = (int)d;
  const int64_t ld = ns.ld, ldl = round_up(d, 16), slab = d * ldl;
  const int n_rb = (int)((n + 127) / 128), cs_gx = (D + 127) / 128;
  const int splits = gram_host ? gram_splits(ctx, D, n) : 0;
  int64_t off = 0;
  auto carve = [&off](int64_t doubles) {
    const int64_t o = off;
    off += round_up(doubles, 16);
    return o;
  };
  const int64_t o_col = carve((int64_t)n_rb * ld), o_f = carve((int64_t)n_rb * cs_gx), o_cpart = carve((int64_t)splits * slab),
                o_sums = carve(16 + ld + slab);
  VB_TRY(ensure(ctx, ctx->scratch, (size_t)off * sizeof(double)));
  double* base = (double*)ctx->scratch.ptr;
  const double* E = (const double*)ns.buf.ptr;
  hipStream_t st = ctx->stream;
  VB_TRY(fr_colsum_enqueue(ctx, E, E, ld, n, D, 0, nullptr, base + o_col, base + o_f, nullptr, 0));
  if (gram_host) VB_TRY(gram_lower_enqueue(ctx, E, E, ld, D, n, splits, base + o_cpart, ldl, slab));
  FrSums S;
  S.sums = base + o_sums;
  S.off_col = 16;
  S.off_c = 16 + ld;
  S.len = 16 + ld + slab;
  VB_TRY(fr_reduce_enqueue(ctx, base + o_cpart, splits, slab, D, ldl, base + o_col, n_rb, ld, base + o_f, 0, S));
  if (ctx->comm) VB_TRY(comm_allreduce_sum(ctx, st, S.sums, (size_t)(gram_host ? S.len : 16 + ld)));
  VB_HIP(ctx, hipMemcpyAsync(colsum_host, S.sums + S.off_col, (size_t)d * sizeof(double), hipMemcpyDeviceToHost, st));
  if (gram_host)
    VB_HIP(ctx, hipMemcpy2DAsync(gram_host, (size_t)d * sizeof(double), S.sums + S.off_c, (size_t)ldl * sizeof(double),
                                 (size_t)d * sizeof(double), (size_t)d, hipMemcpyDeviceToHost, st));
  VB_HIP(ctx, hipStreamSynchronize(st));
  if (gram_host)      // the product fills the lower triangle: mirror it
    for (int64_t i = 0; i < d; ++i)
      for (int64_t j = i + 1; j < d; ++j) gram_host[i * d + j] = gram_host[j * d + i];
  return VB_OK;
}

int gram_splits(vb_ctx* ctx, int d, int64_t n) {
  const int tiles = gemm_tiles(d, 128);
  const int lower_tiles = tiles * (tiles + 1) / 2;
  int splits = ctx->prop.multiProcessorCount / lower_tiles;
  const int max_splits = (int)(n / 256) > 0 ? (int)(n / 256) : 1;
  if (splits > max_splits) splits = max_splits;
  static const int forced = [] {
    const char* e = getenv("VB_GRAM_SPLITS");
    return e ? atoi(e) : 0;
  }();
  if (forced > 0 && forced < splits) splits = forced;
  return splits < 1 ? 1 : splits;
}

// ---- XCD-aware tile order of the lower-triangular product C = G' E ------------------------------------------
// Block x of a launch runs on XCD x % 8 and every XCD has its own 4-MiB L2.  In row-by-row order the tiles that share
// an operand panel (same row block: the same 128 columns of G; same column block: the same 64 columns of E) land on
// eight different XCDs and each XCD streams nearly every panel: 355 MB cross the fabric per launch against 67 MB of
// operands (PMC FETCH_SIZE, D = 1024).  Here the tiles are walked in bands of two row blocks, column by column, and
// cut into eight runs of equal length -- compact 2 x ~4.5 patches that touch few panels -- and run c gets the blocks
// x = c, c + 8, c + 16, ...
static int tri2_tile_map(vb_ctx* ctx, int d, int bm_rows, int bn_cols, const int** map_out, int* blocks_out) {
  if (ctx->tri_map.ptr && ctx->tri_map_key[0] == d && ctx->tri_map_key[1] == bm_rows && ctx->tri_map_key[2] == bn_cols) {
    *map_out = (const int*)ctx->tri_map.ptr;
    *blocks_out = ctx->tri_map_blocks;
    return VB_OK;
  }
  const int tm = gemm_tiles(d, bm_rows), tn = gemm_tiles(d, bn_cols);
  std::vector<std::pair<int, int>> order;
  for (int band = 0; band < tm; band += 2)
    for (int bn = 0; bn < tn; ++bn)
      for (int bm = band; bm < band + 2 && bm < tm; ++bm)
        if (bn * bn_cols <= bm * bm_rows + bm_rows - 1) order.push_back({bm, bn});
  const int total = (int)order.size(), per = (total + 7) / 8, blocks = per * 8;
  std::vector<int> host((size_t)2 * blocks, -1);
  for (int c = 0; c < 8; ++c)
    for (int i = 0; i < per; ++i) {
      const int src = c * per + i;
      if (src >= total) break;
      host[2 * (8 * i + c)] = order[src].first;
      host[2 * (8 * i + c) + 1] = order[src].second;
    }
  VB_TRY(ensure(ctx, ctx->tri_map, host.size() * sizeof(int)));
  VB_HIP(ctx, hipMemcpyAsync(ctx->tri_map.ptr, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice,
                             ctx->stream));
  VB_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `host` is stack-scoped
  ctx->tri_map_key[0] = d;
  ctx->tri_map_key[1] = bm_rows;
  ctx->tri_map_key[2] = bn_cols;
  ctx->tri_map_blocks = blocks;
  *map_out = (const int*)ctx->tri_map.ptr;
  *blocks_out = blocks;
  return VB_OK;
}

// ---- short shards: split-k for the two N x D x D products ------------------------------------------------------------
// With few sample rows (a strong-scaling shard: 512 rows at D = 1024 give 128 tiles of 64 x 64 for 256 CUs) the 64-slab k
// range of ONE tile is the critical path of Z = E L' and of G = -(Z - m) P: 47 and 51 us with three quarters of the chip
// idle.  The k range is then cut into `parts` pieces that run as separate workgroups into slabs of partial products
// (EpiSplitSlab, the triangular one simply finds some of its pieces empty), and these kernels add the slabs in fixed
// order and apply the epilogue of the unsplit product: Z = sum + mu - m, and G = -sum with the per-workgroup sum of
// f = 1/2 (z - m)' g.
__global__ void __launch_bounds__(256) fr_zsum_kernel(const double* __restrict__ P, int parts, int64_t pslab, int64_t n,
                                                      int d, int64_t ldz, const double* __restrict__ mu,
                                                      const double* __restrict__ shift, double* __restrict__ Z) {
  const int64_t idx = 2 * ((int64_t)blockIdx.x * 256 + threadIdx.x);
  if (idx >= n * ldz) return;
  const int c = (int)(idx % ldz);
  fr_d2 s = (fr_d2){0.0, 0.0};
  for (int k = 0; k < parts; ++k) s += *reinterpret_cast<const fr_d2*>(P + k * pslab + idx);
  if (c < d) s.x += mu[c] - (shift ? shift[c] : 0.0);
  else s.x = 0.0;
  if (c + 1 < d) s.y += mu[c + 1] - (shift ? shift[c + 1] : 0.0);
  else s.y = 0.0;
  *reinterpret_cast<fr_d2*>(Z + idx) = s;
}

__global__ void __launch_bounds__(256) fr_gsum_kernel(const double* __restrict__ P, int parts, int64_t pslab, int64_t n,
                                                      int d, int64_t ldz, const double* __restrict__ Zc,
                                                      double* __restrict__ G, double* __restrict__ fpart) {
  __shared__ double sh[4];
  const int64_t idx = 2 * ((int64_t)blockIdx.x * 256 + threadIdx.x);
  double f = 0.0;
  if (idx < n * ldz) {
    const int c = (int)(idx % ldz);
    fr_d2 s = (fr_d2){0.0, 0.0};
    for (int k = 0; k < parts; ++k) s += *reinterpret_cast<const fr_d2*>(P + k * pslab + idx);
    const fr_d2 z = *reinterpret_cast<const fr_d2*>(Zc + idx);
    if (c >= d) s.x = 0.0;
    if (c + 1 >= d) s.y = 0.0;
    *reinterpret_cast<fr_d2*>(G + idx) = (fr_d2){-s.x, -s.y};
    f = -0.5 * s.x * z.x - 0.5 * s.y * z.y;
  }
  f = fr_wave_sum(f);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = f;
  __syncthreads();
  if (threadIdx.x == 0) fpart[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ---- the fused evaluation's launch (vb_fullrank_fused.h) --------------------------------------------------------------
// phases == 2: Z and G by one persistent launch (returns the number of sum-f partials); phases == 3: C as well.
static int fr_fused_enqueue(vb_ctx* ctx, hipStream_t st, int phases, GemmArgs g1, GemmArgs g2, GemmArgs g3, int splits,
                            double* Z, double* G, int64_t ldz, const double* mu, const double* shift, double* fpart,
                            const EpiSplitSlabCs& e3, unsigned* tiles2_out) {
  const int n_cu = ctx->prop.multiProcessorCount;
  const int n = g1.M, D = g1.N;
  auto fill = [](GemmArgs& g, int splits_) {
    const int ks = gemm_tiles(g.K, splits_);
    g.k_split = gemm_tiles(ks, kGemmBK) * kGemmBK;
    g.tiles_m = gemm_tiles(g.M, 128);
    g.tiles_n = gemm_tiles(g.N, 64);
    g.prio_div = 0;
    g.ev0 = g.ev1 = nullptr;
  };
  fill(g1, 1);
  fill(g2, 1);
  fill(g3, splits);
  const int tm = g1.tiles_m, tn = g1.tiles_n;
  const int n_p1 = tm * tn, n_p2 = tm * tn;
  const int n_p3 = g3.tile_map ? g3.tile_blocks : (int)gemm_count_blocks(g3, 128, 64);
  // shared words: [head | err | 14 pad | zflag tm x tn | gflag tm x tn]
  const size_t words = 16 + 2 * (size_t)tm * tn;
  VB_TRY(ensure(ctx, ctx->fz_words, words * sizeof(unsigned)));      // (a new allocation is zeroed by ensure)
  unsigned* wbase = (unsigned*)ctx->fz_words.ptr;
  const int64_t key[5] = {n, D, splits, phases, n_p3};      // (the group count is read once per process)
  if (memcmp(key, ctx->fz_key, sizeof key) != 0 || !ctx->fz_items.ptr) {
    // the list, in a topological order: every Z tile heaviest k range first; then row block by row block the G tiles,
    // and behind the last row block of a split the C tiles of that split
    std::vector<int> host;
    auto push = [&host](int phase, int bx, int bz) {
      host.push_back(phase), host.push_back(bx), host.push_back(bz), host.push_back(0);
    };
    // (VB_FR_FUSED_GROUPS = g: the row blocks in g groups, each group's Z tiles followed by its G tiles -- measured, see
    // profiles/r04_fused_timeline.txt)
    static const int groups_env = getenv("VB_FR_FUSED_GROUPS") ? atoi(getenv("VB_FR_FUSED_GROUPS")) : 1;
    const int groups = groups_env < 1 ? 1 : (groups_env > tm ? tm : groups_env);
    std::vector<int> map;
    if (phases == 3 && g3.tile_map) {
      map.resize((size_t)2 * n_p3);
      VB_HIP(ctx, hipMemcpy(map.data(), g3.tile_map, map.size() * sizeof(int), hipMemcpyDeviceToHost));
    }
    int z_next = 0;
    for (int grp = 0; grp < groups; ++grp) {
      const int rb0 = (int)((int64_t)tm * grp / groups), rb1 = (int)((int64_t)tm * (grp + 1) / groups);
      // block x of the triangular product: idx = x / tm selects the column block (heaviest first), x % tm the row block
      for (int idx = 0; idx < tn; ++idx)
        for (int rb = rb0; rb < rb1; ++rb) push(0, idx * tm + rb, 0);
      for (int rb = rb0; rb < rb1; ++rb) {
        for (int cb = 0; cb < tn; ++cb) push(1, cb * tm + rb, 0);
        while (phases == 3 && z_next < splits) {
          const int64_t last_row = std::min<int64_t>(n, (int64_t)(z_next + 1) * g3.k_split) - 1;
          if (last_row / 128 > rb) break;
          for (int bx = 0; bx < n_p3; ++bx)
            if (map.empty() || map[2 * bx] >= 0) push(2, bx, z_next);
          ++z_next;
        }
      }
    }
    VB_TRY(ensure(ctx, ctx->fz_items, host.size() * sizeof(int)));
    VB_HIP(ctx, hipMemcpyAsync(ctx->fz_items.ptr, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice, st));
    VB_HIP(ctx, hipStreamSynchronize(st));      // `host` is stack-scoped
    memcpy(ctx->fz_key, key, sizeof key);
    ctx->fz_n_items = (int)(host.size() / 4);
  }
  if (++ctx->fz_epoch == 0) ctx->fz_epoch = 1;
  FzArgs a;
  a.g1 = g1, a.g2 = g2, a.g3 = g3;
  a.e1 = EpiStoreZPub{Z, ldz, mu, shift};
  a.G = G, a.ldz = ldz, a.Zc = Z, a.fpart = fpart;
  a.e3 = e3;
  a.s.head = wbase, a.s.err = wbase + 1, a.s.zflag = wbase + 16, a.s.gflag = wbase + 16 + (size_t)tm * tn;
  a.s.epoch = ctx->fz_epoch, a.s.tiles_n = tn;
  a.items = (const int4*)ctx->fz_items.ptr;
  a.n_items = ctx->fz_n_items, a.n_p1 = n_p1, a.n_p2 = n_p2, a.n_p3 = n_p3;
  a.clk = nullptr;
#ifdef VB_FUSED_CLOCK
  VB_TRY(ensure(ctx, ctx->scratch2, (size_t)4 * a.n_items * sizeof(long long)));
  a.clk = (long long*)ctx->scratch2.ptr;
#endif
  constexpr size_t lds = (size_t)3 * (128 * kGemmBK + kGemmBK * 64) * sizeof(double) + 16;
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fr_fused_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fr_fused_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    configured = true;
  }
  static const int wgs_env = getenv("VB_FR_FUSED_WGS") ? atoi(getenv("VB_FR_FUSED_WGS")) : 0;
  const unsigned grid = (unsigned)(wgs_env > 0 ? wgs_env : 2 * n_cu);
  if (phases == 3) hipLaunchKernelGGL(fr_fused_kernel<3>, dim3(grid), dim3(256), lds, st, a);
  else hipLaunchKernelGGL(fr_fused_kernel<2>, dim3(grid), dim3(256), lds, st, a);
  VB_HIP(ctx, hipGetLastError());
#ifdef VB_FUSED_CLOCK
  if (const char* path = getenv("VB_FUSED_CLOCK_DUMP")) {      // per-item clocks of THIS launch (tools/fused_clock.py)
    VB_HIP(ctx, hipStreamSynchronize(st));
    std::vector<long long> h((size_t)4 * a.n_items);
    VB_HIP(ctx, hipMemcpy(h.data(), a.clk, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(path, "wb")) {
      fwrite(h.data(), sizeof(long long), h.size(), f);
      fclose(f);
    }
  }
#endif
  *tiles2_out = (unsigned)n_p2;
  return VB_OK;
}

// ---- host orchestration ------------------------------------------------------------------------------
// theta_dev != nullptr: full-rank Gaussian (Z = E L' + mu, lower-triangular gradient, epilogue into the flat layout).
// theta_dev == nullptr: multivariate t (X = (E R) / s + mu with the dense symmetric root R and the per-row scale
// `row_scale`; the caller gets the raw sums [F | sum g | sum_n g_n (e_n / s_n)' (full D x D)] in `sums_out`).
int fr_pipeline_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int64_t n_total,
                        const double* theta_dev, double* out_dev, const double* mu_dev, const double* root_dev,
                        const double* row_scale, FrSums* sums_out, unsigned flags, const FrWeighted* weighted) {
  const ModelDev& m = ctx->model;
  const bool mvt = theta_dev == nullptr;
  FrWeighted wm;
  wm.roww = nullptr;
  wm.scale = 0.0;
  wm.wsum = wm.value = nullptr;
  if (weighted) {   // the t family takes the weighted raw sums (wm.scale / wsum / value are the caller's there)
    if (flags & VB_FLAG_PATH_DERIV)
      return fail(ctx, VB_ERR_UNSUPPORTED, "weighted sums: entropy-form pipeline only");
    wm = *weighted;
  }
  const bool pd = (flags & VB_FLAG_PATH_DERIV) != 0;
  if (pd && mvt) return fail(ctx, VB_ERR_UNSUPPORTED, "path derivative: dense Gaussian family only");
  const bool glm = m.id == VB_MODEL_LOGISTIC;
  const bool source = m.id == VB_MODEL_SOURCE;
  if (m.id != VB_MODEL_GAUSS_DIAG && m.id != VB_MODEL_FUNNEL && m.id != VB_MODEL_GAUSS_FULL && !glm && !source)
    return fail(ctx, VB_ERR_UNSUPPORTED, "full-rank path: unsupported model id %d", m.id);
  if (m.dim != d)
    return fail(ctx, VB_ERR_INVALID, "model dimension %d != family dimension %lld", m.dim, (long long)d);
  if (n <= 0 || d <= 0 || n > ns.n || d != ns.d)
    return fail(ctx, VB_ERR_INVALID, "noise slot holds %lld x %lld, evaluation asks %lld x %lld",
                (long long)ns.n, (long long)ns.d, (long long)n, (long long)d);
  hipStream_t st = ctx->stream;
  const int D = (int)d;
  // tile-configuration overrides of the three GEMMs (experiments; 0 = the launcher's choice)
  static const int cfg1 = getenv("VB_FR_G1_CFG") ? atoi(getenv("VB_FR_G1_CFG")) : 0;
  static const int cfg2 = getenv("VB_FR_G2_CFG") ? atoi(getenv("VB_FR_G2_CFG")) : 0;
  static const int cfg3 = getenv("VB_FR_G3_CFG") ? atoi(getenv("VB_FR_G3_CFG")) : 0;
  const int64_t ldl = round_up(d, 16), ldz = round_up(d, 16);
  const int n_cu = ctx->prop.multiProcessorCount;
  const int tiles = gemm_tiles(D, 128);
  const int lower_tiles = mvt ? tiles * tiles : tiles * (tiles + 1) / 2;
  int splits = n_cu / lower_tiles;   // one wave of workgroups: no second, mostly empty round
  // at least this many sample rows per split: 192 lets D = 512 take 21 splits of its 36 lower 64 x 64 tiles (756
  // workgroups on the 768 resident slots: gradient GEMM 38.9 -> 33.7 us, split reduction 8.5 -> 10.4 us; 256 rows
  // stopped it at 16 splits); D = 1024 takes 7 either way
  static const int split_rows = getenv("VB_FR_SPLIT_ROWS") ? atoi(getenv("VB_FR_SPLIT_ROWS")) : 192;
  const int max_splits = (int)(n / split_rows) > 0 ? (int)(n / split_rows) : 1;
  static const int splits_env = getenv("VB_FR_SPLITS") ? atoi(getenv("VB_FR_SPLITS")) : 0;      // experiment
  if (splits_env > 0) splits = splits_env;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int n_rb = (int)((n + 127) / 128);
  const int cs_gx = (D + 127) / 128;
  // regression targets: the log-likelihood partials of the eta GEMM follow the column-sum kernel's f partials
  const int64_t glm_part = glm ? gemm_max_blocks(n, m.n_data) : 0;
  const int64_t ldr = glm ? round_up(m.n_data, 16) : 0;
  const int n_fpart = m.id == VB_MODEL_FUNNEL ? (int)((n + 3) / 4) : source ? (int)n : n_rb * cs_gx + (int)glm_part;
  const int64_t slab = d * ldl;

  // device buffers (one allocation, carved)
  int64_t off = 0;
  auto carve = [&off](int64_t doubles) {
    const int64_t o = off;
    off += round_up(doubles, 16);
    return o;
  };
  const int64_t o_mu = carve(ldz), o_lt = carve(slab), o_z = carve(n * ldz), o_g = carve(n * ldz),
                o_cpart = carve((int64_t)(splits + 1) * slab), o_col = carve((int64_t)(n_rb + 1) * ldz),
                o_fpart = carve((int64_t)n_fpart + (int64_t)n_rb * cs_gx + gemm_max_blocks(n, D) + (n * ldz) / 512 + 1),
                o_r = carve(glm ? n * ldr : 0);
  // path derivative: (L')^-1 (Xa), a product buffer T (then L^-1), partial sums of squares of the noise
  const int64_t o_xa = pd ? carve(slab) : 0, o_t = pd ? carve(slab) : 0, o_m2 = pd ? carve(256) : 0;
  // sum vector: the t family takes the full D x ldl matrix; the Gaussian family packs the lower triangle in
  // theta's own order, twice over when the all-reduce of one evaluation overlaps the kernels of the next
  const int64_t np = d * (d + 1) / 2;
  FrSums S;
  S.off_col = 16;
  S.off_c = 16 + ldz;
  S.len = 16 + ldz + (mvt ? slab : round_up(np, 16));
  static const bool overlap_env = !(getenv("VB_COMM_OVERLAP") && atoi(getenv("VB_COMM_OVERLAP")) == 0);
  const bool overlap = !mvt && ctx->comm != nullptr && overlap_env;
  const int64_t o_sums = carve(S.len * (overlap ? 2 : 1));
  VB_TRY(ensure(ctx, ctx->fr_work, (size_t)off * sizeof(double)));
  double* base = (double*)ctx->fr_work.ptr;
  double *mu = base + o_mu, *Lt = base + o_lt, *Z = base + o_z, *G = base + o_g, *Cpart = base + o_cpart,
         *colpart = base + o_col, *fpart = base + o_fpart;
  // the dense family keeps mu and L' in a buffer of their own: when the parameter is the resident one
  // (vb_fullrank_set_theta) it is unpacked once per upload, not once per evaluation
  bool lt_cached = false;
  if (!mvt) {
    VB_TRY(ensure(ctx, ctx->fr_lt, (size_t)(ldz + slab) * sizeof(double)));
    mu = (double*)ctx->fr_lt.ptr;
    Lt = mu + ldz;
    static const bool cache_env = !(getenv("VB_FR_UNPACK_CACHE") && atoi(getenv("VB_FR_UNPACK_CACHE")) == 0);
    const bool resident = theta_dev == (const double*)ctx->fr_theta.ptr;
    // ... or vb_fit's parameter, whose step kernel wrote mu and L' of the stepped value itself (fr_step_unpack_enqueue)
    const bool stepped = ctx->fr_lt_owner != nullptr && ctx->fr_lt_owner == theta_dev;
    lt_cached = cache_env && (resident || stepped) && ctx->fr_lt_d == d;
    if (!stepped) ctx->fr_lt_owner = nullptr;
    ctx->fr_lt_d = (resident || stepped) ? d : 0;     // (a foreign parameter leaves the copy stale for the resident one)
  }
  // stream plan (as mf_enqueue's `overlap`): everything up to the split reduction stays in order on the main
  // stream; the all-reduce and the epilogue go to `post` behind one event, into sum set `seq & 1`, and the main
  // stream only waits for them when that set comes round again two evaluations later
  Pipeline& P = ctx->pipe;
  int set = 0;
  if (overlap) {
    VB_TRY(pipe_init(ctx));
    set = (int)(ctx->fr_seq++ & 1);
    if (P.fin_valid[set]) VB_HIP(ctx, hipStreamWaitEvent(st, P.ev_fin[set], 0));
  } else if (P.post_pending) {   // order this in-order evaluation after everything `post` has in flight
    VB_HIP(ctx, hipStreamWaitEvent(st, P.ev_fin[P.last_set], 0));
    P.post_pending = false;
  }
  S.sums = base + o_sums + (int64_t)set * S.len;

  // a pipelined upload of this evaluation's parameter is in flight (vb_elbo_grad_fullrank): the sampling product below may
  // consume it chunk by chunk; whoever else reads the parameter or its unpacked copy first waits for the last chunk
  vb_ctx::FrUpload* up = (ctx->fr_up_active && !mvt && lt_cached) ? &ctx->fr_up : nullptr;
  ctx->fr_up_active = false;
  bool up_waited = false;
  auto up_wait_all = [&]() -> int {
    if (up && !up_waited) {
      VB_HIP(ctx, hipStreamWaitEvent(st, up->ev[up->n_chunks - 1], 0));
      up_waited = true;
    }
    return VB_OK;
  };
  if (up && pd) VB_TRY(up_wait_all());      // (the triangular inverse reads the whole parameter at once)

  if (mvt) {
    VB_HIP(ctx, hipMemcpyAsync(mu, mu_dev, (size_t)d * sizeof(double), hipMemcpyDeviceToDevice, st));
    VB_HIP(ctx, hipMemcpy2DAsync(Lt, (size_t)ldl * sizeof(double), root_dev, (size_t)ldl * sizeof(double),
                                 (size_t)d * sizeof(double), (size_t)d, hipMemcpyDeviceToDevice, st));
  } else if (!lt_cached) {
    hipLaunchKernelGGL(fr_unpack_kernel, dim3((unsigned)((d + 31) / 32), (unsigned)((d + 31) / 32)), dim3(256), 0, st, theta_dev,
                       D, ldl, Lt, mu);
    VB_HIP(ctx, hipGetLastError());
  }

  if (pd) {
    // Path derivative (objectives.py:166-168: the score's own parameter dependence is stopped): with z = mu + L eps the
    // gradient of -log q along the path is L^-T eps, so every row of G takes that term -- G~ = G + E L^-1, one more
    // N x D x D / 2 product -- and the usual sums of G~ finish the job: no entropy term, no noise Gram matrix, no
    // D x D x D product (round 4; rounds 2-3 formed L^-T (E'E / N): 620 -> ~500 us at D = 1024, N = 4096).
    // (L')^-1 = U^-1 by recursive doubling: diagonal blocks of kTriLeaf rows are inverted by back substitution
    // (one wave per column), then [[A, B], [0, C]]^-1 = [[A^-1, -A^-1 B C^-1], [0, C^-1]] level by level --
    // two GEMMs per pair of blocks, D^3 / 3 flops in all instead of a triangular solve
    double *Xa = base + o_xa, *T = base + o_t, *sq = base + o_m2;
    VB_TRY(fr_tri_inverse_enqueue(ctx, st, theta_dev, Lt, D, ldl, Xa, T));
    VB_HIP(ctx, hipGetLastError());
    // L^-1 = (U^-1)' with its rows k-major for the product below (T is free again)
    hipLaunchKernelGGL(fr_transpose_kernel, dim3((unsigned)((D + 31) / 32), (unsigned)((D + 31) / 32)), dim3(256), 0, st,
                       (const double*)Xa, T, D, ldl);
    // sum ||eps_n||^2 -> sums[1]: the value's mean log q of the samples (:167)
    const int sq_blocks = (int)(n < 256 ? n : 256);
    hipLaunchKernelGGL(fr_sumsq_kernel, dim3((unsigned)sq_blocks), dim3(256), 0, st, (const double*)ns.buf.ptr, ns.ld, n, D, sq);
    hipLaunchKernelGGL(fr_sumsq_final_kernel, dim3(1), dim3(256), 0, st, (const double*)sq, sq_blocks, S.sums + 1);
    VB_HIP(ctx, hipGetLastError());
  }

  // GEMM 1: Z[n][j] = sum_k E[n][k] Lt[k][j]   (Lt[k][j] = 0 for k > j)
  GemmArgs g1;
  g1.A = (const double*)ns.buf.ptr;
  g1.lda = ns.ld;
  g1.B = Lt;
  g1.ldb = ldl;
  g1.M = (int)n;
  g1.N = D;
  g1.K = D;
  g1.tri_mode = mvt ? 0 : 1;
  prof_events(ctx, &g1.ev0, &g1.ev1, 1, VB_PROF_FR_SAMPLE_GEMM);
  int fmode = 0;
  // correlated-Gaussian target under the dense Gaussian family: no pass over G and Z between the GEMMs -- sum f comes
  // out of the model GEMM's epilogue (EpiNegateF) and the column sums of G out of the gradient GEMM (EpiSplitSlabCs)
  static const bool fast_env = !(getenv("VB_FR_FUSED_SUMS") && atoi(getenv("VB_FR_FUSED_SUMS")) == 0);
  const bool fused_sums = fast_env && !mvt && m.id == VB_MODEL_GAUSS_FULL && !wm.roww && !row_scale &&
                          n % kGemmBK == 0 && gemm_uses_dma(g1) && (int64_t)splits <= n_rb;
  // short shards (fewer than two 64 x 64 tiles per CU): the N x D x D products with their k range cut into `kparts`
  // pieces (see fr_zsum_kernel); the slabs of partial products live in the split area of the gradient product, which
  // is not in use yet
  static const int kparts_env = getenv("VB_FR_KPARTS") ? atoi(getenv("VB_FR_KPARTS")) : -1;     // experiments; 1 = off
  int kparts = 1;
  const int64_t pslab = n * ldz;
  if (!row_scale && !pd && cfg1 == 0 && cfg2 == 0 && m.id != VB_MODEL_GAUSS_DIAG) {      // (pd keeps a slab of its own there)
    // measured (tools/fr_bench.py, D = 1024): 512 rows 126 -> 85 us per evaluation, 256 rows 118 -> 61 us, 1 024 rows
    // 133 -> 128 us, 2 048 rows unchanged (not split).  At D = 512 a tile's 32 slabs are no longer than a piece plus
    // the extra kernel: not split (pieces of at least 16 slabs out of at least 48).
    const long tiles64 = gemm_count_blocks(g1, 64, 64);
    if (tiles64 < 2L * n_cu && D >= 48 * kGemmBK) kparts = (int)((2L * n_cu + tiles64 - 1) / tiles64);
    if (kparts > 4) kparts = 4;
    if (kparts_env >= 1) kparts = kparts_env;
    while (kparts > 1 && (D % (kGemmBK * kparts) != 0 || D / kparts < 16 * kGemmBK)) --kparts;
    if ((int64_t)kparts * pslab > (int64_t)(splits + 1) * slab) kparts = 1;
  }
  // the fused evaluation: 2 = Z and G in one persistent launch, 3 = the gradient product's split slabs as well
  int fz_mode = ctx->fr_fused_mode;
  if (fz_mode < 0) {
    const char* e = getenv("VB_FR_FUSED");
    fz_mode = e ? atoi(e) : 0;
  }
  if (!(fused_sums && kparts == 1 && cfg1 == 0 && cfg2 == 0 && cfg3 == 0 && n % 128 == 0 && D % 64 == 0 &&
        (int64_t)n * ldz * 8 < ((int64_t)1 << 31)))
    fz_mode = 0;
  if ((fz_mode != 2 && fz_mode != 3) || pd) fz_mode = 0;      // (the path derivative changes G between the products)
  const unsigned sum_blocks = (unsigned)((pslab / 2 + 255) / 256);
  // Z = E L' + mu - shift into `Z` (the samples, or z - m for the correlated Gaussian target)
  auto sample_gemm = [&](const double* shift, int cfg) {
    if (wm.z_ready && !shift) {      // (the caller's samples: see FrWeighted)
      Z = const_cast<double*>(wm.z_ready);
      return;
    }
    if (up && !up_waited && kparts == 1 && cfg == 0 && !row_scale && gemm_uses_dma(g1) && up->n_chunks > 1 &&
        gemm_count_blocks(g1, 128, 64) < 4L * n_cu) {
      // chunk 0 (the heaviest column blocks) on the main stream behind its event; the lighter chunks on side streams
      // behind theirs (and behind everything the main stream had queued before: the noise), joined below
      up->consumed = true;
      (void)hipEventRecord(ctx->up_ev_main, st);
      for (int c = 0; c < up->n_chunks; ++c) {
        hipStream_t sc = c == 0 ? st : ctx->up_side[(c - 1) & 1];
        if (c > 0) (void)hipStreamWaitEvent(sc, ctx->up_ev_main, 0);
        (void)hipStreamWaitEvent(sc, up->ev[c], 0);
        GemmArgs gc = g1;
        gc.ev0 = gc.ev1 = nullptr;
        gc.bn_begin = up->bn_begin[c];
        gc.bn_count = up->bn_count[c];
        gemm_f64_launch<true>(sc, gc, 1, n_cu, EpiStoreZ{Z, ldz, mu, shift, row_scale}, 4);
      }
      for (int i = 0; i < 2 && i < up->n_chunks - 1; ++i) {
        (void)hipEventRecord(ctx->up_ev_join[i], ctx->up_side[i]);
        (void)hipStreamWaitEvent(st, ctx->up_ev_join[i], 0);
      }
      up_waited = true;      // (the main stream is now behind the last chunk's event as well)
      return;
    }
    (void)up_wait_all();
    {
      // experiment: two-way k split of the heavy column blocks only
      const char* he = getenv("VB_FR_HSPLIT");
      const int k_half = gemm_tiles(gemm_tiles(D, 2), kGemmBK) * kGemmBK;
      if (he && atoi(he) != 0 && kparts == 1 && !mvt && !row_scale && gemm_uses_dma(g1) && D % 2 == 0 && D >= 384 &&
          gemm_count_blocks(g1, 64, 64) <= 2L * n_cu && k_half % 64 == 0 && pslab <= (int64_t)(splits + 1) * slab) {
        const int hc = atoi(he) == 2 ? 4 : 8;
        gemm_f64_launch<true>(st, g1, 2, n_cu, EpiStoreZHeavy{Z, ldz, mu, shift, Cpart, k_half}, cfg ? cfg : hc);
        const int64_t items = n * ((D - k_half + 1) / 2);
        hipLaunchKernelGGL(fr_zfix_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, (const double*)Cpart, n, D, ldz,
                           k_half, Z);
        return;
      }
    }
    if (kparts > 1) {
      gemm_f64_launch<true>(st, g1, kparts, n_cu, EpiSplitSlab{Cpart, ldz, pslab});
      hipLaunchKernelGGL(fr_zsum_kernel, dim3(sum_blocks), dim3(256), 0, st, (const double*)Cpart, kparts, pslab, n, D, ldz,
                         (const double*)mu, shift, Z);
    } else {
      gemm_f64_launch<true>(st, g1, 1, n_cu, EpiStoreZ{Z, ldz, mu, shift, row_scale}, cfg);
    }
  };
  unsigned tiles2 = 0;
  bool g_prescaled = false;      // the target's own kernel wrote G already scaled by the row weights
  // diagonal Gaussian target under the dense Gaussian family: sum f out of the sampling product's epilogue
  const bool diag_f = fast_env && !mvt && m.id == VB_MODEL_GAUSS_DIAG && !wm.roww && !row_scale &&
                      n % kGemmBK == 0 && gemm_uses_dma(g1) && (int64_t)splits <= n_rb;
  if (m.id == VB_MODEL_GAUSS_DIAG) VB_TRY(up_wait_all());      // (a reducing epilogue: one launch)
  if (m.id == VB_MODEL_GAUSS_DIAG && diag_f) {
    tiles2 = gemm_f64_launch<true>(st, g1, 1, n_cu, EpiGaussDiagF{G, ldz, mu, m.p0, m.p1, fpart});
    fmode = 1;
  } else if (m.id == VB_MODEL_GAUSS_DIAG) {
    gemm_f64_launch<true>(st, g1, 1, n_cu, EpiGaussDiag{G, ldz, mu, m.p0, m.p1, row_scale});
    fmode = 1;
  } else if (m.id == VB_MODEL_FUNNEL) {
    sample_gemm(nullptr, 0);
    VB_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(fr_funnel_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, (const double*)Z,
                       G, ldz, n, D, m, fpart, wm.roww);
    g_prescaled = wm.roww != nullptr;
  } else if (source) {        // the user's row kernel: G and one f per sample (summed with the other f partials)
    sample_gemm(nullptr, 0);
    VB_HIP(ctx, hipGetLastError());
    VB_TRY(user_rows_enqueue(ctx, st, Z, ldz, n, D, G, ldz, fpart));
  } else if (glm) {
    sample_gemm(nullptr, 0);
    VB_HIP(ctx, hipGetLastError());
    double* Rm = base + o_r;
    double* part = fpart + (int64_t)n_rb * cs_gx;
    VB_HIP(ctx, hipMemsetAsync(part, 0, (size_t)glm_part * sizeof(double), st));
    GemmArgs gh;                                   // eta = Z X'   [n x n_data x d]
    gh.A = Z;
    gh.lda = ldz;
    gh.B = m.p1;
    gh.ldb = m.ldq;
    gh.M = (int)n;
    gh.N = (int)m.n_data;
    gh.K = D;
    gh.tri_mode = 0;
    gemm_f64_launch<true>(st, gh, 1, n_cu, EpiGlm{Rm, ldr, m.p2, part, m.link, m.aux});
    VB_HIP(ctx, hipGetLastError());
    VB_TRY(glm_grad_enqueue(ctx, st, m, Rm, ldr, Z, G, ldz, n, D));   // G = R X - Z / sd^2
    fmode = 3;
  } else {
    GemmArgs g2;                                   // G = -(Z - m) P,  P symmetric: B[k][j] = P[k][j]
    g2.A = Z;
    g2.lda = ldz;
    g2.B = m.p1;
    g2.ldb = m.ldp;
    g2.M = (int)n;
    g2.N = D;
    g2.K = D;
    g2.tri_mode = 0;
    if (wm.g_ready) {
      G = const_cast<double*>(wm.g_ready);      // (the caller's G of these very samples: see FrWeighted)
    } else if (fz_mode >= 2) {
      // one persistent launch for Z - m and G (and, fz_mode 3, the split slabs of C): vb_fullrank_fused.h
      EpiSplitSlabCs e3{Cpart, ldl, slab, colpart, ldz};
      GemmArgs g3f;
      g3f.A = G, g3f.lda = ldz, g3f.B = (const double*)ns.buf.ptr, g3f.ldb = ns.ld;
      g3f.M = D, g3f.N = D, g3f.K = (int)n, g3f.tri_mode = 2;
      if (fz_mode == 3) VB_TRY(tri2_tile_map(ctx, D, 128, 64, &g3f.tile_map, &g3f.tile_blocks));
      VB_TRY(up_wait_all());
      VB_TRY(fr_fused_enqueue(ctx, st, fz_mode, g1, g2, g3f, splits, Z, G, ldz, mu, m.p0, fpart, e3, &tiles2));
    } else {
    sample_gemm(m.p0, cfg1);                         // Z - m
    VB_HIP(ctx, hipGetLastError());
    }
    if (fz_mode >= 2 || wm.g_ready) {
    } else if (kparts > 1 && fused_sums) {
      gemm_f64_launch<true>(st, g2, kparts, n_cu, EpiSplitSlab{Cpart, ldz, pslab});
      hipLaunchKernelGGL(fr_gsum_kernel, dim3(sum_blocks), dim3(256), 0, st, (const double*)Cpart, kparts, pslab, n, D, ldz,
                         (const double*)Z, G, fpart);
      tiles2 = sum_blocks;
    } else {
      prof_events(ctx, &g2.ev0, &g2.ev1, 1, VB_PROF_FR_MODEL_GEMM);
      if (fused_sums) tiles2 = gemm_f64_launch<true>(st, g2, 1, n_cu, EpiNegateF{G, ldz, Z, fpart}, cfg2);
      else gemm_f64_launch<true>(st, g2, 1, n_cu, EpiNegate{G, ldz}, cfg2);
    }
    fmode = wm.g_ready ? 0 : 2;      // (weighted sums take their value from the weights: no f here)
  }
  VB_HIP(ctx, hipGetLastError());

#ifdef VB_DBG_IDLE
  {
    static const int idle_us = getenv("VB_FR_IDLE_US") ? atoi(getenv("VB_FR_IDLE_US")) : 0;
    if (idle_us > 0) hipLaunchKernelGGL(dbg_idle_kernel, dim3(1), dim3(64), 0, st, (long long)idle_us * 100);
  }
#endif
  if (wm.roww && !g_prescaled) {   // weighted sums: scale the rows of G before anything is summed
    hipLaunchKernelGGL(fr_rowscale_kernel, dim3((unsigned)n, (unsigned)((D + 255) / 256)), dim3(256), 0, st, G, ldz, n,
                       D, wm.roww);
    VB_HIP(ctx, hipGetLastError());
  }
  // C = G' E: lower-triangular tiles (all tiles for the t family), split over the sample axis
  GemmArgs g3;
  g3.A = G;
  g3.lda = ldz;
  g3.B = (const double*)ns.buf.ptr;
  g3.ldb = ns.ld;
  g3.M = D;
  g3.N = D;
  g3.K = (int)n;
  g3.tri_mode = mvt ? 0 : 2;
  // targets whose row kernel leaves sum f behind already (funnel, source models): the column sums of G are all the pass
  // below would add, and they come out of the gradient product's LDS tiles as for the correlated Gaussian
  // (weighted sums: the same, once G is scaled -- the column sums of the gradient product's operand tiles ARE sum w g)
  const bool cs_only = diag_f || (fast_env && !fused_sums && !mvt && (m.id == VB_MODEL_FUNNEL || source) &&
                                 (!wm.roww || g_prescaled) &&
                                 !row_scale && n % kGemmBK == 0 && gemm_uses_dma(g3) && (int64_t)splits <= n_rb);
  // path derivative: sum f belongs to the model's G, the column sums and the gradient product to G~ = G + E L^-1.  Where
  // the pass below forms f from G (every target but the funnel and source models, whose own kernels left it behind) it
  // runs once before the score is added -- for f -- and once after, for the column sums only
  const bool f_from_pass = !(m.id == VB_MODEL_FUNNEL || source);
  const bool two_passes = pd && !fused_sums && !cs_only && f_from_pass;
  if (two_passes) {
    hipLaunchKernelGGL(fr_colsum_kernel, dim3((unsigned)cs_gx, (unsigned)n_rb), dim3(256), 0, st,
                       (const double*)G, (const double*)Z, ldz, n, D, fmode, m.p1, colpart, fpart,
                       glm ? 1.0 / (m.tau * m.tau) : 0.0, (const double*)nullptr, 0, (double*)nullptr,
                       (const double*)nullptr);
    VB_HIP(ctx, hipGetLastError());
  }
  if (pd) {       // G~ = G + E L^-1   (L^-1[k][j] = 0 for k < j: tri_mode 3)
    GemmArgs gy;
    gy.A = (const double*)ns.buf.ptr;
    gy.lda = ns.ld;
    gy.B = base + o_t;
    gy.ldb = ldl;
    gy.M = (int)n;
    gy.N = D;
    gy.K = D;
    gy.tri_mode = 3;
    gemm_f64_launch<true>(st, gy, 1, n_cu, EpiAccumulate{G, ldz});
    VB_HIP(ctx, hipGetLastError());
  }
  if (!fused_sums && !cs_only) {
    hipLaunchKernelGGL(fr_colsum_kernel, dim3((unsigned)cs_gx, (unsigned)n_rb), dim3(256), 0, st,
                       (const double*)G, (const double*)Z, ldz, n, D, two_passes ? 0 : fmode, m.p1, colpart,
                       (!f_from_pass || two_passes) ? fpart + n_fpart /*unused tail*/ : fpart,
                       glm ? 1.0 / (m.tau * m.tau) : 0.0, (const double*)nullptr, 0, row_scale ? G : (double*)nullptr,
                       row_scale);
    VB_HIP(ctx, hipGetLastError());
  } else if (row_scale) {
    hipLaunchKernelGGL(fr_rowscale_kernel, dim3((unsigned)n, (unsigned)((D + 255) / 256)), dim3(256), 0, st, G, ldz, n,
                       D, row_scale);
    VB_HIP(ctx, hipGetLastError());
  }
  // GEMM 3: C[i][j] = sum_n G[n][i] E[n][j]
  prof_events(ctx, &g3.ev0, &g3.ev1, 1, VB_PROF_FR_GRAD_GEMM);
  int n_rb_red = n_rb, n_fpart_red = n_fpart;
  if (fz_mode == 3) {          // the fused launch has written the split slabs and their column sums
    n_rb_red = splits;
    n_fpart_red = (int)tiles2;
  } else if (fused_sums || cs_only) {
    static const bool map_env = !(getenv("VB_FR_TILE_MAP") && atoi(getenv("VB_FR_TILE_MAP")) == 0);
    int cfg3_used = cfg3;
    if (map_env && cfg3 == 0 && gemm_count_blocks(g3, 128, 64) * splits * 100 >= 190L * n_cu) {
      cfg3_used = 2;     // the launcher's own choice for this shape (128 x 64 tiles), made here so that the list fits it
      VB_TRY(tri2_tile_map(ctx, D, 128, 64, &g3.tile_map, &g3.tile_blocks));
    }
    gemm_f64_launch<false>(st, g3, splits, n_cu, EpiSplitSlabCs{Cpart, ldl, slab, colpart, ldz}, cfg3_used);
    n_rb_red = splits;                  // one row of column sums per split
    if (fused_sums || diag_f) n_fpart_red = (int)tiles2;          // one partial of sum f per tile of the model GEMM
  } else {
    gemm_f64_launch<false>(st, g3, splits, n_cu, EpiSplitSlab{Cpart, ldl, slab});
  }
  VB_HIP(ctx, hipGetLastError());

  const int64_t red_items = slab / 2 > ldz ? slab / 2 : ldz;
  const dim3 red_grid((unsigned)((red_items + 255) / 256));
  VB_TRY(up_wait_all());      // (the epilogue reads the flat parameter's diagonal)
  if (mvt) {
    fr_reduce_launch(ctx, st, (const double*)Cpart, splits, slab, D, ldl, (const double*)colpart, n_rb, ldz, (const double*)fpart,
                     n_fpart, S, 1, nullptr);
    VB_HIP(ctx, hipGetLastError());
    if (ctx->comm) VB_TRY(comm_allreduce_sum(ctx, st, S.sums, (size_t)S.len));
    *sums_out = S;
    return VB_OK;
  }
  if (!ctx->comm) {   // single GPU: the split reduction writes (value, grad) itself
    hipLaunchKernelGGL(fr_reduce_packed_kernel<true>, red_grid, dim3(256), 0, st, (const double*)Cpart,
                       splits, slab, D, ldl, (const double*)colpart, n_rb_red, ldz,
                       (const double*)fpart, n_fpart_red, S, theta_dev, (double)n_total, (double)n_total, m.c0, out_dev,
                       pd ? 1 : 0, wm);
    VB_HIP(ctx, hipGetLastError());
    return VB_OK;
  }
  hipLaunchKernelGGL(fr_reduce_packed_kernel<false>, red_grid, dim3(256), 0, st, (const double*)Cpart,
                     splits, slab, D, ldl, (const double*)colpart, n_rb_red, ldz,
                     (const double*)fpart, n_fpart_red, S, theta_dev, (double)n_total, (double)n_total, m.c0, out_dev,
                     pd ? 1 : 0, wm);
  VB_HIP(ctx, hipGetLastError());
  hipStream_t st_post = st;
  if (overlap) {
    VB_HIP(ctx, hipEventRecord(P.ev_k1[set], st));
    st_post = P.post;
    VB_HIP(ctx, hipStreamWaitEvent(st_post, P.ev_k1[set], 0));
  }
  VB_TRY(comm_allreduce_sum(ctx, st_post, S.sums, (size_t)S.len));
  // one workgroup per CU: 328 -> 310 us per evaluation with a one-rank communicator (64: 312, 16: 350 -- the epilogue
  // then is what the evaluation after next waits for; unlimited = 2 050 workgroups: 328)
  static const int epi_env = getenv("VB_FR_EPI_WGS") ? atoi(getenv("VB_FR_EPI_WGS")) : -1;
  const int epi_cap = epi_env >= 0 ? epi_env : n_cu;
  const int64_t epi_full = (np + 255) / 256;
  const unsigned epi_grid = (unsigned)((overlap && epi_cap > 0 && epi_full > epi_cap) ? epi_cap : epi_full);
  hipLaunchKernelGGL(fr_epilogue_packed_kernel, dim3(epi_grid), dim3(256), 0, st_post, S,
                     theta_dev, D, (double)n_total, (double)n_total, m.c0, out_dev, pd ? 1 : 0, wm);
  VB_HIP(ctx, hipGetLastError());
  if (overlap) {
    VB_HIP(ctx, hipEventRecord(P.ev_fin[set], st_post));
    P.fin_valid[set] = true;
    P.post_pending = true;
    P.last_set = set;
  }
  return VB_OK;
}

// Z = E L' + mu into `Z` (n x ldz, ldz = round_up(d, 16)): the samples themselves, for per-row evaluations.
// theta_dev == nullptr: Z = (E root) * row_scale + mu with a full `root` (row stride ldz) as the t family has it
int fr_sample_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, const double* theta_dev, double* Z,
                      const double* mu_dev, const double* root_dev, const double* row_scale) {
  if (n <= 0 || d <= 0 || n > ns.n || d != ns.d) return fail(ctx, VB_ERR_INVALID, "noise slot shape mismatch");
  const int D = (int)d;
  const int64_t ldl = round_up(d, 16), ldz = ldl;
  hipStream_t st = ctx->stream;
  GemmArgs g1;
  g1.A = (const double*)ns.buf.ptr;
  g1.lda = ns.ld;
  g1.ldb = ldl;
  g1.M = (int)n;
  g1.N = D;
  g1.K = D;
  if (theta_dev) {
    VB_TRY(ensure(ctx, ctx->fr_work, (size_t)(ldz + d * ldl) * sizeof(double)));
    double* mu = (double*)ctx->fr_work.ptr;
    double* Lt = mu + ldz;
    hipLaunchKernelGGL(fr_unpack_kernel, dim3((unsigned)((d + 31) / 32), (unsigned)((d + 31) / 32)), dim3(256), 0, st, theta_dev, D, ldl,
                       Lt, mu);
    g1.B = Lt;
    g1.tri_mode = 1;
    gemm_f64_launch<true>(st, g1, 1, ctx->prop.multiProcessorCount, EpiStoreZ{Z, ldz, mu, nullptr, row_scale});
  } else {
    g1.B = root_dev;
    g1.tri_mode = 0;
    gemm_f64_launch<true>(st, g1, 1, ctx->prop.multiProcessorCount, EpiStoreZ{Z, ldz, mu_dev, nullptr, row_scale});
  }
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

int fr_elbo_grad_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int64_t n_total,
                         const double* theta_dev, double* out_dev, unsigned flags, const FrWeighted* weighted) {
  return fr_pipeline_enqueue(ctx, ns, n, d, n_total, theta_dev, out_dev, nullptr, nullptr, nullptr, nullptr, flags,
                             weighted);
}

}  // namespace vb
