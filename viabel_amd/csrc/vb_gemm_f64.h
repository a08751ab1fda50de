// fp64 MFMA GEMM building block for gfx950 (v_mfma_f64_16x16x4_f64), used by the dense-covariance
// paths (full-rank Gaussian: Z = E L', G' E; correlated-Gaussian target: (Z - m) P).
//
//   C[M x N] (+)= A[M x K] * B[K x N]
//   B is row-major [k][n] (ldb).  A is either row-major [m][k] (A_KCONTIG, e.g. the noise matrix
//   E[n][k]) or k-major [k][m] (e.g. G'[i][n] given as G[n][i]).
//
// Workgroup = 256 threads = 4 waves (2 x 2); block tile 128 x 128 x 16; each wave owns a 64 x 64
// sub-tile = 4 x 4 MFMA tiles (64 fp64 accumulators per lane).  Both operand tiles live in LDS
// k-major ([k][m], row stride 128 + 16 doubles = 1152 B, i.e. +32 banks per k row, so the two k rows
// a ds_read_b64 half-wave touches fall on disjoint banks); the A_KCONTIG loader transposes on the
// way in with 16 distinct rows per 16-lane store group (conflict-free ds_write_b64).  Global loads
// for slab k+1 are issued into registers before the MFMAs of slab k (register double buffering),
// LDS is double buffered: one barrier per slab.  The fp64 MFMA issues once per 64 cycles per SIMD,
// so 16 MFMAs per 8 ds_read_b64 keep the matrix pipe saturated from one wave per SIMD.
//
// Fragment layouts (cdna_hip_programming.md section 3): A lane l -> A[i = l & 15][k = l >> 4],
// B lane l -> B[k = l >> 4][j = l & 15], C/D lane l, register r -> C[i = (l >> 4) + 4 r][j = l & 15].
#pragma once

#include "vb_common.h"

namespace vb {

typedef double d2v __attribute__((ext_vector_type(2)));
typedef double d4v __attribute__((ext_vector_type(4)));

constexpr int kGemmBM = 128, kGemmBN = 128, kGemmBK = 16;
constexpr int kGemmLds = kGemmBM + 16;   // LDS row stride in doubles

struct GemmArgs {
  const double* A;
  const double* B;
  int64_t lda, ldb;
  int M, N, K;
  int tiles_m, tiles_n;
  int tri_mode;     // 0: dense; 1: B[k][j] == 0 for k > j (k-range cut per column block);
                    // 2: only output tiles with bm >= bn (lower triangle of a square C)
  int k_split;      // K range per blockIdx.z (multiple of kGemmBK); splits = gridDim.z
};

// Epilogue functor interface:  void operator()(int split, int row, int col, double acc) const;
// called for every in-range element of the block tile.
template <bool A_KCONTIG, class Epi>
__global__ void __launch_bounds__(256) gemm_f64_kernel(const GemmArgs g, const Epi epi) {
  __shared__ double As[2][kGemmBK][kGemmLds];
  __shared__ double Bs[2][kGemmBK][kGemmLds];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile assignment --------------------------------------------------------------------------
  int bm, bn;
  if (g.tri_mode == 2) {        // linear index over the lower-triangular tiles, row by row
    int idx = blockIdx.x;
    bm = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
    while ((bm + 1) * (bm + 2) / 2 <= idx) ++bm;
    while (bm * (bm + 1) / 2 > idx) --bm;
    bn = idx - bm * (bm + 1) / 2;
  } else if (g.tri_mode == 1) { // heaviest column blocks (largest k range) first
    bn = g.tiles_n - 1 - (int)(blockIdx.x / g.tiles_m);
    bm = blockIdx.x % g.tiles_m;
  } else {
    bn = blockIdx.x / g.tiles_m;
    bm = blockIdx.x % g.tiles_m;
  }
  const int m0 = bm * kGemmBM, n0 = bn * kGemmBN;
  int k_begin = blockIdx.z * g.k_split;
  int k_end = k_begin + g.k_split < g.K ? k_begin + g.k_split : g.K;
  if (g.tri_mode == 1) {
    const int kmax = n0 + kGemmBN;     // B[k][j] == 0 for k > j
    if (k_end > kmax) k_end = kmax;
  }

  d4v acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (d4v){0.0, 0.0, 0.0, 0.0};

  // ---- global -> register staging ---------------------------------------------------------------
  d2v ra[4], rb[4];
  // Branch-free: indices are clamped into range and the value zeroed afterwards (every operand row
  // is padded to a multiple of 16 doubles, so the 16-B load of a pair that straddles the logical
  // edge stays inside the allocation).
  auto load_slab = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // B (and k-major A): one 128-double row per wave-load
      const int p = i * 256 + t;
      const int krow = p >> 6, c = (p & 63) * 2;
      const int k = k0 + krow;
      const int kc = k < k_end ? k : k_end - 1;
      {
        const int n = n0 + c;
        const int nc = n < g.N ? n : 0;
        d2v v = *reinterpret_cast<const d2v*>(g.B + (int64_t)kc * g.ldb + nc);
        v.x = (k < k_end && n < g.N) ? v.x : 0.0;
        v.y = (k < k_end && n + 1 < g.N) ? v.y : 0.0;
        rb[i] = v;
      }
      if (!A_KCONTIG) {
        const int m = m0 + c;
        const int mc = m < g.M ? m : 0;
        d2v v = *reinterpret_cast<const d2v*>(g.A + (int64_t)kc * g.lda + mc);
        v.x = (k < k_end && m < g.M) ? v.x : 0.0;
        v.y = (k < k_end && m + 1 < g.M) ? v.y : 0.0;
        ra[i] = v;
      } else {
        // A[m][k]: wave-load q covers 16 rows x 4 k-pairs; 16 consecutive lanes = 16 distinct rows
        const int q = i * 4 + wave;
        const int row = (q >> 1) * 16 + (lane & 15);
        const int kp = (q & 1) * 4 + (lane >> 4);
        const int m = m0 + row, ka = k0 + 2 * kp;
        const int mc = m < g.M ? m : 0;
        const int kac = ka < k_end ? ka : 0;
        d2v v = *reinterpret_cast<const d2v*>(g.A + (int64_t)mc * g.lda + kac);
        v.x = (m < g.M && ka < k_end) ? v.x : 0.0;
        v.y = (m < g.M && ka + 1 < k_end) ? v.y : 0.0;
        ra[i] = v;
      }
    }
  };
  auto store_slab = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = i * 256 + t;
      const int krow = p >> 6, c = (p & 63) * 2;
      *reinterpret_cast<d2v*>(&Bs[buf][krow][c]) = rb[i];
      if (!A_KCONTIG) {
        *reinterpret_cast<d2v*>(&As[buf][krow][c]) = ra[i];
      } else {
        const int q = i * 4 + wave;
        const int row = (q >> 1) * 16 + (lane & 15);
        const int kp = (q & 1) * 4 + (lane >> 4);
        As[buf][2 * kp][row] = ra[i].x;
        As[buf][2 * kp + 1][row] = ra[i].y;
      }
    }
  };

  const int fi = lane & 15, fk = lane >> 4;
  auto compute_slab = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < kGemmBK / 4; ++kk) {
      double af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = As[buf][4 * kk + fk][wm * 64 + i * 16 + fi];
        bf[i] = Bs[buf][4 * kk + fk][wn * 64 + i * 16 + fi];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };

  if (k_begin < k_end) {
    load_slab(k_begin);
    store_slab(0);
    __syncthreads();
    int buf = 0;
    // steady state: a single basic block per slab (keeps the 128 accumulator registers in place)
    for (int k0 = k_begin + kGemmBK; k0 < k_end; k0 += kGemmBK) {
      load_slab(k0);          // in flight while the MFMAs below run
      compute_slab(buf);
      store_slab(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
    compute_slab(buf);
  }

  // ---- epilogue -----------------------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wn * 64 + j * 16 + fi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 64 + i * 16 + fk + 4 * r;
        if (row < g.M && col < g.N) epi((int)blockIdx.z, row, col, acc[i][j][r]);
      }
    }
}

inline int gemm_tiles(int x, int b) { return (x + b - 1) / b; }

template <bool A_KCONTIG, class Epi>
inline void gemm_f64_launch(hipStream_t st, GemmArgs g, int splits, const Epi& epi) {
  g.tiles_m = gemm_tiles(g.M, kGemmBM);
  g.tiles_n = gemm_tiles(g.N, kGemmBN);
  if (splits < 1) splits = 1;
  int ks = gemm_tiles(g.K, splits);
  g.k_split = gemm_tiles(ks, kGemmBK) * kGemmBK;
  const unsigned gx = g.tri_mode == 2 ? (unsigned)(g.tiles_m * (g.tiles_m + 1) / 2)
                                      : (unsigned)(g.tiles_m * g.tiles_n);
  hipLaunchKernelGGL((gemm_f64_kernel<A_KCONTIG, Epi>), dim3(gx, 1, (unsigned)splits), dim3(256), 0, st,
                     g, epi);
}

}  // namespace vb
