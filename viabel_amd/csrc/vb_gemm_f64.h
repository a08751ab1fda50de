// fp64 MFMA GEMM building block for gfx950, used by the dense-covariance paths (full-rank Gaussian:
// Z = E L', G' E; correlated-Gaussian target: (Z - m) P).
//
//   C[M x N] (+)= A[M x K] * B[K x N]
//   B is row-major [k][n] (ldb).  A is either row-major [m][k] (A_KCONTIG, e.g. the noise matrix
//   E[n][k]) or k-major [k][m] (e.g. G'[i][n] given as G[n][i]).
//
// Matrix instruction: v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per issue).  Measured on
// this MI355X (tools/fp64_peak.hip): the 4x4x4 4-block form sustains 63-73 TFLOP/s, the 16x16x4 form
// only 33-36 TFLOP/s, so the tile loop is built on the former.  Lane layout (probed with
// tools/mfma_f64_4x4x4_layout.hip): operand lane s holds A_blk[i][k] / B_blk[k][j] with
// i or j = s & 3, blk = (s >> 2) & 3, k = s >> 4; result lane l holds D_blk[i][j] with j = l & 3,
// blk = (l >> 2) & 3, i = l >> 4.
//
// Workgroup = 256 threads = 4 waves (2 x 2); block tile 128 x 128 x 16; each wave owns a 64 x 64
// sub-tile = 16 x 16 blocks of 4 x 4.  Per k-step of 4 a wave loads 4 A fragments (16 rows each: the 4
// blocks of an instruction are 4 consecutive row blocks) and 16 B fragments (arrangement r puts column
// block (blk + r) mod 16 into block blk), and issues 4 x 16 = 64 MFMAs: instruction (a, r) produces the
// blocks (4a + blk, (blk + r) mod 16), so every (row block, column block) pair is covered exactly once
// and each lane accumulates 64 doubles.  20 ds_read_b64 per 64 MFMAs (1024 matrix-pipe cycles).
// Operand tiles live in LDS k-major ([k][m], row stride 128 + 16 doubles = 1152 B = +32 banks per k
// row, so the two k rows a ds_read_b64 half-wave touches fall on disjoint banks); the A_KCONTIG loader
// transposes on the way in with 16 distinct rows per 16-lane store group (conflict-free ds_write_b64).
// Global loads for slab k+1 are issued into registers before the MFMAs of slab k, LDS is double
// buffered: one barrier per slab; edge handling is branch-free (clamped index + select).
#pragma once

#include "vb_common.h"

#include <type_traits>
#include <utility>

namespace vb {

typedef double d2v __attribute__((ext_vector_type(2)));
typedef double d4v __attribute__((ext_vector_type(4)));

constexpr int kGemmBN = 128, kGemmBK = 16;   // block tile: (32 AF) x 128 x 16, AF = A fragments per wave
constexpr int kGemmLds = 128 + 16;           // LDS row stride in doubles

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
// called for every in-range element of the block tile.  A functor that defines `double* part` and
// returns double from operator() additionally gets the per-workgroup sum of its return values written to
// part[blockIdx.z * gridDim.x + blockIdx.x] (workgroups that exit early write nothing: zero `part` first).
template <class E, class = void>
struct EpiReduces : std::false_type {};
template <class E>
struct EpiReduces<E, std::void_t<decltype(std::declval<E>().part)>> : std::true_type {};

// AF = 4: 128-row block tile, 64 accumulators per lane, one workgroup per CU;
// AF = 2:  64-row block tile, 32 accumulators per lane, two workgroups per CU (stalls of one are
//          covered by the MFMAs of the other) -- used when 128-row tiles would not fill the chip twice.
template <bool A_KCONTIG, int AF, class Epi>
__global__ void __launch_bounds__(256, AF == 2 ? 2 : 1) gemm_f64_kernel(const GemmArgs g, const Epi epi) {
  constexpr int BM = 32 * AF;
  constexpr int NA = BM / 32;   // staging iterations for the A tile
  __shared__ double As[2][kGemmBK][BM + 16];
  __shared__ double Bs[2][kGemmBK][kGemmLds];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile assignment --------------------------------------------------------------------------
  int bm, bn;
  if (g.tri_mode == 2) {
    // compact enumeration of the tiles that touch the lower triangle, row by row: consecutive block
    // ids (= consecutive XCDs) get equal work instead of XCD x owning tile row x
    int idx = blockIdx.x;
    bm = 0;
    for (;;) {
      const int cnt = min(g.tiles_n, (bm * BM + BM - 1) / kGemmBN + 1);
      if (idx < cnt) break;
      idx -= cnt;
      ++bm;
    }
    bn = idx;
  } else if (g.tri_mode == 1) {   // heaviest column blocks (largest k range) first
    bn = g.tiles_n - 1 - (int)(blockIdx.x / g.tiles_m);
    bm = blockIdx.x % g.tiles_m;
  } else {
    bn = blockIdx.x / g.tiles_m;
    bm = blockIdx.x % g.tiles_m;
  }
  const int m0 = bm * BM, n0 = bn * kGemmBN;
  int k_begin = blockIdx.z * g.k_split;
  int k_end = k_begin + g.k_split < g.K ? k_begin + g.k_split : g.K;
  if (g.tri_mode == 1) {
    const int kmax = n0 + kGemmBN;     // B[k][j] == 0 for k > j
    if (k_end > kmax) k_end = kmax;
  }

  double acc[AF][16];
#pragma unroll
  for (int i = 0; i < AF; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.0;

  // ---- global -> register staging ---------------------------------------------------------------
  d2v ra[NA], rb[4];
  // Branch-free: indices are clamped into range and the value zeroed afterwards (every operand row
  // is padded to a multiple of 16 doubles, so the 16-B load of a pair that straddles the logical
  // edge stays inside the allocation).
  auto load_slab = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // B: one 128-double row per wave-load
      const int p = i * 256 + t;
      const int krow = p >> 6, c = (p & 63) * 2;
      const int k = k0 + krow;
      const int kc = k < k_end ? k : k_end - 1;
      const int n = n0 + c;
      const int nc = n < g.N ? n : 0;
      d2v v = *reinterpret_cast<const d2v*>(g.B + (int64_t)kc * g.ldb + nc);
      v.x = (k < k_end && n < g.N) ? v.x : 0.0;
      v.y = (k < k_end && n + 1 < g.N) ? v.y : 0.0;
      rb[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (!A_KCONTIG) {         // A[k][m]: BM/2 pairs per k row
        const int p = i * 256 + t;
        const int krow = p / (BM / 2), c = (p % (BM / 2)) * 2;
        const int k = k0 + krow;
        const int kc = k < k_end ? k : k_end - 1;
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
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (!A_KCONTIG) {
        const int p = i * 256 + t;
        const int krow = p / (BM / 2), c = (p % (BM / 2)) * 2;
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

  const int fi = lane & 15, fk = lane >> 4;          // A fragment: 16 consecutive rows, k = lane >> 4
  const int fblk = (lane >> 2) & 3, fj = lane & 3;   // B fragment / result: block and column in block
  auto compute_slab = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < kGemmBK / 4; ++kk) {
      double af[AF], bf[16];
#pragma unroll
      for (int a = 0; a < AF; ++a) af[a] = As[buf][4 * kk + fk][wm * (16 * AF) + a * 16 + fi];
#pragma unroll
      for (int r = 0; r < 16; ++r) bf[r] = Bs[buf][4 * kk + fk][wn * 64 + 4 * ((fblk + r) & 15) + fj];
#pragma unroll
      for (int a = 0; a < AF; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          acc[a][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[a], bf[r], acc[a][r], 0, 0, 0);
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
  double local = 0.0;
  // acc[a][r], lane l: row = 16 a + 4 blk + (l >> 4), col = 4 ((blk + r) mod 16) + (l & 3)
#pragma unroll
  for (int a = 0; a < AF; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * (16 * AF) + a * 16 + 4 * fblk + fk;
      const int col = n0 + wn * 64 + 4 * ((fblk + r) & 15) + fj;
      if (row < g.M && col < g.N) {
        if constexpr (EpiReduces<Epi>::value)
          local += epi((int)blockIdx.z, row, col, acc[a][r]);
        else
          epi((int)blockIdx.z, row, col, acc[a][r]);
      }
    }
  // optional per-workgroup reduction of the values the epilogue returns (e.g. sum of log-likelihood terms)
  if constexpr (EpiReduces<Epi>::value) {
    __syncthreads();                       // the LDS slabs are free: reuse the first words
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    double* red = &As[0][0][0];
    if (lane == 0) red[wave] = local;
    __syncthreads();
    if (t == 0) epi.part[(int64_t)blockIdx.z * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

inline int gemm_tiles(int x, int b) { return (x + b - 1) / b; }

template <bool A_KCONTIG, class Epi>
inline void gemm_f64_launch(hipStream_t st, GemmArgs g, int splits, int n_cu, const Epi& epi) {
  g.tiles_n = gemm_tiles(g.N, kGemmBN);
  if (splits < 1) splits = 1;
  int ks = gemm_tiles(g.K, splits);
  g.k_split = gemm_tiles(ks, kGemmBK) * kGemmBK;
  // 128-row tiles (one workgroup per CU, 64 accumulators per lane) unless they would leave CUs idle
  const long tm128 = gemm_tiles(g.M, 128);
  const long tiles128 = (g.tri_mode == 2 ? tm128 * (tm128 + 1) / 2 : tm128 * g.tiles_n) * splits;
  const int bm_rows = (20 * tiles128 >= 17L * n_cu) ? 128 : 64;
  g.tiles_m = gemm_tiles(g.M, bm_rows);
  long blocks = (long)g.tiles_m * g.tiles_n;
  if (g.tri_mode == 2) {
    blocks = 0;
    for (int bm = 0; bm < g.tiles_m; ++bm) {
      const int cnt = (bm * bm_rows + bm_rows - 1) / kGemmBN + 1;
      blocks += cnt < g.tiles_n ? cnt : g.tiles_n;
    }
  }
  const dim3 grid((unsigned)blocks, 1, (unsigned)splits);
  if (bm_rows == 128)
    hipLaunchKernelGGL((gemm_f64_kernel<A_KCONTIG, 4, Epi>), grid, dim3(256), 0, st, g, epi);
  else
    hipLaunchKernelGGL((gemm_f64_kernel<A_KCONTIG, 2, Epi>), grid, dim3(256), 0, st, g, epi);
}

}  // namespace vb
