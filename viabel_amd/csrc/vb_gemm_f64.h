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

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace vb {

typedef double d2v __attribute__((ext_vector_type(2)));
typedef double d4v __attribute__((ext_vector_type(4)));

constexpr int kGemmBK = 16;   // block tile: (32 AF) x (8 NB) x 16; AF / NB = A / B fragments per wave and k-step

#ifdef VB_GEMM_CLOCK
__device__ long long vb_gemm_dbg[8 * 4096];
#endif

struct GemmArgs {
  const double* A;
  const double* B;
  int64_t lda, ldb;
  int M, N, K;
  int tiles_m, tiles_n;
  int tri_mode;     // 0: dense; 1: B[k][j] == 0 for k > j (k-range cut per column block);
                    // 2: only output tiles with bm >= bn (lower triangle of a square C);
                    // 3: B[k][j] == 0 for k < j (the mirror image of 1; LDS-DMA kernel, otherwise computed densely)
  int k_split;      // K range per blockIdx.z (multiple of kGemmBK); splits = gridDim.z
  // batch mode (batch != 0): blockIdx.z selects one of gridDim.z independent products of the same shape -- operand
  // z starts batch_a / batch_b doubles after operand z - 1, every product runs over the whole K range and the
  // epilogue receives z as its `split` argument
  int batch = 0;
  int64_t batch_a = 0, batch_b = 0;
  // tri_mode 2, optional: explicit tile list -- block x computes tile (tile_map[2 x], tile_map[2 x + 1]) (a negative
  // row block: no tile), gridDim.x = tile_blocks.  Lets the caller place the tiles that share operand panels on the
  // same XCD (block x runs on XCD x % 8 and each XCD has its own L2)
  const int* tile_map = nullptr;
  int tile_blocks = 0;
  // tri_mode 1, optional: only the column blocks [bn_begin, bn_begin + bn_count) of the product (bn_count == 0: all) --
  // the launch covers tiles_m x bn_count tiles in the usual heavy-first order; rows, columns and k ranges keep their
  // global indices.  (The blocking full-rank call starts the sampling product of the heaviest column blocks while the rest
  // of the parameter is still crossing PCIe: vb_fullrank.hip, FrUpload.)
  int bn_begin = 0, bn_count = 0;
  // split products, optional (LDS-DMA kernel; the launcher clears it unless gridDim.z % 8 == 0): the workgroups of ONE split
  // -- which read the same k range of both operands -- are dispatched to ONE XCD.  Workgroups go to the XCDs round robin
  // in linear order (x fastest), so the gridDim.x tiles of a split land on all eight L2s and every L2 streams every
  // panel; remapped, workgroup L computes tile (L / 8) % gridDim.x of split L % 8 + 8 (L / (8 gridDim.x)).
  int xcd_group = 0;
  // wave-priority alternation (LDS-DMA kernel): workgroups of generation (linear id / prio_div) & 1 raise their
  // wave priority on even slabs, the others on odd slabs (0: off).  Set by the launcher to the number of CUs.
  int prio_div = 0;
  // optional start / stop events of the launch (hipExtLaunchKernel: the kernel's own begin / end timestamps)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
#ifdef VB_GEMM_CLOCK
  // probe builds only (tools/gemm_bench.hip): tri_mode 1 tiles whose column block lies outside [dbg_bn_min, dbg_bn_max]
  // leave at once -- "what does the heavy half of the triangle cost alone?"
  int dbg_bn_min = 0, dbg_bn_max = 1 << 30;
  int dbg_prio_slabs = 0;      // > 0: tiles of at most that many k slabs run at wave priority 3; < 0: tiles of more than -that many
#endif
};

// Epilogue functor interface:  void operator()(int split, int row, int col, double acc) const;
// called for every in-range element of the block tile.  A functor that defines `double* part` and
// returns double from operator() additionally gets the per-workgroup sum of its return values written to
// part[blockIdx.z * gridDim.x + blockIdx.x] (workgroups that exit early write nothing: zero `part` first).
template <class E, class = void>
struct EpiReduces : std::false_type {};
template <class E>
struct EpiReduces<E, std::void_t<decltype(std::declval<E>().part)>> : std::true_type {};

// A functor that defines `double* colsum` and `int64_t colsum_ld` (LDS-DMA kernel, A given k-major) additionally gets
// the column sums of A over the workgroup's k range, colsum[blockIdx.z * colsum_ld + m] = sum_k A[k][m], written once per
// row block and split: the operand tiles are in LDS anyway, so the sums cost a few LDS reads per slab instead of another
// pass over A.  tri_mode 2: by the waves of the row block's diagonal tile that lie above the diagonal and multiply
// nothing; otherwise by the workgroups of column block 0.
// EpiPairs: epilogues that can take two horizontally adjacent elements at once --
//   d2v pair(int split, int row, int col, double acc0, double acc1) const   (col even, col + 1 < N)
// stores (row, col) and (row, col + 1) as one 16-byte access and returns what operator() would have returned for each
// (ignored by non-reducing epilogues).  The LDS-DMA kernel's fragment assignment puts fragments 2q and 2q + 1 of a lane
// on adjacent columns, so its epilogue issues half the store instructions, each lane covering 16 contiguous bytes.
// Output bases are 256-B aligned and every row stride is even (round_up(., 16)), which is what the access relies on.
template <class E, class = void>
struct EpiPairs : std::false_type {};
template <class E>
struct EpiPairs<E, std::void_t<decltype(std::declval<E>().pair(0, 0, 0, 0.0, 0.0))>> : std::true_type {};

// EpiRowSums (LDS-DMA kernel, round 6): a functor that defines `double* rowpart` gets, besides its stores, the sum over the
// TILE's columns of a pair of per-element terms for every row of the tile:
//   d2v rows_pair(int row, int col, double acc0, double acc1) const   -- elements (row, col), (row, col + 1): stores, returns
//                                                                        the two terms, each summed over the two elements
//   d2v rows_one(int row, int col, double acc) const                  -- one element
//   void rows_out(int row, int bn, d2v total) const                   -- the row's terms over column block bn
// Order of a row's sum: a lane's column pairs in fragment order, the four lanes of the row (quad shuffles), the wave of the
// left half of the tile then the right one -- fixed, so the sums are reproducible.  (The multivariate t's DIS refresh takes
// log p and log prior of a diagonal-Gaussian target and prior out of its sampling product this way: vb_mvt.hip.)
template <class E, class = void>
struct EpiRowSums : std::false_type {};
template <class E>
struct EpiRowSums<E, std::void_t<decltype(std::declval<E>().rowpart)>> : std::true_type {};

template <class E, class = void>
struct EpiColsum : std::false_type {};
template <class E>
struct EpiColsum<E, std::void_t<decltype(std::declval<E>().colsum)>> : std::true_type {};

// Block tile (32 AF) x (8 NB): (AF, NB) = (4, 16) 128 x 128, (4, 8) 128 x 64, (2, 8) 64 x 64.  One wave per
// SIMD issues a 4x4x4 MFMA every 16.5 cycles, two waves sharing a SIMD one every 12.4 (measured,
// tools/gemm_bench.hip): every variant fits two workgroups per CU (<= 256 registers, <= 74 KB LDS) and the
// launcher picks the largest tile that still gives the chip two workgroups per CU.
template <bool A_KCONTIG, int AF, int NB, class Epi>
__global__ void __launch_bounds__(256, AF * NB > 32 ? 1 : 2) gemm_f64_kernel(const GemmArgs g, const Epi epi) {
  constexpr int BM = 32 * AF, BN = 8 * NB;
  constexpr int NA = BM / 32;   // staging iterations for the A tile
  constexpr int NBL = BN / 32;  // staging iterations for the B tile
  __shared__ double As[2][kGemmBK][BM + 16];
  __shared__ double Bs[2][kGemmBK][BN + 16];

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
      const int cnt = min(g.tiles_n, (bm * BM + BM - 1) / BN + 1);
      if (idx < cnt) break;
      idx -= cnt;
      ++bm;
    }
    bn = idx;
  } else if (g.tri_mode == 1) {
    // first half of the grid: heaviest column blocks (largest k range) in descending order; second half:
    // the light ones ascending, so the two workgroups a CU ends up with sum to the same k range
    const int tn = g.bn_count ? g.bn_count : g.tiles_n;
    const int idx = blockIdx.x / g.tiles_m, half = (tn + 1) / 2;
    bn = g.bn_begin + (idx < half ? tn - 1 - idx : idx - half);
    bm = blockIdx.x % g.tiles_m;
  } else {
    bn = blockIdx.x / g.tiles_m;
    bm = blockIdx.x % g.tiles_m;
  }
  const int m0 = bm * BM, n0 = bn * BN;
  int k_begin = g.batch ? 0 : blockIdx.z * g.k_split;
  int k_end = (g.batch || k_begin + g.k_split >= g.K) ? g.K : k_begin + g.k_split;
  const double* __restrict__ gA = g.A + (g.batch ? (int64_t)blockIdx.z * g.batch_a : 0);
  const double* __restrict__ gB = g.B + (g.batch ? (int64_t)blockIdx.z * g.batch_b : 0);
  if (g.tri_mode == 1) {
    const int kmax = n0 + BN;     // B[k][j] == 0 for k > j
    if (k_end > kmax) k_end = kmax;
  }

  double acc[AF][NB];
#pragma unroll
  for (int i = 0; i < AF; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = 0.0;

  // ---- global -> register staging ---------------------------------------------------------------
  d2v ra[NA], rb[NBL];
  unsigned keep = 0;   // validity bits of the staged values (2 per load: .x, .y), applied at the LDS store
  // Branch-free: indices are clamped into range and out-of-range values are zeroed when they are
  // written to LDS one iteration later -- nothing touches a loaded register before that, so the loads
  // have a whole slab of MFMAs to land (every operand row is padded to a multiple of 16 doubles, so the
  // 16-B load of a pair that straddles the logical edge stays inside the allocation).
  auto load_slab = [&](int k0) __attribute__((always_inline)) {
    keep = 0;
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      // B: BN / 2 pairs per k row
      const int p = i * 256 + t;
      const int krow = p / (BN / 2), c = (p % (BN / 2)) * 2;
      const int k = k0 + krow;
      const int kc = k < k_end ? k : k_end - 1;
      const int n = n0 + c;
      const int nc = n < g.N ? n : 0;
      rb[i] = *reinterpret_cast<const d2v*>(gB + (int64_t)kc * g.ldb + nc);
      keep |= (unsigned)(k < k_end && n < g.N) << (2 * i);
      keep |= (unsigned)(k < k_end && n + 1 < g.N) << (2 * i + 1);
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
        ra[i] = *reinterpret_cast<const d2v*>(gA + (int64_t)kc * g.lda + mc);
        keep |= (unsigned)(k < k_end && m < g.M) << (2 * (NBL + i));
        keep |= (unsigned)(k < k_end && m + 1 < g.M) << (2 * (NBL + i) + 1);
      } else {
        // A[m][k]: wave-load q covers 16 rows x 4 k-pairs; 16 consecutive lanes = 16 distinct rows
        const int q = i * 4 + wave;
        const int row = (q >> 1) * 16 + (lane & 15);
        const int kp = (q & 1) * 4 + (lane >> 4);
        const int m = m0 + row, ka = k0 + 2 * kp;
        const int mc = m < g.M ? m : 0;
        const int kac = ka < k_end ? ka : 0;
        ra[i] = *reinterpret_cast<const d2v*>(gA + (int64_t)mc * g.lda + kac);
        keep |= (unsigned)(m < g.M && ka < k_end) << (2 * (NBL + i));
        keep |= (unsigned)(m < g.M && ka + 1 < k_end) << (2 * (NBL + i) + 1);
      }
    }
  };
  auto store_slab = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      const int p = i * 256 + t;
      const int krow = p / (BN / 2), c = (p % (BN / 2)) * 2;
      d2v v = rb[i];
      v.x = (keep >> (2 * i)) & 1u ? v.x : 0.0;
      v.y = (keep >> (2 * i + 1)) & 1u ? v.y : 0.0;
      *reinterpret_cast<d2v*>(&Bs[buf][krow][c]) = v;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      d2v v = ra[i];
      v.x = (keep >> (2 * (NBL + i))) & 1u ? v.x : 0.0;
      v.y = (keep >> (2 * (NBL + i) + 1)) & 1u ? v.y : 0.0;
      if (!A_KCONTIG) {
        const int p = i * 256 + t;
        const int krow = p / (BM / 2), c = (p % (BM / 2)) * 2;
        *reinterpret_cast<d2v*>(&As[buf][krow][c]) = v;
      } else {
        const int q = i * 4 + wave;
        const int row = (q >> 1) * 16 + (lane & 15);
        const int kp = (q & 1) * 4 + (lane >> 4);
        As[buf][2 * kp][row] = v.x;
        As[buf][2 * kp + 1][row] = v.y;
      }
    }
  };

  const int fi = lane & 15, fk = lane >> 4;          // A fragment: 16 consecutive rows, k = lane >> 4
  const int fblk = (lane >> 2) & 3, fj = lane & 3;   // B fragment / result: block and column in block
  // Software pipeline of one slab iteration (slab s in LDS buffer `buf`, KS = 4 k-steps of 4):
  //   top     ds_write slab s+1 (global loads issued one iteration ago) into buf^1, then issue the
  //           global loads of slab s+2 into the same staging registers (the scheduler spreads the writes over
  //           k-step 0 and the loads over k-step 1, one per fragment read, so no queue fills up)
  //   kk<KS-1 MFMAs of k-step kk interleaved with the fragment ds_reads of k-step kk+1
  //   barrier (all waves have issued every read of slab s and every write of slab s+1)
  //   kk=KS-1 MFMAs interleaved with the fragment reads of k-step 0 of slab s+1 (from buf^1)
  // so neither the LDS fill nor the first fragment fetch of a slab is exposed; the only idle time is the
  // barrier skew.  buf^1 is free at the top of the iteration: its last readers issued their reads before
  // the previous barrier.  sched_group_barrier pins the read/MFMA interleave: left to itself the
  // scheduler hoists every fragment read of a slab in front of the first MFMA and the matrix pipe idles
  // while the LDS queue drains (20.6 instead of 16 cycles per MFMA, tools/gemm_bench.hip).
  double fa[2][AF], fb[2][NB];
  auto load_frags = [&](int buf, int kk, int set) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < AF; ++a) fa[set][a] = As[buf][4 * kk + fk][wm * (16 * AF) + a * 16 + fi];
#pragma unroll
    for (int r = 0; r < NB; ++r)
      fb[set][r] = Bs[buf][4 * kk + fk][wn * (4 * NB) + 4 * ((fblk + r) & (NB - 1)) + fj];
  };
  auto mfma_step = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < AF; ++a)
#pragma unroll
      for (int r = 0; r < NB; ++r)
        acc[a][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[set][a], fb[set][r], acc[a][r], 0, 0, 0);
  };
  constexpr int KS = kGemmBK / 4;
  constexpr int kReads = AF + NB, kMfma = AF * NB;
  constexpr int kPer = kMfma / kReads;        // MFMAs issued after each fragment read
  // extra: 0 = none, 1 = also one LDS write per group (the slab fill), 2 = also one global load per group
  auto interleave = [&](int extra) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < kReads; ++i) {
      if (extra == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // one DS write
      if (extra == 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one VMEM read
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one DS read
      __builtin_amdgcn_sched_group_barrier(0x008, kPer, 0);   // kPer MFMAs
    }
    __builtin_amdgcn_sched_group_barrier(0x008, kMfma - kPer * kReads, 0);
  };

#ifdef VB_GEMM_CLOCK
  const long long dbg_c0 = clock64(), dbg_w0 = wall_clock64();
#endif
  if (k_begin < k_end) {
    load_slab(k_begin);
    store_slab(0);
    load_slab(k_begin + kGemmBK);
    __syncthreads();
    load_frags(0, 0, 0);
    int buf = 0;
    // One uniform basic block per slab (keeps the accumulators in place; no peeled tail: loads beyond the
    // last slab are clamped into range and zeroed, and their LDS copy is never used).
    for (int k0 = k_begin; k0 < k_end; k0 += kGemmBK) {
#ifndef VB_GEMM_SKIP
#define VB_GEMM_SKIP 0     // tools/gemm_bench.hip builds ablation variants (bits: 1 global loads, 2 LDS fill, 4 barrier, 8 fragment reads)
#endif
      if (!(VB_GEMM_SKIP & 2)) store_slab(buf ^ 1);
      if (!(VB_GEMM_SKIP & 1)) load_slab(k0 + 2 * kGemmBK);
#pragma unroll
      for (int kk = 0; kk < KS - 1; ++kk) {
        if (!(VB_GEMM_SKIP & 8)) load_frags(buf, kk + 1, (kk + 1) & 1);
        mfma_step(kk & 1);
        interleave(kk == 0 ? 1 : (kk == 1 ? 2 : 0));
      }
      if (!(VB_GEMM_SKIP & 4)) __syncthreads();
      if (!(VB_GEMM_SKIP & 8)) load_frags(buf ^ 1, 0, KS & 1);
      mfma_step((KS - 1) & 1);
      interleave(0);
      buf ^= 1;
    }
  }

#ifdef VB_GEMM_CLOCK
  if (t == 0) {
    vb_gemm_dbg[2 * blockIdx.x] = clock64() - dbg_c0;
    vb_gemm_dbg[2 * blockIdx.x + 1] = wall_clock64() - dbg_w0;
  }
#endif
  // ---- epilogue -----------------------------------------------------------------------------------
  double local = 0.0;
  // acc[a][r], lane l: row = 16 a + 4 blk + (l >> 4), col = 4 ((blk + r) mod NB) + (l & 3)
#pragma unroll
  for (int a = 0; a < AF; ++a)
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      const int row = m0 + wm * (16 * AF) + a * 16 + 4 * fblk + fk;
      const int col = n0 + wn * (4 * NB) + 4 * ((fblk + r) & (NB - 1)) + fj;
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

}  // namespace vb

#include "vb_gemm_f64_dma.h"

namespace vb {

inline int gemm_tiles(int x, int b) { return (x + b - 1) / b; }

// upper bound on gridDim.x of any tile choice (sizes the `part` array of reducing epilogues)
inline int64_t gemm_max_blocks(int64_t M, int64_t N) { return ((M + 63) / 64) * ((N + 63) / 64); }

inline long gemm_count_blocks(const GemmArgs& g, int bm_rows, int bn_cols) {
  const int tm = gemm_tiles(g.M, bm_rows), tn = gemm_tiles(g.N, bn_cols);
  if (g.tri_mode != 2) return (long)tm * tn;
  long blocks = 0;
  for (int bm = 0; bm < tm; ++bm) {
    const int cnt = (bm * bm_rows + bm_rows - 1) / bn_cols + 1;
    blocks += cnt < tn ? cnt : tn;
  }
  return blocks;
}

// cfg: 0 = automatic, 1 = 128 x 128, 2 = 128 x 64, 3 = 64 x 64 block tiles (three LDS stages);
//      4 = 64 x 64, 5 = 128 x 64, 6 = 128 x 128 with two LDS stages (more workgroups per CU); LDS-DMA kernel only
//      7 = 64 x 32 with three, 8 = 64 x 32 with two LDS stages: LDS-DMA kernel, A[m][k], epilogues that neither reduce nor
//      take column sums (element values do not depend on the tile: bit-identical to the 64 x 64 tiles, tools/gemm_bench.hip)
// flags bit 0: force the register-staged kernel
// does a product of this shape go to the LDS-DMA kernel (which alone implements EpiColsum)?
inline bool gemm_uses_dma(const GemmArgs& g) {
  static const bool dma_ok = !(getenv("VB_GEMM_DMA") && atoi(getenv("VB_GEMM_DMA")) == 0);
  return dma_ok && g.K % kGemmBK == 0 && g.M > 0 && g.N > 0;
}

// returns gridDim.x of the launch (the number of output tiles; reducing epilogues write one partial per tile and split)
template <bool A_KCONTIG, class Epi>
inline unsigned gemm_f64_launch(hipStream_t st, GemmArgs g, int splits, int n_cu, const Epi& epi, int cfg = 0,
                                int flags = 0) {
  // 64 x 32 tiles: only where the result cannot depend on the tiling (no per-tile partial sums, no column sums)
  // (round 6: a REDUCING epilogue may take them too when the caller asks for it by cfg = 7 / 8 -- its per-tile partial sums
  // are then grouped by 64 x 32 tiles: another, equally fixed, summation order)
  constexpr bool kNarrowOk = A_KCONTIG && !EpiColsum<Epi>::value;
  constexpr bool kNarrowAuto = kNarrowOk && !EpiReduces<Epi>::value && !EpiRowSums<Epi>::value;
  if (splits < 1) splits = 1;
  int ks = g.batch ? g.K : gemm_tiles(g.K, splits);   // batch mode: `splits` is the number of products
  g.k_split = gemm_tiles(ks, kGemmBK) * kGemmBK;
  static const int cfg_env = getenv("VB_GEMM_CFG") ? atoi(getenv("VB_GEMM_CFG")) : 0;   // experiments: force a tile
  if (cfg == 0 && cfg_env >= 1 && cfg_env <= 8) cfg = cfg_env;
  // operands straight into LDS (vb_gemm_f64_dma.h) when every k range is a whole number of slabs
  const bool dma = gemm_uses_dma(g) && !(flags & 1);
  if (g.tri_mode == 3 && (!dma || splits != 1 || g.batch)) g.tri_mode = 0;      // the zeros are multiplied instead of skipped
  if (cfg == 0) {
    if (dma && (g.tri_mode == 1 || g.tri_mode == 3) && splits == 1 && gemm_count_blocks(g, 128, 64) < 4L * n_cu) {
      // k ranges grow with the column block: with only a couple of tiles per CU the long ones finish alone.  The
      // 64 x 64 tiles (three or four workgroups per CU, heaviest first) even that out: 89.9 -> 85.9 us at
      // 4096 x 1024 x 1024 (tools/gemm_bench.hip; pairing a long and a short block inside one workgroup measured
      // the same 85.9 us, so the simpler launch order is kept)
      // round 3: with the priority alternation off for these launches (below) the two-stage variant -- 32 KB, four
      // workgroups per CU instead of three -- is the faster one: 82.4 -> 78.2 us in the headline pipeline
      cfg = 4;
      // at most two 64 x 64 tiles per CU (D <= 512 at 4096 samples): the launch is one heavy tile's k loop, and half
      // as wide a tile halves the MFMAs of every slab of it -- 4096 x 256 x 256: 15.2 -> 11.9 us, 4096 x 512 x 512:
      // 28.9 -> 26.9 us; with more tiles per CU it loses (16 384 x 256 x 256: 29.3 -> 30.8 us, D = 1024: 82 -> 92 us)
      // (tools/tile32_probe.sh)
      if (kNarrowAuto && !g.batch && gemm_count_blocks(g, 64, 64) <= 2L * n_cu) cfg = 8;
    } else if (dma && g.tri_mode == 0 && g.batch == 0 && gemm_count_blocks(g, 128, 128) * splits >= 2L * n_cu) {
      // large dense products: 128 x 128 tiles with TWO LDS stages (64 KB: two workgroups per CU).  Per MFMA a third
      // fewer fragment reads and LDS-DMA pieces than 128 x 64: 69.3 - 70.2 against 66.0 - 67.5 TFLOP/s on
      // 4096 x {2048, 4096, 8192} x {2000 ... 4096} (tools/gemm_bench.hip).  Below two tiles per CU it loses
      // (4096 x 1024 x 1024: 148 against 136 us), and the three-stage variant (96 KB, one workgroup per CU) always did
      cfg = 6;
    } else if (!dma && gemm_count_blocks(g, 128, 128) * splits >= 2L * n_cu) {
      // largest tile that gives every CU two workgroups; the LDS-DMA kernel's 128 x 128 tile needs 96 KB of LDS
      // (one workgroup per CU) and measures slower than its 128 x 64 tile at every shape tried, so it is skipped
      cfg = 1;
    } else if (gemm_count_blocks(g, 128, 64) * splits * 100 >= 190L * n_cu) {
      cfg = 2;    // (lower-triangular D = 1024 product in 7 row slabs: 504 workgroups on 256 CUs)
    } else {
      cfg = 3;
      // dense products with at most one 64 x 64 tile per CU: 4096 x 256 x 256 17.2 -> 15.8 us (at two per CU it loses:
      // 4096 x 512 x 512 42.9 -> 45.4 us)
      if (kNarrowAuto && dma && g.tri_mode == 0 && !g.batch && gemm_count_blocks(g, 64, 64) * splits <= (long)n_cu) cfg = 7;
    }
  }
  if (!kNarrowOk && cfg >= 7) cfg = 4;
  if (!dma && cfg > 3) cfg = (cfg == 4 || cfg >= 7) ? 3 : cfg == 6 ? 1 : 2;
  const int bm_rows = (cfg == 3 || cfg == 4 || cfg >= 7) ? 64 : 128, bn_cols = cfg >= 7 ? 32 : (cfg == 1 || cfg == 6) ? 128 : 64;
  g.tiles_m = gemm_tiles(g.M, bm_rows);
  g.tiles_n = gemm_tiles(g.N, bn_cols);
  if (g.bn_count && (g.tri_mode != 1 || g.bn_begin + g.bn_count > g.tiles_n)) g.bn_begin = g.bn_count = 0;
  if (splits % 8 != 0 || g.batch) g.xcd_group = 0;
  const dim3 grid(g.tile_map ? (unsigned)g.tile_blocks
                             : g.bn_count ? (unsigned)(g.tiles_m * g.bn_count) : (unsigned)gemm_count_blocks(g, bm_rows, bn_cols), 1,
                  (unsigned)splits);
  static const int prio_env = getenv("VB_GEMM_PRIO") ? atoi(getenv("VB_GEMM_PRIO")) : 1;
  // (triangular k ranges: the tiles of a CU differ in length anyway, and the alternation costs 1 - 3 us there --
  // 4096 x 768 x 768: 55.5 -> 52.7 us, 512: 30.1 -> 29.1 us, 1024: unchanged, tools/gemm_bench.hip with VB_GEMM_PRIO=0)
  g.prio_div = (prio_env && g.tri_mode != 1 && g.tri_mode != 3) ? n_cu : 0;
  if (dma) {
    if (cfg == 1) gemm_f64_dma_launch<A_KCONTIG, 4, 16, 3, Epi>(st, g, grid, epi);
    else if (cfg == 2) gemm_f64_dma_launch<A_KCONTIG, 4, 8, 3, Epi>(st, g, grid, epi);
    else if (cfg == 3) gemm_f64_dma_launch<A_KCONTIG, 2, 8, 3, Epi>(st, g, grid, epi);
    else if (cfg == 4) gemm_f64_dma_launch<A_KCONTIG, 2, 8, 2, Epi>(st, g, grid, epi);
    else if (cfg == 6) gemm_f64_dma_launch<A_KCONTIG, 4, 16, 2, Epi>(st, g, grid, epi);
    else if (cfg >= 7) {
      if constexpr (kNarrowOk) {
        if (cfg == 7) gemm_f64_dma_launch<A_KCONTIG, 2, 4, 3, Epi>(st, g, grid, epi);
        else gemm_f64_dma_launch<A_KCONTIG, 2, 4, 2, Epi>(st, g, grid, epi);
      }
    }
    else gemm_f64_dma_launch<A_KCONTIG, 4, 8, 2, Epi>(st, g, grid, epi);
    return grid.x;
  }
  if (cfg == 1)
    hipExtLaunchKernelGGL((gemm_f64_kernel<A_KCONTIG, 4, 16, Epi>), grid, dim3(256), 0, st, g.ev0, g.ev1, 0, g, epi);
  else if (cfg == 2)
    hipExtLaunchKernelGGL((gemm_f64_kernel<A_KCONTIG, 4, 8, Epi>), grid, dim3(256), 0, st, g.ev0, g.ev1, 0, g, epi);
  else
    hipExtLaunchKernelGGL((gemm_f64_kernel<A_KCONTIG, 2, 8, Epi>), grid, dim3(256), 0, st, g.ev0, g.ev1, 0, g, epi);
  return grid.x;
}

}  // namespace vb
