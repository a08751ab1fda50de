// fp64 MFMA GEMM, operand tiles moved global -> LDS by `global_load_lds_dwordx4` (no VGPR staging, no
// ds_write, no zeroing selects): the variant the launcher uses when every k range is a whole number of
// 16-deep slabs (K % 16 == 0).  Same tiles, fragment arrangement and epilogue contract as gemm_f64_kernel in
// vb_gemm_f64.h (included from there); what changes is how a slab gets into LDS:
//
//  * a wave-load moves 64 x 16 B = 1 KiB to LDS addresses M0 + 16 * lane, i.e. LDS rows are dense (no pad
//    columns).  Bank conflicts are a matter of which lanes the hardware services together (MI355X_MICROARCH.md, LDS):
//      k-major tiles [k][W] (B, and A given as A[k][m]) are read with ds_read_b128, fragment pairs per lane (see
//        frag_row / frag_col); a service group of that instruction is 16 lanes of four different block slots, and
//        the fragment assignment gives the four slots four different 64-B column groups: conflict-free with the tile
//        stored as it is in memory (SQ_LDS_BANK_CONFLICT = 0; an XOR of odd k rows -- what ds_read_b64 needed --
//        makes it 2-way);
//      A given as A[m][k], tile [m][16], read with ds_read2st64_b64:  pair slot p of row m holds k-pair
//        p ^ ((m >> 1) & 7), applied on the *global* side (each lane picks which 16-B pair it fetches) and undone in
//        the fragment-read addresses (lane constants) -- 2-way instead of 16-way (16-B pieces cannot separate the
//        two rows of a pair).
//  * three LDS stages, prefetch distance two: at the top of iteration s the loads of slab s+2 are issued into
//    the stage whose last readers passed the previous barrier; before the barrier of iteration s each wave
//    waits with a counted vmcnt until only the loads of slab s+2 are outstanding, so a slab has a whole
//    iteration (~2 us) more than in the register-staged kernel to arrive.  The barrier is a bare s_barrier:
//    __syncthreads() would add a fence that drains vmcnt to zero and with it the prefetch.
//  * rows / columns beyond M / N are fetched from clamped (in-range) addresses; they only feed output rows /
//    columns that the epilogue masks.
#pragma once

namespace vb {

typedef __attribute__((address_space(1))) const void* gemm_gptr;
typedef __attribute__((address_space(3))) void* gemm_lptr;

// In-launch dependencies of a tile (the fused full-rank evaluation, vb_fullrank_fused.h): a `Dep` object is consulted
// by every wave before it requests a k slab (`before_slab(bm, bn, slab, first)`: a consumer polls the producer's flag there), names the
// cache policy of the A operand's loads (`kAuxA`: 16 = sc1, reads data another workgroup of the same launch stored
// write-through) and is told when the tile's results have left (`publish`).  Stand-alone launches use GemmNoDep: no
// code.
struct GemmNoDep {
  static constexpr int kAuxA = 0;
  __device__ __forceinline__ void before_slab(int, int, int, bool) const {}
  __device__ __forceinline__ void publish(int, int, int) const {}
};

// One output tile of the product: block x of a launch with gx blocks per split, split (or batch index) bz.
template <bool A_KCONTIG, int AF, int NB, int STAGES, class Epi, class Dep>
__device__ __forceinline__ void gemm_f64_dma_tile(const GemmArgs& g, const Epi& epi, const Dep& dep, const int bx,
                                                  const int bz, const int gx) {
  constexpr int BM = 32 * AF, BN = 8 * NB;
  constexpr int kStages = STAGES;
  static_assert(STAGES == 2 || STAGES == 3, "two or three LDS stages");
  constexpr int kATile = BM * kGemmBK, kBTile = kGemmBK * BN;      // doubles per stage
  constexpr int kAUnits = kATile / 128, kBUnits = kBTile / 128;    // 1-KiB wave-loads per stage
  constexpr int UPW = (kAUnits + kBUnits) / 4;                     // wave-loads per wave per slab
  static_assert((kAUnits + kBUnits) % 4 == 0, "units must divide over the 4 waves");
  extern __shared__ double gemm_lds[];
  double* As = gemm_lds;                         // [kStages][kATile]
  double* Bs = gemm_lds + kStages * kATile;      // [kStages][kBTile]

  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
#ifdef VB_GEMM_CLOCK
  const long long dbg_t0 = clock64(), dbg_w0 = wall_clock64();
  long long dbg_t1 = dbg_t0, dbg_t2 = dbg_t0;
#endif

  // ---- tile assignment (as gemm_f64_kernel) -----------------------------------------------------
  double local = 0.0;
  int bm, bn;
  if (g.tri_mode == 2 && g.tile_map) {
    bm = g.tile_map[2 * bx];
    bn = g.tile_map[2 * bx + 1];
    if (bm < 0) {                        // padding entry of the tile list (uniform for the workgroup)
      if constexpr (EpiReduces<Epi>::value)
        if (t == 0) epi.part[(int64_t)bz * gx + bx] = 0.0;
      return;
    }
  } else if (g.tri_mode == 2) {
    int idx = bx;
    bm = 0;
    for (;;) {
      const int cnt = min(g.tiles_n, (bm * BM + BM - 1) / BN + 1);
      if (idx < cnt) break;
      idx -= cnt;
      ++bm;
    }
    bn = idx;
  } else if (g.tri_mode == 1 || g.tri_mode == 3) {
    const int tn = (g.tri_mode == 1 && g.bn_count) ? g.bn_count : g.tiles_n;      // (a column-block sub-range: GemmArgs)
    const int idx = bx / g.tiles_m, half = (tn + 1) / 2;
    bn = idx < half ? tn - 1 - idx : idx - half;
    if (g.tri_mode == 3) bn = g.tiles_n - 1 - bn;      // mirrored triangle: column block 0 has the longest k range
    else bn += g.bn_begin;
    bm = bx % g.tiles_m;
#ifdef VB_GEMM_CLOCK
    if (bn < g.dbg_bn_min || bn > g.dbg_bn_max) return;
#endif
  } else {
    bn = bx / g.tiles_m;
    bm = bx % g.tiles_m;
  }
  const int m0 = bm * BM, n0 = bn * BN;
  int k_begin = g.batch ? 0 : bz * g.k_split;
  int k_end = (g.batch || k_begin + g.k_split >= g.K) ? g.K : k_begin + g.k_split;
  if (g.tri_mode == 3) {         // B[k][j] == 0 for k < j: the k range of column block bn starts at its first column
    const int kmin = n0 / kGemmBK * kGemmBK;
    if (k_begin < kmin) k_begin = kmin < k_end ? kmin : k_end;
  }
  const double* __restrict__ gA = g.A + (g.batch ? (int64_t)bz * g.batch_a : 0);
  const double* __restrict__ gB = g.B + (g.batch ? (int64_t)bz * g.batch_b : 0);
  if (g.tri_mode == 1) {
    const int kmax = n0 + BN;
    if (k_end > kmax) k_end = kmax;
  }
  // tri_mode 2: a wave whose whole sub-tile lies above the diagonal (first column > last row) has nothing to
  // compute -- it still moves its share of the operands and keeps the barriers, but issues no MFMAs, which
  // leaves its SIMD's matrix pipe to the co-resident workgroup
  const bool idle_wave = g.tri_mode == 2 && n0 + wn * (4 * NB) > m0 + wm * (16 * AF) + 16 * AF - 1;
  // column sums of the A operand (see EpiColsum): thread -> column cs_m of the tile, k rows cs_k0 .. cs_k0 + kCsRows - 1
  constexpr bool kColsum = !A_KCONTIG && EpiColsum<Epi>::value;
  constexpr int kCsRows = kGemmBK * BM / 256;
  const bool cs_wg = kColsum && g.tri_mode != 2 && bn == 0;
  const int cs_m = t & (BM - 1), cs_k0 = (t / BM) * kCsRows;
  double cs = 0.0;
  // tri_mode 2: the column sums of row block bm are formed in the row's LAST tile (the one on the diagonal) by its
  // waves wm == 0 (for 64-row tiles: the one wave wm == 0, wn == 1) -- exactly the waves whose sub-tile lies above the
  // diagonal and that issue no MFMAs there (128 x 64 and 64 x 64 tiles): one lane per column, all 16 k rows of a slab
  // per lane, so no cross-thread combine and nothing added to the waves that
  // multiply.  (Done by the workgroups of column block 0, the 8 LDS reads per thread and slab made those 56
  // workgroups the last to finish: 88 against 81 us for the whole product at D = 1024.)
  bool cs_lane = false;
  int cs_col = 0;
  if constexpr (kColsum) {
    if (g.tri_mode == 2) {
      const int cnt = min(g.tiles_n, (bm * BM + BM - 1) / BN + 1);
      const bool mine = BM == 64 ? (wm == 0 && wn == 1) : wm == 0;
      cs_lane = bn == cnt - 1 && mine;
      cs_col = BM == 64 ? lane : wn * 64 + lane;
    }
  }
  const int nslabs = (k_end - k_begin) / kGemmBK;
  // tri_mode 1: B[k][j] == 0 for k > j, so a wave whose last column is j_last has nothing to multiply in the slabs
  // that start beyond it -- in a tile's diagonal block the waves of the left half skip the slabs of the lower half
  // (1/34 of the MFMAs at D = 1024); like an idle wave it keeps moving operands and meeting the barriers
  int my_nslabs = nslabs;
  if (g.tri_mode == 1) {
    const int j_last = n0 + wn * (4 * NB) + 4 * NB - 1;
    const int need = (j_last - k_begin) / kGemmBK + 1;       // slabs with a k <= j_last
    my_nslabs = need < nslabs ? (need > 0 ? need : 0) : nslabs;
  }

  double acc[AF][NB];
#pragma unroll
  for (int i = 0; i < AF; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = 0.0;

  // ---- global source of each of this wave's units (slab 0), advanced by one slab per issue -----------
  // unit q = u * 4 + wave; q < kAUnits: A unit, else B unit.
  const double* src[UPW];
  int64_t step[UPW];
  int lds_off[UPW];          // doubles from the start of the stage region (A region then B region)
#pragma unroll
  for (int u = 0; u < UPW; ++u) {
    const int q = u * 4 + wave;
    if (q < kAUnits) {
      if (A_KCONTIG) {       // 8 rows x 8 pairs per unit
        const int row = q * 8 + (lane >> 3), p = lane & 7;
        const int kp = p ^ ((row >> 1) & 7);
        int m = m0 + row;
        m = m < g.M ? m : g.M - 1;
        src[u] = gA + (int64_t)m * g.lda + k_begin + 2 * kp;
        step[u] = kGemmBK;
      } else {               // k-major: (BM / 2) pairs per k row
        constexpr int PPR = BM / 2, RPU = 64 / PPR > 0 ? 64 / PPR : 1;     // pairs per row, rows per unit
        constexpr int UPR = PPR >= 64 ? PPR / 64 : 1;                      // units per row (wide tiles)
        const int krow = (PPR >= 64) ? q / UPR : q * RPU + lane / PPR;
        const int p = (PPR >= 64) ? (q % UPR) * 64 + lane : lane % PPR;
        const int c = p;
        int64_t col = m0 + 2 * c;
        col = col < g.lda - 1 ? col : g.lda - 2;
        src[u] = gA + (int64_t)(k_begin + krow) * g.lda + col;
        step[u] = (int64_t)kGemmBK * g.lda;
      }
      lds_off[u] = q * 128;
    } else {
      const int qb = q - kAUnits;
      constexpr int PPR = BN / 2, RPU = 64 / PPR > 0 ? 64 / PPR : 1;
      constexpr int UPR = PPR >= 64 ? PPR / 64 : 1;
      const int krow = (PPR >= 64) ? qb / UPR : qb * RPU + lane / PPR;
      const int p = (PPR >= 64) ? (qb % UPR) * 64 + lane : lane % PPR;
      const int c = p;
      int64_t col = n0 + 2 * c;
      col = col < g.ldb - 1 ? col : g.ldb - 2;
      src[u] = gB + (int64_t)(k_begin + krow) * g.ldb + col;
      step[u] = (int64_t)kGemmBK * g.ldb;
      lds_off[u] = kStages * kATile + qb * 128;
    }
  }
  // stage `st` of the A region starts at st * kATile, of the B region at kStages * kATile + st * kBTile
  auto issue = [&](int st) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
      const int q = u * 4 + wave;
      double* dst = gemm_lds + lds_off[u] + st * (q < kAUnits ? kATile : kBTile);
      if (q < kAUnits) __builtin_amdgcn_global_load_lds((gemm_gptr)src[u], (gemm_lptr)dst, 16, 0, Dep::kAuxA);
      else __builtin_amdgcn_global_load_lds((gemm_gptr)src[u], (gemm_lptr)dst, 16, 0, 0);
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < UPW; ++u) src[u] += step[u];
  };

  // ---- fragment addresses (lane constants; doubles) -------------------------------------------------------
  const int fi = lane & 15, fk = lane >> 4;
  const int fblk = (lane >> 2) & 3, fj = lane & 3;
  // Which tile row / column a lane's fragment element stands for is bookkeeping: block slot blk of the instruction
  // multiplies whatever 4 rows of A and 4 columns of B the lanes of that slot supply.  The assignment is chosen so
  // that the two values a lane needs for fragments 2q and 2q + 1 are NEIGHBOURS in the k-major LDS row, i.e. one
  // ds_read_b128 instead of two ds_read_b64 (half the LDS instructions of a k-step; the 16 lanes of one service
  // pass still cover 256 consecutive bytes = every bank once):
  //   B: fragment r, slot blk, lane column j -> column 8 ((blk + r / 2) mod NB/2) + 2 j + (r & 1) of the wave's 4 NB
  //      (for a fixed slot, r = 0 .. NB-1 visits every column of the wave exactly once);
  //   A given as A[k][m]: fragment a, lane row i (slot i / 4) -> row 32 (a / 2) + 2 i + (a & 1) of the wave's 16 AF.
  // A given as A[m][k] keeps 16 consecutive rows per fragment (its reads are already merged, two fragments per
  // ds_read2st64_b64).  The per-element order of the k sum does not depend on the assignment: results are bit-identical
  // to the register-staged kernel's.
  auto frag_row = [&](int a, int i) __attribute__((always_inline)) {
    return A_KCONTIG ? wm * (16 * AF) + a * 16 + i : wm * (16 * AF) + 32 * (a >> 1) + 2 * i + (a & 1);
  };
  auto frag_col = [&](int r, int blk, int j) __attribute__((always_inline)) {
    return wn * (4 * NB) + 8 * ((blk + (r >> 1)) & (NB / 2 - 1)) + 2 * j + (r & 1);
  };
  static_assert(AF % 2 == 0 && NB % 2 == 0, "fragments are read in pairs");
  // A[m][k] tile: fragment a of k-step kk at a_off[kk] + 256 a (16 rows further: (row >> 1) & 7 unchanged);
  // A[k][m] tile: fragment a of k-step kk at a_off[a] + 4 kk BM (parity of k = parity of fk)
  constexpr int kAOffs = A_KCONTIG ? kGemmBK / 4 : AF;
  int a_off[kAOffs];
  int b_off[NB];                 // B fragment r of k-step 0; k-step kk adds 4 * kk * BN
  if (A_KCONTIG) {
#pragma unroll
    for (int kk = 0; kk < kGemmBK / 4; ++kk) {
      const int k = 4 * kk + fk;
      const int row = wm * (16 * AF) + fi;
      a_off[kk < kAOffs ? kk : 0] = row * kGemmBK + 2 * ((k >> 1) ^ ((row >> 1) & 7)) + (k & 1);
    }
  } else {
#pragma unroll
    for (int a = 0; a < AF; ++a) {
      const int m = frag_row(a, fi);
      a_off[a < kAOffs ? a : 0] = fk * BM + m;
    }
  }
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    const int col = frag_col(r, fblk, fj);
    b_off[r] = fk * BN + col;
  }

  double fa[2][AF], fb[2][NB];
  auto load_frags = [&](int st, int kk, int set) __attribute__((always_inline)) {
#ifdef VB_ABL_NO_READS
    if (kk >= 0) return;
#endif
    const double* as = As + st * kATile;
    const double* bs = Bs + st * kBTile;
    // The compiler merges the AF reads below into ds_read2st64_b64 pairs, which are serviced 16 lanes at a time
    // on 32 banks: 2-way conflicts for the [m][16] tile (SQ_LDS_BANK_CONFLICT 8.4 M of 25 M LDS cycles per
    // 4096 x 1024 x 1024 launch).  Forcing single ds_read_b64 (opaque addresses) removed the conflicts and halved
    // the LDS cycles but ran 5 % slower (157 vs 149 us): the extra address arithmetic and the lost scheduling
    // freedom cost more than the LDS time, which the MFMAs hide anyway.
    if constexpr (A_KCONTIG) {
#pragma unroll
      for (int a = 0; a < AF; ++a) fa[set][a] = as[a_off[kk < kAOffs ? kk : 0] + a * (16 * kGemmBK)];
    } else {
#pragma unroll
      for (int a = 0; a < AF; a += 2) {
        const d2v v = *reinterpret_cast<const d2v*>(as + a_off[a < kAOffs ? a : 0] + 4 * kk * BM);
        fa[set][a] = v.x, fa[set][a + 1] = v.y;
      }
    }
#pragma unroll
    for (int r = 0; r < NB; r += 2) {
      const d2v v = *reinterpret_cast<const d2v*>(bs + b_off[r] + 4 * kk * BN);
      fb[set][r] = v.x, fb[set][r + 1] = v.y;
    }
  };
  auto mfma_step = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < AF; ++a)
#pragma unroll
      for (int r = 0; r < NB; ++r)
        acc[a][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[set][a], fb[set][r], acc[a][r], 0, 0, 0);
  };
  constexpr int KS = kGemmBK / 4;
  // LDS *instructions* per k-step as the compiler emits them: AF / 2 for the A fragments (ds_read2st64_b64 pairs of
  // the [m][16] tile, ds_read_b128 pairs of the k-major tile) and NB / 2 ds_read_b128 for the B fragments.  The
  // interleave pattern below must ask for exactly that many DS groups: when it asked for more than a step has, the
  // scheduler filled the surplus with the NEXT step's first read, and that read then sat right in front of the step
  // boundary's s_waitcnt lgkmcnt(0) -- one exposed LDS round trip per k-step.
  // (Measured and dropped: spreading the reads over only the first two thirds of a step's MFMAs, and a scheduling
  // fence at the step boundary -- DESIGN 4.3.)
  constexpr int kReads = AF / 2 + NB / 2, kMfma = AF * NB;
  constexpr int kPer = kMfma / kReads;
  constexpr int kAhead = STAGES - 1;          // slabs in flight ahead of the one being multiplied
  auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < kReads; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, kPer, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, kMfma - kPer * kReads, 0);
  };
  // s_waitcnt vmcnt(n) only (gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14)
  auto wait_vm = [&](auto n) __attribute__((always_inline)) {
    constexpr int N = decltype(n)::value;
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
  };

  const int slab0 = k_begin / kGemmBK;
  if (nslabs > 0) {
    dep.before_slab(bm, bn, slab0, true);
    issue(0);                          // slab 0
    if (kAhead == 2) {
      if (nslabs > 1) {
        advance();
        dep.before_slab(bm, bn, slab0 + 1, false);
      }
      issue(1);                        // slab 1 (or slab 0 again: keeps the vmcnt arithmetic uniform)
    }
    wait_vm(std::integral_constant<int, UPW*(kAhead - 1)>());
    __asm__ volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __asm__ volatile("" ::: "memory");
    if (!idle_wave && my_nslabs > 0) load_frags(0, 0, 0);
    int st = 0;
#ifdef VB_GEMM_CLOCK
    dbg_t1 = clock64();
#endif
    // Two (or three) workgroups share a CU, one wave of each per SIMD, and the SIMD's issue arbiter serves the
    // oldest wave first: left alone, the older workgroup runs at its single-wave pace, the younger one gets the
    // gaps, and after the older one is gone the younger finishes alone at ~60 % matrix-pipe utilisation (measured:
    // workgroup lifetimes of 100 and 140 us side by side in the dense 4096 x 1024 x 1024 product).  Alternating the
    // wave priority slab by slab, in opposite phase for the two generations of workgroups, lets them progress at
    // the same pace and finish together.
    const int prio_phase = g.prio_div > 0 ? (int)((bx + gx * bz) / (unsigned)g.prio_div) & 1 : -1;
#ifdef VB_GEMM_CLOCK
    if (g.dbg_prio_slabs > 0 && nslabs <= g.dbg_prio_slabs) __builtin_amdgcn_s_setprio(3);
    if (g.dbg_prio_slabs < 0 && nslabs > -g.dbg_prio_slabs) __builtin_amdgcn_s_setprio(3);
#endif
    for (int s = 0; s < nslabs; ++s) {
      if (prio_phase >= 0) {
        if ((s ^ prio_phase) & 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
      const int st1 = st == kStages - 1 ? 0 : st + 1;
      const int st_new = kAhead == 2 ? (st1 == kStages - 1 ? 0 : st1 + 1) : st1;   // the stage slab s + kAhead goes to
      if (s + kAhead < nslabs) {            // beyond the end: re-fetch the last slab into a free stage
        advance();
        dep.before_slab(bm, bn, slab0 + s + kAhead, false);
      }
#ifndef VB_ABL_NO_DMA     // timing ablations (tools/gemm_bench.hip): results are wrong with any of them defined
      issue(st_new);
#endif
      if constexpr (kColsum) {
        if (cs_wg) {
          const double* as = As + st * kATile;
#pragma unroll
          for (int i = 0; i < kCsRows; ++i) {
            const int k = cs_k0 + i;
            cs += as[k * BM + cs_m];
          }
        }
        if (cs_lane) {
          const double* as = As + st * kATile + cs_col;
#pragma unroll
          for (int k0 = 0; k0 < kGemmBK; k0 += 4) {      // four reads in flight (the 128 x 128 tile has no registers to spare)
            double v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = as[(k0 + k) * BM];
#pragma unroll
            for (int k = 0; k < 4; ++k) cs += v[k];
          }
        }
      }
      if (!idle_wave && s < my_nslabs) {
#pragma unroll
        for (int kk = 0; kk < KS - 1; ++kk) {
          load_frags(st, kk + 1, (kk + 1) & 1);
          mfma_step(kk & 1);
          interleave();
        }
      }
#ifndef VB_ABL_NO_DMA
      wait_vm(std::integral_constant<int, UPW*(kAhead - 1)>());     // slab s + 1 has landed (this wave's share)
#endif
#ifndef VB_ABL_NO_LGKM
      __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): this wave's reads of stage st are done
#endif
      __asm__ volatile("" ::: "memory");
#ifndef VB_ABL_NO_BARRIER
      __builtin_amdgcn_s_barrier();
#endif
      __asm__ volatile("" ::: "memory");
      if (!idle_wave && s < my_nslabs) {
        load_frags(st1, 0, KS & 1);
        mfma_step((KS - 1) & 1);
        interleave();
      }
      st = st1;
    }
    __builtin_amdgcn_s_waitcnt(0);     // nothing in flight into LDS when the epilogue reuses it
    __builtin_amdgcn_s_barrier();
#ifdef VB_GEMM_CLOCK
    dbg_t2 = clock64();
#endif
  }

  if constexpr (kColsum) {
    if (cs_lane && m0 + cs_col < g.M) epi.colsum[(int64_t)bz * epi.colsum_ld + m0 + cs_col] = cs;
    if (cs_wg) {                 // combine the k groups in fixed order (the slabs are no longer needed)
      gemm_lds[t] = cs;
      __syncthreads();
      if (t < BM && m0 + t < g.M) {
        double tot = gemm_lds[t];
#pragma unroll
        for (int q = 1; q < 256 / BM; ++q) tot += gemm_lds[q * BM + t];
        epi.colsum[(int64_t)bz * epi.colsum_ld + m0 + t] = tot;
      }
      __syncthreads();
    }
  }
  // ---- epilogue with per-row sums over the tile's columns (EpiRowSums, vb_gemm_f64.h) -----------------------------------
  if constexpr (EpiRowSums<Epi>::value) {
    d2v* rsum = reinterpret_cast<d2v*>(gemm_lds);      // [2 (left / right half of the tile)][BM]; the stages are free by now
#pragma unroll
    for (int a = 0; a < AF; ++a) {
      const int lrow = frag_row(a, 4 * fblk + fk), row = m0 + lrow;      // the same row for every fragment r of this lane
      d2v s = (d2v){0.0, 0.0};
#pragma unroll
      for (int r = 0; r < NB; r += 2) {
        const int col = n0 + frag_col(r, fblk, fj);
        if (row < g.M && col + 1 < g.N) {
          s += epi.rows_pair(row, col, acc[a][r], acc[a][r + 1]);
        } else {
#pragma unroll
          for (int q = 0; q < 2; ++q)
            if (row < g.M && col + q < g.N) s += epi.rows_one(row, col + q, acc[a][r + q]);
        }
      }
      s.x += __shfl_xor(s.x, 1, 64), s.y += __shfl_xor(s.y, 1, 64);      // the row's four lanes: fj = lane & 3
      s.x += __shfl_xor(s.x, 2, 64), s.y += __shfl_xor(s.y, 2, 64);
      if (fj == 0) rsum[wn * BM + lrow] = s;
    }
    __syncthreads();
    if (t < BM && m0 + t < g.M) epi.rows_out(m0 + t, bn, rsum[t] + rsum[BM + t]);
    dep.publish(bm, bn, bz);
    return;
  }
  // ---- epilogue (as gemm_f64_kernel) -------------------------------------------------------------------------
#pragma unroll
  for (int a = 0; a < AF; ++a)
#pragma unroll
    for (int r = 0; r < NB; r += 2) {
      const int row = m0 + frag_row(a, 4 * fblk + fk);      // output lane (slot fblk, row fk, column fj)
      const int col = n0 + frag_col(r, fblk, fj);           // even; fragment r + 1 is column col + 1
      if constexpr (EpiPairs<Epi>::value && AF * NB <= 32) {     // (the 128 x 128 tile has no registers to spare)
        if (row < g.M && col + 1 < g.N) {
          const d2v v = epi.pair((int)bz, row, col, acc[a][r], acc[a][r + 1]);
          if constexpr (EpiReduces<Epi>::value) {
            local += v.x;
            local += v.y;
          }
          continue;
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (row < g.M && col + q < g.N) {
          if constexpr (EpiReduces<Epi>::value)
            local += epi((int)bz, row, col + q, acc[a][r + q]);
          else
            epi((int)bz, row, col + q, acc[a][r + q]);
        }
      }
    }
  if constexpr (EpiReduces<Epi>::value) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if (lane == 0) gemm_lds[wave] = local;
    __syncthreads();
    if (t == 0)
      epi.part[(int64_t)bz * gx + bx] = (gemm_lds[0] + gemm_lds[1]) + (gemm_lds[2] + gemm_lds[3]);
  }
  dep.publish(bm, bn, bz);
#ifdef VB_GEMM_CLOCK
  __builtin_amdgcn_s_waitcnt(0);       // the epilogue's stores have left
  if (lane == 0 && bx < 1024) {
    long long* o = vb_gemm_dbg + 8 * (4 * bx + wave);
    o[0] = dbg_t1 - dbg_t0;            // prologue (tile setup + first two slab fetches)
    o[1] = dbg_t2 - dbg_t1;            // main loop
    o[2] = clock64() - dbg_t2;         // epilogue
    o[3] = wall_clock64() - dbg_w0;    // whole workgroup lifetime, 100 MHz ticks
    o[4] = dbg_w0;
  }
#endif
}


template <bool A_KCONTIG, int AF, int NB, int STAGES, class Epi>
__global__ void __launch_bounds__(256, AF * NB > 32 ? (STAGES > 2 ? 1 : 2) : (AF * NB > 16 || STAGES > 2 ? 2 : 4))
    gemm_f64_dma_kernel(const GemmArgs g, const Epi epi) {
  int bx = (int)blockIdx.x, bz = (int)blockIdx.z;
  if (g.xcd_group) {      // (GemmArgs::xcd_group: the tiles of a split on one XCD)
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.z, slot = lin >> 3;
    bz = (int)((lin & 7) + 8 * (slot / gridDim.x));
    bx = (int)(slot % gridDim.x);
  }
  gemm_f64_dma_tile<A_KCONTIG, AF, NB, STAGES, Epi, GemmNoDep>(g, epi, GemmNoDep{}, bx, bz, (int)gridDim.x);
}

template <bool A_KCONTIG, int AF, int NB, int STAGES, class Epi>
inline void gemm_f64_dma_launch(hipStream_t st, const GemmArgs& g, dim3 grid, const Epi& epi) {
  constexpr size_t lds = (size_t)STAGES * (32 * AF * kGemmBK + kGemmBK * 8 * NB) * sizeof(double);
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f64_dma_kernel<A_KCONTIG, AF, NB, STAGES, Epi>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    configured = true;
  }
  hipExtLaunchKernelGGL((gemm_f64_dma_kernel<A_KCONTIG, AF, NB, STAGES, Epi>), grid, dim3(256), lds, st, g.ev0, g.ev1,
                        0, g, epi);
}

}  // namespace vb
