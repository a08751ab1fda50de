// Z = E L' for a lower-triangular L (tri_mode 1: B[k][j] == 0 for k > j, A given as A[m][k]) with every workgroup
// doing the same amount of work.
//
// Column block j of Z needs the k range [0, 64 (j + 1)): 4 ... 4 tn slabs of 16.  However the tiles are ordered, a
// 64-slab tile that shares its SIMDs fairly with two or three other waves lives for the whole kernel, and the slots
// that short tiles leave behind cannot be refilled with anything of the right size -- the 64 x 64 launch of
// gemm_f64_dma_kernel takes 82 us for 59 us of MFMA time at D = 1024 and no priority or launch order changes that
// (tools/tri_sched.sh: a wave gets at most ~45 % of its SIMD's matrix pipe however it is prioritised, so a tile's
// lifetime is bounded below by its own length).  Here a workgroup is TWO four-wave teams and owns a PAIR of tiles of
// one row block: the heavy column block tn - 1 - p (n_h slabs) and the light one p (n_l slabs), n_h + n_l = 4 (tn + 1)
// for every pair.  With H = ceil((n_h + n_l) / 2):
//     team 0:  the light tile, all n_l slabs -> epilogue of the light tile -> the first H - n_l slabs of the heavy tile
//     team 1:  the remaining n_h - (H - n_l) (= H or H - 1) slabs of the heavy tile
// and at the end team 1 hands its accumulators to team 0 through its (now idle) LDS stages: heavy = acc0 + acc1, a fixed
// order, deterministic.  Every workgroup runs H slab iterations, the grid is tiles_m x ceil(tn / 2) workgroups (512 at
// 4096 x 1024: exactly two per CU, four waves per SIMD), all start together and end together.
//
// Tile, LDS layout, fragment assignment and the software pipeline are those of gemm_f64_dma_kernel<true, 2, 8, 2>
// (64 x 64 x 16 tile, two LDS stages per team, operands by global_load_lds_dwordx4, A[m][16] tile XOR-swizzled on the
// global side, B tile k-major read with ds_read_b128 fragment pairs).  Epilogue contract: as gemm_f64_kernel, including
// `pair` and reducing epilogues; a reducing epilogue receives one partial per workgroup (both tiles of the pair).
#pragma once

namespace vb {

template <class Epi>
__global__ void __launch_bounds__(512, 4) gemm_tripair_kernel(const GemmArgs g, const Epi epi) {
  constexpr int AF = 2, NB = 8, BM = 64, BN = 64, kStages = 2;
  constexpr int kATile = BM * kGemmBK, kBTile = kGemmBK * BN;      // 1024 doubles each
  constexpr int kTeamLds = kStages * (kATile + kBTile);            // 4096 doubles = 32 KB per team
  constexpr int KS = kGemmBK / 4;
  extern __shared__ double tp_lds_all[];
  const int t = threadIdx.x & 255;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int team = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  double* lds = tp_lds_all + team * kTeamLds;
  double* As = lds;                              // [kStages][kATile]
  double* Bs = lds + kStages * kATile;           // [kStages][kBTile]
  const int wm = wave >> 1, wn = wave & 1;
#ifdef VB_GEMM_CLOCK
  const long long dbg_t0 = clock64(), dbg_w0 = wall_clock64();
  long long dbg_t1 = dbg_t0, dbg_t2 = dbg_t0;
#endif

  // ---- the pair of tiles and this team's slab program ---------------------------------------------------------
  const int tn = g.tiles_n;
  const int bm = blockIdx.x % g.tiles_m, p = blockIdx.x / g.tiles_m;
  const int bn_h = tn - 1 - p, bn_l = p;
  const int m0 = bm * BM, n0_h = bn_h * BN, n0_l = bn_l * BN;
  auto k_slabs = [&](int n0) { const int ke = n0 + BN < g.K ? n0 + BN : g.K; return ke / kGemmBK; };
  const int n_h = k_slabs(n0_h), n_l = bn_l != bn_h ? k_slabs(n0_l) : 0;
  const int H = (n_h + n_l + 1) / 2;             // slab iterations of the workgroup
  const int h0 = H - n_l;                        // heavy slabs of team 0 (n_l <= n_h, so 0 <= h0 <= n_h)
  // team 0: sequence index s < n_l -> light slab s, else heavy slab s - n_l;  team 1: heavy slab h0 + s
  const int cnt = team == 0 ? H : n_h - h0;      // slabs this team multiplies (team 1: H or H - 1, possibly 0)
  const int sw = team == 0 ? n_l : 0;            // sequence index at which the tile changes (team 0 only; 0: never)

  double acc[AF][NB];
#pragma unroll
  for (int i = 0; i < AF; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = 0.0;

  // ---- global sources of this wave's four 1-KiB units per slab: u = 0, 1 the A tile, u = 2, 3 the B tile ---------
  const double* src[4];      // next slab to fetch
  const int64_t step_b = (int64_t)kGemmBK * g.ldb;
  // sources of k slab `ks` of the tile whose first column is n0 (recomputed at the one tile switch of team 0 rather
  // than kept in registers: the kernel has 128 of them)
  auto set_sources = [&](int n0, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = u * 4 + wave;                  // A unit: 8 rows x 8 pairs
      const int row = q * 8 + (lane >> 3), pr = lane & 7;
      const int kp = pr ^ ((row >> 1) & 7);
      int m = m0 + row;
      m = m < g.M ? m : g.M - 1;
      src[u] = g.A + (int64_t)m * g.lda + 2 * kp + (int64_t)ks * kGemmBK;
      const int krow = q * 2 + (lane >> 5), c = lane & 31;      // B unit: 2 k rows x 32 pairs
      int64_t col = n0 + 2 * c;
      col = col < g.ldb - 1 ? col : g.ldb - 2;
      src[2 + u] = g.B + (int64_t)krow * g.ldb + col + (int64_t)ks * step_b;
    }
  };
  if (team == 0) set_sources(n_l > 0 ? n0_l : n0_h, 0);
  else set_sources(n0_h, h0 < n_h ? h0 : n_h - 1);      // (nothing to multiply: re-fetch team 0's last slab, stay in range)
  auto issue = [&](int st) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = (u & 1) * 4 + wave;
      double* dst = lds + (u < 2 ? q * 128 + st * kATile : kStages * kATile + q * 128 + st * kBTile);
      __builtin_amdgcn_global_load_lds((gemm_gptr)src[u], (gemm_lptr)dst, 16, 0, 0);
    }
  };
  // make src point at sequence index `next` (called with next = previous + 1, next < cnt)
  auto advance_to = [&](int next) __attribute__((always_inline)) {
    if (next == sw && sw > 0) {
      set_sources(n0_h, 0);
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) src[u] += kGemmBK, src[2 + u] += step_b;
    }
  };

  // ---- fragment addresses (as gemm_f64_dma_kernel, A[m][k] tile) ------------------------------------------------
  const int fi = lane & 15, fk = lane >> 4;
  const int fblk = (lane >> 2) & 3, fj = lane & 3;
  auto frag_row = [&](int a, int i) __attribute__((always_inline)) { return wm * (16 * AF) + a * 16 + i; };
  auto frag_col = [&](int r, int blk, int j) __attribute__((always_inline)) {
    return wn * (4 * NB) + 8 * ((blk + (r >> 1)) & (NB / 2 - 1)) + 2 * j + (r & 1);
  };
  // A fragment of k-step kk: row * 16 + 2 * (pair ^ swizzle) + (k & 1) with pair = 2 kk + (fk >> 1); the pair index sits
  // in address bits 1..3 and 2 kk in bits 2..3, so k-step kk is the address of k-step 0 XOR 4 kk (one register, not KS)
  int a_off0, b_off[NB / 2];
  {
    const int row = wm * (16 * AF) + fi;
    a_off0 = row * kGemmBK + 2 * ((fk >> 1) ^ ((row >> 1) & 7)) + (fk & 1);
  }
#pragma unroll
  for (int r = 0; r < NB; r += 2) b_off[r / 2] = fk * BN + frag_col(r, fblk, fj);

  double fa[2][AF], fb[2][NB];
  auto load_frags = [&](int st, int kk, int set) __attribute__((always_inline)) {
    const double* as = As + st * kATile;
    const double* bs = Bs + st * kBTile;
#pragma unroll
    for (int a = 0; a < AF; ++a) fa[set][a] = as[(a_off0 ^ (4 * kk)) + a * (16 * kGemmBK)];
#pragma unroll
    for (int r = 0; r < NB; r += 2) {
      const d2v v = *reinterpret_cast<const d2v*>(bs + b_off[r / 2] + 4 * kk * BN);
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
  constexpr int kReads = AF / 2 + NB / 2, kMfma = AF * NB, kPer = kMfma / kReads;
  auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < kReads; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, kPer, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, kMfma - kPer * kReads, 0);
  };

  // ---- epilogue of one tile (team 0 only) ----------------------------------------------------------------------
  double local = 0.0;
  auto store_tile = [&](int n0, int m0, int ln) __attribute__((always_inline)) {
    const int fk = ln >> 4, fblk = (ln >> 2) & 3, fj = ln & 3;
#pragma unroll
    for (int a = 0; a < AF; ++a)
#pragma unroll
      for (int r = 0; r < NB; r += 2) {
        const int row = m0 + frag_row(a, 4 * fblk + fk);
        const int col = n0 + frag_col(r, fblk, fj);
        if constexpr (EpiPairs<Epi>::value) {
          if (row < g.M && col + 1 < g.N) {
            const d2v v = epi.pair(0, row, col, acc[a][r], acc[a][r + 1]);
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
              local += epi(0, row, col + q, acc[a][r + q]);
            else
              epi(0, row, col + q, acc[a][r + q]);
          }
        }
      }
  };

  // s_waitcnt vmcnt(0) only
  auto wait_vm0 = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8)); };
  // B[k][j] == 0 for k > j: a wave whose last column is j_last multiplies nothing in the slabs that start beyond it
  // (the lower half of a tile's diagonal block for the waves of its left half)
  const int j_last_h = n0_h + wn * (4 * NB) + 4 * NB - 1, j_last_l = n0_l + wn * (4 * NB) + 4 * NB - 1;
  auto live = [&](int s) __attribute__((always_inline)) {      // does sequence index s of this team carry MFMAs for this wave?
    if (s >= cnt) return false;
    const bool light = s < sw;
    const int ks = light ? s : (team == 0 ? s - sw : h0 + s);
    return kGemmBK * ks <= (light ? j_last_l : j_last_h);
  };

  if (H > 0) {
    issue(0);
    wait_vm0();
    __asm__ volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __asm__ volatile("" ::: "memory");
    if (live(0)) load_frags(0, 0, 0);
#ifdef VB_GEMM_CLOCK
    dbg_t1 = clock64();
#endif
    int st = 0;
    // the two workgroups of a CU (block b and b + prio_div) trade the higher wave priority slab by slab: left to the
    // arbiter's oldest-first rule the older one finishes at 60 us and the younger, alone on the CU with two waves per
    // SIMD, at 78 us (measured; as in gemm_f64_dma_kernel)
    const int prio_phase = g.prio_div > 0 ? (int)(blockIdx.x / (unsigned)g.prio_div) & 1 : -1;
    for (int s = 0; s < H; ++s) {
      if (prio_phase >= 0) {
        if (g.tri_flags & 4) {                   // experiment: the younger generation permanently at the higher priority
          if (s == 0 && prio_phase) __builtin_amdgcn_s_setprio(1);
        } else if (g.tri_flags & 8) {            // experiment: the younger generation high on three slabs out of four
          if (prio_phase ? (s & 3) != 0 : (s & 3) == 0) __builtin_amdgcn_s_setprio(1);
          else __builtin_amdgcn_s_setprio(0);
        } else {
          if ((s ^ prio_phase) & 1) __builtin_amdgcn_s_setprio(1);
          else __builtin_amdgcn_s_setprio(0);
        }
      }
      const int st1 = st ^ 1;
      if (s > 0 && s == sw) {            // team 0: the light tile is complete -- store it and start the heavy one
        // opaque copies: the addresses of this store must be formed here, not hoisted out of the slab loop (where
        // they would occupy 30 registers for the whole kernel)
        int n0x = n0_l, m0x = m0, lx = lane;
        __asm__ volatile("" : "+s"(n0x), "+s"(m0x), "+v"(lx));
        store_tile(n0x, m0x, lx);
#pragma unroll
        for (int i = 0; i < AF; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[i][j] = 0.0;
        // (the first fragments of the heavy tile were not fetched ahead: nothing but the accumulators is live across
        // the store, which keeps the kernel inside its 128 registers)
        if (live(s)) load_frags(st, 0, 0);
      }
      if (s + 1 < cnt) advance_to(s + 1);      // beyond the end: re-fetch the last slab into the free stage
      issue(st1);
      const bool on = live(s);
      if (on) {
#pragma unroll
        for (int kk = 0; kk < KS - 1; ++kk) {
          load_frags(st, kk + 1, (kk + 1) & 1);
          mfma_step(kk & 1);
          interleave();
        }
      }
      wait_vm0();                              // slab s + 1 has landed (this wave's share)
      __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's reads of stage st are done
      __asm__ volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __asm__ volatile("" ::: "memory");
      const bool on1 = live(s + 1) && !(sw > 0 && s + 1 == sw);
      if (on) {
        if (on1) load_frags(st1, 0, KS & 1);
        mfma_step((KS - 1) & 1);
        interleave();
      } else if (on1) {
        load_frags(st1, 0, KS & 1);
      }
      st = st1;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_s_barrier();
#ifdef VB_GEMM_CLOCK
    dbg_t2 = clock64();
#endif
    // team 1's half of the heavy tile to team 0: [wave][fragment][lane] in team 1's stages (4 x 16 x 64 doubles)
    static_assert(4 * AF * NB * 64 <= kTeamLds, "hand-off buffer");
    double* xch = tp_lds_all + kTeamLds + wave * (AF * NB * 64) + lane;
    if (team == 1) {
#pragma unroll
      for (int a = 0; a < AF; ++a)
#pragma unroll
        for (int r = 0; r < NB; ++r) xch[(a * NB + r) * 64] = acc[a][r];
    }
    __syncthreads();
    if (team == 0) {
#pragma unroll
      for (int a = 0; a < AF; ++a)
#pragma unroll
        for (int r = 0; r < NB; ++r) acc[a][r] = acc[a][r] + xch[(a * NB + r) * 64];
      store_tile(n0_h, m0, lane);
    }
  }
  if constexpr (EpiReduces<Epi>::value) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    __syncthreads();
    if (lane == 0 && team == 0) tp_lds_all[wave] = local;
    __syncthreads();
    if (threadIdx.x == 0)
      epi.part[blockIdx.x] = (tp_lds_all[0] + tp_lds_all[1]) + (tp_lds_all[2] + tp_lds_all[3]);
  }
#ifdef VB_GEMM_CLOCK
  __builtin_amdgcn_s_waitcnt(0);
  if (lane == 0 && blockIdx.x < 1024 && team == 0) {
    long long* o = vb_gemm_dbg + 8 * (4 * blockIdx.x + wave);
    o[0] = dbg_t1 - dbg_t0;
    o[1] = dbg_t2 - dbg_t1;
    o[2] = clock64() - dbg_t2;
    o[3] = wall_clock64() - dbg_w0;
    o[4] = dbg_w0;
  }
#endif
}

// does gemm_f64_launch hand a product of this shape to the pair kernel?
inline bool gemm_uses_tripair(const GemmArgs& g, int splits) {
  static const bool on = !(getenv("VB_GEMM_TRIPAIR") && atoi(getenv("VB_GEMM_TRIPAIR")) == 0);
  return on && g.tri_mode == 1 && splits == 1 && !g.batch && g.K % kGemmBK == 0 && g.M > 0 && g.N > 0 && g.K >= g.N;
}

template <class Epi>
inline unsigned gemm_tripair_launch(hipStream_t st, GemmArgs g, int n_cu, const Epi& epi) {
  static const int prio_env = getenv("VB_GEMM_PRIO") ? atoi(getenv("VB_GEMM_PRIO")) : 1;
  g.prio_div = prio_env ? n_cu : 0;
  static const int tri_env = getenv("VB_GEMM_TRI") ? atoi(getenv("VB_GEMM_TRI")) : 0;
  g.tri_flags = tri_env;
  g.tiles_m = (g.M + 63) / 64;
  g.tiles_n = (g.N + 63) / 64;
  const dim3 grid((unsigned)(g.tiles_m * ((g.tiles_n + 1) / 2)), 1, 1);
  constexpr size_t lds = (size_t)2 * 2 * (64 * kGemmBK + kGemmBK * 64) * sizeof(double);      // 64 KB: two per CU
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tripair_kernel<Epi>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    configured = true;
  }
  hipExtLaunchKernelGGL((gemm_tripair_kernel<Epi>), grid, dim3(512), lds, st, g.ev0, g.ev1, 0, g, epi);
  return grid.x;
}

}  // namespace vb
