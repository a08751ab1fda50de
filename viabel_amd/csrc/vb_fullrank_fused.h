// One persistent launch for the dense Gaussian family's evaluation on the correlated-Gaussian target (north_star: "fused
// into one launch"; viabel/objectives.py:154-168 with approximations.py's z = mu + L eps written as GEMMs):
//
//   phase 1   Z - m = E L' + mu - m          128 x 64 tiles, triangular k range        (EpiStoreZ)
//   phase 2   G = -(Z - m) P, sum f          128 x 64 tiles, k = the columns of Z      (EpiNegateF)
//   phase 3   C_z = G_z' E_z (lower tiles)   128 x 64 tiles, k = the sample rows of split z (EpiSplitSlabCs)
//
// The three products are a dependent chain, but not tile for tile: a phase-2 tile of row block rb walks its k range in
// the order in which the 64-column blocks of Z[rb] are finished by phase 1 (short k ranges first), and a phase-3 tile
// walks the sample rows in the order in which the row blocks of G are finished.  So instead of three launches that
// each end with a tail of idle CUs, ONE launch of resident workgroups pulls tiles from a list in dependency order
// (one returning atomic per tile) and a tile whose input is not there yet polls a flag:
//
//   * hand-off (cdna_hip_programming.md, Guideline 16, form R1): the producer's epilogue stores write-through (sc1),
//     every storing wave drains vmcnt, one lane stores the tile's flag (relaxed, agent scope); the consumer polls that
//     one word relaxed and reads the operand with sc1 loads (global_load_lds ... sc1: L1 bypassed) -- no fences.
//   * flags hold the evaluation's epoch (a counter of the context, never 0): nothing has to be zeroed between
//     evaluations; the tile counter is reset by the workgroup that draws the last ticket.
//   * progress: the list is in a topological order and every workgroup only waits for tiles EARLIER in the list, which
//     are held by workgroups that are running -- no assumption about dispatch order, residency or placement.  Every
//     poll is bounded; a give-up sets an error word that vb_fullrank_get reports.
//
// The arithmetic of a tile is gemm_f64_dma_tile's, the same instantiation the stand-alone launches use: results are
// bit-identical to the three-launch path.
#pragma once

namespace vb {

typedef __attribute__((address_space(1))) unsigned fz_gu32;
typedef __attribute__((address_space(1))) unsigned long long fz_gu64;

// write-through 16-byte store (the compiler's own waits stay conservative with it in flight: vmcnt retires in order).
// The s_nop is the ISA's "VMEM store of more than 64 bits followed by a VALU write of its data VGPRs: 1 wait state"
// rule, which the compiler's hazard recogniser applies to its own stores but not to an asm statement (without it the
// first half of a pair was overwritten before the store had read it).
__device__ __forceinline__ void fz_store_sc1(double* p, d2v v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ double fz_load_sc1(const double* p) {
  const unsigned long long u = __hip_atomic_load((fz_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_bit_cast(double, u);
}

struct FzShared {            // words every workgroup of the launch shares (device memory, one set per context)
  unsigned* head;            // next ticket
  unsigned* zflag;           // [tiles_m][tiles_n]: epoch when Z tile (rb, j) has been stored
  unsigned* gflag;           // [tiles_m][tiles_n]: ... G tile (rb, cb)
  unsigned* err;             // != 0: a poll gave up
  unsigned epoch;
  int tiles_n;               // 64-column blocks
};

__device__ __forceinline__ void fz_wait(const unsigned* f, unsigned epoch, unsigned* err) {
  // every lane reads the same word (one request); bounded: ~2^22 polls of >= 0.3 us each is seconds, then give up
  unsigned spins = 0;
  while (__hip_atomic_load((fz_gu32*)f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
    __builtin_amdgcn_s_sleep(4);
    if (++spins > (1u << 22)) {
      __hip_atomic_store((fz_gu32*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
}

__device__ __forceinline__ void fz_publish(unsigned* f, unsigned epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY storing wave drains its write-through stores ...
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store((fz_gu32*)f, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ... then ONE lane
}

// ---- epilogues: the stand-alone ones with write-through stores ----------------------------------------------------------
struct EpiStoreZPub {        // Z = acc + mu - shift (EpiStoreZ without the t family's row scale)
  double* Z;
  int64_t ldz;
  const double* mu;
  const double* shift;
  __device__ void operator()(int, int row, int col, double acc) const {
    double z = acc + mu[col];
    if (shift) z -= shift[col];
    __hip_atomic_store((fz_gu64*)(Z + (int64_t)row * ldz + col), __builtin_bit_cast(unsigned long long, z), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ d2v pair(int, int row, int col, double a0, double a1) const {
    const d2v m = *reinterpret_cast<const d2v*>(mu + col);
    d2v z = (d2v){a0 + m.x, a1 + m.y};
    if (shift) {
      const d2v sh = *reinterpret_cast<const d2v*>(shift + col);
      z.x -= sh.x, z.y -= sh.y;
    }
    fz_store_sc1(Z + (int64_t)row * ldz + col, z);
    return z;
  }
};

template <bool PUB>
struct EpiNegateFPub {       // EpiNegateF; z - m is another workgroup's store of this launch: read past the L1
  double* G;
  int64_t ldz;
  const double* Zc;
  double* part;
  __device__ double operator()(int, int row, int col, double acc) const {
    const double z = fz_load_sc1(Zc + (int64_t)row * ldz + col);
    if (PUB)
      __hip_atomic_store((fz_gu64*)(G + (int64_t)row * ldz + col), __builtin_bit_cast(unsigned long long, -acc),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      G[(int64_t)row * ldz + col] = -acc;
    return -0.5 * acc * z;
  }
  __device__ d2v pair(int, int row, int col, double a0, double a1) const {
    const double* zp = Zc + (int64_t)row * ldz + col;
    const double z0 = fz_load_sc1(zp), z1 = fz_load_sc1(zp + 1);
    if (PUB) fz_store_sc1(G + (int64_t)row * ldz + col, (d2v){-a0, -a1});
    else *reinterpret_cast<d2v*>(G + (int64_t)row * ldz + col) = (d2v){-a0, -a1};
    return (d2v){-0.5 * a0 * z0, -0.5 * a1 * z1};
  }
};

// ---- dependencies -----------------------------------------------------------------------------------------------------
struct DepProduceZ {         // phase 1: reads launch inputs only, publishes Z tile (bm, bn)
  static constexpr int kAuxA = 0;
  FzShared s;
  __device__ __forceinline__ void before_slab(int, int, int, bool) const {}
  __device__ __forceinline__ void publish(int bm, int bn, int) const { fz_publish(s.zflag + bm * s.tiles_n + bn, s.epoch); }
};

template <bool PUB>
struct DepConsumeZ {         // phase 2: k slab `slab` of row block bm lies in Z tile (bm, slab / 4); publishes G tile (bm, bn)
  static constexpr int kAuxA = 16;
  FzShared s;
  __device__ __forceinline__ void before_slab(int bm, int, int slab, bool first) const {
    if ((slab & 3) == 0 || first) fz_wait(s.zflag + bm * s.tiles_n + (slab >> 2), s.epoch, s.err);
  }
  __device__ __forceinline__ void publish(int bm, int bn, int) const {
    if (PUB) fz_publish(s.gflag + bm * s.tiles_n + bn, s.epoch);
  }
};

struct DepConsumeG {         // phase 3: A = G given k-major; k slab `slab` (16 sample rows) lies in row block slab / 8, the
#ifdef VB_FZ_P3_ACQUIRE
  static constexpr int kAuxA = 0;
#else
  static constexpr int kAuxA = 16;      // tile's 128 columns of G in column blocks 2 bm and 2 bm + 1
#endif
  FzShared s;
  __device__ __forceinline__ void before_slab(int bm, int, int slab, bool first) const {
    if ((slab & 7) == 0 || first) {
      const unsigned* f = s.gflag + (slab >> 3) * s.tiles_n + 2 * bm;
      fz_wait(f, s.epoch, s.err);
      if (2 * bm + 1 < s.tiles_n) fz_wait(f + 1, s.epoch, s.err);
#ifdef VB_FZ_P3_ACQUIRE
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    }
  }
  __device__ __forceinline__ void publish(int, int, int) const {}
};

struct FzArgs {
  GemmArgs g1, g2, g3;
  EpiStoreZPub e1;
  double* G;                 // phase 2 epilogue: (G, ldz, Zc, part)
  int64_t ldz;
  const double* Zc;
  double* fpart;
  EpiSplitSlabCs e3;
  FzShared s;
  const int4* items;         // (phase, block x, block z, -)
  int n_items, n_p1, n_p2, n_p3;      // tiles per phase: gridDim.x of the stand-alone launch (the partial-sum layouts)
  long long* clk;            // per-item clocks (VB_FUSED_CLOCK builds): [start, end, phase, x] x n_items
};

// PHASES = 2: phases 1 and 2 (the gradient product and the split reduction stay launches); 3: all three products
//
// The argument block is read through the kernarg pointer, phase by phase and tile by tile: taken by value the compiler
// hoists every field of all three products out of the ticket loop and spills 100-180 SGPRs into the tiles' inner loops.
typedef const __attribute__((address_space(4))) FzArgs* fz_kernarg_ptr;

template <class T, class S>
__device__ __forceinline__ T fz_arg(const S& src) {      // a by-value copy of one member of the argument block
  T out;
  __builtin_memcpy(&out, &src, sizeof(T));
  return out;
}

template <int PHASES>
__global__ void __launch_bounds__(256, 2) fr_fused_kernel(const FzArgs) {
  constexpr int kLdsDoubles = 3 * (128 * kGemmBK + kGemmBK * 64);      // the 128 x 64 tile's three stages
  extern __shared__ double gemm_lds[];
  const int t = threadIdx.x;
  int* ticket = reinterpret_cast<int*>(gemm_lds + kLdsDoubles);
  const unsigned n_wg = gridDim.x;
  fz_kernarg_ptr ap = (fz_kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  for (;;) {
    asm volatile("" : "+s"(ap));      // (opaque per iteration: nothing of the argument block is loop-invariant to the compiler)
    __syncthreads();           // the previous tile's last LDS reads (epilogue partials) are done
    const int n_items = ap->n_items;
    if (t == 0) {
      unsigned* head = ap->s.head;
      const unsigned it = __hip_atomic_fetch_add((fz_gu32*)head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // every workgroup leaves on its first ticket beyond the list: the last such ticket is n_items + n_wg - 1, and
      // whoever draws it re-arms the counter for the next launch (stream order does the rest)
      if (it == (unsigned)n_items + n_wg - 1u)
        __hip_atomic_store((fz_gu32*)head, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *ticket = (int)it;
    }
    __syncthreads();
    const int it = __builtin_amdgcn_readfirstlane(*ticket);
    if (it >= n_items) return;
    const int4 w = ap->items[it];
    const int phase = __builtin_amdgcn_readfirstlane(w.x), bx = __builtin_amdgcn_readfirstlane(w.y),
              bz = __builtin_amdgcn_readfirstlane(w.z);
#ifdef VB_FUSED_CLOCK
    const long long c0 = wall_clock64();
#endif
    const FzShared sh = fz_arg<FzShared>(ap->s);
    if (phase == 0) {
      __builtin_amdgcn_s_setprio(2);      // producers first: what they finish is what everybody else waits for
      gemm_f64_dma_tile<true, 4, 8, 3, EpiStoreZPub, DepProduceZ>(fz_arg<GemmArgs>(ap->g1), fz_arg<EpiStoreZPub>(ap->e1),
                                                                  DepProduceZ{sh}, bx, bz, ap->n_p1);
    } else if (phase == 1) {
      __builtin_amdgcn_s_setprio(1);
      gemm_f64_dma_tile<true, 4, 8, 3, EpiNegateFPub<PHASES == 3>, DepConsumeZ<PHASES == 3>>(
          fz_arg<GemmArgs>(ap->g2), EpiNegateFPub<PHASES == 3>{ap->G, ap->ldz, ap->Zc, ap->fpart},
          DepConsumeZ<PHASES == 3>{sh}, bx, bz, ap->n_p2);
    } else if (PHASES == 3) {
      __builtin_amdgcn_s_setprio(0);
      gemm_f64_dma_tile<false, 4, 8, 3, EpiSplitSlabCs, DepConsumeG>(fz_arg<GemmArgs>(ap->g3), fz_arg<EpiSplitSlabCs>(ap->e3),
                                                                     DepConsumeG{sh}, bx, bz, ap->n_p3);
    }
#ifdef VB_FUSED_CLOCK
    long long* clk = ap->clk;
    if (t == 0 && clk) {
      long long* o = clk + 4 * (int64_t)it;
      o[0] = c0, o[1] = wall_clock64(), o[2] = phase, o[3] = ((long long)bz << 32) | (unsigned)bx;
    }
#endif
  }
}

template __global__ void fr_fused_kernel<2>(const FzArgs);
template __global__ void fr_fused_kernel<3>(const FzArgs);

}  // namespace vb
