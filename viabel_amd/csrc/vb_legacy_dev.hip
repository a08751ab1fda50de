// numpy's legacy normal stream on the DEVICE: RandomState(seed).randn(N, D) generated straight into a noise slot, bit
// for bit -- the draws the reference's families consume (viabel/approximations.py:203, :213-216, :343-347), without the
// 11.6 ms of host generation and the 33.5 MB upload of a (4096, 1024) matrix (second half).
//
// The stream is MT19937 + the polar method (numpy/random/src/legacy/legacy-distributions.c: legacy_gauss): attempt i
// consumes words [4 i, 4 i + 4) of the generator's output whether it is accepted or not, and the k-th ACCEPTED attempt
// yields outputs 2 k (f x2, returned first) and 2 k + 1 (f x1, the cached value).  Everything except one logarithm is
// integer or exactly-rounded IEEE arithmetic and is done here:
//
//   * the word stream, in parallel.  MT19937 is one linear recurrence over GF(2); the state J words ahead of a known
//     block is the correlation of the jump polynomial x^J mod phi (vb_mt_jump.h, made and checked against numpy by
//     tools/make_mt_jump.py) with 20 560 words generated from that block.  Streams of 256 blocks each: the known
//     stream starts quadruple every round (mtd_seq_kernel + mtd_corr_kernel: three jumps per known stream), then one workgroup per stream runs the recurrence
//     (mtd_stream_kernel: 227-way parallel inside a block, the block in registers, one barrier per block).
//   * the attempts: words -> two 53-bit doubles -> x1, x2, r2, accepted? (exact arithmetic, no fused multiply-adds:
//     this file is compiled with -ffp-contract=off, as numpy's build of that code has none), acceptance counts per
//     workgroup, their prefix sums, and the scatter of the accepted pairs into the slot's row-major layout.
//   * f = sqrt(-2 log(r2) / r2): division and square root are correctly rounded on both sides; the logarithm is the
//     host C library's (glibc: <= 0.52 ulp, not correctly rounded, and not reproducible instruction for instruction
//     here).  The device computes log(r2) in double-double (error < 2^-90), which IS the correctly rounded value unless
//     the true value lies within 0.03 ulp of a rounding boundary; only there can glibc's result differ.  For those
//     attempts both candidates are carried through f and the two products; if the outputs still differ (about one
//     attempt in forty) the attempt goes on a list that the host finishes with its own log (vb_legacy_finish_pairs,
//     vb_legacy_rng.cpp) and a small scatter kernel writes back.
//
// The generator's state afterwards is numpy's: position just behind the last consumed attempt, the odd value cached.
// Anything this path cannot take (more than 1024 streams, a list overflow, an attempt budget that fell short -- 10
// sigma above the mean) returns VB_ERR_UNSUPPORTED with the generator untouched, and the caller draws on the host.
#include "vb_common.h"
#include "vb_glibc_log.h"
#include "vb_mt_jump.h"

#include <cmath>
#include <vector>

namespace vb {

namespace {

constexpr int kN = 624, kM = 397;
constexpr int kCorrSlices = 63;                   // 63 x 320 = 20 160 >= 19 937 coefficients (the table's words beyond 623 are zero)
constexpr int kAttemptsPerWg = 1024;              // 256 threads x 4 attempts
constexpr int kHardPerWg = 64;                    // list segment per workgroup: ~28 expected, sigma ~5

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// The recurrence by 256 threads, thread t < 227 owning words t, t + 227 and (t < 170) t + 454 of the block in REGISTERS:
// key[k] <- key[k + 397] ^ mix(key[k], key[k + 1]) reads, besides the thread's own words, its right neighbour's three
// words and one word 397 ahead -- old values, taken from an LDS copy of the block -- while the new values it needs
// (key[k - 227] for the second and third word) are the thread's own results; the last word needs the new key[0], which
// thread 169 recomputes from old words instead of waiting for thread 0.  New blocks go to the other half of a
// ping-pong LDS buffer: ONE barrier per block.
struct MtRegs {
  uint32_t w0, w1, w2;
};
__device__ __forceinline__ MtRegs mt_load(const uint32_t* key, int t) {
  MtRegs r = {0u, 0u, 0u};
  if (t < 227) r.w0 = key[t], r.w1 = key[t + 227];
  if (t < 170) r.w2 = key[t + 454];
  return r;
}
// cur: LDS copy of the block held in `r` (complete, visible); next: the other buffer.  Returns with `next` written and
// NOT yet synchronised: the caller's barrier comes before anybody reads it.
__device__ __forceinline__ void mt_step(MtRegs& r, const uint32_t* cur, uint32_t* next, int t) {
  if (t < 227) {
    // every LDS read of the step is issued up front (round 6): the third word's operand sat behind the first two words'
    // write in program order -- a second LDS round trip per block on the loop's critical path
    const uint32_t a1 = cur[t + 1], m = cur[t + kM], b1 = cur[t + 228];
    const uint32_t c_in = cur[t < 169 ? t + 455 : kN - 1];                     // (clamped: lanes 169 .. 226 do not use it)
    const uint32_t k0 = cur[0], k1 = cur[1], km = cur[kM];                       // broadcast reads: the NEW key[0], for k = 623
    const uint32_t n0 = m ^ mt_mix(r.w0, a1);
    const uint32_t n1 = n0 ^ mt_mix(r.w1, b1);
    next[t] = n0, next[t + 227] = n1;
    if (t < 170) {
      const uint32_t c1 = t < 169 ? c_in : (km ^ mt_mix(k0, k1));
      const uint32_t n2 = n1 ^ mt_mix(r.w2, c1);
      next[t + 454] = n2;
      r.w2 = n2;
    }
    r.w0 = n0, r.w1 = n1;
  }
}
__device__ __forceinline__ void mt_store(const MtRegs& r, uint32_t* dst, int t) {      // the block to global memory
  if (t < 227) dst[t] = r.w0, dst[t + 227] = r.w1;
  if (t < 170) dst[t + 454] = r.w2;
}

// words[0 .. pre) = the unread rest of the current block (untempered), state0 = the next block
__global__ void __launch_bounds__(256) mtd_first_kernel(const uint32_t* __restrict__ key_in, int pos,
                                                        uint32_t* __restrict__ words, uint32_t* __restrict__ state0) {
  __shared__ uint32_t key[2][kN];
  const int t = threadIdx.x;
  for (int i = t; i < kN; i += 256) key[0][i] = key_in[i];
  __syncthreads();
  for (int i = pos + t; i < kN; i += 256) words[i - pos] = key[0][i];
  MtRegs r = mt_load(key[0], t);
  mt_step(r, key[0], key[1], t);
  mt_store(r, state0, t);
}

// state[count + s] = the block 624 * 256 * count words behind state[s] (one round of the ladder, two kernels):
// mtd_seq_kernel writes the 34 blocks that start with state[s] (every u[i + j], i < 20 160, j < 624) and
// mtd_corr_kernel's workgroup (s, slice, a) XORs its 320 coefficients' share of the 624 words of the correlation into the target.
constexpr int kSeqBlocks = 34;                    // 34 x 624 = 21 216 >= 20 160 + 623 words
constexpr int kSeqWords = kSeqBlocks * kN;
constexpr int kSlice = 320;                       // coefficients per slice (10 words of the polynomial)
__global__ void __launch_bounds__(256) mtd_seq_kernel(const uint32_t* __restrict__ state, uint32_t* __restrict__ seq) {
  __shared__ uint32_t key[2][kN];
  const int t = threadIdx.x;
  const uint32_t* src = state + (size_t)blockIdx.x * kN;
  uint32_t* dst = seq + (size_t)blockIdx.x * kSeqWords;
  for (int i = t; i < kN; i += 256) key[0][i] = src[i];
  __syncthreads();
  MtRegs r = mt_load(key[0], t);
  for (int b = 0; b < kSeqBlocks; ++b) {
    mt_store(r, dst + (size_t)b * kN, t);
    if (b + 1 < kSeqBlocks) {
      mt_step(r, key[b & 1], key[(b + 1) & 1], t);
      __syncthreads();      // (an LDS-counter-only barrier -- no wait for the block's global stores -- measured the same: round 6)
    }
  }
}

// blockIdx.z = a - 1: the jump of a * count streams (polynomial row `poly_row0 + a - 1`), target stream a * count + s.
// Thread tt forms FOUR consecutive words 4 tt .. 4 tt + 3 of the correlation: coefficient bit i contributes u[i + j] to
// word j, so the four words share a sliding window of the sequence -- 36 words from LDS (nine 16-byte reads) serve 32
// coefficients x 4 outputs, a quarter of the LDS traffic of one output per thread (the kernel's bound).
constexpr int kCorrThreads = 192;                 // 156 of them own outputs (624 / 4)
__global__ void __launch_bounds__(kCorrThreads) mtd_corr_kernel(const uint32_t* __restrict__ polys, int poly_row0,
                                                                const uint32_t* __restrict__ seq, uint32_t* __restrict__ state,
                                                                int count, int streams) {
  __shared__ __attribute__((aligned(16))) uint32_t u[kSlice + kN + 8];
  __shared__ uint32_t g[kSlice / 32];
  const int s = blockIdx.x, slice = blockIdx.y, a = blockIdx.z + 1, tt = threadIdx.x;
  const int target = a * count + s;
  if (target >= streams) return;
  const uint32_t* poly = polys + (size_t)(poly_row0 + a - 1) * kN;
  const uint32_t* src = seq + (size_t)s * kSeqWords + slice * kSlice;
  for (int i = tt; i < kSlice + kN + 8; i += kCorrThreads) u[i] = i < kSlice + kN ? src[i] : 0u;
  if (tt < kSlice / 32) g[tt] = slice * (kSlice / 32) + tt < kN ? poly[slice * (kSlice / 32) + tt] : 0u;
  __syncthreads();
  if (tt >= kN / 4) return;
  uint32_t acc[4] = {0u, 0u, 0u, 0u};
#pragma unroll 1
  for (int w = 0; w < kSlice / 32; ++w) {
    // (the coefficient word is the same for every lane: in a scalar register the bit masks are scalar work)
    const uint32_t gw = __builtin_amdgcn_readfirstlane(g[w]);
    uint32_t r[36];
    const uint4* win = reinterpret_cast<const uint4*>(u + 32 * w + 4 * tt);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const uint4 v = win[q];
      r[4 * q] = v.x, r[4 * q + 1] = v.y, r[4 * q + 2] = v.z, r[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int b = 0; b < 32; ++b) {
      const uint32_t m = (uint32_t)((int32_t)(gw << (31 - b)) >> 31);      // scalar: all ones where coefficient bit b is set
#pragma unroll
      for (int k = 0; k < 4; ++k)      // acc ^ (m & r) in ONE three-input bit operation (truth table 0xF0 ^ (0xCC & 0xAA))
        acc[k] = __builtin_amdgcn_bitop3_b32(acc[k], m, r[b + k], 0x78);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    atomicXor(&state[(size_t)target * kN + 4 * tt + k], acc[k]);      // integer: the result does not depend on the order
}

// stream s: blocks [s B, min((s + 1) B, n_blocks)) of the stream into words[pre + block * 624 ...] (untempered)
__global__ void __launch_bounds__(256) mtd_stream_kernel(const uint32_t* __restrict__ state, uint32_t* __restrict__ words,
                                                         int64_t pre, int64_t n_blocks) {
  __shared__ uint32_t key[2][kN];
  const int t = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * kMtBlocksPerStream;
  const int64_t b1 = b0 + kMtBlocksPerStream < n_blocks ? b0 + kMtBlocksPerStream : n_blocks;
  const uint32_t* src = state + (size_t)blockIdx.x * kN;
  for (int i = t; i < kN; i += 256) key[0][i] = src[i];
  __syncthreads();
  MtRegs r = mt_load(key[0], t);
  for (int64_t b = b0; b < b1; ++b) {
    mt_store(r, words + pre + b * kN, t);
    if (b + 1 < b1) {
      mt_step(r, key[(b - b0) & 1], key[(b - b0 + 1) & 1], t);
      __syncthreads();      // (an LDS-counter-only barrier -- no wait for the block's global stores -- measured the same: round 6)
    }
  }
}

// ---- attempts --------------------------------------------------------------------------------------------------------
struct Attempt {
  double x1, x2, r2;
  bool ok;
};

__device__ __forceinline__ double words_to_double(uint32_t w0, uint32_t w1) {      // numpy's 53-bit double from two words
  const int32_t a = (int32_t)(w0 >> 5), b = (int32_t)(w1 >> 6);
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

__device__ __forceinline__ Attempt attempt_at(const uint32_t* __restrict__ words, int64_t i) {
  const uint4 w = *reinterpret_cast<const uint4*>(words + 4 * i);
  Attempt a;
  a.x1 = 2.0 * words_to_double(mt_temper(w.x), mt_temper(w.y)) - 1.0;
  a.x2 = 2.0 * words_to_double(mt_temper(w.z), mt_temper(w.w)) - 1.0;
  a.r2 = a.x1 * a.x1 + a.x2 * a.x2;
  a.ok = !(a.r2 >= 1.0 || a.r2 == 0.0);
  return a;
}

__global__ void __launch_bounds__(256) mtd_count_kernel(const uint32_t* __restrict__ words, int64_t attempts,
                                                        int* __restrict__ cnt) {
  __shared__ int sh[4];
  const int t = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * kAttemptsPerWg + 4 * t;
  int c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (i0 + k < attempts) c += attempt_at(words, i0 + k).ok ? 1 : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((t & 63) == 0) sh[t >> 6] = c;
  __syncthreads();
  if (t == 0) cnt[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// exclusive prefix sums of cnt[0 .. n) into base[0 .. n], base[n] = total (one workgroup, int64 sums)
__global__ void __launch_bounds__(1024) mtd_scan_kernel(const int* __restrict__ cnt, int64_t n, int64_t* __restrict__ base) {
  __shared__ int64_t wave_tot[16];
  __shared__ int64_t carry;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int64_t b0 = 0; b0 < n; b0 += 1024) {
    const int64_t i = b0 + t;
    const int64_t own = i < n ? cnt[i] : 0;
    int64_t v = own;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int64_t u = __shfl_up(v, off, 64);
      if (lane >= off) v += u;
    }
    if (lane == 63) wave_tot[wv] = v;
    __syncthreads();
    int64_t before = carry;
    for (int q = 0; q < wv; ++q) before += wave_tot[q];
    if (i < n) base[i] = before + v - own;
    __syncthreads();
    if (t == 1023) carry = before + v;
    __syncthreads();
  }
  if (t == 0) base[n] = carry;
}

// ---- log(x) in double-double, 0 < x < 1 ------------------------------------------------------------------------------
struct dd {
  double hi, lo;
};
__host__ __device__ __forceinline__ dd two_sum(double a, double b) {
  const double s = a + b, bb = s - a;
  return {s, (a - (s - bb)) + (b - bb)};
}
__host__ __device__ __forceinline__ dd quick_two_sum(double a, double b) {      // |a| >= |b|
  const double s = a + b;
  return {s, b - (s - a)};
}
__host__ __device__ __forceinline__ dd two_prod(double a, double b) {
  const double p = a * b;
  return {p, fma(a, b, -p)};
}
__host__ __device__ __forceinline__ dd dd_add(dd a, dd b) {
  dd s = two_sum(a.hi, b.hi);
  const dd t = two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = quick_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return quick_two_sum(s.hi, s.lo);
}
__host__ __device__ __forceinline__ dd dd_mul(dd a, dd b) {
  dd p = two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return quick_two_sum(p.hi, p.lo);
}

// log c for c = k / 32, k = 23 .. 45, as double-doubles (filled once by the host from the same series, 2^-100)
__constant__ double kLogTabHi[23];
__constant__ double kLogTabLo[23];

// log x for 0 < x < 1 as a double-double with relative error < 2^-66 (6e-5 ulp of the rounded result): x = m 2^e with m in
// [0.7071, 1.4142); c = the nearest multiple of 1/32, so that s = (m - c) / (m + c) has |s| <= 0.0112 and
// log m = log c + 2 atanh(s) = log c + 2 s (1 + s^2 / 3 + s^4 / 5 + ...): s in double-double (the quotient's remainder
// through one fma), the bracket's tail -- below 2^-14 -- in plain double.
__host__ __device__ __forceinline__ dd dd_log_core(double m, int e, double c, dd logc) {
  const double num = m - c;                      // exact (Sterbenz: c / 2 <= m <= 2 c)
  const dd den = two_sum(m, c);
  const double q1 = num / den.hi;
  const double rem = fma(-q1, den.hi, num) - q1 * den.lo;      // num - q1 den, to double accuracy
  const double q2 = rem / den.hi;
  const dd s = quick_two_sum(q1, q2);
  const double z = s.hi * s.hi;
  const double tail = z * (1.0 / 3.0 + z * (1.0 / 5.0 + z * (1.0 / 7.0 + z * (1.0 / 9.0 + z * (1.0 / 11.0 + z * (1.0 / 13.0))))));
  dd lm = dd_add(s, dd_mul(s, dd{tail, 0.0}));
  lm.hi *= 2.0, lm.lo *= 2.0;
  const dd ln2 = {0x1.62e42fefa39efp-1, 0x1.abc9e3b39803fp-56};
  return dd_add(dd_add(dd_mul(ln2, dd{(double)e, 0.0}), logc), lm);
}

__device__ __forceinline__ dd dd_log(double x) {
  int e;
  double m = frexp(x, &e);                      // x = m 2^e, m in [0.5, 1)
  if (m < 0.70710678118654752440) m *= 2.0, e -= 1;      // m in [0.7071, 1.4142)
  const int k = (int)rint(32.0 * m);            // 23 .. 45
  return dd_log_core(m, e, (double)k * 0.03125, dd{kLogTabHi[k - 23], kLogTabLo[k - 23]});
}

struct EmitArgs {
  const uint32_t* words;
  int64_t attempts;
  const int64_t* base;            // exclusive acceptance prefix per workgroup
  int64_t pairs, n_vals, first;   // pairs wanted; values wanted (after the cached one); index of the first value (0 / 1)
  int64_t d, row_begin, rows, ld;
  double* slot;
  double* hard_seg;               // [n_wg][kHardPerWg][4]: q, x1, x2, r2 of attempts the host finishes, per workgroup
  int* hard_cnt;                  // [n_wg]: entries of each segment (may exceed kHardPerWg: overflow, the host falls back)
  int64_t* a_star;                // attempt that produced the last pair
  double* last_x1f;               // its f x1 (the value numpy caches when the count is odd)
};

__device__ __forceinline__ void put_value(const EmitArgs& a, int64_t o, double v) {
  const int64_t row = o / a.d, col = o - row * a.d;
  if (row >= a.row_begin && row < a.row_begin + a.rows) a.slot[(row - a.row_begin) * a.ld + col] = v;
}

// kExact: log(r2) by the host C library's own sequence of operations on its own table (vb_glibc_log.h; `logtab` proven
// against the host's log by vb_glibc_log_locate): every value is final here, no list, nothing for the host to finish.
template <bool kExact>
__global__ void __launch_bounds__(256) mtd_emit_kernel(const EmitArgs a, const GlibcLogData* __restrict__ logtab) {
  __shared__ int wave_cnt[4];
  __shared__ int hard_fill;
  __shared__ GlibcLogData lt;
  if (kExact) {
    const double* src = reinterpret_cast<const double*>(logtab);
    double* dst = reinterpret_cast<double*>(&lt);
    for (int i = threadIdx.x; i < (int)(sizeof(GlibcLogData) / sizeof(double)); i += 256) dst[i] = src[i];
  }
  if (threadIdx.x == 0) hard_fill = 0;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * kAttemptsPerWg + 4 * t;
  Attempt at[4];
  int mine = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    at[k].ok = false;
    if (i0 + k < a.attempts) at[k] = attempt_at(a.words, i0 + k);
    mine += at[k].ok ? 1 : 0;
  }
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int u = __shfl_up(incl, off, 64);
    if (lane >= off) incl += u;
  }
  if (lane == 63) wave_cnt[wv] = incl;
  __syncthreads();
  int64_t q = a.base[blockIdx.x] + (incl - mine);
  for (int w = 0; w < wv; ++w) q += wave_cnt[w];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!at[k].ok) continue;
    if (q < a.pairs) {
      const double x1 = at[k].x1, x2 = at[k].x2, r2 = at[k].r2;
      if (kExact) {
        const double f = sqrt(-2.0 * glibc_log(r2, lt) / r2);
        const double v0 = f * x2, v1 = f * x1;
        put_value(a, a.first + 2 * q, v0);
        if (2 * q + 1 < a.n_vals) put_value(a, a.first + 2 * q + 1, v1);
        if (q == a.pairs - 1) {
          a.a_star[0] = i0 + k;
          a.last_x1f[0] = v1;
          a.a_star[1] = 0;
        }
        ++q;
        continue;
      }
      const dd L = dd_log(r2);
      // correctly rounded log = L.hi; the C library's may be the neighbour on the side of L.lo when the true value is
      // within 0.03 ulp of the midpoint (its error bound is 0.52 ulp)
      const double f = sqrt(-2.0 * L.hi / r2);
      double v0 = f * x2, v1 = f * x1;
      const double ulp = ldexp(1.0, ilogb(L.hi) - 52);
      bool hard = false;
      // (L.hi a power of two: the spacing below it is half of `ulp`, the test would look at the wrong boundary -- a
      // one-in-2^52 event, simply handed to the host)
      int e_hi;
      const bool pow2 = frexp(L.hi, &e_hi) == -0.5;
      if (fabs(L.lo) > 0.478 * ulp || pow2) {
        const double alt = L.lo > 0.0 ? nextafter(L.hi, INFINITY) : nextafter(L.hi, -INFINITY);
        const double f2 = sqrt(-2.0 * alt / r2);
        hard = (f2 * x2 != v0) || (f2 * x1 != v1) || pow2;
      }
      if (hard) {
        const int h = atomicAdd(&hard_fill, 1);      // LDS: the order inside a segment is arbitrary, its content is not
        if (h < kHardPerWg) {
          double* o = a.hard_seg + 4 * ((int64_t)blockIdx.x * kHardPerWg + h);
          o[0] = (double)q, o[1] = x1, o[2] = x2, o[3] = r2;
        }
      } else {
        put_value(a, a.first + 2 * q, v0);
        if (2 * q + 1 < a.n_vals) put_value(a, a.first + 2 * q + 1, v1);
      }
      if (q == a.pairs - 1) {
        a.a_star[0] = i0 + k;
        a.last_x1f[0] = v1;          // (replaced by the host's value if this attempt is on the list)
        a.a_star[1] = hard ? 1 : 0;
      }
    }
    ++q;
  }
  __syncthreads();
  if (!kExact && threadIdx.x == 0) a.hard_cnt[blockIdx.x] = hard_fill;
}

// the segments, one behind the other: list[hbase[wg] + i] = segment wg's entry i
__global__ void __launch_bounds__(64) mtd_compact_kernel(const double* __restrict__ seg, const int* __restrict__ cnt,
                                                         const int64_t* __restrict__ hbase, double* __restrict__ list,
                                                         int64_t* __restrict__ overflow) {
  const int n = min(cnt[blockIdx.x], kHardPerWg);
  const int t = threadIdx.x;
  if (t == 0 && cnt[blockIdx.x] > kHardPerWg) overflow[0] = 1;
  if (t < n) {
    const double* src = seg + 4 * ((int64_t)blockIdx.x * kHardPerWg + t);
    double* dst = list + 4 * (hbase[blockIdx.x] + t);
    dst[0] = src[0], dst[1] = src[1], dst[2] = src[2], dst[3] = src[3];
  }
}

// values the host finished: list entries (q, v0, v1)
__global__ void __launch_bounds__(256) mtd_patch_kernel(const EmitArgs a, const double* __restrict__ fixed, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t q = (int64_t)fixed[3 * i];
  put_value(a, a.first + 2 * q, fixed[3 * i + 1]);
  if (2 * q + 1 < a.n_vals) put_value(a, a.first + 2 * q + 1, fixed[3 * i + 2]);
}

__global__ void mtd_first_value_kernel(const EmitArgs a, double v) { put_value(a, 0, v); }

}  // namespace

// The generator's next `n_words` output words (untempered) on the device, from the state (key, pos): out->words[0] is the
// word at `pos`.  legacy_work is laid out [polynomials | key | 16 scalars (zeroed) | log table | words | stream states |
// ladder sequences | `extra_u32` words for the caller]; every region starts 16-byte aligned.  VB_ERR_UNSUPPORTED: more
// streams than the jump ladder reaches.
int legacy_mt_words(vb_ctx* ctx, const uint32_t key[624], int pos, int64_t n_words, size_t extra_u32, LegacyWords* out) {
  hipStream_t st = ctx->stream;
  const int64_t pre = kN - pos;                                      // unread words of the current block
  const int64_t n_blocks = n_words > pre ? (n_words - pre + kN - 1) / kN : 1;
  const int64_t streams = (n_blocks + kMtBlocksPerStream - 1) / kMtBlocksPerStream;
  if (streams > ((int64_t)1 << (2 * kMtJumpRounds))) return VB_ERR_UNSUPPORTED;
  size_t off = 0;
  auto carve = [&off](size_t words32) {
    const size_t o = off;
    off += (words32 + 3) & ~(size_t)3;
    return o;
  };
  int64_t pow2 = 1;      // (capacity of the state / sequence areas: a power of four)
  while (pow2 < streams) pow2 <<= 2;
  // (fixed-size regions first: the polynomials and the log table stay where they were uploaded from call to call)
  const size_t o_poly = carve((size_t)kMtJumpPolys * kN), o_key = carve(kN), o_scal = carve(16), o_meta = carve(4),
               o_log = carve(sizeof(GlibcLogData) / sizeof(uint32_t)),
               o_words = carve((size_t)(pre + n_blocks * kN) + 8), o_state = carve((size_t)pow2 * kN),
               o_seq = carve((size_t)(pow2 / 4 > 0 ? pow2 / 4 : 1) * kSeqWords), o_extra = carve(extra_u32);
  VB_TRY(ensure(ctx, ctx->legacy_work, off * sizeof(uint32_t)));
  uint32_t* base = (uint32_t*)ctx->legacy_work.ptr;
  uint32_t *words = base + o_words, *state = base + o_state, *seq = base + o_seq, *poly = base + o_poly, *key_dev = base + o_key;
  const GlibcLogData* host_tab = vb_glibc_log_locate();
  if (ctx->legacy_poly_at != (const void*)poly || ctx->legacy_poly_bytes != ctx->legacy_work.bytes) {      // (a new allocation is zeroed)
    VB_HIP(ctx, hipMemcpyAsync(poly, kMtJump, sizeof kMtJump, hipMemcpyHostToDevice, st));
    if (host_tab) VB_HIP(ctx, hipMemcpyAsync(base + o_log, host_tab, sizeof(GlibcLogData), hipMemcpyHostToDevice, st));
    ctx->legacy_poly_at = poly;
    ctx->legacy_poly_bytes = ctx->legacy_work.bytes;
  }
  // (the key through a mapped staging slot and a copy kernel: a pageable source makes the runtime stage and wait)
  VB_TRY(push_small(ctx, st, key, kN * sizeof(uint32_t), key_dev));
  VB_HIP(ctx, hipMemsetAsync(state, 0, (size_t)pow2 * kN * sizeof(uint32_t), st));
  VB_HIP(ctx, hipMemsetAsync(base + o_scal, 0, 16 * sizeof(uint32_t), st));
  hipLaunchKernelGGL(mtd_first_kernel, dim3(1), dim3(256), 0, st, (const uint32_t*)key_dev, pos, words, state);
  int r = 0;
  for (int64_t count = 1; count < streams; count <<= 2, ++r) {      // radix 4: the known stream starts quadruple per round
    const int n_src = (int)std::min<int64_t>(count, streams - count);      // sources with at least one target
    hipLaunchKernelGGL(mtd_seq_kernel, dim3((unsigned)n_src), dim3(256), 0, st, (const uint32_t*)state, seq);
    hipLaunchKernelGGL(mtd_corr_kernel, dim3((unsigned)n_src, kCorrSlices, 3), dim3(kCorrThreads), 0, st, (const uint32_t*)poly, 3 * r,
                       (const uint32_t*)seq, state, (int)count, (int)streams);
  }
  hipLaunchKernelGGL(mtd_stream_kernel, dim3((unsigned)streams), dim3(256), 0, st, (const uint32_t*)state, words, pre, n_blocks);
  VB_HIP(ctx, hipGetLastError());
  out->words = words;
  out->scal = (int64_t*)(base + o_scal);
  out->extra = base + o_extra;
  out->logtab = host_tab ? (const void*)(base + o_log) : nullptr;
  out->pre = pre;
  out->n_words = pre + n_blocks * kN;
  out->key_io = key_dev;
  out->meta = (int64_t*)(base + o_meta);
  return VB_OK;
}

// Where the generator stands after consuming `w_star` of those words: the new position and, when the position has left
// the block the call started in, that block's words into key[] (one small copy down + a synchronisation).
int legacy_mt_finish(vb_ctx* ctx, const LegacyWords& lw, int64_t w_star, uint32_t key[624], int* pos) {
  if (w_star <= lw.pre) {
    *pos = (int)(*pos + w_star);               // still inside the block the call started in
    return VB_OK;
  }
  const int64_t offw = w_star - lw.pre;
  int64_t key_block = offw / kN;
  int new_pos = (int)(offw % kN);
  if (new_pos == 0) key_block -= 1, new_pos = kN;      // exactly at a block end: numpy refreshes lazily
  if (key_block < 0 || lw.pre + (key_block + 1) * kN > lw.n_words)
    return fail(ctx, VB_ERR_STATE, "legacy generator: inconsistent end position");
  VB_HIP(ctx, hipMemcpyAsync(key, lw.words + lw.pre + key_block * kN, kN * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *pos = new_pos;
  return VB_OK;
}

namespace {
// the end block of a draw whose consumed word count is still on the device: meta = [status, new position]; status 0: the
// block's words are in key_out; 2: the draw ended inside the block it started in (position += w_star, key unchanged);
// 1: inconsistent end position
__global__ void __launch_bounds__(256) mtd_key_gather_kernel(const uint32_t* __restrict__ words, int64_t pre, int64_t n_words,
                                                             const int64_t* __restrict__ src, int64_t mult, int64_t add,
                                                             uint32_t* __restrict__ key_out, int64_t* __restrict__ meta) {
  const int64_t w_star = mult * (src[0] + add);
  int64_t status = 0, new_pos = 0, key_block = 0;
  if (w_star <= pre) {
    status = 2;
    new_pos = w_star;
  } else {
    const int64_t offw = w_star - pre;
    key_block = offw / kN;
    new_pos = offw % kN;
    if (new_pos == 0) key_block -= 1, new_pos = kN;      // exactly at a block end: numpy refreshes lazily
    if (key_block < 0 || pre + (key_block + 1) * kN > n_words) status = 1;
  }
  if (status == 0)
    for (int t = threadIdx.x; t < kN; t += 256) key_out[t] = words[pre + key_block * kN + t];
  if (threadIdx.x == 0) meta[0] = status, meta[1] = new_pos;
}
}  // namespace

int legacy_mt_finish_fetch(vb_ctx* ctx, const LegacyWords& lw, const int64_t* src_dev, int64_t mult, int64_t add,
                           const FetchSeg* extra, int n_extra, bool (*accept)(void*), void* accept_arg, uint32_t key[624],
                           int* pos) {
  if (n_extra > 5) return fail(ctx, VB_ERR_INVALID, "legacy_mt_finish_fetch: too many segments");
  hipLaunchKernelGGL(mtd_key_gather_kernel, dim3(1), dim3(256), 0, ctx->stream, lw.words, lw.pre, lw.n_words, src_dev, mult, add,
                     lw.key_io, lw.meta);
  VB_HIP(ctx, hipGetLastError());
  uint32_t block[kN];
  int64_t meta[2] = {1, 0};
  int64_t wsrc = 0;
  FetchSeg segs[8];
  for (int k = 0; k < n_extra; ++k) segs[k] = extra[k];
  segs[n_extra] = FetchSeg{lw.key_io, kN * sizeof(uint32_t), block};
  segs[n_extra + 1] = FetchSeg{lw.meta, sizeof meta, meta};
  segs[n_extra + 2] = FetchSeg{src_dev, sizeof wsrc, &wsrc};
  VB_TRY(fetch_blocking(ctx, ctx->stream, segs, n_extra + 3));
  if (accept && !accept(accept_arg)) return VB_ERR_UNSUPPORTED;
  if (meta[0] == 1) return fail(ctx, VB_ERR_STATE, "legacy generator: inconsistent end position");
  if (meta[0] == 2) {
    *pos = (int)(*pos + mult * (wsrc + add));      // still inside the block the call started in
    return VB_OK;
  }
  memcpy(key, block, sizeof block);
  *pos = (int)meta[1];
  return VB_OK;
}

namespace {
// the deferred finish's landing area in mapped host memory (uint64 words): [0, 8) extras 0 | [8, 16) extras 1 | [16, 328) the
// end block's 624 words | [328, 330) meta | [330] consumed-word source | [344] completion word.  One workgroup.
__global__ void __launch_bounds__(256) mtd_finish_copy_kernel(const unsigned long long* __restrict__ e0, int n0,
                                                              const unsigned long long* __restrict__ e1, int n1,
                                                              const unsigned long long* __restrict__ block,
                                                              const unsigned long long* __restrict__ meta,
                                                              const unsigned long long* __restrict__ src,
                                                              unsigned long long* __restrict__ out, unsigned long long seq) {
  const int t = threadIdx.x;
  if (t < n0) out[t] = e0[t];
  if (t < n1) out[8 + t] = e1[t];
  for (int i = t; i < kN / 2; i += 256) out[16 + i] = block[i];
  if (t < 2) out[328 + t] = meta[t];
  if (t == 0) out[330] = src[0];
  __threadfence_system();
  __syncthreads();
  if (t == 0) __hip_atomic_store(out + 344, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

int legacy_finish_launch(vb_ctx* ctx, hipStream_t st, const LegacyFinish& f, unsigned long long* landing_dev, unsigned long long seq) {
  hipLaunchKernelGGL(mtd_key_gather_kernel, dim3(1), dim3(256), 0, st, f.lw.words, f.lw.pre, f.lw.n_words, f.src_dev, f.mult, f.add,
                     f.lw.key_io, f.lw.meta);
  hipLaunchKernelGGL(mtd_finish_copy_kernel, dim3(1), dim3(256), 0, st, (const unsigned long long*)f.extra_src[0], f.extra_words[0],
                     (const unsigned long long*)f.extra_src[1], f.extra_words[1], (const unsigned long long*)f.lw.key_io,
                     (const unsigned long long*)f.lw.meta, (const unsigned long long*)f.src_dev, landing_dev, seq);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

int legacy_finish_complete(vb_ctx* ctx, const LegacyFinish& f, const unsigned long long* landing, uint32_t key[624], int* pos,
                           int* has_gauss, double* gauss) {
  int64_t e0[8], e1[8], meta[2], wsrc;
  memcpy(e0, landing, sizeof e0);
  memcpy(e1, landing + 8, sizeof e1);
  memcpy(meta, landing + 328, sizeof meta);
  memcpy(&wsrc, landing + 330, sizeof wsrc);
  if (f.kind == 0) {
    if (e1[0] < f.pairs) return VB_ERR_UNSUPPORTED;      // the word budget fell short
  } else {
    if (!(e0[0] >= 0 && e0[3] == 0)) return VB_ERR_UNSUPPORTED;
  }
  if (meta[0] == 1) return fail(ctx, VB_ERR_STATE, "legacy generator: inconsistent end position");
  if (meta[0] == 2) {
    *pos = (int)(f.pos_in + f.mult * (wsrc + f.add));      // still inside the block the draw started in: key unchanged
  } else {
    memcpy(key, landing + 16, kN * sizeof(uint32_t));
    *pos = (int)meta[1];
  }
  if (f.kind == 0) {
    double last_x1f;
    memcpy(&last_x1f, &e0[3], sizeof last_x1f);
    *has_gauss = (f.n_vals & 1) ? 1 : 0;
    *gauss = (f.n_vals & 1) ? last_x1f : 0.0;
  } else {
    *has_gauss = e0[1] ? 1 : 0;
    double cached = 0.0;
    if (e0[1]) memcpy(&cached, &e0[2], sizeof cached);
    *gauss = cached;
  }
  return VB_OK;
}

// The draws s.randn(n_total, d) of the generator whose state is (key, pos, has_gauss, gauss): rows [row_begin,
// row_begin + rows) into `ns`; the state afterwards in the same variables.  VB_ERR_UNSUPPORTED: nothing changed, draw on
// the host.
int legacy_dev_randn(vb_ctx* ctx, uint32_t key[624], int* pos, int* has_gauss, double* gauss, const NoiseSlot& ns,
                     int64_t n_total, int64_t d, int64_t row_begin, int64_t rows, LegacyFinish* defer) {
  hipStream_t st = ctx->stream;
  if (!ctx->legacy_table_ready) {      // (constant memory is per device: once per context, not once per process)
    // log(k / 32) in double-double by the long series on the host: 2 atanh(s), s = (c - 1) / (c + 1), |s| <= 0.17,
    // 40 terms (s^2 <= 0.03: 2^-200), every operation in double-double
    double hi[23], lo[23];
    for (int k = 23; k <= 45; ++k) {
      const double c = k * 0.03125;
      const dd num = {c - 1.0, 0.0}, den = two_sum(c, 1.0);
      const double q1 = num.hi / den.hi;
      const double q2 = (std::fma(-q1, den.hi, num.hi) - q1 * den.lo) / den.hi;
      const dd sv = quick_two_sum(q1, q2), z = dd_mul(sv, sv);
      dd p = {0.0, 0.0};
      for (int j = 40; j >= 0; --j) {
        const double n = 2.0 * j + 1.0, ch = 1.0 / n, cl = std::fma(-ch, n, 1.0) / n;
        p = dd_add(dd_mul(p, z), dd{ch, cl});
      }
      dd l = dd_mul(sv, p);
      hi[k - 23] = 2.0 * l.hi, lo[k - 23] = 2.0 * l.lo;
    }
    VB_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(kLogTabHi), hi, sizeof hi));
    VB_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(kLogTabLo), lo, sizeof lo));
    ctx->legacy_table_ready = true;
  }
  const int64_t total = n_total * d;
  const int64_t first = (*has_gauss && total > 0) ? 1 : 0;
  const int64_t n_vals = total - first;
  if (n_vals <= 0) return VB_ERR_UNSUPPORTED;
  const int64_t pairs = (n_vals + 1) / 2;
  const double p_acc = 0.78539816339744830962;
  const double sigma = std::sqrt((double)pairs * (1.0 - p_acc)) / p_acc;
  const int64_t attempts = (int64_t)((double)pairs / p_acc + 10.0 * sigma) + 64;
  const int64_t n_wg = (attempts + kAttemptsPerWg - 1) / kAttemptsPerWg;
  const int64_t hard_cap = pairs / 8 + 1024;
  const bool exact = vb_glibc_log_locate() != nullptr;
  if (defer && !exact) return VB_ERR_UNSUPPORTED;

  // the caller's part of the scratch (uint32 units, every region 16-byte aligned): cnt | base | hard | ...
  size_t off = 0;
  auto carve = [&off](size_t words32) {
    const size_t o = off;
    off += (words32 + 3) & ~(size_t)3;
    return o;
  };
  const size_t o_cnt = carve((size_t)n_wg), o_base = carve(2 * (size_t)(n_wg + 1));
  size_t o_hard = 0, o_hseg = 0, o_hcnt = 0, o_hbase = 0, o_fixed = 0;
  if (!exact) {
    o_hard = carve(2 * 4 * (size_t)hard_cap), o_hseg = carve(2 * 4 * (size_t)n_wg * kHardPerWg), o_hcnt = carve((size_t)n_wg);
    o_hbase = carve(2 * (size_t)(n_wg + 1)), o_fixed = carve(2 * 3 * (size_t)hard_cap);
  }
  LegacyWords lw;
  VB_TRY(legacy_mt_words(ctx, key, *pos, 4 * attempts, off, &lw));
  uint32_t* base = lw.extra;
  const uint32_t* words = lw.words;
  int* cnt = (int*)(base + o_cnt);
  int64_t* pbase = (int64_t*)(base + o_base);
  hipLaunchKernelGGL(mtd_count_kernel, dim3((unsigned)n_wg), dim3(256), 0, st, words, attempts, cnt);
  hipLaunchKernelGGL(mtd_scan_kernel, dim3(1), dim3(1024), 0, st, (const int*)cnt, n_wg, pbase);
  EmitArgs a;
  a.words = words, a.attempts = attempts, a.base = pbase;
  a.pairs = pairs, a.n_vals = n_vals, a.first = first;
  a.d = d, a.row_begin = row_begin, a.rows = rows, a.ld = ns.ld;
  a.slot = (double*)ns.buf.ptr;
  a.hard_seg = exact ? nullptr : (double*)(base + o_hseg);
  a.hard_cnt = exact ? nullptr : (int*)(base + o_hcnt);
  double* hard_list = (double*)(base + o_hard);
  int64_t* hbase = (int64_t*)(base + o_hbase);
  int64_t* scal = lw.scal;      // [0] a*, [1] last pair on the list?, [3] f x1 of the last pair
  a.a_star = scal;
  a.last_x1f = (double*)(scal + 3);
  if (first) hipLaunchKernelGGL(mtd_first_value_kernel, dim3(1), dim3(1), 0, st, a, *gauss);
  if (exact) {
    hipLaunchKernelGGL(mtd_emit_kernel<true>, dim3((unsigned)n_wg), dim3(256), 0, st, a, (const GlibcLogData*)lw.logtab);
  } else {
    hipLaunchKernelGGL(mtd_emit_kernel<false>, dim3((unsigned)n_wg), dim3(256), 0, st, a, (const GlibcLogData*)nullptr);
    hipLaunchKernelGGL(mtd_scan_kernel, dim3(1), dim3(1024), 0, st, (const int*)a.hard_cnt, n_wg, hbase);
    hipLaunchKernelGGL(mtd_compact_kernel, dim3((unsigned)n_wg), dim3(64), 0, st, (const double*)a.hard_seg,
                       (const int*)a.hard_cnt, (const int64_t*)hbase, hard_list, scal + 4);
  }
  VB_HIP(ctx, hipGetLastError());
  if (exact && defer) {      // (look-ahead draw: the finish is launched, polled and completed by the caller)
    defer->lw = lw;
    defer->src_dev = scal, defer->mult = 4, defer->add = 1;
    defer->extra_src[0] = scal, defer->extra_words[0] = 4;
    defer->extra_src[1] = pbase + n_wg, defer->extra_words[1] = 1;
    defer->kind = 0, defer->pairs = pairs, defer->n_vals = n_vals, defer->pos_in = *pos;
    return VB_OK;
  }
  if (exact) {
    // no list, no host arithmetic: the scalars, the acceptance count and the generator's end block in ONE fetch (the
    // end position is a device scalar: legacy_mt_finish_fetch finds and gathers the block there)
    int64_t res4[4] = {0, 0, 0, 0}, got = 0;
    struct Check {
      const int64_t* got;
      int64_t pairs;
    } chk{&got, pairs};
    const FetchSeg extra[2] = {{scal, sizeof res4, res4}, {pbase + n_wg, sizeof got, &got}};
    int new_pos = *pos;
    const int rc = legacy_mt_finish_fetch(ctx, lw, scal, 4, 1, extra, 2,
                                          [](void* p) { return *((Check*)p)->got >= ((Check*)p)->pairs; }, &chk, key, &new_pos);
    if (rc == VB_ERR_UNSUPPORTED) return rc;      // (the word budget fell short: host path, nothing changed)
    VB_TRY(rc);
    double last_x1f;
    memcpy(&last_x1f, &res4[3], sizeof last_x1f);
    *pos = new_pos;
    *has_gauss = (n_vals & 1) ? 1 : 0;
    *gauss = (n_vals & 1) ? last_x1f : 0.0;
    return VB_OK;
  }
  // results through one pinned buffer: [scalars 4 | accepted | list (first `spec` entries, speculatively) | fixed]
  const int64_t spec = std::min<int64_t>(hard_cap, pairs / 20 + 256);      // ~1.7 x the expected list length
  const size_t pin_doubles = 8 + (exact ? 0 : (size_t)4 * hard_cap + (size_t)3 * hard_cap) + 8;
  if (ctx->legacy_pin_doubles < pin_doubles) {
    if (ctx->legacy_pin) VB_HIP(ctx, hipHostFree(ctx->legacy_pin));
    ctx->legacy_pin = nullptr;
    VB_HIP(ctx, hipHostMalloc((void**)&ctx->legacy_pin, pin_doubles * sizeof(double), hipHostMallocDefault));
    ctx->legacy_pin_doubles = pin_doubles;
  }
  int64_t* res = (int64_t*)ctx->legacy_pin;                 // [0] a*, [1] last pair listed?, [2] list length, [3] f x1 bits
  int64_t* accepted = res + 4;
  double* list = ctx->legacy_pin + 8;
  double* fixed = list + 4 * hard_cap;
  res[2] = 0, res[5] = 0;
  VB_HIP(ctx, hipMemcpyAsync(res, scal, 4 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  VB_HIP(ctx, hipMemcpyAsync(accepted, pbase + n_wg, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  if (!exact) {
    VB_HIP(ctx, hipMemcpyAsync(res + 2, hbase + n_wg, sizeof(int64_t), hipMemcpyDeviceToHost, st));      // list length
    VB_HIP(ctx, hipMemcpyAsync(res + 5, scal + 4, sizeof(int64_t), hipMemcpyDeviceToHost, st));          // segment overflow?
    VB_HIP(ctx, hipMemcpyAsync(list, hard_list, (size_t)4 * spec * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  VB_HIP(ctx, hipStreamSynchronize(st));
  const int64_t n_hard = exact ? 0 : res[2];
  // (budget, list or a workgroup's segment fell short: host path)
  if (*accepted < pairs || n_hard > hard_cap || (!exact && res[5] != 0)) return VB_ERR_UNSUPPORTED;
  double last_x1f;
  memcpy(&last_x1f, &res[3], sizeof last_x1f);
  const int64_t w_star = 4 * (res[0] + 1);      // the generator stands just behind the last consumed attempt
  if (n_hard > spec) {
    VB_HIP(ctx, hipMemcpyAsync(list + 4 * spec, hard_list + 4 * spec, (size_t)4 * (n_hard - spec) * sizeof(double),
                               hipMemcpyDeviceToHost, st));
    VB_HIP(ctx, hipStreamSynchronize(st));
  }
  if (n_hard > 0) {
    vb_legacy_finish_pairs(list, n_hard, fixed);
    for (int64_t i = 0; i < n_hard; ++i)
      if ((int64_t)fixed[3 * i] == pairs - 1) last_x1f = fixed[3 * i + 2];
    double* fixed_dev = (double*)(base + o_fixed);
    VB_HIP(ctx, hipMemcpyAsync(fixed_dev, fixed, (size_t)3 * n_hard * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(mtd_patch_kernel, dim3((unsigned)((n_hard + 255) / 256)), dim3(256), 0, st, a,
                       (const double*)fixed_dev, n_hard);
    VB_HIP(ctx, hipStreamSynchronize(st));          // the patch has read the pinned list
  }
  int new_pos = *pos;
  VB_TRY(legacy_mt_finish(ctx, lw, w_star, key, &new_pos));
  *pos = new_pos;
  *has_gauss = (n_vals & 1) ? 1 : 0;
  *gauss = (n_vals & 1) ? last_x1f : 0.0;
  return VB_OK;
}

}  // namespace vb
