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
//     stream starts double every round (mtd_seq_kernel + mtd_corr_kernel), then one workgroup per stream runs the
//     recurrence (mtd_stream_kernel: 227-way parallel inside a block, three barriers per block).
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
#include "vb_mt_jump.h"

#include <cmath>
#include <vector>

namespace vb {

namespace {

constexpr int kN = 624, kM = 397;
constexpr int kSeqBlocks = 33;                    // 33 x 624 = 20 592 >= 19 937 + 623 words
constexpr int kSeqWords = kSeqBlocks * kN;
constexpr int kCorrSlices = 16;
constexpr int kAttemptsPerWg = 1024;              // 256 threads x 4 attempts

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

// The next 624 words of the recurrence, in place in LDS, by 256 threads (key consistent on entry).  key[k] <- key[k + 397]
// ^ mix(key[k], key[k + 1]) for k < 227 only reads old words; k in [227, 454) reads the NEW key[k - 227], which is the
// same thread's own first result; k in [454, 623) the same thread's second result; k = 623 needs the new key[0].
__device__ __forceinline__ void mt_refresh_lds(uint32_t* key, int t) {
  uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0, c0 = 0, c1 = 0, m = 0;
  if (t < 227) {
    a0 = key[t], a1 = key[t + 1], m = key[t + kM];
    b0 = key[t + 227], b1 = key[t + 228];
  }
  if (t < 170) {
    c0 = key[t + 454];
    c1 = t + 455 < kN ? key[t + 455] : 0u;
  }
  __syncthreads();
  uint32_t n1 = 0;
  if (t < 227) {
    const uint32_t n0 = m ^ mt_mix(a0, a1);
    key[t] = n0;
    n1 = n0 ^ mt_mix(b0, b1);
    key[t + 227] = n1;
  }
  if (t < 169) key[t + 454] = n1 ^ mt_mix(c0, c1);
  __syncthreads();
  if (t == 169) key[kN - 1] = n1 ^ mt_mix(c0, key[0]);
  __syncthreads();
}

// words[0 .. pre) = the unread rest of the current block (untempered), state0 = the next block
__global__ void __launch_bounds__(256) mtd_first_kernel(const uint32_t* __restrict__ key_in, int pos,
                                                        uint32_t* __restrict__ words, uint32_t* __restrict__ state0) {
  __shared__ uint32_t key[kN];
  const int t = threadIdx.x;
  for (int i = t; i < kN; i += 256) key[i] = key_in[i];
  __syncthreads();
  for (int i = pos + t; i < kN; i += 256) words[i - pos] = key[i];
  mt_refresh_lds(key, t);
  for (int i = t; i < kN; i += 256) state0[i] = key[i];
}

// seq[s] = 33 consecutive blocks starting with state[s]
__global__ void __launch_bounds__(256) mtd_seq_kernel(const uint32_t* __restrict__ state, uint32_t* __restrict__ seq) {
  __shared__ uint32_t key[kN];
  const int t = threadIdx.x;
  const uint32_t* src = state + (size_t)blockIdx.x * kN;
  uint32_t* dst = seq + (size_t)blockIdx.x * kSeqWords;
  for (int i = t; i < kN; i += 256) key[i] = src[i];
  __syncthreads();
  for (int b = 0; b < kSeqBlocks; ++b) {
    for (int i = t; i < kN; i += 256) dst[(size_t)b * kN + i] = key[i];
    if (b + 1 < kSeqBlocks) mt_refresh_lds(key, t);
  }
}

// state[count + s][j] ^= XOR over the coefficients i of this slice of seq[s][i + j]   (dst zeroed beforehand)
__global__ void __launch_bounds__(640) mtd_corr_kernel(const uint32_t* __restrict__ poly, const uint32_t* __restrict__ seq,
                                                       uint32_t* __restrict__ state, int count, int n_new) {
  constexpr int kSlice = (19937 + kCorrSlices - 1) / kCorrSlices;      // 1247 coefficients
  __shared__ uint32_t u[kSlice + kN];
  __shared__ uint32_t g[(kSlice + 31) / 32 + 2];
  const int s = blockIdx.x, slice = blockIdx.y, t = threadIdx.x;
  if (s >= n_new) return;
  const int i0 = slice * kSlice, i1 = min(19937, i0 + kSlice);
  const uint32_t* src = seq + (size_t)s * kSeqWords + i0;
  for (int i = t; i < i1 - i0 + kN - 1; i += 640) u[i] = src[i];
  const int w0 = i0 >> 5, nw = ((i1 + 31) >> 5) - w0;
  for (int i = t; i < nw; i += 640) g[i] = poly[w0 + i];
  __syncthreads();
  if (t >= kN) return;
  uint32_t acc = 0;
  for (int i = i0; i < i1; ++i) {
    const uint32_t bit = (g[(i >> 5) - w0] >> (i & 31)) & 1u;      // uniform across the workgroup
    if (bit) acc ^= u[i - i0 + t];
  }
  atomicXor(&state[(size_t)(count + s) * kN + t], acc);      // integer: the result does not depend on the order
}

// stream s: blocks [s B, min((s + 1) B, n_blocks)) of the stream into words[pre + block * 624 ...] (untempered)
__global__ void __launch_bounds__(256) mtd_stream_kernel(const uint32_t* __restrict__ state, uint32_t* __restrict__ words,
                                                         int64_t pre, int64_t n_blocks) {
  __shared__ uint32_t key[kN];
  const int t = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * kMtBlocksPerStream;
  const int64_t b1 = b0 + kMtBlocksPerStream < n_blocks ? b0 + kMtBlocksPerStream : n_blocks;
  const uint32_t* src = state + (size_t)blockIdx.x * kN;
  for (int i = t; i < kN; i += 256) key[i] = src[i];
  __syncthreads();
  for (int64_t b = b0; b < b1; ++b) {
    uint32_t* dst = words + pre + b * kN;
    for (int i = t; i < kN; i += 256) dst[i] = key[i];
    if (b + 1 < b1) mt_refresh_lds(key, t);
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
__device__ __forceinline__ dd two_sum(double a, double b) {
  const double s = a + b, bb = s - a;
  return {s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd quick_two_sum(double a, double b) {      // |a| >= |b|
  const double s = a + b;
  return {s, b - (s - a)};
}
__device__ __forceinline__ dd two_prod(double a, double b) {
  const double p = a * b;
  return {p, fma(a, b, -p)};
}
__device__ __forceinline__ dd dd_add(dd a, dd b) {
  dd s = two_sum(a.hi, b.hi);
  const dd t = two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = quick_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return quick_two_sum(s.hi, s.lo);
}
__device__ __forceinline__ dd dd_mul(dd a, dd b) {
  dd p = two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return quick_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd dd_div(dd a, dd b) {      // three quotient digits
  const double q1 = a.hi / b.hi;
  dd r = dd_add(a, dd_mul(b, dd{-q1, 0.0}));
  const double q2 = r.hi / b.hi;
  r = dd_add(r, dd_mul(b, dd{-q2, 0.0}));
  const double q3 = r.hi / b.hi;
  dd q = quick_two_sum(q1, q2);
  return dd_add(q, dd{q3, 0.0});
}

__constant__ double kLogCoefHi[24];      // 1 / (2 k + 1) as a double-double, filled once by the host
__constant__ double kLogCoefLo[24];

__device__ __forceinline__ dd dd_log(double x) {
  int e;
  double m = frexp(x, &e);                      // x = m 2^e, m in [0.5, 1)
  if (m < 0.70710678118654752440) m *= 2.0, e -= 1;      // m in [0.7071, 1.4142)
  // log m = 2 atanh(s), s = (m - 1) / (m + 1): |s| <= 0.1716, s^2 <= 0.0295, 22 terms reach 2^-112
  const dd num = {m - 1.0, 0.0};                // exact (Sterbenz)
  const dd den = two_sum(m, 1.0);
  const dd s = dd_div(num, den);
  const dd s2 = dd_mul(s, s);
  dd p = {kLogCoefHi[22], kLogCoefLo[22]};
#pragma unroll
  for (int k = 21; k >= 0; --k) p = dd_add(dd_mul(p, s2), dd{kLogCoefHi[k], kLogCoefLo[k]});
  dd lm = dd_mul(s, p);
  lm.hi *= 2.0, lm.lo *= 2.0;
  const dd ln2 = {0x1.62e42fefa39efp-1, 0x1.abc9e3b39803fp-56};
  const dd el = dd_mul(ln2, dd{(double)e, 0.0});
  return dd_add(el, lm);
}

struct EmitArgs {
  const uint32_t* words;
  int64_t attempts;
  const int64_t* base;            // exclusive acceptance prefix per workgroup
  int64_t pairs, n_vals, first;   // pairs wanted; values wanted (after the cached one); index of the first value (0 / 1)
  int64_t d, row_begin, rows, ld;
  double* slot;
  double* hard;                   // [cap][4]: q, x1, x2, r2 of attempts the host finishes
  int64_t hard_cap;
  unsigned long long* hard_n;
  int64_t* a_star;                // attempt that produced the last pair
  double* last_x1f;               // its f x1 (the value numpy caches when the count is odd)
};

__device__ __forceinline__ void put_value(const EmitArgs& a, int64_t o, double v) {
  const int64_t row = o / a.d, col = o - row * a.d;
  if (row >= a.row_begin && row < a.row_begin + a.rows) a.slot[(row - a.row_begin) * a.ld + col] = v;
}

__global__ void __launch_bounds__(256) mtd_emit_kernel(const EmitArgs a) {
  __shared__ int wave_cnt[4];
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
      const dd L = dd_log(r2);
      // correctly rounded log = L.hi; the C library's may be the neighbour on the side of L.lo when the true value is
      // within 0.03 ulp of the midpoint (its error bound is 0.52 ulp)
      const double f = sqrt(-2.0 * L.hi / r2);
      double v0 = f * x2, v1 = f * x1;
      const double ulp = ldexp(1.0, ilogb(L.hi) - 52);
      bool hard = false;
      if (fabs(L.lo) > 0.47 * ulp) {
        const double alt = L.lo > 0.0 ? nextafter(L.hi, INFINITY) : nextafter(L.hi, -INFINITY);
        const double f2 = sqrt(-2.0 * alt / r2);
        hard = (f2 * x2 != v0) || (f2 * x1 != v1);
      }
      if (hard) {
        const unsigned long long h = atomicAdd(a.hard_n, 1ull);
        if ((int64_t)h < a.hard_cap) {
          double* o = a.hard + 4 * h;
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

// The draws s.randn(n_total, d) of the generator whose state is (key, pos, has_gauss, gauss): rows [row_begin,
// row_begin + rows) into `ns`; the state afterwards in the same variables.  VB_ERR_UNSUPPORTED: nothing changed, draw on
// the host.
int legacy_dev_randn(vb_ctx* ctx, uint32_t key[624], int* pos, int* has_gauss, double* gauss, const NoiseSlot& ns,
                     int64_t n_total, int64_t d, int64_t row_begin, int64_t rows) {
  hipStream_t st = ctx->stream;
  static bool coef_ready = false;
  if (!coef_ready) {
    double hi[24], lo[24];
    for (int k = 0; k < 24; ++k) {
      const double n = 2.0 * k + 1.0;
      hi[k] = 1.0 / n;
      lo[k] = std::fma(-hi[k], n, 1.0) / n;      // the exact remainder of the division, divided once more
    }
    VB_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(kLogCoefHi), hi, sizeof hi));
    VB_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(kLogCoefLo), lo, sizeof lo));
    coef_ready = true;
  }
  const int64_t total = n_total * d;
  const int64_t first = (*has_gauss && total > 0) ? 1 : 0;
  const int64_t n_vals = total - first;
  if (n_vals <= 0) return VB_ERR_UNSUPPORTED;
  const int64_t pairs = (n_vals + 1) / 2;
  const double p_acc = 0.78539816339744830962;
  const double sigma = std::sqrt((double)pairs * (1.0 - p_acc)) / p_acc;
  const int64_t attempts = (int64_t)((double)pairs / p_acc + 10.0 * sigma) + 64;
  const int64_t n_words = 4 * attempts;
  const int64_t pre = kN - *pos;                                      // unread words of the current block
  const int64_t n_blocks = n_words > pre ? (n_words - pre + kN - 1) / kN : 1;
  const int64_t streams = (n_blocks + kMtBlocksPerStream - 1) / kMtBlocksPerStream;
  if (streams > ((int64_t)1 << kMtJumpPolys)) return VB_ERR_UNSUPPORTED;
  const int64_t n_wg = (attempts + kAttemptsPerWg - 1) / kAttemptsPerWg;
  const int64_t hard_cap = pairs / 8 + 1024;

  // scratch (uint32 units, every region 16-byte aligned): words | states | seq | polys | key | cnt | base | hard | scalars
  size_t off = 0;
  auto carve = [&off](size_t words32) {
    const size_t o = off;
    off += (words32 + 3) & ~(size_t)3;
    return o;
  };
  int64_t pow2 = 1;
  while (pow2 < streams) pow2 <<= 1;
  // (fixed-size regions first: the polynomials stay where they were uploaded from call to call)
  const size_t o_poly = carve((size_t)kMtJumpPolys * kN), o_key = carve(kN), o_scal = carve(16),
               o_words = carve((size_t)(pre + n_blocks * kN) + 8), o_state = carve((size_t)pow2 * kN),
               o_seq = carve((size_t)(pow2 / 2 > 0 ? pow2 / 2 : 1) * kSeqWords), o_cnt = carve((size_t)n_wg),
               o_base = carve(2 * (size_t)(n_wg + 1)), o_hard = carve(2 * 4 * (size_t)hard_cap),
               o_fixed = carve(2 * 3 * (size_t)hard_cap);
  VB_TRY(ensure(ctx, ctx->legacy_work, off * sizeof(uint32_t)));
  uint32_t* base = (uint32_t*)ctx->legacy_work.ptr;
  uint32_t *words = base + o_words, *state = base + o_state, *seq = base + o_seq, *poly = base + o_poly, *key_dev = base + o_key;
  if (ctx->legacy_poly_at != (const void*)poly || ctx->legacy_poly_bytes != ctx->legacy_work.bytes) {      // (a new allocation is zeroed)
    VB_HIP(ctx, hipMemcpyAsync(poly, kMtJump, sizeof kMtJump, hipMemcpyHostToDevice, st));
    ctx->legacy_poly_at = poly;
    ctx->legacy_poly_bytes = ctx->legacy_work.bytes;
  }
  VB_HIP(ctx, hipMemcpyAsync(key_dev, key, kN * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  VB_HIP(ctx, hipMemsetAsync(state, 0, (size_t)pow2 * kN * sizeof(uint32_t), st));
  VB_HIP(ctx, hipMemsetAsync(base + o_scal, 0, 16 * sizeof(uint32_t), st));
  hipLaunchKernelGGL(mtd_first_kernel, dim3(1), dim3(256), 0, st, (const uint32_t*)key_dev, *pos, words, state);
  int k = 0;
  for (int64_t count = 1; count < streams; count <<= 1, ++k) {
    const int n_new = (int)std::min<int64_t>(count, streams - count);
    hipLaunchKernelGGL(mtd_seq_kernel, dim3((unsigned)n_new), dim3(256), 0, st, (const uint32_t*)state, seq);
    hipLaunchKernelGGL(mtd_corr_kernel, dim3((unsigned)n_new, kCorrSlices), dim3(640), 0, st,
                       (const uint32_t*)(poly + (size_t)k * kN), (const uint32_t*)seq, state, (int)count, n_new);
  }
  hipLaunchKernelGGL(mtd_stream_kernel, dim3((unsigned)streams), dim3(256), 0, st, (const uint32_t*)state, words, pre, n_blocks);
  int* cnt = (int*)(base + o_cnt);
  int64_t* pbase = (int64_t*)(base + o_base);
  hipLaunchKernelGGL(mtd_count_kernel, dim3((unsigned)n_wg), dim3(256), 0, st, (const uint32_t*)words, attempts, cnt);
  hipLaunchKernelGGL(mtd_scan_kernel, dim3(1), dim3(1024), 0, st, (const int*)cnt, n_wg, pbase);
  EmitArgs a;
  a.words = words, a.attempts = attempts, a.base = pbase;
  a.pairs = pairs, a.n_vals = n_vals, a.first = first;
  a.d = d, a.row_begin = row_begin, a.rows = rows, a.ld = ns.ld;
  a.slot = (double*)ns.buf.ptr;
  a.hard = (double*)(base + o_hard), a.hard_cap = hard_cap;
  int64_t* scal = (int64_t*)(base + o_scal);      // [0] a*, [1] last pair on the list?, [2] hard count, [3] f x1 of the last pair
  a.hard_n = (unsigned long long*)(scal + 2);
  a.a_star = scal;
  a.last_x1f = (double*)(scal + 3);
  if (first) hipLaunchKernelGGL(mtd_first_value_kernel, dim3(1), dim3(1), 0, st, a, *gauss);
  hipLaunchKernelGGL(mtd_emit_kernel, dim3((unsigned)n_wg), dim3(256), 0, st, a);
  VB_HIP(ctx, hipGetLastError());
  int64_t res[4];
  int64_t accepted = 0;
  VB_HIP(ctx, hipMemcpyAsync(res, scal, sizeof res, hipMemcpyDeviceToHost, st));
  VB_HIP(ctx, hipMemcpyAsync(&accepted, pbase + n_wg, sizeof accepted, hipMemcpyDeviceToHost, st));
  VB_HIP(ctx, hipStreamSynchronize(st));
  const int64_t n_hard = res[2];
  if (accepted < pairs || n_hard > hard_cap) return VB_ERR_UNSUPPORTED;      // (budget or list fell short: host path)
  double last_x1f;
  memcpy(&last_x1f, &res[3], sizeof last_x1f);
  if (n_hard > 0) {
    std::vector<double> list((size_t)4 * n_hard), fixed((size_t)3 * n_hard);
    VB_HIP(ctx, hipMemcpy(list.data(), a.hard, list.size() * sizeof(double), hipMemcpyDeviceToHost));
    vb_legacy_finish_pairs(list.data(), n_hard, fixed.data());
    for (int64_t i = 0; i < n_hard; ++i)
      if ((int64_t)fixed[3 * i] == pairs - 1) last_x1f = fixed[3 * i + 2];
    double* fixed_dev = (double*)(base + o_fixed);
    VB_HIP(ctx, hipMemcpyAsync(fixed_dev, fixed.data(), fixed.size() * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(mtd_patch_kernel, dim3((unsigned)((n_hard + 255) / 256)), dim3(256), 0, st, a,
                       (const double*)fixed_dev, n_hard);
    VB_HIP(ctx, hipStreamSynchronize(st));      // `fixed` is stack-scoped
  }
  // numpy's state: just behind the last consumed attempt
  const int64_t w_star = 4 * (res[0] + 1);
  if (w_star <= pre) {
    *pos = (int)(*pos + w_star);                  // still inside the block the call started in
  } else {
    const int64_t offw = w_star - pre;            // words consumed from block 1 onward (blocks counted from 0 here)
    int64_t blk = offw / kN;
    int p = (int)(offw % kN);
    if (p == 0) blk -= 1, p = kN;                 // exactly at a block end: numpy refreshes lazily
    if (blk < 0) {                                // (pre == 0 and nothing of block 1 consumed: cannot happen, w_star >= 4)
      return fail(ctx, VB_ERR_STATE, "legacy generator: inconsistent end position");
    }
    VB_HIP(ctx, hipMemcpy(key, words + pre + blk * kN, kN * sizeof(uint32_t), hipMemcpyDeviceToHost));
    *pos = p;
  }
  *has_gauss = (n_vals & 1) ? 1 : 0;
  *gauss = (n_vals & 1) ? last_x1f : 0.0;
  return VB_OK;
}

}  // namespace vb
