// Pareto-smoothed importance sampling of N log weights on the device.
// Reference: viabel/_psis.py:113-209 (psislw), :212-332 (gpdfitnew), :335-377 (gpinv), :380-396 (sumlogs);
// caller viabel/convenience.py:166-179 (psis_correction / samples_and_log_weights).
//
// One workgroup of 1024 threads does the whole smoothing for one weight vector (N up to a few million; the
// diagnostics use N = 1e5): max shift, exact selection of the tail cut-off (the (M+1)-th largest value,
// M = ceil(min(0.2 N, 3 sqrt(N / Reff))), _psis.py:158) by an 8-pass byte-wise radix select on
// order-preserving keys with an LDS histogram, gather + rank sort of the <= M tail values in LDS, the
// Zhang-Stephens empirical-Bayes GPD fit (30 + sqrt(M) quadrature points), replacement of the tail by the
// fitted quantiles, truncation at 0 and log-sum-exp normalisation.  Integer atomics only (histogram, gather
// counter); the sorted order is by (value, index), so the result does not depend on the gather order.
#include "vb_common.h"

#include <cfloat>
#include <cmath>

namespace vb {

constexpr int kPsisThreads = 1024;
constexpr int kPsisTailCap = 4096;     // tail values sorted in LDS (M <= 4096  <=>  N <= 1.86e6 at Reff = 1)
constexpr int kPsisQuadCap = 128;      // 30 + sqrt(4096) = 94 quadrature points at most

__device__ __forceinline__ double ps_wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}
__device__ __forceinline__ double ps_wave_max(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x = fmax(x, __shfl_down(x, off, 64));
  return x;
}

// block-wide sum / max over 1024 threads; every thread gets the result (sh: 16 doubles + 1)
__device__ double ps_block_sum(double x, double* sh) {
  x = ps_wave_sum(x);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < kPsisThreads / 64; ++w) t += sh[w];
  return t;
}
__device__ double ps_block_max(double x, double* sh) {
  x = ps_wave_max(x);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
  __syncthreads();
  double t = sh[0];
#pragma unroll
  for (int w = 1; w < kPsisThreads / 64; ++w) t = fmax(t, sh[w]);
  return t;
}

__device__ __forceinline__ unsigned long long ps_key(double v) {   // ascending order-preserving key
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double ps_unkey(unsigned long long k) {
  const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)u);
}

// (value, index) lexicographic "a after b"
__device__ __forceinline__ bool ps_after(double av, int ai, double bv, int bi) {
  return (av > bv) | ((av == bv) & (ai > bi));      // no short circuit: a branch per comparison cost 25 us per call
}

// lw = f - b + sum(log sigma): the log importance weights log p(z_n) - log q(z_n) of the mean-field families
__global__ void __launch_bounds__(256) psis_lw_kernel(const double* __restrict__ f, const double* __restrict__ b,
                                                      const double* __restrict__ scal, int64_t n,
                                                      double* __restrict__ lw) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) lw[i] = f[i] - b[i] + scal[0];
}

// The passes over the N weights are one workgroup's strided loops: a thread's loads are independent, but left as
// `for (i = t; i < n; i += 1024) use(x[i])` every trip waits out its own L2 round trip (16 trips x 8 radix passes were
// most of the kernel's time at N = 16 384).  ps_for_each requests kPsisBatch elements before it uses the first.
constexpr int kPsisBatch = 8;
template <class F>
__device__ __forceinline__ void ps_for_each(const double* __restrict__ x, int64_t n, F&& f) {
  for (int64_t i0 = threadIdx.x; i0 < n; i0 += (int64_t)kPsisBatch * kPsisThreads) {
    double v[kPsisBatch];
#pragma unroll
    for (int u = 0; u < kPsisBatch; ++u) {
      const int64_t i = i0 + (int64_t)u * kPsisThreads;
      v[u] = i < n ? x[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kPsisBatch; ++u) {
      const int64_t i = i0 + (int64_t)u * kPsisThreads;
      if (i < n) f(i, v[u]);
    }
  }
}

// Up to kPsisRegs * 1024 weights (16 384: BASELINE configs[3]) stay in registers from the shift to the tail gather and
// again through the renormalisation: the eight radix passes, the gather and the three renormalisation sweeps then read
// no memory at all.  Longer vectors take the batched loops.
constexpr int kPsisRegs = 16;
template <bool REGS, class F>
__device__ __forceinline__ void ps_each(const double (&r)[kPsisRegs], const double* __restrict__ x, int64_t n, F&& f) {
  if constexpr (REGS) {
#pragma unroll
    for (int u = 0; u < kPsisRegs; ++u) {
      const int64_t i = threadIdx.x + (int64_t)u * kPsisThreads;
      if (i < n) f(i, r[u]);
    }
  } else {
    ps_for_each(x, n, f);
  }
}

// x: N log weights, smoothed in place.  out = [khat, n_tail, xcutoff (shifted), sigma]
template <bool REGS>
__global__ void __launch_bounds__(kPsisThreads) psis_kernel(double* __restrict__ x, int64_t n, int m_tail,
                                                            double* __restrict__ out) {
  __shared__ double sh[17];
  __shared__ int hist[256];
  __shared__ unsigned long long sel_prefix;
  __shared__ long long sel_rank;
  __shared__ int sel_bin_count;
  __shared__ int tail_count;
  __shared__ double tv[kPsisTailCap];
  __shared__ int ti[kPsisTailCap];
  __shared__ double q_bs[kPsisQuadCap], q_ks[kPsisQuadCap], q_L[kPsisQuadCap], q_w[kPsisQuadCap];
  __shared__ int rank_sh[kPsisThreads / 2];
  __shared__ double bc[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#ifdef VB_PSIS_CLOCK
  int dbg_k = 4;
#define PSIS_MARK() do { __syncthreads(); if (t == 0) out[dbg_k] = (double)wall_clock64(); ++dbg_k; } while (0)
#else
#define PSIS_MARK() do { } while (0)
#endif
  PSIS_MARK();

  // 1. improve numerical accuracy: x -= max(x)   (_psis.py:166)
  double r[kPsisRegs];
  if constexpr (REGS) {
#pragma unroll
    for (int u = 0; u < kPsisRegs; ++u) {
      const int64_t i = t + (int64_t)u * kPsisThreads;
      r[u] = i < n ? x[i] : 0.0;
    }
  }
  double mx = -INFINITY;
  ps_each<REGS>(r, x, n, [&](int64_t, double v) { mx = fmax(mx, v); });
  mx = ps_block_max(mx, sh);
  if constexpr (REGS) {
#pragma unroll
    for (int u = 0; u < kPsisRegs; ++u) {
      const int64_t i = t + (int64_t)u * kPsisThreads;
      r[u] -= mx;
      if (i < n) x[i] = r[u];
    }
  } else {
    ps_for_each(x, n, [&](int64_t i, double v) { x[i] = v - mx; });
  }
  __syncthreads();

  PSIS_MARK();
  // 2. x_sorted[n - m_tail - 1] by radix select (8 bits per pass, most significant first)   (:170-173)
  if (t == 0) {
    sel_prefix = 0ull;
    sel_rank = n - (long long)m_tail - 1;     // 0-based ascending rank
  }
  // (early exit: once the bin that holds the wanted rank has at most 1024 members -- two or three passes for log weights,
  // whose top bytes are sign and exponent -- the bin's keys are gathered and the rank is found by counting: every thread
  // counts the members below and up to its own key; the same key as eight passes deliver, ~15 us sooner at N = 16 384)
  int pass_done = -1;
  for (int pass = 7; pass >= 0; --pass) {
    if (t < 256) hist[t] = 0;
    __syncthreads();
    const unsigned long long prefix = sel_prefix;
    const unsigned long long mask = pass == 7 ? 0ull : (~0ull << (8 * (pass + 1)));
    ps_each<REGS>(r, x, n, [&](int64_t, double v) {
      const unsigned long long k = ps_key(v);
      if ((k & mask) == prefix) atomicAdd(&hist[(int)((k >> (8 * pass)) & 255ull)], 1);
    });
    __syncthreads();
    if (wave == 0) {
      // the bin that holds rank r: wave 0 scans the 256 counts, four consecutive bins per lane (a serial walk by one
      // thread was 3.5 us of every pass)
      const long long r = sel_rank;
      const int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
      long long incl = (long long)c0 + c1 + c2 + c3;          // inclusive prefix over lanes
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const long long up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
      }
      const long long excl = incl - ((long long)c0 + c1 + c2 + c3);
      // exactly one lane has excl <= r < incl (the total is > r); the last lane also takes r beyond the total's reach
      const bool mine = (excl <= r && r < incl) || (lane == 63 && r >= incl);
      if (mine) {
        long long rr = r - excl;
        int bin = 4 * lane, cb = c0;
        if (rr >= c0 && bin < 255) { rr -= c0; ++bin; cb = c1;
          if (rr >= c1 && bin < 255) { rr -= c1; ++bin; cb = c2;
            if (rr >= c2 && bin < 255) { rr -= c2; ++bin; cb = c3; } } }
        sel_rank = rr;
        sel_prefix = prefix | ((unsigned long long)bin << (8 * pass));
        sel_bin_count = cb;
      }
    }
    __syncthreads();
    pass_done = pass;
    if (pass > 0 && sel_bin_count <= kPsisThreads) break;      // (uniform: a shared value read behind the barrier)
  }
  if (pass_done > 0) {
    // the members of the selected bin (keys that agree with sel_prefix down to byte pass_done), counted against each other
    unsigned long long* bk = reinterpret_cast<unsigned long long*>(tv);
    if (t == 0) tail_count = 0;
    __syncthreads();
    const unsigned long long prefix = sel_prefix, mask = ~0ull << (8 * pass_done);
    ps_each<REGS>(r, x, n, [&](int64_t, double v) {
      const unsigned long long k = ps_key(v);
      if ((k & mask) == prefix) {
        const int p = atomicAdd(&tail_count, 1);
        if (p < kPsisThreads) bk[p] = k;
      }
    });
    __syncthreads();
    const int nb = tail_count < kPsisThreads ? tail_count : kPsisThreads;
    const long long want = sel_rank;
    if (t < nb) {
      const unsigned long long mk = bk[t];
      int lt = 0, le = 0;
      for (int j = 0; j < nb; ++j) {
        const unsigned long long o = bk[j];
        lt += o < mk ? 1 : 0;
        le += o <= mk ? 1 : 0;
      }
      if (lt <= want && want < le) sel_prefix = mk;      // (every thread that holds this key writes the same value)
    }
    __syncthreads();
  }
  PSIS_MARK();
  const double cutoffmin = log(DBL_MIN);                       // :159
  const double xcutoff = fmax(ps_unkey(sel_prefix), cutoffmin);
  const double expxc = exp(xcutoff);

  // 3. right tail: x > xcutoff   (:175-177)
  if (t == 0) tail_count = 0;
  __syncthreads();
  ps_each<REGS>(r, x, n, [&](int64_t i, double v) {
    if (v > xcutoff) {
      const int p = atomicAdd(&tail_count, 1);
      if (p < kPsisTailCap) {
        tv[p] = v;
        ti[p] = (int)i;
      }
    }
  });
  __syncthreads();
  const int n2 = tail_count < kPsisTailCap ? tail_count : kPsisTailCap;
  PSIS_MARK();
  double k = INFINITY, sigma = NAN;
  if (n2 > 4) {                                                 // :178-180
    // 4. order of the tail samples by (value, index): every element counts the elements before it (n2^2 / 1024
    // comparisons per thread on LDS broadcasts: 0.15 M in all at n2 = 384) and moves to that position -- two
    // barriers instead of the 45 of a bitonic network
    constexpr int kPerThread = kPsisTailCap / kPsisThreads;
    if (n2 <= kPsisThreads / 2) {
      // few tail values (n2 = 384 at N = 16 384): the counting is a chain of dependent compare / select instructions,
      // so it goes to ALL waves -- R = 1024 / round_up(n2, 64) threads per element, thread (e, rep) counts the comparands
      // j = rep, rep + R, ... and the partial counts meet in an LDS integer (30 -> 15 us at R = 2)
      const int n2p = (n2 + 63) & ~63, R = kPsisThreads / n2p;
      const int e = t % n2p, rep = t / n2p;
      int* rk = rank_sh;
      if (t < n2) rk[t] = 0;
      __syncthreads();
      const double mv = e < n2 ? tv[e] : INFINITY;
      const int mi = e < n2 ? ti[e] : 0x7fffffff;
      int cnt = 0;
      if (rep < R) {
        for (int j0 = rep; j0 < n2; j0 += 8 * R) {
          double v8[8];
          int i8[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {        // unconditional reads (clamped), the sentinel selected afterwards
            const int jx = j0 + q * R;
            const int jc = jx < n2 ? jx : n2 - 1;
            const double vv = tv[jc];
            const int ii = ti[jc];
            v8[q] = jx < n2 ? vv : INFINITY;
            i8[q] = jx < n2 ? ii : 0x7fffffff;
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) cnt += ps_after(mv, mi, v8[q], i8[q]) ? 1 : 0;
        }
        if (e < n2) atomicAdd(&rk[e], cnt);               // integer: order does not matter
      }
      __syncthreads();
      const int my_rank = t < n2 ? rk[t] : 0;
      const double keep_v = t < n2 ? tv[t] : 0.0;
      const int keep_i = t < n2 ? ti[t] : 0;
      __syncthreads();
      if (t < n2) {
        tv[my_rank] = keep_v;
        ti[my_rank] = keep_i;
      }
      __syncthreads();
    } else {
    double my_v[kPerThread];
    int my_i[kPerThread], my_r[kPerThread];
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
      const int e = t + u * kPsisThreads;
      my_v[u] = e < n2 ? tv[e] : INFINITY;
      my_i[u] = e < n2 ? ti[e] : 0x7fffffff;
      my_r[u] = 0;
    }
    // eight comparands are read from LDS before the first comparison; beyond n2 a sentinel nothing comes after
    for (int j0 = 0; j0 < n2; j0 += 8) {
      double v8[8];
      int i8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int jx = j0 + q;
        const int jc = jx < n2 ? jx : n2 - 1;
        const double vv = tv[jc];
        const int ii = ti[jc];
        v8[q] = jx < n2 ? vv : INFINITY;
        i8[q] = jx < n2 ? ii : 0x7fffffff;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int u = 0; u < kPerThread; ++u)
          if (u * kPsisThreads < n2) my_r[u] += ps_after(my_v[u], my_i[u], v8[q], i8[q]) ? 1 : 0;     // (uniform condition)
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
      if (t + u * kPsisThreads < n2) {
        tv[my_r[u]] = my_v[u];
        ti[my_r[u]] = my_i[u];
      }
    }
    __syncthreads();
    }
    PSIS_MARK();
    // x2 = exp(x2) - exp(xcutoff)   (:185-186)
    for (int i = t; i < n2; i += kPsisThreads) tv[i] = exp(tv[i]) - expxc;
    __syncthreads();

    PSIS_MARK();
    // 5. gpdfitnew (:266-325): PRIOR = 3, m = 30 + int(sqrt(n2))
    const int m = 30 + (int)sqrt((double)n2);
    const double xq = tv[(int)(n2 / 4.0 + 0.5) - 1], xl = tv[n2 - 1];
    if (t < m) q_bs[t] = (1.0 - sqrt((double)m / ((double)(t + 1) - 0.5))) / (3.0 * xq) + 1.0 / xl;
    __syncthreads();
    // sixteen lanes per quadrature point (64 points at a time: m <= 94), each a fixed stride of the tail, combined by a
    // fixed butterfly -- the m n2 log1p evaluations spread over the whole workgroup instead of one wave per point
    const int grp = t >> 4, gl = t & 15;
    for (int j = grp; j < m; j += kPsisThreads / 16) {          // ks_j = mean log1p(-bs_j x)
      const double nb = -q_bs[j];
      double s = 0.0;
      // log(1 + y) for log1p(y): the ABSOLUTE error of a term is what the mean sees, and that is one rounding of 1 + y
      // (<= 1.1e-16) either way; log costs half of what log1p does, and these m n2 evaluations are the phase's time
      // four logarithms in flight per lane (independent chains: the evaluation is latency-bound), added in the same order
      int i = gl;
      for (; i + 48 < n2; i += 64) {
        const double l0 = log(fma(nb, tv[i], 1.0)), l1 = log(fma(nb, tv[i + 16], 1.0)), l2 = log(fma(nb, tv[i + 32], 1.0)),
                     l3 = log(fma(nb, tv[i + 48], 1.0));
        s += l0;
        s += l1;
        s += l2;
        s += l3;
      }
      for (; i < n2; i += 16) s += log(fma(nb, tv[i], 1.0));
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      if (gl == 0) {
        const double ks = s / n2;
        q_ks[j] = ks;
        q_L[j] = n2 * (log(-(q_bs[j] / ks)) - ks - 1.0);
      }
    }
    __syncthreads();
    for (int j = grp; j < m; j += kPsisThreads / 16) {
      const double lj = q_L[j];
      double s = 0.0;
      for (int i = gl; i < m; i += 16) s += exp(q_L[i] - lj);
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      if (gl == 0) {
        const double w = 1.0 / s;
        q_w[j] = w >= 10.0 * DBL_EPSILON ? w : 0.0;              // remove negligible weights
      }
    }
    __syncthreads();
    if (t == 0) {
      double ws = 0.0, bsum = 0.0;
      for (int i = 0; i < m; ++i) ws += q_w[i];
      for (int i = 0; i < m; ++i) bsum += q_bs[i] * (q_w[i] / ws);
      bc[0] = bsum;                                              // posterior mean of b
    }
    __syncthreads();
    const double b = bc[0];
    double s = 0.0;
    for (int i = t; i < n2; i += kPsisThreads) s += log1p(-b * tv[i]);
    s = ps_block_sum(s, sh);
    k = s / n2;
    sigma = -k / b;
    k = k * n2 / (n2 + 10.0) + 10.0 * 0.5 / (n2 + 10.0);        // weakly informative prior, a = 10

    PSIS_MARK();
    // 6. smoothed tail (:188-199): order statistics of the fitted GPD, truncated at the largest raw weight
    if (k >= 1.0 / 3.0 && !isinf(k)) {
      for (int i = t; i < n2; i += kPsisThreads) {
        const double p = ((double)i + 0.5) / n2;
        double qq = NAN;
        if (sigma > 0.0) {
          const double l = log1p(-p);
          qq = (fabs(k) < DBL_EPSILON ? -l : expm1(-k * l) / k) * sigma;
        }
        double v = log(qq + expxc);
        if (v > 0.0) v = 0.0;
        x[ti[i]] = v;
      }
    }
  }
  __syncthreads();

  PSIS_MARK();
  // 7. renormalise: x -= sumlogs(x)   (:201, :380-396)
  if constexpr (REGS) {             // (the tail was rewritten in memory: read the vector once more)
#pragma unroll
    for (int u = 0; u < kPsisRegs; ++u) {
      const int64_t i = t + (int64_t)u * kPsisThreads;
      r[u] = i < n ? x[i] : 0.0;
    }
  }
  double m2 = -INFINITY;
  ps_each<REGS>(r, x, n, [&](int64_t, double v) { m2 = fmax(m2, v); });
  m2 = ps_block_max(m2, sh);
  double se = 0.0;
  ps_each<REGS>(r, x, n, [&](int64_t, double v) { se += exp(v - m2); });
  se = ps_block_sum(se, sh);
  const double lse = log(se) + m2;
  ps_each<REGS>(r, x, n, [&](int64_t i, double v) { x[i] = v - lse; });
  PSIS_MARK();
  if (t == 0) {
    out[0] = k;
    out[1] = (double)tail_count;
    out[2] = xcutoff;
    out[3] = sigma;
  }
}

// log importance weights of the mean-field families for the noise staged in `ns`; left in ctx->psis_lw
int log_weights_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int family, double df,
                        const double* theta_src) {
  if (ctx->model.id != VB_MODEL_GAUSS_DIAG && ctx->model.id != VB_MODEL_FUNNEL && ctx->model.id != VB_MODEL_SOURCE)
    return fail(ctx, VB_ERR_UNSUPPORTED, "row-statistics path supports the gauss_diag, funnel and source models");
  if (ctx->model.dim != d) return fail(ctx, VB_ERR_INVALID, "model dimension != family dimension");
  if (n <= 0 || n > ns.n || d != ns.d) return fail(ctx, VB_ERR_INVALID, "noise slot shape mismatch");
  const int student = family == VB_FAMILY_MF_STUDENT_T;
  const int64_t o_scal = 2 * ns.ld, o_f = o_scal + 16, o_b = o_f + round_up(n, 16);
  VB_TRY(ensure(ctx, ctx->rowvec, (size_t)(o_b + round_up(n, 16)) * sizeof(double)));
  VB_TRY(ensure(ctx, ctx->psis_lw, (size_t)(round_up(n, 16) + 16) * sizeof(double)));
  double* base = (double*)ctx->rowvec.ptr;
  VB_TRY(rowstats_enqueue(ctx, ns, n, d, theta_src, ctx->model, student, df, base, base + o_scal, base + o_f,
                          base + o_b));
  hipLaunchKernelGGL(psis_lw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const double*)(base + o_f), (const double*)(base + o_b), (const double*)(base + o_scal), n,
                     (double*)ctx->psis_lw.ptr);
  VB_HIP(ctx, hipGetLastError());
  ctx->psis_n = n;
  return VB_OK;
}

int psis_tail_size(int64_t n, double reff) {   // _psis.py:158
  const double a = 0.2 * (double)n, b = 3.0 * sqrt((double)n / reff);
  return (int)ceil(a < b ? a : b);
}

// smooth the n log weights in ctx->psis_lw in place; out_dev = [khat, n_tail, xcutoff, sigma]
int psis_enqueue(vb_ctx* ctx, int64_t n, double reff) {
  if (n <= 1) return fail(ctx, VB_ERR_INVALID, "More than one log-weight needed.");
  if (!(reff > 0.0)) return fail(ctx, VB_ERR_INVALID, "Reff must be positive");
  const int m_tail = psis_tail_size(n, reff);
  if (m_tail > kPsisTailCap)
    return fail(ctx, VB_ERR_UNSUPPORTED, "PSIS tail of %d values exceeds the on-chip sort capacity %d", m_tail,
                kPsisTailCap);
  double* lw = (double*)ctx->psis_lw.ptr;
  if (n <= (int64_t)kPsisRegs * kPsisThreads)
    hipLaunchKernelGGL(psis_kernel<true>, dim3(1), dim3(kPsisThreads), 0, ctx->stream, lw, n, m_tail,
                       lw + round_up(n, 16));
  else
    hipLaunchKernelGGL(psis_kernel<false>, dim3(1), dim3(kPsisThreads), 0, ctx->stream, lw, n, m_tail,
                       lw + round_up(n, 16));
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

}  // namespace vb
