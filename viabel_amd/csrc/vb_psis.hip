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
#include <cstdlib>

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
    if constexpr (REGS) {
      // the top bytes of log weights are sign and exponent: a wave's 64 keys fall into one or two bins, and 64 LDS
      // atomics on one address are served one after the other.  Two rounds of "the first lane's bin: everybody in it
      // is counted by one add"; what is left (the low bytes' spread-out bins) goes lane by lane.
#pragma unroll
      for (int u = 0; u < kPsisRegs; ++u) {
        const int64_t i = t + (int64_t)u * kPsisThreads;
        const unsigned long long k = ps_key(r[u]);
        const int bin = (int)((k >> (8 * pass)) & 255ull);
        bool act = i < n && (k & mask) == prefix;
#pragma unroll
        for (int round = 0; round < 2; ++round) {
          const unsigned long long todo = __ballot(act);
          if (!todo) break;
          const int leader = __builtin_ctzll(todo);
          const int b = __shfl(bin, leader, 64);
          const bool same = act && bin == b;
          const unsigned long long m = __ballot(same);
          if (lane == leader) atomicAdd(&hist[b], __builtin_popcountll(m));
          act = act && !same;
        }
        if (act) atomicAdd(&hist[bin], 1);
      }
    } else {
      ps_each<REGS>(r, x, n, [&](int64_t, double v) {
        const unsigned long long k = ps_key(v);
        if ((k & mask) == prefix) atomicAdd(&hist[(int)((k >> (8 * pass)) & 255ull)], 1);
      });
    }
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

// ---- the same smoothing on G workgroups (round 4) --------------------------------------------------------------------
// One workgroup spends its 80 us on dependent chains (sixteen exponentials per thread, 384^2 comparisons, 49 x 384
// logarithms).  Here workgroup g of G = ceil(N / 1024) (at most 64) owns the weights [g S, (g + 1) S), E <= 4 per thread,
// and the phases are separated by grid barriers (a returning atomic + a bounded poll: the G workgroups of a launch are
// co-resident, one per CU):
//   max                          | per-workgroup maxima, combined by everybody
//   radix select, 12 bits a pass | LDS histogram (wave-aggregated adds) -> integer atomics on the global one; everybody
//                                | scans it; two or three passes until the bin of the wanted rank has <= 1024 members
//   gather                       | members of that bin and everything above it, per-workgroup lists -> one range
//                                | reservation per workgroup; the cut-off is found among the bin's members by counting
//   rank                         | workgroup g ranks its share of the tail by (value, index) and stores it sorted
//   quadrature                   | point j on workgroup j mod G, sixteen lanes per point: the single-workgroup kernel's order
//   fit, quantiles, log-sum-exp  | everybody fits (same numbers), patches its own slice; the new maximum is the largest
//                                | smoothed value, so the sum of exponentials needs one more barrier only
// Every floating-point sum is formed in a fixed order: results do not depend on timing; k-hat, sigma and the smoothed
// tail are bit-identical to the single-workgroup kernel's, the normalising constant agrees to rounding (its terms are
// added slice by slice instead of strided).
constexpr int kPgBits = 12, kPgBins = 1 << kPgBits, kPgPasses = 6;
constexpr int kPgMaxWg = 64, kPgRegs = 4;
constexpr int kPgBinCap = kPsisThreads;
constexpr int kPgBarriers = 12;            // per launch and workgroup: max 1 + select 6 + gather 1 + rank 1 + quadrature 1 + sum 1, padded
// scratch (bytes from the base): [0] barrier counter (runs on from launch to launch: every launch adds exactly
// G * kPgBarriers, its first target is a kernel argument), [16] poison
constexpr size_t kPgOffVal = 64;                                              // double [3][64]
constexpr size_t kPgOffCnt = kPgOffVal + 3 * kPgMaxWg * sizeof(double);       // int [16]: 0 bin members, 1 tail candidates
constexpr size_t kPgOffHist = kPgOffCnt + 16 * sizeof(int);                   // int [6][4096]
constexpr size_t kPgOffBk = kPgOffHist + (size_t)kPgPasses * kPgBins * sizeof(int);      // u64 [1024]
constexpr size_t kPgOffBi = kPgOffBk + kPgBinCap * sizeof(unsigned long long);           // int [1024]
constexpr size_t kPgOffTv = kPgOffBi + kPgBinCap * sizeof(int);                          // double [4096] candidates
constexpr size_t kPgOffTi = kPgOffTv + kPsisTailCap * sizeof(double);                    // int [4096]
constexpr size_t kPgOffSv = kPgOffTi + kPsisTailCap * sizeof(int);                       // double [4096] sorted
constexpr size_t kPgOffSi = kPgOffSv + kPsisTailCap * sizeof(double);                    // int [4096]
constexpr size_t kPgOffQ = kPgOffSi + kPsisTailCap * sizeof(int);                        // double [2][128]: L, ks
constexpr size_t kPgBytes = kPgOffQ + 2 * kPsisQuadCap * sizeof(double);

// Every word the workgroups exchange is stored and loaded as an agent-scope atomic (write-through `sc1` stores, `sc1`
// loads): the eight XCDs' L2s are not coherent with each other, and a line this XCD has read or partly written before
// -- a neighbour's slot of a shared array, a histogram it zeroed -- would be served stale to a plain load whatever
// fences surround it.  With write-through payloads the barrier needs no fence: every wave drains its stores, one lane
// adds to the counter and polls it.
__device__ __forceinline__ void pg_st(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pg_st(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pg_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double pg_ld(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ unsigned long long pg_ld(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int pg_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct PgBarrier {
  unsigned long long* bar;
  unsigned long long target, tag;       // tag: this launch's mark in the poison word (nothing is reset between launches)
  int g_count, used;
  __device__ __forceinline__ void wait() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave: its write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
      target += (unsigned long long)g_count;
      ++used;
#ifdef VB_PSIS_FENCE
      __threadfence();
#endif
      atomicAdd(bar, 1ull);
      int spins = 0;
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 24)) {        // (a workgroup that never arrives: poison the results instead of hanging)
          atomicExch((unsigned long long*)(bar + 2), tag);
          break;
        }
      }
#ifdef VB_PSIS_FENCE
      __threadfence();
#endif
    }
    __syncthreads();
  }
};

// Round 6: weights in, weights out.  The DIS objectives smooth WEIGHTS (normalised, non-negative), not log weights: a prep
// launch took their logarithms and slice totals, an apply launch turned the smoothed log weights back (total x exp) --
// 4 + 5 us of kernels and two launch gaps around a 43-us kernel.  With io.w_in set the kernel reads the weights itself
// (x = log w, the workgroups' totals exchanged at the first barrier, added in workgroup order) and writes
// io.w_out[i] = total * exp(x_i) and k-hat at the end: the same values (for N <= 16 384 the same bits: one element per thread,
// the prep kernel's slices and order).
struct PsisWeightsIo {
  const double* w_in = nullptr;
  double* w_out = nullptr;
  double* khat_out = nullptr;
};

template <int E>
__global__ void __launch_bounds__(kPsisThreads) psis_grid_kernel(double* __restrict__ x, int64_t n, int m_tail,
                                                                 double* __restrict__ out, char* __restrict__ work,
                                                                 unsigned long long bar_base, PsisWeightsIo io) {
  __shared__ double sh[17];
  __shared__ int hist[kPgBins];
  __shared__ double tv[kPsisTailCap];
  __shared__ int ti[kPsisTailCap];
  __shared__ unsigned long long bk[kPgBinCap];
  __shared__ int bi[kPgBinCap];
  __shared__ double q_bs[kPsisQuadCap], q_L[kPsisQuadCap], q_w[kPsisQuadCap];
  __shared__ int rank_sh[kPsisThreads];
  __shared__ double lbuf[kPsisTailCap];
  __shared__ double bc[4];
  __shared__ unsigned long long sel_prefix;
  __shared__ long long sel_rank;
  __shared__ int sel_bin_count, cnt_a, cnt_b, base_a, base_b;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int G = gridDim.x, g = blockIdx.x;
  unsigned long long* bar = reinterpret_cast<unsigned long long*>(work);
  double* wgval = reinterpret_cast<double*>(work + kPgOffVal);
  int* gcnt = reinterpret_cast<int*>(work + kPgOffCnt);
  int* ghist = reinterpret_cast<int*>(work + kPgOffHist);
  unsigned long long* gbk = reinterpret_cast<unsigned long long*>(work + kPgOffBk);
  int* gbi = reinterpret_cast<int*>(work + kPgOffBi);
  double* gtv = reinterpret_cast<double*>(work + kPgOffTv);
  int* gti = reinterpret_cast<int*>(work + kPgOffTi);
  double* gsv = reinterpret_cast<double*>(work + kPgOffSv);
  int* gsi = reinterpret_cast<int*>(work + kPgOffSi);
  double* gq = reinterpret_cast<double*>(work + kPgOffQ);
  PgBarrier barrier{bar, bar_base, bar_base + 1, G, 0};
#ifdef VB_PSIS_CLOCK
  int dbg_k = 4;
#define PG_MARK() do { if (g == 0 && t == 0) out[dbg_k] = (double)wall_clock64(); ++dbg_k; } while (0)
#else
#define PG_MARK() do { } while (0)
#endif
  PG_MARK();
  const int64_t slice = (int64_t)E * kPsisThreads;
  const int64_t i0 = (int64_t)g * slice;

  // 0. this launch's counters and histograms (used behind the first barrier only)
  for (int e = g * kPsisThreads + t; e < kPgPasses * kPgBins; e += G * kPsisThreads) pg_st(&ghist[e], 0);
  if (g == 0 && t < 16) pg_st(&gcnt[t], 0);

  // 1. x -= max(x)   (_psis.py:166)
  double r[E];
  double mx = -INFINITY;
  double sw = 0.0;
#pragma unroll
  for (int u = 0; u < E; ++u) {
    const int64_t i = i0 + t + (int64_t)u * kPsisThreads;
    if (io.w_in) {
      const double wv = i < n ? io.w_in[i] : 0.0;
      sw += wv;
      r[u] = i < n ? log(wv) : 0.0;      // log 0 = -inf: a weight that stays zero
    } else {
      r[u] = i < n ? x[i] : 0.0;
    }
    if (i < n) mx = fmax(mx, r[u]);
  }
  mx = ps_block_max(mx, sh);
  if (t == 0) pg_st(&wgval[g], mx);
  if (io.w_in) {      // this workgroup's share of sum w: the wave by shuffles, the sixteen waves in order (mvt_psis_prep_kernel's)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sw += __shfl_down(sw, off, 64);
    __syncthreads();
    if (lane == 0) sh[wave] = sw;
    __syncthreads();
    if (t == 0) {
      double tw = 0.0;
      for (int q = 0; q < 16; ++q) tw += sh[q];
      pg_st(&wgval[2 * kPgMaxWg + g], tw);
    }
    __syncthreads();
  }
  barrier.wait();
  mx = pg_ld(&wgval[0]);
  for (int q = 1; q < G; ++q) mx = fmax(mx, pg_ld(&wgval[q]));
#pragma unroll
  for (int u = 0; u < E; ++u) {
    const int64_t i = i0 + t + (int64_t)u * kPsisThreads;
    r[u] -= mx;
    if (i < n) x[i] = r[u];
  }

  PG_MARK();
  // 2. the key of x_sorted[n - m_tail - 1]   (:170-173)
  if (t == 0) {
    sel_prefix = 0ull;
    sel_rank = n - (long long)m_tail - 1;
    sel_bin_count = 0x7fffffff;
  }
  __syncthreads();
  int shift_done = 64;            // bits [shift_done, 64) of the key are fixed
  for (int pass = 0; pass < kPgPasses; ++pass) {
    const int shift = pass < 5 ? 52 - kPgBits * pass : 0;
    const int width = shift_done - shift;
    const unsigned long long fixed_mask = shift_done == 64 ? 0ull : (~0ull << shift_done);
    for (int e = t; e < kPgBins; e += kPsisThreads) hist[e] = 0;
    __syncthreads();
    const unsigned long long prefix = sel_prefix;
#pragma unroll
    for (int u = 0; u < E; ++u) {
      const int64_t i = i0 + t + (int64_t)u * kPsisThreads;
      const unsigned long long k = ps_key(r[u]);
      const int bin = (int)((k >> shift) & ((1ull << width) - 1ull));
      bool act = i < n && (k & fixed_mask) == prefix;
#pragma unroll
      for (int round = 0; round < 2; ++round) {      // (see psis_kernel: a wave's keys share their top bits)
        const unsigned long long todo = __ballot(act);
        if (!todo) break;
        const int leader = __builtin_ctzll(todo);
        const int b = __shfl(bin, leader, 64);
        const bool same = act && bin == b;
        const unsigned long long m = __ballot(same);
        if (lane == leader) atomicAdd(&hist[b], __builtin_popcountll(m));
        act = act && !same;
      }
      if (act) atomicAdd(&hist[bin], 1);
    }
    __syncthreads();
    int* gh = ghist + pass * kPgBins;
    for (int e = t; e < kPgBins; e += kPsisThreads)
      if (hist[e]) atomicAdd(&gh[e], hist[e]);
    barrier.wait();
    // everybody scans the global histogram: four consecutive bins per thread
    {
      const long long want = sel_rank;
      const int c0 = pg_ld(&gh[4 * t]), c1 = pg_ld(&gh[4 * t + 1]), c2 = pg_ld(&gh[4 * t + 2]), c3 = pg_ld(&gh[4 * t + 3]);
      const long long mine4 = (long long)c0 + c1 + c2 + c3;
      long long incl = mine4;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const long long up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
      }
      long long* wtot = reinterpret_cast<long long*>(tv);       // 16 wave totals (tv is free here)
      __syncthreads();
      if (lane == 63) wtot[wave] = incl;
      __syncthreads();
      long long before = 0;
      for (int w = 0; w < wave; ++w) before += wtot[w];
      incl += before;
      const long long excl = incl - mine4;
      const bool last_thread = t == kPsisThreads - 1;
      if ((excl <= want && want < incl) || (last_thread && want >= incl)) {
        long long rr = want - excl;
        int bin = 4 * t, cb = c0;
        if (rr >= c0 && bin < kPgBins - 1) { rr -= c0; ++bin; cb = c1;
          if (rr >= c1 && bin < kPgBins - 1) { rr -= c1; ++bin; cb = c2;
            if (rr >= c2 && bin < kPgBins - 1) { rr -= c2; ++bin; cb = c3; } } }
        sel_rank = rr;
        sel_prefix = prefix | ((unsigned long long)bin << shift);
        sel_bin_count = cb;
      }
      __syncthreads();
    }
    shift_done = shift;
    if (shift > 0 && sel_bin_count <= kPgBinCap) break;
  }
  PG_MARK();
  // 3. members of the selected bin (when bits are left to decide) and everything above it
  {
    const unsigned long long prefix = sel_prefix, fixed_mask = shift_done == 0 ? ~0ull : (~0ull << shift_done);
    const bool need_bin = shift_done > 0;
    if (t == 0) cnt_a = 0, cnt_b = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < E; ++u) {
      const int64_t i = i0 + t + (int64_t)u * kPsisThreads;
      if (i < n) {
        const unsigned long long k = ps_key(r[u]);
        const unsigned long long top = k & fixed_mask;
        if (top > prefix) {
          const int p = atomicAdd(&cnt_a, 1);
          tv[p] = r[u];
          ti[p] = (int)i;
        } else if (need_bin && top == prefix) {
          const int p = atomicAdd(&cnt_b, 1);
          if (p < kPgBinCap) {
            bk[p] = k;
            bi[p] = (int)i;
          }
        }
      }
    }
    __syncthreads();
    if (t == 0) {
      base_a = cnt_a ? atomicAdd(&gcnt[1], cnt_a) : 0;
      base_b = cnt_b ? atomicAdd(&gcnt[0], cnt_b) : 0;
    }
    __syncthreads();
    for (int e = t; e < cnt_a; e += kPsisThreads)
      if (base_a + e < kPsisTailCap) pg_st(&gtv[base_a + e], tv[e]), pg_st(&gti[base_a + e], ti[e]);
    for (int e = t; e < cnt_b; e += kPsisThreads)
      if (base_b + e < kPgBinCap) pg_st(&gbk[base_b + e], bk[e]), pg_st(&gbi[base_b + e], bi[e]);
    barrier.wait();
    int nb = pg_ld(&gcnt[0]);
    nb = nb < kPgBinCap ? nb : kPgBinCap;
    if (need_bin) {
      if (t < nb) bk[t] = pg_ld(&gbk[t]), bi[t] = pg_ld(&gbi[t]);
      __syncthreads();
      const long long want = sel_rank;
      if (t < nb) {
        const unsigned long long mk = bk[t];
        int lt = 0, le = 0;
        for (int j = 0; j < nb; ++j) {
          const unsigned long long o = bk[j];
          lt += o < mk ? 1 : 0;
          le += o <= mk ? 1 : 0;
        }
        if (lt <= want && want < le) sel_prefix = mk;
      }
      __syncthreads();
    }
  }
  PG_MARK();
  const double cutoffmin = log(DBL_MIN);                       // :159
  const double xcutoff = fmax(ps_unkey(sel_prefix), cutoffmin);
  const double expxc = exp(xcutoff);
  // right tail: x > xcutoff (:175-177): the candidates above the bin, and the bin's members above the cut-off
  int na = pg_ld(&gcnt[1]);
  na = na < kPsisTailCap ? na : kPsisTailCap;
  int nb = shift_done > 0 ? pg_ld(&gcnt[0]) : 0;
  nb = nb < kPgBinCap ? nb : kPgBinCap;
  {
    if (t == 0) cnt_a = 0;
    __syncthreads();
    for (int e = t; e < na; e += kPsisThreads) {
      const double v = pg_ld(&gtv[e]);
      if (v > xcutoff) {
        const int p = atomicAdd(&cnt_a, 1);
        if (p < kPsisTailCap) {
          tv[p] = v;
          ti[p] = pg_ld(&gti[e]);
        }
      }
    }
    __syncthreads();
    for (int e = t; e < nb; e += kPsisThreads) {
      const double v = ps_unkey(bk[e]);
      if (v > xcutoff) {
        const int p = atomicAdd(&cnt_a, 1);
        if (p < kPsisTailCap) {
          tv[p] = v;
          ti[p] = bi[e];
        }
      }
    }
    __syncthreads();
  }
  const int n2 = cnt_a < kPsisTailCap ? cnt_a : kPsisTailCap;
  const int tail_count = cnt_a;
  PG_MARK();
  double k = INFINITY, sigma = NAN, new_max = 0.0;
  if (n2 > 4) {                                                 // :178-180
    // 4. this workgroup ranks its share of the tail by (value, index) and stores it at its ranks.  The share is a range
    // of the GLOBAL candidate lists (the same for everybody); the LDS copy, whose order differs from workgroup to
    // workgroup, only supplies the comparands
    {
      const int total = na + nb, per = (total + G - 1) / G, c0 = g * per;
      const int cnt = c0 < total ? (c0 + per < total ? per : total - c0) : 0;
      int R = cnt > 0 ? kPsisThreads / cnt : 1;       // threads per element
      R = R < 1 ? 1 : (R > 64 ? 64 : R);
      const int slots = kPsisThreads / R;
      for (int eb = 0; eb < cnt; eb += slots) {
        const int slot = t / R, el = eb + slot, rep = t % R;
        bool live = el < cnt && slot < slots;
        double mv = INFINITY;
        int mi = 0x7fffffff;
        if (live) {
          const int c = c0 + el;
          mv = c < na ? pg_ld(&gtv[c]) : ps_unkey(bk[c - na]);
          mi = c < na ? pg_ld(&gti[c]) : bi[c - na];
          live = mv > xcutoff;
        }
        if (t < slots) rank_sh[t] = 0;
        __syncthreads();
        int c = 0;
        if (live)
          for (int j = rep; j < n2; j += R) c += ps_after(mv, mi, tv[j], ti[j]) ? 1 : 0;
        if (live) atomicAdd(&rank_sh[slot], c);
        __syncthreads();
        if (live && rep == 0) {
          const int rk = rank_sh[slot];
          pg_st(&gsv[rk], mv);
          pg_st(&gsi[rk], mi);
        }
        __syncthreads();
      }
    }
    barrier.wait();
    PG_MARK();
    // x2 = exp(x2) - exp(xcutoff)   (:185-186)
    for (int i = t; i < n2; i += kPsisThreads) {
      tv[i] = exp(pg_ld(&gsv[i])) - expxc;
      ti[i] = pg_ld(&gsi[i]);
    }
    __syncthreads();
    // 5. gpdfitnew (:266-325): PRIOR = 3, m = 30 + int(sqrt(n2)); point j on workgroup j mod G
    const int m = 30 + (int)sqrt((double)n2);
    const double xq = tv[(int)(n2 / 4.0 + 0.5) - 1], xl = tv[n2 - 1];
    if (t < m) q_bs[t] = (1.0 - sqrt((double)m / ((double)(t + 1) - 0.5))) / (3.0 * xq) + 1.0 / xl;
    __syncthreads();
    const int grp = t >> 4, gl = t & 15;
    // ks_j = mean log1p(-bs_j x) for the points j = g, g + G, ...: the logarithms by all threads into LDS (as many points
    // at a time as fit), then sixteen lanes per point add them in psis_kernel's order (lane l takes i = l, l + 16, ..., a
    // fixed butterfly joins the lanes)
    const int npts = g < m ? (m - g + G - 1) / G : 0;
    const int pc = kPsisTailCap / n2 > 0 ? kPsisTailCap / n2 : 1;
    for (int p0 = 0; p0 < npts; p0 += pc) {
      const int np = npts - p0 < pc ? npts - p0 : pc;
      for (int e = t; e < np * n2; e += kPsisThreads) {
        const int pp = e / n2, i = e - pp * n2;
        lbuf[e] = log(fma(-q_bs[g + (p0 + pp) * G], tv[i], 1.0));
      }
      __syncthreads();
      for (int pp = grp; pp < np; pp += kPsisThreads / 16) {
        const double* lb = lbuf + pp * n2;
        double s = 0.0;
        int i = gl;
        for (; i + 112 < n2; i += 128) {           // eight reads in flight, added in index order
          double v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = lb[i + 16 * q];
#pragma unroll
          for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; i < n2; i += 16) s += lb[i];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (gl == 0) {
          const int j = g + (p0 + pp) * G;
          const double ks = s / n2;
          pg_st(&gq[j], n2 * (log(-(q_bs[j] / ks)) - ks - 1.0));
        }
      }
      __syncthreads();
    }
    barrier.wait();
    PG_MARK();
    if (t < m) q_L[t] = pg_ld(&gq[t]);
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

    // 6. smoothed tail (:188-199); every workgroup forms all of it and keeps what falls into its slice
    if (k >= 1.0 / 3.0 && !isinf(k)) {
      double vmax = -INFINITY;
      for (int i = t; i < n2; i += kPsisThreads) {
        const double p = ((double)i + 0.5) / n2;
        double qq = NAN;
        if (sigma > 0.0) {
          const double l = log1p(-p);
          qq = (fabs(k) < DBL_EPSILON ? -l : expm1(-k * l) / k) * sigma;
        }
        double v = log(qq + expxc);
        if (v > 0.0) v = 0.0;
        vmax = fmax(vmax, v);          // (fmax drops a NaN: the reference's max of the vector would not -- sigma <= 0 only)
        const int64_t idx = ti[i];
        if (idx >= i0 && idx < i0 + slice) x[idx] = v;
      }
      new_max = ps_block_max(vmax, sh);
      // the untouched weights are <= xcutoff <= every smoothed value (the quantiles are >= 0); without a tail to the
      // right of the cut-off the old maximum (0) stays
      if (!(new_max > -INFINITY)) new_max = 0.0;
      __syncthreads();
#pragma unroll
      for (int u = 0; u < E; ++u) {
        const int64_t i = i0 + t + (int64_t)u * kPsisThreads;
        if (i < n) r[u] = x[i];
      }
    }
  }
  PG_MARK();
  // 7. renormalise: x -= sumlogs(x)   (:201, :380-396)
  double se = 0.0;
#pragma unroll
  for (int u = 0; u < E; ++u) {
    const int64_t i = i0 + t + (int64_t)u * kPsisThreads;
    if (i < n) se += exp(r[u] - new_max);
  }
  se = ps_block_sum(se, sh);
  if (t == 0) pg_st(&wgval[kPgMaxWg + g], se);
  barrier.wait();
  double tot = 0.0;
  for (int q = 0; q < G; ++q) tot += pg_ld(&wgval[kPgMaxWg + q]);
  const double lse = log(tot) + new_max;
  const bool poisoned = pg_ld(&bar[2]) == bar_base + 1;
  double wtot = 0.0;
  if (io.w_out)
    for (int q = 0; q < G; ++q) wtot += pg_ld(&wgval[2 * kPgMaxWg + q]);
#pragma unroll
  for (int u = 0; u < E; ++u) {
    const int64_t i = i0 + t + (int64_t)u * kPsisThreads;
    if (i < n) {
      const double v = poisoned ? NAN : r[u] - lse;
      x[i] = v;
      if (io.w_out) io.w_out[i] = wtot * exp(v);      // (mvt_psis_apply_kernel's expression)
    }
  }
  PG_MARK();
  if (g == 0 && t == 0) {
    if (io.khat_out) io.khat_out[0] = poisoned ? NAN : k;
    out[0] = poisoned ? NAN : k;
    out[1] = poisoned ? NAN : (double)tail_count;      // (k-hat AND the tail count NaN: the poison's signature)
    out[2] = xcutoff;
    out[3] = sigma;
#ifdef VB_PSIS_DEBUG
    out[4] = n2, out[5] = pg_ld(&gcnt[1]), out[6] = pg_ld(&gcnt[0]), out[7] = shift_done, out[8] = (double)sel_rank;
    out[9] = mx, out[10] = new_max, out[11] = tot;
#endif
  }
  // the barriers this launch did not need (the select leaves early): the counter advances by the same amount every launch
  if (t == 0 && barrier.used < kPgBarriers) atomicAdd(bar, (unsigned long long)(kPgBarriers - barrier.used));
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
// weights_in != nullptr (round 6): smooth WEIGHTS -- their logarithms go to ctx->psis_lw as before, weights_out[i] = sum(w) x
// exp(smoothed log weight), khat_out[0] = k-hat -- in the same launch, when the multi-workgroup kernel applies; *fused_out says
// whether it did (false: nothing was launched for them, the caller takes the three-launch route)
int psis_enqueue(vb_ctx* ctx, int64_t n, double reff, const double* weights_in, double* weights_out, double* khat_out,
                 bool* fused_out) {
  if (fused_out) *fused_out = false;
  const char* fuse_s = getenv("VB_PSIS_FUSED_IO");      // 0: prep / apply launches around the kernel (cross-check)
  if (weights_in && fuse_s && atoi(fuse_s) == 0) return VB_OK;
  if (n <= 1) return fail(ctx, VB_ERR_INVALID, "More than one log-weight needed.");
  if (!(reff > 0.0)) return fail(ctx, VB_ERR_INVALID, "Reff must be positive");
  const int m_tail = psis_tail_size(n, reff);
  if (m_tail > kPsisTailCap)
    return fail(ctx, VB_ERR_UNSUPPORTED, "PSIS tail of %d values exceeds the on-chip sort capacity %d", m_tail,
                kPsisTailCap);
  double* lw = (double*)ctx->psis_lw.ptr;
  const char* grid_s = getenv("VB_PSIS_GRID");             // 0: the single-workgroup kernel (cross-check)
  const int grid_env = grid_s ? atoi(grid_s) : 1;
  int wgs = (int)((n + kPsisThreads - 1) / kPsisThreads);
  wgs = wgs > kPgMaxWg ? kPgMaxWg : wgs;
  const int per_thread = (int)((n + (int64_t)wgs * kPsisThreads - 1) / ((int64_t)wgs * kPsisThreads));
  // The grid barrier needs every workgroup of the launch resident at once: ask the runtime how many of these 1024-thread,
  // ~115 KB-LDS workgroups one CU takes (once per context and variant) and keep to the single-workgroup kernel when the
  // grid would not fit -- other streams' work or another process can still delay residency, which the bounded polls turn
  // into a poisoned (NaN) result and vb_psis_smooth into VB_ERR_STATE, not into a hang.
  static int per_cu[3] = {-1, -1, -1};
  const int variant = per_thread == 1 ? 0 : (per_thread == 2 ? 1 : 2);
  if (grid_env && wgs > 1 && per_thread <= kPgRegs && per_cu[variant] < 0) {
    int nb = 0;
    hipError_t e = variant == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, psis_grid_kernel<1>, kPsisThreads, 0)
                   : variant == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, psis_grid_kernel<2>, kPsisThreads, 0)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, psis_grid_kernel<4>, kPsisThreads, 0);
    per_cu[variant] = e == hipSuccess ? nb : 0;
  }
  if (grid_env && wgs > 1 && per_thread <= kPgRegs && (int64_t)wgs <= (int64_t)per_cu[variant] * ctx->prop.multiProcessorCount) {
    if (!ctx->psis_work.ptr) {
      VB_TRY(ensure(ctx, ctx->psis_work, kPgBytes));
      VB_HIP(ctx, hipMemsetAsync(ctx->psis_work.ptr, 0, 64, ctx->stream));        // barrier counter, poison
      ctx->psis_bar_base = 0;
    }
    const unsigned long long bar_base = ctx->psis_bar_base;
    char* work = (char*)ctx->psis_work.ptr;
    double* out = lw + round_up(n, 16);
    PsisWeightsIo io;
    if (weights_in) io.w_in = weights_in, io.w_out = weights_out, io.khat_out = khat_out;
    if (per_thread == 1)
      hipLaunchKernelGGL(psis_grid_kernel<1>, dim3(wgs), dim3(kPsisThreads), 0, ctx->stream, lw, n, m_tail, out, work, bar_base, io);
    else if (per_thread == 2)
      hipLaunchKernelGGL(psis_grid_kernel<2>, dim3(wgs), dim3(kPsisThreads), 0, ctx->stream, lw, n, m_tail, out, work, bar_base, io);
    else
      hipLaunchKernelGGL(psis_grid_kernel<4>, dim3(wgs), dim3(kPsisThreads), 0, ctx->stream, lw, n, m_tail, out, work, bar_base, io);
    VB_HIP(ctx, hipGetLastError());
    ctx->psis_bar_base += (unsigned long long)wgs * kPgBarriers;      // (a launch that did not happen adds nothing)
    if (fused_out) *fused_out = io.w_in != nullptr;
    return VB_OK;
  }
  if (weights_in) return VB_OK;      // (the single-workgroup kernel reads log weights: the caller prepares them and calls again)
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
