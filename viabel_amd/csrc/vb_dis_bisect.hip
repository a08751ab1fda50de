// DISInclusiveKL: the tempering parameter's bisection (objectives.py:338-368), speculative version (round 4).
//
// The reference walks 50 levels of a binary tree of intervals: at each level ESS((lower + upper) / 2) is compared
// with the target and one end moves.  The look-ahead version (vb_rowstats.hip) evaluates all 63 midpoints six
// levels can visit in one launch: nine launches + a final one.  Here a launch evaluates the midpoints along the TWO
// root-to-leaf paths that lead to `p - delta` and `p + delta`, p being the root of an inverse-interpolation model of
// ESS(eps) - target through the (up to four) evaluated ends next to the current interval, delta the disagreement of
// the two highest model orders.  The next launch replays the decisions on those numbers -- the reference's
// comparisons at the reference's midpoints, formed by its own expression from the same ends, so the walk is the
// reference's walk whatever the model predicted -- for as long as the visited child was one of the candidates: with
// the true root between the two targets that is until the cells are as small as 2 delta.  The model then has ends
// that close and its error falls with their distance to the third or fourth power: 6 levels (a first round without
// a model: all 63 midpoints of six levels), ~20, 50.  A wrong prediction costs progress, never correctness; the last
// kernel finishes whatever is left level by level (all workgroups alike) and writes the weights of its own slice.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vb_common.h"

namespace vb {

namespace {

constexpr int kBsParts = 2;                       // workgroups per candidate (the N samples in two blocks): with up to 112
                                                  // candidates one 1024-thread workgroup per CU holds the whole launch.
                                                  // (Round 6, VB_DIS_CLOCK: at N = 16 384 a round is ~8 us of replay + planning by
                                                  // one wave and ~5.3 us of sums -- 8 192 exponentials per workgroup; FOUR parts
                                                  // halve the sums and double the workgroups that replay: rounds 9.2 -> 12.7 us, C3
                                                  // call 192 -> 208 us.  Two it stays.)
constexpr int kBsHeapLevels = 6;                  // a round without a model: every midpoint of six levels
constexpr int kBsPath = 56;                       // nodes of a predicted path per round (one lane of a wave per level)
constexpr int kBsMaxCand = 2 * kBsPath;           // >= 2^6 - 1
constexpr int kBsHdr = 24;
constexpr int kBsPlan = kBsHdr + kBsMaxCand;      // doubles per round: header + the candidates' eps
constexpr int kBsRes = kBsMaxCand * kBsParts * 3; // [candidate][part]{sum w, sum w^2, max log w}
static_assert(kBsMaxCand >= (1 << kBsHeapLevels) - 1, "the heap round must fit the candidate list");
static_assert(kBsPath <= 64, "one lane per path level");

// header of a round's plan: the walk's state BEFORE the round's candidates are consumed, and their arrangement
enum {
  H_LOWER = 0, H_UPPER, H_LO2, H_UP2,             // interval and the evaluated ends behind its ends
  H_ESS_LO, H_ESS_UP, H_ESS_LO2, H_ESS_UP2,       // ESS at those four (NaN: never evaluated)
  H_LEVEL, H_STATUS, H_NCAND, H_MODE,             // mode 0: nothing to evaluate, 1: heap of H_LEN_A levels, 2: paths
  H_LEN_A, H_LEN_B, H_DIV, H_DIR_A, H_DIR_B,      // path lengths, first level at which B leaves A, direction bits
  H_FIN, H_FIN_S1, H_FIN_S2, H_FIN_MX, H_FIN_EPS  // the final midpoint's sums once evaluated (:358-359)
};

#ifdef VB_DIS_CLOCK
__device__ long long bs_dbg[8];
#define BS_MARK(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) bs_dbg[k] = wall_clock64(); } while (0)
#else
#define BS_MARK(k) do { } while (0)
#endif

struct BsState {
  double lower, upper, lo2, up2, ess_lo, ess_up, ess_lo2, ess_up2;
  int level, status, fin;
  double fin_s1, fin_s2, fin_mx, fin_eps;
};

__device__ __forceinline__ double bs_wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}
__device__ __forceinline__ double bs_wave_max(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x = fmax(x, __shfl_down(x, off, 64));
  return x;
}

// the sums of candidate idx over its parts, in fixed order (the ESS of a candidate is reproducible)
__device__ __forceinline__ void bs_node(const double* tab, int idx, double& t1, double& t2, double& mx) {
  t1 = 0.0, t2 = 0.0, mx = -INFINITY;
#pragma unroll
  for (int p = 0; p < kBsParts; ++p) {
    const double* e = tab + (idx * kBsParts + p) * 3;
    t1 += e[0];
    t2 += e[1];
    mx = fmax(mx, e[2]);
  }
}

__device__ __forceinline__ void bs_move(BsState& s, bool up_moves, double guess, double ess) {
  if (up_moves) {                 // :352-353
    s.up2 = s.upper, s.ess_up2 = s.ess_up;
    s.upper = guess, s.ess_up = ess;
  } else {                        // :354-355
    s.lo2 = s.lower, s.ess_lo2 = s.ess_lo;
    s.lower = guess, s.ess_lo = ess;
  }
  ++s.level;
}

// Replay a finished round (wave 0 of the workgroup, all lanes; the result is wave-uniform).
// plan / tab: that round's plan and result table in LDS.
__device__ void bs_walk(BsState& s, const double* plan, const double* tab, double ess_target, int max_its) {
  const int lane = threadIdx.x & 63;
  const int mode = (int)plan[H_MODE];
  const double* eps = plan + kBsHdr;
  if (mode == 1) {                // heap: a handful of dependent steps (uniform: every lane walks)
    const int hl = (int)plan[H_LEN_A];
    int node = 1;
    for (int l = 0; l < hl; ++l) {
      double t1, t2, mx;
      bs_node(tab, node - 1, t1, t2, mx);
      if (s.level == max_its) {   // the final midpoint (:358-359)
        s.fin = 1, s.fin_s1 = t1, s.fin_s2 = t2, s.fin_mx = mx, s.fin_eps = eps[node - 1];
        break;
      }
      if (mx == -INFINITY) s.status = 1;
      const double ess = t1 * t1 / t2;
      const bool up = ess > ess_target;
      bs_move(s, up, eps[node - 1], ess);
      node = 2 * node + (up ? 0 : 1);
    }
    return;
  }
  if (mode != 2) return;
  const int len = (int)plan[H_LEN_A], div = (int)plan[H_DIV];
  const unsigned long long dir_a = (unsigned long long)__double_as_longlong(plan[H_DIR_A]);
  const unsigned long long dir_b = (unsigned long long)__double_as_longlong(plan[H_DIR_B]);
  int n_dec = max_its - s.level;            // decision nodes on a path; the node behind them is the final midpoint
  n_dec = n_dec < len ? n_dec : len;
  const int k = lane;
  double a1 = 0.0, a2 = 1.0, am = 0.0, b1 = 0.0, b2 = 1.0, bm = 0.0, eps_a = 0.0, eps_b = 0.0;
  if (k < len) {
    bs_node(tab, k, a1, a2, am);
    eps_a = eps[k];
    if (k >= div) {
      bs_node(tab, len + k - div, b1, b2, bm);
      eps_b = eps[len + k - div];
    }
  }
  const double ess_a = a1 * a1 / a2, ess_b = b1 * b1 / b2;
  const bool dec_a = ess_a > ess_target, dec_b = ess_b > ess_target;
  const bool ok_a = k < n_dec && dec_a == (bool)((dir_a >> k) & 1);
  const bool ok_b = k < div || (k < n_dec && dec_b == (bool)((dir_b >> k) & 1));
  const unsigned long long m_a = __ballot(ok_a), m_b = __ballot(ok_b);
  const int k_a = m_a == ~0ull ? 64 : __builtin_ctzll(~m_a);          // <= n_dec
  const bool took_b = k_a < n_dec && div == k_a + 1 && div < len;
  const int k_b = m_b == ~0ull ? 64 : __builtin_ctzll(~m_b);
  const int end_k = took_b ? k_b : k_a;     // the last visited decision node, or n_dec: all of them agreed
  const int last = end_k < n_dec ? end_k : n_dec - 1;
  const bool on_a = k <= (k_a < n_dec ? k_a : n_dec - 1);
  const bool on_b = took_b && k >= div && k <= last;
  const bool on = on_a || on_b;
  const double eps_k = on_b ? eps_b : eps_a, ess_k = on_b ? ess_b : ess_a;
  const bool dec_k = on_b ? dec_b : dec_a, zero_k = (on_b ? bm : am) == -INFINITY;
  if (__ballot(on && zero_k)) s.status = 1;
  const unsigned long long m_up = __ballot(on && dec_k), m_lo = __ballot(on && !dec_k);
  auto take = [&](unsigned long long m, double& e1, double& s1, double& e2, double& s2) {
    if (!m) return;
    const int k1 = 63 - __builtin_clzll(m);
    const unsigned long long rest = m & ~(1ull << k1);
    const double ne1 = __shfl(eps_k, k1, 64), ns1 = __shfl(ess_k, k1, 64);
    if (rest) {
      const int k2 = 63 - __builtin_clzll(rest);
      e2 = __shfl(eps_k, k2, 64), s2 = __shfl(ess_k, k2, 64);
    } else {
      e2 = e1, s2 = s1;
    }
    e1 = ne1, s1 = ns1;
  };
  take(m_up, s.upper, s.ess_up, s.up2, s.ess_up2);
  take(m_lo, s.lower, s.ess_lo, s.lo2, s.ess_lo2);
  s.level += n_dec > 0 ? (end_k + 1 < n_dec ? end_k + 1 : n_dec) : 0;
  if (end_k == n_dec && n_dec < len) {      // every decision agreed and the node behind them was evaluated too
    const bool via_b = took_b || false;
    const int src = n_dec;
    const double f1 = via_b ? b1 : a1, f2 = via_b ? b2 : a2, fm = via_b ? bm : am, fe = via_b ? eps_b : eps_a;
    s.fin = 1;
    s.fin_s1 = __shfl(f1, src, 64), s.fin_s2 = __shfl(f2, src, 64), s.fin_mx = __shfl(fm, src, 64);
    s.fin_eps = __shfl(fe, src, 64);
  }
}

// root of the inverse-interpolation polynomial through (f_i, x_i), i < n, at f = 0
template <int N>
__device__ __forceinline__ double bs_inverse_root_n(const double* x, const double* f) {
  double p = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double t = x[i];
#pragma unroll
    for (int j = 0; j < N; ++j)
      if (j != i) t *= f[j] / (f[j] - f[i]);
    p += t;
  }
  return p;
}
__device__ __forceinline__ double bs_inverse_root(const double* x, const double* f, int n) {
  return n == 4 ? bs_inverse_root_n<4>(x, f) : (n == 3 ? bs_inverse_root_n<3>(x, f) : bs_inverse_root_n<2>(x, f));
}

// Where will the walk end?  [a, b] from the evaluated ends around the interval; false: no usable model.
__device__ bool bs_predict(const BsState& s, double target, double& a, double& b) {
  double x[4] = {0.0, 0.0, 0.0, 0.0}, f[4] = {1.0, 2.0, 3.0, 4.0};
  int n = 0;
  const double f_lo = s.ess_lo - target, f_up = s.ess_up - target, f_lo2 = s.ess_lo2 - target, f_up2 = s.ess_up2 - target;
  const bool v_lo = isfinite(f_lo), v_up = isfinite(f_up);
  const bool v_lo2 = v_lo && isfinite(f_lo2) && f_lo2 < f_lo && s.lo2 < s.lower;
  const bool v_up2 = v_up && isfinite(f_up2) && f_up2 > f_up && s.up2 > s.upper;
  // nearest ends first: dropping the last point gives the model of one order less
  if (v_lo && v_up) {
    if (!(f_lo < f_up)) return false;
    x[0] = s.lower, f[0] = f_lo, x[1] = s.upper, f[1] = f_up, n = 2;
    const bool lo_first = v_lo2 && (!v_up2 || fabs(f_lo2) <= fabs(f_up2));
    if (v_lo2 && v_up2) {
      x[2] = lo_first ? s.lo2 : s.up2, f[2] = lo_first ? f_lo2 : f_up2;
      x[3] = lo_first ? s.up2 : s.lo2, f[3] = lo_first ? f_up2 : f_lo2;
      n = 4;
    } else if (v_lo2 || v_up2) {
      x[2] = v_lo2 ? s.lo2 : s.up2, f[2] = v_lo2 ? f_lo2 : f_up2;
      n = 3;
    }
  } else if (v_lo && v_lo2) {           // the upper end is the caller's (never evaluated): extrapolate upwards
    x[0] = s.lower, f[0] = f_lo, x[1] = s.lo2, f[1] = f_lo2, n = 2;
  } else if (v_up && v_up2) {
    x[0] = s.upper, f[0] = f_up, x[1] = s.up2, f[1] = f_up2, n = 2;
  } else if (!v_up && s.level > 0) {      // (the upper end never moved; the lower end's ESS may even be NaN: underflown weights)
    // every decision so far moved the lower end and the evaluated points give no usable model (ESS falling towards the
    // upper end, say): the walk most likely goes on hugging that end -- one path straight to it (a heap round would
    // cover six levels; a wrong guess here costs progress, never correctness: the replay makes the reference's decisions)
    a = b = s.upper;
    return true;
  } else if (!v_lo && s.level > 0) {
    a = b = s.lower;
    return true;
  } else {
    return false;
  }
  double p = bs_inverse_root(x, f, n);
  if (!isfinite(p)) return false;
  // one-sided models: every decision so far moved the same end, the root lies beyond the evaluated one
  if (!v_up && !(p > s.lower)) p = s.upper;
  if (!v_lo && !(p < s.upper)) p = s.lower;
  const double width = s.upper - s.lower;
  // error of the highest-order root: successive orders' corrections d_k = |p_k - p_(k-1)| fall geometrically while the
  // model converges, so the next (unseen) correction is about d_n * d_n / d_(n-1); four times that.  A delta that is
  // too small costs levels (the walk leaves both paths where the cells get as small as its distance to them), one that
  // is too large costs the same levels for certain.
  double delta;
  if (n > 2) {
    const double p_prev = bs_inverse_root(x, f, n - 1);
    const double d_n = fabs(p - p_prev);
    const double d_prev = n > 3 ? fabs(p_prev - bs_inverse_root(x, f, n - 2)) : 0.25 * width;
    const double ratio = d_prev > 0.0 ? d_n / d_prev : 1.0;
    delta = 4.0 * d_n * (ratio < 1.0 ? ratio : 1.0);
  } else {
    delta = (v_lo && v_up) ? 0.125 * width : 0.5 * fabs(p - x[0]);
  }
  if (!isfinite(delta)) delta = 0.125 * width;
  delta += 4.0 * 2.220446049250313e-16 * fabs(s.upper);
  a = p - delta, b = p + delta;
  a = a > s.lower ? (a < s.upper ? a : s.upper) : s.lower;
  b = b > s.lower ? (b < s.upper ? b : s.upper) : s.lower;
  return true;
}

// Plan the next round from the walk's state (wave 0; plan in LDS, complete after the function's last barrier-free
// write: the caller synchronises).
__device__ void bs_build(const BsState& s, double* plan, double ess_target, int max_its) {
  const int lane = threadIdx.x & 63;
  double* eps = plan + kBsHdr;
  int mode = 0, len_a = 0, len_b = 0, div = 0, ncand = 0;
  unsigned long long dir_a = 0, dir_b = 0;
  double a = 0.0, b = 0.0;
  if (!s.fin) {
    const int remaining = max_its - s.level + 1;          // decisions + the final midpoint
    const bool no_model = remaining > 1 && !bs_predict(s, ess_target, a, b);
    BS_MARK(3);
    if (no_model) {
      mode = 1;
      len_a = remaining < kBsHeapLevels ? remaining : kBsHeapLevels;
      ncand = (1 << len_a) - 1;
      // heap node -> interval: one lane per node walks down from the root (<= 5 steps)
      if (lane < ncand) {
        const int node = lane + 1;
        int depth = 0;
        while ((node >> (depth + 1)) != 0) ++depth;
        double lower = s.lower, upper = s.upper;
        for (int l = depth - 1; l >= 0; --l) {
          const double guess = (lower + upper) / 2.0;
          if ((node >> l) & 1) lower = guess;
          else upper = guess;
        }
        eps[lane] = (lower + upper) / 2.0;
      }
    } else {
      mode = 2;
      if (remaining <= 1) a = b = s.lower;                // only the final midpoint is left
      len_a = len_b = remaining < kBsPath ? remaining : kBsPath;
      int n_dec = max_its - s.level;
      n_dec = n_dec < len_a ? n_dec : len_a;
      // lanes 0 and 1: the path to a and the path to b, midpoints by the reference's expression (:347, :356)
      unsigned long long dir = 0;
      if (lane < 2) {
        const double x = lane == 0 ? a : b;
        double lower = s.lower, upper = s.upper;
        double* out = eps + lane * kBsPath;               // B's nodes are compacted below
        // (VB_DIS_CLOCK: ~67 ns a level -- a dozen dependent fp64 / select instructions of ONE wave --, 3 us of every path round
        // at 45 levels, with selects as with the branches this loop had first; together with the table's reload (1.6 us), the
        // replay (1-1.8 us) and the model (0.5-1.3 us) the ~8 us a path round spends before its sums)
        for (int k = 0; k < len_a; ++k) {
          const double guess = (lower + upper) / 2.0;
          out[k] = guess;
          const bool dec = k < n_dec;
          const bool left = x < guess;                    // the cell that holds x: `upper = guess` (:353)
          dir |= (unsigned long long)(dec && left) << k;
          upper = dec && left ? guess : upper;
          lower = dec && !left ? guess : lower;
        }
      }
      BS_MARK(4);
      const unsigned lo32 = (unsigned)dir, hi32 = (unsigned)(dir >> 32);
      dir_a = ((unsigned long long)__shfl(hi32, 0, 64) << 32) | __shfl(lo32, 0, 64);
      dir_b = ((unsigned long long)__shfl(hi32, 1, 64) << 32) | __shfl(lo32, 1, 64);
      const unsigned long long diff = dir_a ^ dir_b;
      div = diff ? __builtin_ctzll(diff) + 1 : len_a;
      if (div > len_a) div = len_a;
      // compact: B's own nodes (levels div ..) directly behind A's
      const double vb = (lane >= div && lane < len_b) ? eps[kBsPath + lane] : 0.0;
      __builtin_amdgcn_wave_barrier();
      if (lane >= div && lane < len_b) eps[len_a + lane - div] = vb;
      ncand = len_a + (len_b - div);
    }
  }
  if (lane == 0) {
    plan[H_LOWER] = s.lower, plan[H_UPPER] = s.upper, plan[H_LO2] = s.lo2, plan[H_UP2] = s.up2;
    plan[H_ESS_LO] = s.ess_lo, plan[H_ESS_UP] = s.ess_up, plan[H_ESS_LO2] = s.ess_lo2, plan[H_ESS_UP2] = s.ess_up2;
    plan[H_LEVEL] = s.level, plan[H_STATUS] = s.status, plan[H_NCAND] = ncand, plan[H_MODE] = mode;
    plan[H_LEN_A] = len_a, plan[H_LEN_B] = len_b, plan[H_DIV] = div;
    plan[H_DIR_A] = __longlong_as_double((long long)dir_a), plan[H_DIR_B] = __longlong_as_double((long long)dir_b);
    plan[H_FIN] = s.fin, plan[H_FIN_S1] = s.fin_s1, plan[H_FIN_S2] = s.fin_s2, plan[H_FIN_MX] = s.fin_mx;
    plan[H_FIN_EPS] = s.fin_eps;
  }
}

__device__ __forceinline__ BsState bs_load_state(const double* plan) {
  BsState s;
  s.lower = plan[H_LOWER], s.upper = plan[H_UPPER], s.lo2 = plan[H_LO2], s.up2 = plan[H_UP2];
  s.ess_lo = plan[H_ESS_LO], s.ess_up = plan[H_ESS_UP], s.ess_lo2 = plan[H_ESS_LO2], s.ess_up2 = plan[H_ESS_UP2];
  s.level = (int)plan[H_LEVEL], s.status = (int)plan[H_STATUS], s.fin = (int)plan[H_FIN];
  s.fin_s1 = plan[H_FIN_S1], s.fin_s2 = plan[H_FIN_S2], s.fin_mx = plan[H_FIN_MX], s.fin_eps = plan[H_FIN_EPS];
  return s;
}

// Stage the previous round in LDS (all threads), replay it and plan this round (wave 0); the plan is in `plan` for
// everybody after the call.  prev_plan == nullptr: the first round, [0, eps_prev] (:344-346).
__device__ void bs_advance(double* plan, double* tab, const double* __restrict__ prev_plan,
                           const double* __restrict__ prev_res, double eps_prev, double ess_target, int max_its,
                           bool build) {
  BS_MARK(0);
  if (prev_plan) {       // one round trip: the whole table, not only the previous round's candidates (672 doubles)
    for (int e = threadIdx.x; e < kBsPlan; e += blockDim.x) plan[e] = prev_plan[e];
    for (int e = threadIdx.x; e < kBsRes; e += blockDim.x) tab[e] = prev_res[e];
    __syncthreads();
  }
  BS_MARK(1);
  if (threadIdx.x < 64) {
    BsState s;
    if (prev_plan) {
      s = bs_load_state(plan);
      bs_walk(s, plan, tab, ess_target, max_its);
      BS_MARK(2);
    } else {
      s.lower = 0.0, s.upper = eps_prev, s.lo2 = s.up2 = 0.0;
      s.ess_lo = s.ess_up = s.ess_lo2 = s.ess_up2 = NAN;
      s.level = 0, s.status = 0, s.fin = 0;
      s.fin_s1 = s.fin_s2 = s.fin_mx = s.fin_eps = 0.0;
    }
    __builtin_amdgcn_wave_barrier();
    if (build) {
      bs_build(s, plan, ess_target, max_its);
    } else if (threadIdx.x == 0) {
      plan[H_LOWER] = s.lower, plan[H_UPPER] = s.upper;
      plan[H_LEVEL] = s.level, plan[H_STATUS] = s.status;
      plan[H_FIN] = s.fin, plan[H_FIN_S1] = s.fin_s1, plan[H_FIN_S2] = s.fin_s2, plan[H_FIN_MX] = s.fin_mx;
      plan[H_FIN_EPS] = s.fin_eps;
    }
  }
  __syncthreads();
}

// one sample's log weight and log q at tempering `guess` (:317-323)
struct BsSample {
  double lp, b, lprior;
};

// sums of w and w^2 and the maximum log weight over samples [i_begin, i_end) at `guess`; the first trip's operands
// come preloaded (fetched while the walk ran).  Four independent elements per trip: the dependent fp64 chains of
// `exp` overlap.  sh: 48 doubles of LDS; out3 (any memory) = {sum w, sum w^2, max}.
template <bool STORE>
__device__ __forceinline__ void bs_sums(const double* __restrict__ lp, const double* __restrict__ b,
                                        const double* __restrict__ lprior, const BsSample* pre, double sum_ls,
                                        double guess, int64_t i_begin, int64_t i_end, double* __restrict__ w,
                                        double* __restrict__ lq_out, double* sh, double* out3) {
  double s1 = 0.0, s2 = 0.0, mx = -INFINITY;
  bool first = pre != nullptr;
  for (int64_t i0 = i_begin + threadIdx.x; i0 < i_end; i0 += 4 * 1024) {
    double lw[4], lq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * 1024;
      const bool in = i < i_end;
      const double vb = first ? pre[u].b : (in ? b[i] : 0.0);
      const double vp = first ? pre[u].lp : (in ? lp[i] : 0.0);
      const double vr = first ? pre[u].lprior : (in ? lprior[i] : 0.0);
      lq[u] = in ? vb - sum_ls : 0.0;
      lw[u] = in ? guess * vr + (1.0 - guess) * vp - lq[u] : -INFINITY;
    }
    first = false;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * 1024;
      if (i < i_end) {
        const double wv = exp(lw[u]);
        mx = fmax(mx, lw[u]);
        s1 += wv;
        s2 = fma(wv, wv, s2);
        if (STORE) {
          w[i] = wv;
          lq_out[i] = lq[u];
        }
      }
    }
  }
  s1 = bs_wave_sum(s1);
  s2 = bs_wave_sum(s2);
  mx = bs_wave_max(mx);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) {
    sh[wave] = s1;
    sh[16 + wave] = s2;
    sh[32 + wave] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t1 = 0.0, t2 = 0.0, tm = -INFINITY;
    for (int k = 0; k < 16; ++k) {
      t1 += sh[k];
      t2 += sh[16 + k];
      tm = fmax(tm, sh[32 + k]);
    }
    out3[0] = t1;
    out3[1] = t2;
    out3[2] = tm;
  }
}

__device__ __forceinline__ void bs_preload(const double* __restrict__ lp, const double* __restrict__ b,
                                           const double* __restrict__ lprior, int64_t i_begin, int64_t i_end,
                                           BsSample* pre) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = i_begin + threadIdx.x + u * 1024;
    const bool in = i < i_end;
    pre[u].lp = in ? lp[i] : 0.0;
    pre[u].b = in ? b[i] : 0.0;
    pre[u].lprior = in ? lprior[i] : 0.0;
  }
}

// grid = kBsMaxCand x kBsParts; workgroup (c, part) sums block `part` of the samples for candidate c of this round's plan
__global__ void __launch_bounds__(1024) dis_spec_round_kernel(const double* __restrict__ lp, const double* __restrict__ b,
                                                              const double* __restrict__ lprior,
                                                              const double* __restrict__ scal_in, int64_t n,
                                                              double ess_target, int max_its, double eps_prev,
                                                              const double* __restrict__ prev_plan,
                                                              const double* __restrict__ prev_res,
                                                              double* __restrict__ plan_out, double* __restrict__ res_out) {
  __shared__ double plan[kBsPlan];
  __shared__ double tab[kBsRes];
  __shared__ double sh[48];
  const int c = blockIdx.x / kBsParts, part = blockIdx.x % kBsParts;
  const int64_t per = (n + kBsParts - 1) / kBsParts;
  const int64_t i_begin = part * per, i_end = i_begin + per < n ? i_begin + per : n;
#ifdef VB_DIS_CLOCK
  const long long dbg0 = wall_clock64();
#endif
  BsSample pre[4];
  bs_preload(lp, b, lprior, i_begin, i_end, pre);
  const double sum_ls = scal_in[0];
  bs_advance(plan, tab, prev_plan, prev_res, eps_prev, ess_target, max_its, true);
#ifdef VB_DIS_CLOCK
  const long long dbg1 = wall_clock64();
#endif
  if (blockIdx.x == 0)
    for (int e = threadIdx.x; e < kBsPlan; e += blockDim.x) plan_out[e] = plan[e];
  if (c >= (int)plan[H_NCAND]) return;
  const double guess = plan[kBsHdr + c];
  bs_sums<false>(lp, b, lprior, pre, sum_ls, guess, i_begin, i_end, nullptr, nullptr, sh,
                 res_out + (c * kBsParts + part) * 3);
#ifdef VB_DIS_CLOCK      // (tools/build_variant.sh disclk "-DVB_DIS_CLOCK": phase times of a round, 100 MHz ticks)
  if (threadIdx.x == 0 && blockIdx.x == 0)
    printf("[dis round] mode %d ncand %d: advance %lld (load %lld walk %lld predict %lld paths %lld rest %lld) sums %lld ticks (10 ns)\n",
           (int)plan[H_MODE], (int)plan[H_NCAND], dbg1 - dbg0, bs_dbg[1] - bs_dbg[0], bs_dbg[2] - bs_dbg[1], bs_dbg[3] - bs_dbg[2],
           bs_dbg[4] - bs_dbg[3], dbg1 - bs_dbg[4], wall_clock64() - dbg1);
#endif
}

// Last step (:358-366): replay the last round; finish what is left of the walk (normally nothing) level by level,
// every workgroup alike over all samples; the weights of this workgroup's slice at the final midpoint; eps snapped
// to the ends.  scal_out = [eps, ess, status]; status 1 = "all weights zero" (max logw == -inf, :325-328).
__global__ void __launch_bounds__(1024) dis_spec_final_kernel(const double* __restrict__ lp, const double* __restrict__ b,
                                                              const double* __restrict__ lprior,
                                                              const double* __restrict__ scal_in, int64_t n,
                                                              double ess_target, int max_its, double eps_prev,
                                                              const double* __restrict__ prev_plan,
                                                              const double* __restrict__ prev_res, double max_eps,
                                                              double* __restrict__ w, double* __restrict__ lq_out,
                                                              double* __restrict__ scal_out) {
  __shared__ double plan[kBsPlan];
  __shared__ double tab[kBsRes];
  __shared__ double sh[48];
  __shared__ double tot[3];
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t i_begin = blockIdx.x * per, i_end = i_begin + per < n ? i_begin + per : n;
  BsSample pre[4];
  bs_preload(lp, b, lprior, i_begin, i_end, pre);
  const double sum_ls = scal_in[0];
  bs_advance(plan, tab, prev_plan, prev_res, eps_prev, ess_target, max_its, false);
  double lower = plan[H_LOWER], upper = plan[H_UPPER];
  int level = (int)plan[H_LEVEL], status = (int)plan[H_STATUS];
  bool fin = plan[H_FIN] != 0.0;
  double s1 = plan[H_FIN_S1], s2 = plan[H_FIN_S2], mx = plan[H_FIN_MX];
  if (!fin) {
    while (level < max_its) {
      const double guess = (lower + upper) / 2.0;
      __syncthreads();
      bs_sums<false>(lp, b, lprior, nullptr, sum_ls, guess, 0, n, nullptr, nullptr, sh, tot);
      __syncthreads();
      if (tot[2] == -INFINITY) status = 1;
      if (tot[0] * tot[0] / tot[1] > ess_target) upper = guess;
      else lower = guess;
      ++level;
    }
    __syncthreads();
    bs_sums<false>(lp, b, lprior, nullptr, sum_ls, (lower + upper) / 2.0, 0, n, nullptr, nullptr, sh, tot);
    __syncthreads();
    s1 = tot[0], s2 = tot[1], mx = tot[2];
  }
  const double guess = (lower + upper) / 2.0;
  __syncthreads();
  bs_sums<true>(lp, b, lprior, pre, sum_ls, guess, i_begin, i_end, w, lq_out, sh, tot);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double eps = guess;
    if (lower == 0.0) eps = 0.0;          // :363-366
    if (upper == max_eps) eps = max_eps;
    scal_out[0] = eps;
    scal_out[1] = s1 * s1 / s2;
    scal_out[2] = (double)((status != 0 || mx == -INFINITY) ? 1 : 0);
  }
}


// ---- the same walk as ONE resident launch (round 5) ------------------------------------------------------------------------
// The rounds above are launches because round r + 1 replays round r's table.  Here the workgroups of one launch -- two per
// candidate, all resident: checked against the occupancy the runtime reports -- keep the plan in LDS and their first trip
// of samples in registers, exchange only the result table (write-through agent-scope stores, agent-scope loads: the
// XCDs' L2s are not coherent with each other) and meet at a grid barrier per round (a counter that runs on from launch
// to launch: every launch adds exactly G * kBsMaxBarriers, workgroups that leave early add the rest in one step).  Every
// workgroup replays and plans alike, so all of them see `nothing left to evaluate` in the same round and go on to the
// final step together: the spare rounds of the launch chain (two launches of ~5 us that normally find the walk
// finished) do not exist, and a round costs its sums + one barrier instead of a launch's ramp and tail (measured: no
// gain, see dis_bisect_enqueue -- opt-in).  Same functions,
// same partition of the samples, same order of every sum: the results are the launch chain's bit for bit
// (tests/test_gpu_dis_bisect.py runs both).  A workgroup that does not arrive within the poll bound poisons the launch:
// status 3 in scal_out[2], which the callers turn into VB_ERR_STATE.
constexpr int kBsMaxBarriers = 8;

__device__ __forceinline__ void bsp_st(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double bsp_ld(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}

struct BsBarrier {
  unsigned long long* bar;        // [0] counter, [2] poison
  unsigned long long target, tag;
  int g_count, used;
  __device__ __forceinline__ void wait() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave: its write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
      target += (unsigned long long)g_count;
      ++used;
      atomicAdd(bar, 1ull);
      int spins = 0;
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 24)) {
          atomicExch(bar + 2, tag);
          break;
        }
      }
    }
    __syncthreads();
  }
};

__global__ void __launch_bounds__(1024) dis_spec_resident_kernel(const double* __restrict__ lp, const double* __restrict__ b,
                                                                 const double* __restrict__ lprior,
                                                                 const double* __restrict__ scal_in, int64_t n,
                                                                 double ess_target, int max_its, double eps_prev, int rounds,
                                                                 double max_eps, double* __restrict__ res,
                                                                 unsigned long long* __restrict__ bar,
                                                                 unsigned long long bar_base, double* __restrict__ w,
                                                                 double* __restrict__ lq_out, double* __restrict__ scal_out,
                                                                 double* __restrict__ plans_out) {
  __shared__ double plan[kBsPlan];
  __shared__ double tab[kBsRes];
  __shared__ double sh[48];
  __shared__ double tot[3];
  const int c = blockIdx.x / kBsParts, part = blockIdx.x % kBsParts;
  const int64_t per = (n + kBsParts - 1) / kBsParts;
  const int64_t i_begin = part * per, i_end = i_begin + per < n ? i_begin + per : n;
  BsSample pre[4];
  bs_preload(lp, b, lprior, i_begin, i_end, pre);
  const double sum_ls = scal_in[0];
  BsBarrier barrier{bar, bar_base, bar_base + 1, (int)gridDim.x, 0};
  bool have_prev = false;
  int r = 0;
  for (; r < rounds; ++r) {
    if (have_prev) {
      const double* prev = res + (size_t)(r - 1) * kBsRes;
      for (int e = threadIdx.x; e < kBsRes; e += blockDim.x) tab[e] = bsp_ld(prev + e);
      __syncthreads();
    }
    if (threadIdx.x < 64) {
      BsState s;
      if (have_prev) {
        s = bs_load_state(plan);
        bs_walk(s, plan, tab, ess_target, max_its);
      } else {
        s.lower = 0.0, s.upper = eps_prev, s.lo2 = s.up2 = 0.0;
        s.ess_lo = s.ess_up = s.ess_lo2 = s.ess_up2 = NAN;
        s.level = 0, s.status = 0, s.fin = 0;
        s.fin_s1 = s.fin_s2 = s.fin_mx = s.fin_eps = 0.0;
      }
      __builtin_amdgcn_wave_barrier();
      bs_build(s, plan, ess_target, max_its);
    }
    __syncthreads();
    if (plans_out && blockIdx.x == 0)      // (development trace, VB_DIS_TRACE: what each round planned)
      for (int e = threadIdx.x; e < kBsPlan; e += blockDim.x) plans_out[(size_t)r * kBsPlan + e] = plan[e];
    const int ncand = (int)plan[H_NCAND];
    if (ncand == 0) break;            // the walk is over (every workgroup finds the same): on to the final step
    if (c < ncand) {
      bs_sums<false>(lp, b, lprior, pre, sum_ls, plan[kBsHdr + c], i_begin, i_end, nullptr, nullptr, sh, tot);
      if (threadIdx.x == 0) {
        double* out3 = res + (size_t)r * kBsRes + (c * kBsParts + part) * 3;
        bsp_st(out3, tot[0]);
        bsp_st(out3 + 1, tot[1]);
        bsp_st(out3 + 2, tot[2]);
      }
    }
    barrier.wait();
    have_prev = true;
  }
  if (threadIdx.x == 0 && barrier.used < kBsMaxBarriers)
    atomicAdd(bar, (unsigned long long)(kBsMaxBarriers - barrier.used));
  if (r == rounds && have_prev) {     // the last round's candidates were evaluated: replay them (state only)
    const double* prev = res + (size_t)(rounds - 1) * kBsRes;
    __syncthreads();
    for (int e = threadIdx.x; e < kBsRes; e += blockDim.x) tab[e] = bsp_ld(prev + e);
    __syncthreads();
    if (threadIdx.x < 64) {
      BsState s = bs_load_state(plan);
      bs_walk(s, plan, tab, ess_target, max_its);
      __builtin_amdgcn_wave_barrier();
      if (threadIdx.x == 0) {
        plan[H_LOWER] = s.lower, plan[H_UPPER] = s.upper;
        plan[H_LEVEL] = s.level, plan[H_STATUS] = s.status;
        plan[H_FIN] = s.fin, plan[H_FIN_S1] = s.fin_s1, plan[H_FIN_S2] = s.fin_s2, plan[H_FIN_MX] = s.fin_mx;
        plan[H_FIN_EPS] = s.fin_eps;
      }
    }
    __syncthreads();
  }
  // ---- the final step (dis_spec_final_kernel's, on this grid) ----
  double lower = plan[H_LOWER], upper = plan[H_UPPER];
  int level = (int)plan[H_LEVEL], status = (int)plan[H_STATUS];
  const bool fin = plan[H_FIN] != 0.0;
  double s1 = plan[H_FIN_S1], s2 = plan[H_FIN_S2], mx = plan[H_FIN_MX];
  if (!fin) {
    while (level < max_its) {
      const double guess = (lower + upper) / 2.0;
      __syncthreads();
      bs_sums<false>(lp, b, lprior, nullptr, sum_ls, guess, 0, n, nullptr, nullptr, sh, tot);
      __syncthreads();
      if (tot[2] == -INFINITY) status = 1;
      if (tot[0] * tot[0] / tot[1] > ess_target) upper = guess;
      else lower = guess;
      ++level;
    }
    __syncthreads();
    bs_sums<false>(lp, b, lprior, nullptr, sum_ls, (lower + upper) / 2.0, 0, n, nullptr, nullptr, sh, tot);
    __syncthreads();
    s1 = tot[0], s2 = tot[1], mx = tot[2];
  }
  const double guess = (lower + upper) / 2.0;
  const int64_t fper = (n + gridDim.x - 1) / gridDim.x;
  const int64_t f_begin = blockIdx.x * fper, f_end = f_begin + fper < n ? f_begin + fper : n;
  __syncthreads();
  if (f_begin < n) bs_sums<true>(lp, b, lprior, nullptr, sum_ls, guess, f_begin, f_end, w, lq_out, sh, tot);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double eps = guess;
    if (lower == 0.0) eps = 0.0;          // :363-366
    if (upper == max_eps) eps = max_eps;
    const bool poisoned = __hip_atomic_load(bar + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == bar_base + 1;
    scal_out[0] = eps;
    scal_out[1] = s1 * s1 / s2;
    scal_out[2] = poisoned ? 3.0 : (double)((status != 0 || mx == -INFINITY) ? 1 : 0);
  }
}

}  // namespace

int dis_bisect_lookahead_enqueue(vb_ctx* ctx, const double* lp, const double* b, const double* lprior,
                                 const double* scal_in, int64_t n, double eps_prev, double ess_target, int max_its,
                                 double* w, double* lq_out, double* scal_out);

int dis_bisect_enqueue(vb_ctx* ctx, const double* lp, const double* b, const double* lprior, const double* scal_in,
                       int64_t n, double eps_prev, double ess_target, int max_its, double* w, double* lq_out,
                       double* scal_out) {
  if (max_its < 0) return fail(ctx, VB_ERR_INVALID, "max_its must be >= 0");
  const char* env = getenv("VB_DIS_BISECT");                // 0: the look-ahead rounds of round 3 (cross-check)
  if (env && atoi(env) == 0) ctx->bisect_bar_ptr = nullptr;      // (the look-ahead rounds lay the work buffer out their own way)
  if (env && atoi(env) == 0)
    return dis_bisect_lookahead_enqueue(ctx, lp, b, lprior, scal_in, n, eps_prev, ess_target, max_its, w, lq_out, scal_out);
  // rounds: one covers max_its + 1 <= 6 levels outright; the typical 50-level walk is over after three (6, ~20, 50)
  // and the fourth is a spare whose workgroups leave at once; whatever is left the final kernel finishes
  int rounds = max_its + 1 <= kBsHeapLevels ? 1 : 4 + (max_its > 50 ? (max_its - 50 + kBsPath - 1) / kBsPath : 0);
  if (const char* r = getenv("VB_DIS_ROUNDS")) rounds = atoi(r) > 0 ? atoi(r) : rounds;
  const size_t need = (size_t)rounds * (kBsPlan + kBsRes) * sizeof(double);
  VB_TRY(ensure(ctx, ctx->bisect_work, need + 64));
  double* plans = (double*)ctx->bisect_work.ptr;
  double* res = plans + (size_t)rounds * kBsPlan;
  hipStream_t st = ctx->stream;
  // candidates a round can have: the heap's 63, or two paths over what is left of the walk (the first round, which has
  // no model, covers six levels)
  const int heap_cands = (1 << (max_its + 1 < kBsHeapLevels ? max_its + 1 : kBsHeapLevels)) - 1;
  int path_cands = max_its + 1 - kBsHeapLevels;
  path_cands = 2 * (path_cands < 1 ? 1 : (path_cands > kBsPath ? kBsPath : path_cands));
  {
    // VB_DIS_RESIDENT=1: one resident launch when all its workgroups fit the device at once.  OFF by default -- measured
    // (C3 shape, MI355X): 36 us against the chain's 39-40 us of kernels, no difference in the call (0.25-0.27 ms both
    // ways, four A/B runs): a round costs its sums (~1 us) + a 180-workgroup barrier across the XCDs (~3 us) + the table's
    // reload through agent-scope loads (~2.5 us) + replay and planning (~1.5 us), which is what a queued launch costs
    // too.  And two PROCESSES sharing a GPU (tests/test_gpu_two_ranks.py) each hold half of the slots the other's grid
    // waits for: both time out (status 3).  Kept for the record and for tests/test_gpu_dis_bisect.py.
    const char* re = getenv("VB_DIS_RESIDENT");
    static int per_cu = -1;
    if (per_cu < 0) {
      int nb = 0;
      per_cu = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, dis_spec_resident_kernel, 1024, 0) == hipSuccess ? nb : 0;
    }
    const int grid = (heap_cands > path_cands ? heap_cands : path_cands) * kBsParts;
    if (re && atoi(re) != 0 && rounds <= kBsMaxBarriers && grid <= per_cu * ctx->prop.multiProcessorCount) {
      const size_t o_bar = (size_t)rounds * (kBsPlan + kBsRes);      // behind the tables: counter | - | poison
      if (ctx->bisect_bar_words != o_bar || ctx->bisect_bar_ptr != (void*)plans) {      // (a new or re-laid-out allocation)
        VB_HIP(ctx, hipMemsetAsync(plans + o_bar, 0, 64, st));
        ctx->bisect_bar_base = 0;
        ctx->bisect_bar_words = o_bar;
        ctx->bisect_bar_ptr = (void*)plans;
      }
      hipLaunchKernelGGL(dis_spec_resident_kernel, dim3((unsigned)grid), dim3(1024), 0, st, lp, b, lprior, scal_in, n, ess_target,
                         max_its, eps_prev, rounds, 1.0, res, (unsigned long long*)(plans + o_bar), ctx->bisect_bar_base, w,
                         lq_out, scal_out, getenv("VB_DIS_TRACE") ? plans : (double*)nullptr);
      VB_HIP(ctx, hipGetLastError());
      ctx->bisect_bar_base += (unsigned long long)grid * kBsMaxBarriers;
      if (getenv("VB_DIS_TRACE")) {
        std::vector<double> h((size_t)rounds * kBsPlan);
        VB_HIP(ctx, hipStreamSynchronize(st));
        VB_HIP(ctx, hipMemcpy(h.data(), plans, h.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int r = 0; r < rounds; ++r) {
          const double* q = h.data() + (size_t)r * kBsPlan;
          fprintf(stderr, "[dis bisect, resident] round %d: level %d mode %d candidates %d len %d div %d fin %d interval [%.17g, %.17g]\n",
                  r, (int)q[H_LEVEL], (int)q[H_MODE], (int)q[H_NCAND], (int)q[H_LEN_A], (int)q[H_DIV], (int)q[H_FIN], q[H_LOWER],
                  q[H_UPPER]);
        }
      }
      return VB_OK;
    }
  }
  for (int r = 0; r < rounds; ++r) {
    const double* pp = r > 0 ? plans + (size_t)(r - 1) * kBsPlan : nullptr;
    const double* pr = r > 0 ? res + (size_t)(r - 1) * kBsRes : nullptr;
    const int cands = r == 0 || heap_cands > path_cands ? heap_cands : path_cands;
    hipLaunchKernelGGL(dis_spec_round_kernel, dim3((unsigned)(cands * kBsParts)), dim3(1024), 0, st, lp, b, lprior, scal_in, n,
                       ess_target, max_its, eps_prev, pp, pr, plans + (size_t)r * kBsPlan, res + (size_t)r * kBsRes);
  }
  int slices = (int)((n + 1023) / 1024);
  slices = slices < 1 ? 1 : (slices > 64 ? 64 : slices);
  hipLaunchKernelGGL(dis_spec_final_kernel, dim3(slices), dim3(1024), 0, st, lp, b, lprior, scal_in, n, ess_target,
                     max_its, eps_prev, (const double*)(plans + (size_t)(rounds - 1) * kBsPlan),
                     (const double*)(res + (size_t)(rounds - 1) * kBsRes), 1.0, w, lq_out, scal_out);
  VB_HIP(ctx, hipGetLastError());
  if (getenv("VB_DIS_TRACE")) {                               // development: what each round planned
    std::vector<double> h((size_t)rounds * kBsPlan);
    VB_HIP(ctx, hipStreamSynchronize(st));
    VB_HIP(ctx, hipMemcpy(h.data(), plans, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int r = 0; r < rounds; ++r) {
      const double* q = h.data() + (size_t)r * kBsPlan;
      fprintf(stderr, "[dis bisect] round %d: level %d mode %d candidates %d len %d div %d fin %d interval [%.17g, %.17g]\n", r,
              (int)q[H_LEVEL], (int)q[H_MODE], (int)q[H_NCAND], (int)q[H_LEN_A], (int)q[H_DIV], (int)q[H_FIN], q[H_LOWER],
              q[H_UPPER]);
    }
  }
  return VB_OK;
}

}  // namespace vb
