// Mean-field ExclusiveKL / weighted-gradient pipeline for gfx950 (MI355X).
//
// Replaces, for MFGaussian / MFStudentT, the whole body of
//   viabel/objectives.py:154-168 (plain + path-derivative ELBO, autograd backward pass) and
//   viabel/objectives.py:170-271 (RGE control variates),
// i.e. sample (approximations.py:212-216) + model log density and gradient + entropy /
// log q + the Monte-Carlo mean, in ONE streaming pass over the noise matrix.
//
// Kernels (all on the context's stream)
//   mf_prep_kernel    O(D + N).  Reads theta (straight from pinned host memory), writes the
//                     device copy plus per-column constants (mu - m, sigma = exp(log_sigma), 1/sd^2)
//                     and, for row-coupled models (funnel) or weighted sums, per-row scalars
//                     (exp(-2 v_n) * w_n, eps_nk, w_n) and the sums that involve the coupling
//                     column only.  Keeps every transcendental out of the streaming kernel.
//   mf_accum_kernel   HBM-bound, the dominant kernel.  Grid = row-blocks x column-blocks; a
//                     workgroup (4 waves) owns 128 columns (one 1-KiB, 16-B-per-lane coalesced
//                     wave load per row) and a block of rows; a wave issues 16 row loads (16 KiB)
//                     before its first use.  Per-column sums live in registers, per-column
//                     constants are loaded once, per-row scalars arrive through the scalar cache
//                     (s_load: the row index is wave-uniform), the 4 waves are combined through
//                     LDS, one partial per column per workgroup is written.  No atomics: the
//                     result is deterministic.  blockIdx -> (row-block, column-block) keeps the
//                     column blocks of one row block on one XCD (block b runs on XCD b % 8).
//   mf_finalize_kernel  Dp/64 workgroups: fixed-order sum of the row-block partials and, for the
//                     estimators without cross-column coupling, the O(D) epilogue straight into
//                     pinned host memory.  In reduce-only mode it writes the sum vector that a
//                     multi-GPU job all-reduces.
//   mf_epilogue_kernel  one workgroup, O(D): control-variate variants, weighted (alpha / DIS)
//                     gradients and the post-all-reduce epilogue.
//
// Algorithmic HBM bytes per evaluation: N*D*8 (noise) + 4*D*8 (theta, grad) + 8.
#include "vb_common.h"
#include "vb_fit.h"
#include "vb_rng.h"

namespace vb {

typedef double d2 __attribute__((ext_vector_type(2)));

// Up to kMaxBatch independent evaluations (own noise matrix, own theta, own result buffer; same
// shapes and estimator) ride in ONE launch of each kernel: blockIdx.y is the evaluation index.
// Per-evaluation pointers travel in the kernel arguments; per-evaluation work buffers are
// `stride` doubles apart in one workspace allocation.
struct BatchPtrs {
  const double* noise[kMaxBatch];
  const double* theta_src[kMaxBatch];   // [mu | log_sigma], device-visible (pinned host or device)
  double* out[kMaxBatch];               // [value | grad(2D)], device-visible
  const double* roww[kMaxBatch];        // per-row weights or nullptr
};

struct Workspace {
  double* base;
  int64_t stride;          // doubles between consecutive evaluations
  int64_t off_theta;       // [2D]            device copy of theta
  int64_t off_colp;        // [3][Dp]         per-column constants
  int64_t off_rowscal;     // [n][4]          per-row scalars {a, eps_k, w, 0}
  int64_t off_prepscal;    // [PS_NUM][n_prep]
  int64_t off_partials;    // [Dp / 64][n_rb][CF_NUM][64] (partial_index)
  int64_t off_pscal;       // [KS_NUM][n_rb * n_cb]
  double* sums;            // [B][sum_len]: [SF_NUM] | [nf][Dp]   (the all-reduced vector)
  int64_t sum_len;
};

struct Geom {
  int64_t ld, n;
  int d, Dp;
  int n_rb, n_cb, n_prep;
  int rows_per_wg;
  int xcd_map;
  int rows;                // 1: per-row scalars needed
  double df;
  int gen;                 // 1 / 2: Gaussian / Student-t (df = gdf) noise generated in registers (Philox key gk0 / gk1,
                           // stream word gw, first row grow0)
  double gdf;
  uint32_t gk0, gk1, gw;
  int64_t grow0;
  // in-register noise on the funnel: the streaming kernel also forms the per-row scalars of its own rows (the
  // coupling column's draw e_nk and exp(-2 v_n)) and, in column block 0, the row block's partial sums of the terms
  // that involve that column only -- what mf_prep_kernel's row part does for resident noise.  fk / ftau: the
  // funnel's coupling column and log-scale stdev.
  int inline_rows;
  int fk;
  double ftau;
};

constexpr double kLog2Pi = 1.8378770664093454835606594728112;

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

// Words that workgroups of ONE launch hand to each other (the one-launch evaluation, mf_one_kernel): agent-scope atomics --
// write-through `sc1` stores, `sc1` loads.  The eight XCDs' L2s are not coherent with each other; a plain load could be
// served a stale line whatever fences surround it (vb_psis.hip has the long version).
template <bool COH>
__device__ __forceinline__ double cld(const double* p) {
  if (COH)
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
  return *p;
}
template <bool COH>
__device__ __forceinline__ void cst(double* p, double v) {
  if (COH)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  else
    *p = v;
}

// block-wide sum for 256 threads; every thread gets the total
__device__ __forceinline__ double block_sum(double x, double* sh) {
  x = wave_sum(x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = x;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  return t;
}

// ------------------------------------------------------------------------------------------------
// prep
// ------------------------------------------------------------------------------------------------
// column part: device copy of theta and the per-column constants of column i (i < Dp; zeros for the pad columns)
__device__ __forceinline__ void prep_column(int64_t i, double mu, double ls, double* wsb, const Workspace& ws,
                                            const Geom& g, const ModelDev& model) {
  const int d = g.d;
  double c0 = 0.0, c1 = 0.0, c2 = 0.0;
  if (i < d) {
    wsb[ws.off_theta + i] = mu;
    wsb[ws.off_theta + d + i] = ls;
    const double sg = exp(ls);
    if (model.id == kModelLogQ) {   // p0 = mu_r, p1 = sigma_r of the refresh parameter
      c0 = (model.p0[i] - mu) / sg;
      c1 = model.p1[i] / sg;
    } else if (model.id == VB_MODEL_FUNNEL) {
      // the coupling column contributes through the per-row sums below, not elementwise
      c0 = (i == model.k) ? 0.0 : mu;
      c1 = (i == model.k) ? 0.0 : sg;
    } else {
      c0 = mu - model.p0[i];
      c1 = sg;
      c2 = model.p1[i];
    }
  }
  double* colp = wsb + ws.off_colp;
  colp[i] = c0;
  colp[g.Dp + i] = c1;
  colp[2 * (int64_t)g.Dp + i] = c2;
}

// row part of prep block `blk` of `nblk` (rows blk * 256 + t): per-row scalars and the block's partial sums of the
// terms that involve the coupling column only; (muk, lsk) = theta of the funnel's coupling column.  All 256
// threads of the workgroup take part (one barrier inside).
__device__ __forceinline__ void prep_rows_block(int blk, int nblk, double* wsb, const Workspace& ws, const Geom& g,
                                                const ModelDev& model, const double* __restrict__ noise,
                                                const double* __restrict__ roww, double muk, double lsk,
                                                double (*sh)[PS_NUM]) {
  const int64_t i = (int64_t)blk * 256 + threadIdx.x;
  const int d = g.d;
  const bool funnel = model.id == VB_MODEL_FUNNEL;
  double W = 0.0, FK = 0.0, GK = 0.0, GEK = 0.0;
  if (g.rows) {
    if (i < g.n) {
      const double wt = roww ? roww[i] : 1.0;
      double av = wt, ek = 0.0;
      if (funnel) {
        const int k = model.k;
        const double sgk = exp(lsk);
        const double it2 = 1.0 / (model.tau * model.tau), dm1 = (double)(d - 1);
        if (g.gen == 2) {
          ek = student_t_polar(g.gdf, (uint64_t)(g.grow0 + i), (uint32_t)(k >> 1), g.gw, (uint32_t)(k & 1), g.gk0, g.gk1);
        } else if (g.gen) {
          double pa, pb;
          philox_normal_pair(g.gk0, g.gk1, (uint64_t)(g.grow0 + i), (uint32_t)(k >> 1), g.gw, &pa, &pb);
          ek = (k & 1) ? pb : pa;
        } else {
          ek = noise[i * g.ld + k];
        }
        const double v = fma(sgk, ek, muk);
        av = wt * exp(-2.0 * v);
        const double gk = fma(-v, it2, -dm1);
        FK = wt * v * fma(-0.5 * v, it2, -dm1);
        GK = wt * gk;
        GEK = wt * gk * ek;
      }
      W = wt;
      d2* rs = reinterpret_cast<d2*>(wsb + ws.off_rowscal + 4 * i);
      rs[0] = (d2){av, ek};
      rs[1] = (d2){wt, 0.0};
    }
  } else if (i == 0) {
    W = (double)g.n;
  }
  // one combined block reduction (a single barrier)
  W = wave_sum(W);
  FK = wave_sum(FK);
  GK = wave_sum(GK);
  GEK = wave_sum(GEK);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sh[wave][PS_W] = W;
    sh[wave][PS_FK] = FK;
    sh[wave][PS_GK] = GK;
    sh[wave][PS_GEK] = GEK;
  }
  __syncthreads();
  if (threadIdx.x < PS_NUM)
    wsb[ws.off_prepscal + (int64_t)threadIdx.x * nblk + blk] =
        (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

__global__ void __launch_bounds__(256)
mf_prep_kernel(const BatchPtrs bp, const Workspace ws, const Geom g, const ModelDev model) {
  __shared__ double sh[4][PS_NUM];
  const int b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int d = g.d;
  double* wsb = ws.base + b * ws.stride;
  const double* theta_src = bp.theta_src[b];
  if (i < g.Dp) {
    double mu = 0.0, ls = 0.0;
    if (i < d) mu = theta_src[i], ls = theta_src[d + i];
    prep_column(i, mu, ls, wsb, ws, g, model);
  }
  // theta lives in pinned host memory: one lane fetches the coupling column's (mu, log sigma) across PCIe and
  // broadcasts them through LDS instead of every wave issuing its own host reads
  __shared__ double thk[2];
  thk[0] = thk[1] = 0.0;
  if (g.rows && model.id == VB_MODEL_FUNNEL) {
    __syncthreads();
    if (threadIdx.x < 2) thk[threadIdx.x] = theta_src[threadIdx.x * d + model.k];
    __syncthreads();
  }
  if (!g.inline_rows)
    prep_rows_block((int)blockIdx.x, (int)gridDim.x, wsb, ws, g, model, bp.noise[b], bp.roww[b], thk[0], thk[1], sh);
}

// ------------------------------------------------------------------------------------------------
// accumulate (the streaming kernel)
// ------------------------------------------------------------------------------------------------
struct ColAcc {
  double G = 0, GE = 0, E = 0, EE = 0, EK = 0, SC = 0, SCE = 0;
};

// one element: e = base noise, (c0, c1, c2) column constants, (av, ek, wt) row scalars
template <int MODEL, bool MOM, bool TSC, bool WEIGHTED>
__device__ __forceinline__ void accum(const double e, const double c0, const double c1,
                                      const double c2, const double av, const double ek,
                                      const double wt, const double df, ColAcc& A, double& F,
                                      double& Q, double& QE, double& L1P) {
  if (MODEL == kModelLogQ) {
    // DIS (objectives.py:405-414): weighted statistics of log q(z_n; theta) for fixed samples
    // z = mu_r + sigma_r e, i.e. of the standardised residual r = (z - mu)/sigma = c0 + c1 e
    const double r = fma(c1, e, c0);
    const double r2 = r * r;
    double sc = r, l = r2;
    if (TSC) {
      sc = (df + 1.0) * r / (df + r2);
      l = log1p(r2 / df);
    }
    const double wsc = wt * sc;
    A.E += wsc;
    A.EE = fma(wsc, r, A.EE);
    L1P = fma(wt, l, L1P);
    return;
  }
  if (MODEL == VB_MODEL_GAUSS_DIAG) {
    const double dz = fma(c1, e, c0);          // z - m
    double g = -dz * c2;
    if (WEIGHTED) g *= wt;
    A.G += g;
    A.GE = fma(g, e, A.GE);
    F = fma(0.5 * dz, g, F);
  } else {                                       // funnel, non-coupling columns (c0 = c1 = 0 on column k)
    const double z = fma(c1, e, c0);
    const double g = -z * av;                  // av = w_n exp(-2 v_n)
    const double q = -z * g;
    A.G += g;
    A.GE = fma(g, e, A.GE);
    Q += q;
    QE = fma(q, ek, QE);
  }
  if (MOM) {
    A.E += e;
    A.EE = fma(e, e, A.EE);
    if (MODEL == VB_MODEL_FUNNEL) A.EK = fma(e, ek, A.EK);
  }
  if (TSC) {
    const double e2 = e * e;
    const double sc = (df + 1.0) * e / (df + e2);
    A.SC += sc;
    A.SCE = fma(sc, e, A.SCE);
    L1P += log1p(e2 / df);
  }
}

// (row block, column block) of workgroup b
__device__ __forceinline__ void mf_wg_coords(const Geom& g, int b, int* rb, int* cb) {
  if (g.xcd_map) {   // all column blocks of a row block on one XCD (block b -> XCD b % 8)
    const int x = b & 7, j = b >> 3;
    *cb = j % g.n_cb;
    *rb = (j / g.n_cb) * 8 + x;
  } else {
    *rb = b / g.n_cb;
    *cb = b % g.n_cb;
  }
}

// What the one-launch evaluation (mf_one_kernel) hands to the streaming body instead of mf_prep_kernel's arrays: the
// lane's column constants and the funnel's coupling-column parameter, in registers.
struct OnePro {
  d2 cp0, cp1, cp2;
  double muk, lsk;
};

// ONE: the body of mf_one_kernel -- column constants from `pro`, partial sums stored write-through (another workgroup of
// the same launch reads them).
template <int MODEL, bool MOM, bool TSC, bool WEIGHTED, bool GEN, bool ONE>
__device__ __forceinline__ void mf_accum_body(const BatchPtrs& bp, const Workspace& ws, const Geom& g, const OnePro& pro) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bi = blockIdx.y;
  int rb, cb;
  mf_wg_coords(g, (int)blockIdx.x, &rb, &cb);
  constexpr bool ROWS = (MODEL == VB_MODEL_FUNNEL) || WEIGHTED;
  double* wsb = ws.base + bi * ws.stride;
  const double* __restrict__ colp = wsb + ws.off_colp;
  const double* __restrict__ rowscal = wsb + ws.off_rowscal;
  const double* __restrict__ noise_b = bp.noise[bi];
  const int c0i = cb * kMfCols + 2 * lane;
  const int64_t r0 = (int64_t)rb * g.rows_per_wg;
  const int64_t r1 = (r0 + g.rows_per_wg < g.n) ? r0 + g.rows_per_wg : g.n;
  const double* __restrict__ noise = noise_b + c0i;
  const int64_t ld = g.ld;
  const bool lane_ok = c0i < ld;   // columns [d, ld) are zero pads; beyond ld is another row
  const double df = g.df;

  ColAcc A0, A1;
  double F = 0.0, Q = 0.0, QE = 0.0, L1P = 0.0;
  // per-column constants: issued first (vmcnt is in-order and they are needed first)
  d2 cp0, cp1, cp2 = (d2){0.0, 0.0};
  if (ONE) {
    cp0 = pro.cp0, cp1 = pro.cp1, cp2 = pro.cp2;
  } else {
    cp0 = *reinterpret_cast<const d2*>(colp + c0i);
    cp1 = *reinterpret_cast<const d2*>(colp + g.Dp + c0i);
    if (MODEL == VB_MODEL_GAUSS_DIAG) cp2 = *reinterpret_cast<const d2*>(colp + 2 * (int64_t)g.Dp + c0i);
  }
  const bool cols_full = (cb + 1) * kMfCols <= ld;   // every lane's 16-B load stays inside the row

  __shared__ double psw[kMfWaves][PS_NUM];      // inline_rows: per-wave partial sums of the coupling-column terms
  if (GEN) {
    // noise in registers: the lane's two columns of row r are one Philox pair (the same counter layout as
    // rng_normal_kernel, so the values equal what the generator kernel would have stored); nothing is read
    const uint32_t jp = (uint32_t)(c0i >> 1);
    const bool ok0 = c0i < g.d, ok1 = c0i + 1 < g.d;
    auto row_pair = [&](int64_t r) __attribute__((always_inline)) {
      d2 ev = (d2){0.0, 0.0};
      if (ok0 && g.gen == 2) {
        const uint64_t grow = (uint64_t)(g.grow0 + r);
        ev.x = student_t_polar(g.gdf, grow, jp, g.gw, 0, g.gk0, g.gk1);
        if (ok1) ev.y = student_t_polar(g.gdf, grow, jp, g.gw, 1, g.gk0, g.gk1);
      } else if (ok0) {
        double pa, pb;
        philox_normal_pair(g.gk0, g.gk1, (uint64_t)(g.grow0 + r), jp, g.gw, &pa, &pb);
        ev = (d2){pa, ok1 ? pb : 0.0};
      }
      return ev;
    };
    // rows r and r + kMfWaves of a wave's walk are the two halves of one Philox quad when row r's global index has bit
    // 2 clear (vb_rng.h): both come out of one call -- 102 instead of 129 instructions per normal
    static_assert(kMfWaves == 4, "the quad pairing of vb_rng.h is rows g and g ^ 4");
    auto quad_ok = [&](int64_t r) __attribute__((always_inline)) {
      return g.gen == 1 && (((uint64_t)(g.grow0 + r)) & 4) == 0;
    };
    auto row_quad = [&](int64_t r, d2* e0, d2* e1) __attribute__((always_inline)) {
      double q[4] = {0.0, 0.0, 0.0, 0.0};
      if (ok0) philox_normal_quad(g.gk0, g.gk1, philox_quad_id((uint64_t)(g.grow0 + r)), jp, g.gw, q);
      *e0 = (d2){q[0], ok1 ? q[1] : 0.0};
      *e1 = (d2){q[2], ok1 ? q[3] : 0.0};
    };
    if (MODEL == VB_MODEL_FUNNEL && g.inline_rows) {
      // per pass of 256 rows (64 per wave): lane l first forms the row scalars of "its" row cbase + wave + 4 l --
      // the coupling column's draw from the same Philox counter the element kernel uses, v = mu_k + sigma_k e,
      // exp(-2 v) -- into LDS (each wave reads back only what it wrote itself), then the wave walks its 64 rows
      __shared__ double rsc[kMfWaves][kWave][2];
      const int k = g.fk, d = g.d;
      const double* th = wsb + ws.off_theta;
      const double muk = ONE ? pro.muk : th[k], sgk = exp(ONE ? pro.lsk : th[d + k]);
      const double it2 = 1.0 / (g.ftau * g.ftau), dm1 = (double)(d - 1);
      double pW = 0.0, pFK = 0.0, pGK = 0.0, pGEK = 0.0;
      for (int64_t cbase = r0; cbase < r1; cbase += (int64_t)kMfWaves * kWave) {
        const int64_t rl = cbase + wave + (int64_t)kMfWaves * lane;
        double a_l = 0.0, e_l = 0.0;
        if (rl < r1) {
          if (g.gen == 2) {
            e_l = student_t_polar(g.gdf, (uint64_t)(g.grow0 + rl), (uint32_t)(k >> 1), g.gw, (uint32_t)(k & 1), g.gk0,
                                  g.gk1);
          } else {
            double pa, pb;
            philox_normal_pair(g.gk0, g.gk1, (uint64_t)(g.grow0 + rl), (uint32_t)(k >> 1), g.gw, &pa, &pb);
            e_l = (k & 1) ? pb : pa;
          }
          const double v = fma(sgk, e_l, muk);
          a_l = exp(-2.0 * v);
          const double gk = fma(-v, it2, -dm1);
          pW += 1.0;
          pFK += v * fma(-0.5 * v, it2, -dm1);
          pGK += gk;
          pGEK = fma(gk, e_l, pGEK);
        }
        rsc[wave][lane][0] = a_l;
        rsc[wave][lane][1] = e_l;
        for (int i = 0; i < kWave;) {
          const int64_t r = cbase + wave + (int64_t)kMfWaves * i;
          if (r >= r1) break;   // wave-uniform
          if (quad_ok(r) && i + 1 < kWave && r + kMfWaves < r1) {      // (wave-uniform)
            d2 e0, e1;
            row_quad(r, &e0, &e1);
            const double a0 = rsc[wave][i][0], k0_ = rsc[wave][i][1], a1 = rsc[wave][i + 1][0], k1_ = rsc[wave][i + 1][1];
            accum<MODEL, MOM, TSC, WEIGHTED>(e0.x, cp0.x, cp1.x, cp2.x, a0, k0_, 1.0, df, A0, F, Q, QE, L1P);
            accum<MODEL, MOM, TSC, WEIGHTED>(e0.y, cp0.y, cp1.y, cp2.y, a0, k0_, 1.0, df, A1, F, Q, QE, L1P);
            accum<MODEL, MOM, TSC, WEIGHTED>(e1.x, cp0.x, cp1.x, cp2.x, a1, k1_, 1.0, df, A0, F, Q, QE, L1P);
            accum<MODEL, MOM, TSC, WEIGHTED>(e1.y, cp0.y, cp1.y, cp2.y, a1, k1_, 1.0, df, A1, F, Q, QE, L1P);
            i += 2;
            continue;
          }
          const d2 ev = row_pair(r);
          const double a_ = rsc[wave][i][0], k_ = rsc[wave][i][1];
          accum<MODEL, MOM, TSC, WEIGHTED>(ev.x, cp0.x, cp1.x, cp2.x, a_, k_, 1.0, df, A0, F, Q, QE, L1P);
          accum<MODEL, MOM, TSC, WEIGHTED>(ev.y, cp0.y, cp1.y, cp2.y, a_, k_, 1.0, df, A1, F, Q, QE, L1P);
          ++i;
        }
      }
      pW = wave_sum(pW);
      pFK = wave_sum(pFK);
      pGK = wave_sum(pGK);
      pGEK = wave_sum(pGEK);
      if (lane == 0) {
        psw[wave][PS_W] = pW;
        psw[wave][PS_FK] = pFK;
        psw[wave][PS_GK] = pGK;
        psw[wave][PS_GEK] = pGEK;
      }
    } else {
      auto one_row = [&](int64_t r, const d2 ev) __attribute__((always_inline)) {
        double a_ = 1.0, k_ = 0.0, w_ = 1.0;
        if (MODEL == VB_MODEL_FUNNEL) {
          a_ = rowscal[4 * r];
          k_ = rowscal[4 * r + 1];
        } else if (WEIGHTED) {
          w_ = rowscal[4 * r + 2];
        }
        accum<MODEL, MOM, TSC, WEIGHTED>(ev.x, cp0.x, cp1.x, cp2.x, a_, k_, w_, df, A0, F, Q, QE, L1P);
        accum<MODEL, MOM, TSC, WEIGHTED>(ev.y, cp0.y, cp1.y, cp2.y, a_, k_, w_, df, A1, F, Q, QE, L1P);
      };
      for (int64_t r = r0 + wave; r < r1;) {
        if (quad_ok(r) && r + kMfWaves < r1) {      // (wave-uniform)
          d2 e0, e1;
          row_quad(r, &e0, &e1);
          one_row(r, e0);
          one_row(r + kMfWaves, e1);
          r += 2 * kMfWaves;
        } else {
          one_row(r, row_pair(r));
          r += kMfWaves;
        }
      }
    }
  }
  for (int64_t base = r0 + wave; !GEN && base < r1; base += (int64_t)kMfWaves * kMfChunk) {
    d2 e[kMfChunk];
    double av[kMfChunk], ek[kMfChunk], wt[kMfChunk];
    if (cols_full && base + (int64_t)kMfWaves * (kMfChunk - 1) < r1) {
      // ---- full chunk: straight-line code, 16 coalesced 1-KiB row segments in flight -----------
      // wave-uniform row base (SGPRs) + per-lane 32-bit byte offset: global_load saddr form
      const char* rowp = reinterpret_cast<const char*>(noise_b + base * ld + cb * kMfCols);
      const int64_t step = (int64_t)kMfWaves * ld * (int64_t)sizeof(double);
      const unsigned voff = (unsigned)lane * 16u;
#pragma unroll
      for (int j = 0; j < kMfChunk; ++j) {
        e[j] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(rowp + voff));
        rowp += step;
      }
      __builtin_amdgcn_sched_barrier(0);   // all 16 row loads are issued before anything below
      // per-row scalars: wave-uniform addresses => scalar loads, issued per half (SGPR budget)
#pragma unroll
      for (int h = 0; h < kMfChunk; h += kMfChunk / 2) {
        if (ROWS) {
#pragma unroll
          for (int j = h; j < h + kMfChunk / 2; ++j) {
            const double* rs = rowscal + 4 * (base + (int64_t)kMfWaves * j);
            if (MODEL == VB_MODEL_FUNNEL) {
              av[j] = rs[0];
              ek[j] = rs[1];
            } else {
              wt[j] = rs[2];
            }
          }
        }
#pragma unroll
        for (int j = h; j < h + kMfChunk / 2; ++j) {
          const double a_ = MODEL == VB_MODEL_FUNNEL ? av[j] : 1.0;
          const double k_ = MODEL == VB_MODEL_FUNNEL ? ek[j] : 0.0;
          const double w_ = (MODEL != VB_MODEL_FUNNEL && WEIGHTED) ? wt[j] : 1.0;
          accum<MODEL, MOM, TSC, WEIGHTED>(e[j].x, cp0.x, cp1.x, cp2.x, a_, k_, w_, df, A0, F, Q, QE, L1P);
          accum<MODEL, MOM, TSC, WEIGHTED>(e[j].y, cp0.y, cp1.y, cp2.y, a_, k_, w_, df, A1, F, Q, QE, L1P);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // ---- ragged chunk (last rows / last columns): one row at a time, predicated -------------
#pragma unroll 1
      for (int j = 0; j < kMfChunk; ++j) {
        const int64_t r = base + (int64_t)kMfWaves * j;
        if (r >= r1) break;   // wave-uniform
        d2 ev = (d2){0.0, 0.0};
        if (lane_ok) ev = __builtin_nontemporal_load(reinterpret_cast<const d2*>(noise + r * ld));
        double a_ = 1.0, k_ = 0.0, w_ = 1.0;
        if (MODEL == VB_MODEL_FUNNEL) {
          a_ = rowscal[4 * r];
          k_ = rowscal[4 * r + 1];
        } else if (WEIGHTED) {
          w_ = rowscal[4 * r + 2];
        }
        accum<MODEL, MOM, TSC, WEIGHTED>(ev.x, cp0.x, cp1.x, cp2.x, a_, k_, w_, df, A0, F, Q, QE, L1P);
        accum<MODEL, MOM, TSC, WEIGHTED>(ev.y, cp0.y, cp1.y, cp2.y, a_, k_, w_, df, A1, F, Q, QE, L1P);
      }
    }
  }

  // ---- combine the 4 waves through LDS, one partial per column per workgroup ---------------
  constexpr int NF = TSC ? CF_NUM : (MOM ? CF_EK + 1 : CF_GE + 1);
  __shared__ d2 red[NF][kMfWaves][kWave];
  __shared__ double reds[kMfWaves][KS_NUM];
  red[CF_G][wave][lane] = (d2){A0.G, A1.G};
  red[CF_GE][wave][lane] = (d2){A0.GE, A1.GE};
  if (NF > CF_E) {
    red[CF_E][wave][lane] = (d2){A0.E, A1.E};
    red[CF_EE][wave][lane] = (d2){A0.EE, A1.EE};
    red[CF_EK][wave][lane] = (d2){A0.EK, A1.EK};
  }
  if (NF > CF_SC) {
    red[CF_SC][wave][lane] = (d2){A0.SC, A1.SC};
    red[CF_SCE][wave][lane] = (d2){A0.SCE, A1.SCE};
  }
  {
    const double f = wave_sum(F), q = wave_sum(Q), qe = wave_sum(QE), l = wave_sum(L1P),
                 ee = wave_sum(A0.EE + A1.EE);
    if (lane == 0) {   // wave_sum leaves the total in lane 0
      reds[wave][KS_F] = f;
      reds[wave][KS_Q] = q;
      reds[wave][KS_QE] = qe;
      reds[wave][KS_L1P] = l;
      reds[wave][KS_EE] = ee;
    }
  }
  __syncthreads();
  for (int f = wave; f < NF; f += kMfWaves) {
    d2 s = red[f][0][lane];
#pragma unroll
    for (int w = 1; w < kMfWaves; ++w) s += red[f][w][lane];
    double* dst = wsb + ws.off_partials + partial_index(rb, f, c0i, g.n_rb);
    if (ONE) {
      cst<true>(dst, s.x);
      cst<true>(dst + 1, s.y);
    } else {
      *reinterpret_cast<d2*>(dst) = s;
    }
  }
  if (threadIdx.x < KS_NUM) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kMfWaves; ++w) s += reds[w][threadIdx.x];
    cst<ONE>(wsb + ws.off_pscal + (int64_t)threadIdx.x * (g.n_rb * g.n_cb) + (int64_t)rb * g.n_cb + cb, s);
  }
  if (GEN && MODEL == VB_MODEL_FUNNEL && g.inline_rows && cb == 0 && threadIdx.x < PS_NUM)
    cst<ONE>(wsb + ws.off_prepscal + (int64_t)threadIdx.x * g.n_prep + rb,
             (psw[0][threadIdx.x] + psw[1][threadIdx.x]) + (psw[2][threadIdx.x] + psw[3][threadIdx.x]));
}

template <int MODEL, bool MOM, bool TSC, bool WEIGHTED, bool GEN = false>
__global__ void __launch_bounds__(kMfThreads)
mf_accum_kernel(const BatchPtrs bp, const Workspace ws, const Geom g) {
  mf_accum_body<MODEL, MOM, TSC, WEIGHTED, GEN, false>(bp, ws, g, OnePro());
}

// ------------------------------------------------------------------------------------------------
// finalize / epilogue
// ------------------------------------------------------------------------------------------------
struct EpiArgs {
  const double* partials;   // [Dp / 64][n_rb][CF_NUM][64] (partial_index)
  const double* pscal;      // [KS_NUM][n_ps]
  const double* prepscal;   // [PS_NUM][n_prep] or nullptr
  int n_rb, n_ps, n_prep;
  int nf;                   // column fields present
  double* sums;             // reduce-only output: [SF_NUM] | [nf][Dp]
  const double* theta;      // device [mu | log_sigma]
  double* out;              // [value | grad(2D)], device-visible (pinned host or device)
  int d, Dp;
  double n_total;
  int family;
  double df;
  unsigned flags;
  int cv_mode;
  int mode;                 // 0: ELBO; 1: weighted gradient only
  double scale;             // mode 1: grad = scale * [G | GE*sigma + W]
  const double* value_src;  // mode 1: device scalar reported as the value (may be nullptr)
  int reduce_only;
  ModelDev model;
  int has_step;             // device-resident fit: apply the optimiser step to the columns finished here
  FitStep step;
  int prep_next;            // ... and prepare the next iteration (mf_prep_kernel's work) in its workspace set
  double* next_wsb;
  Geom next_g;              // geometry of the next iteration: the same shapes, the next Philox stream
  unsigned long long* done; // blocking call: done[8 * workgroup] = done_seq behind this workgroup's results (or nullptr)
  unsigned long long done_seq;
};

struct Totals {
  double v[SF_NUM];
};

// Sum of the scalar partials of the accumulate and prep kernels, plus sum(log sigma), over the 256 threads of a
// finalize workgroup.  Three phases so that the kernel is ONE memory round trip deep: scalar_fetch issues the first
// 256-entry chunk of every list into registers (no use => no wait; the caller issues its own column-partial loads
// right behind), scalar_accumulate adds them up (and walks the rest of longer lists), scalar_reduce combines the
// threads through LDS.  kNS values per thread: the KS_* and PS_* sums and the log-sigma sum.
constexpr int kNS = KS_NUM + PS_NUM + 1;

// x where `mask` is all ones, +0.0 where it is zero.  The mask comes from opaque_mask so that the compiler cannot
// turn "load, then select" into "branch around the load": a load sunk to its use is one more memory round trip
// (0.7 us here) in a kernel that is nothing but a chain of them.
__device__ __forceinline__ double keep_masked(double x, unsigned long long mask) {
  return __longlong_as_double(__double_as_longlong(x) & (long long)mask);
}
__device__ __forceinline__ unsigned long long opaque_mask(bool c) {
  unsigned long long m = c ? ~0ull : 0ull;
  asm volatile("" : "+v"(m));
  return m;
}
constexpr int kLsChunks = 4;        // log-sigma entries fetched ahead per thread (d <= 1024 in one go)
constexpr int kPsChunks = 2;        // accumulate-kernel scalar partials per thread (512 workgroups = 2 per CU)

struct ScalarFetch {
  double ks[kPsChunks][KS_NUM], ps[PS_NUM], ls[kLsChunks];
};

template <bool COH = false>
__device__ __forceinline__ void scalar_fetch(const EpiArgs& a, bool want_ls, ScalarFetch* f) {
  // unconditional loads from clamped indices, discarded by selects: straight-line code (see mf_finalize_kernel)
  const int e = threadIdx.x;
  const int e_pr = min(e, a.n_prep - 1);
#pragma unroll
  for (int j = 0; j < kPsChunks; ++j) {
    const int e_ps = min(e + 256 * j, a.n_ps - 1);
#pragma unroll
    for (int s = 0; s < KS_NUM; ++s) f->ks[j][s] = cld<COH>(a.pscal + (int64_t)s * a.n_ps + e_ps);
  }
#pragma unroll
  for (int s = 0; s < PS_NUM; ++s) f->ps[s] = 0.0;
#pragma unroll
  for (int j = 0; j < kLsChunks; ++j) f->ls[j] = 0.0;
  if (a.prepscal) {
#pragma unroll
    for (int s = 0; s < PS_NUM; ++s) f->ps[s] = cld<COH>(a.prepscal + (int64_t)s * a.n_prep + e_pr);
  }
  if (want_ls) {
#pragma unroll
    for (int j = 0; j < kLsChunks; ++j) f->ls[j] = cld<COH>(a.theta + a.d + min(e + 256 * j, a.d - 1));
  }
}

// acc[0 .. KS_NUM + PS_NUM) = this thread's share of the scalar sums, acc[kNS - 1] = its share of sum(log sigma);
// entry order e = t, t + 256, ... as a plain strided loop would take them
template <bool COH = false>
__device__ __forceinline__ void scalar_accumulate(const EpiArgs& a, bool want_ls, const ScalarFetch& f, double* acc) {
  // (the entries scalar_fetch took from clamped indices are dropped HERE, at their first use, not next to the loads:
  // a use is a wait, and the caller has more loads to issue in between)
  const int e = threadIdx.x;
  const unsigned long long m_pr = opaque_mask(a.prepscal && e < a.n_prep);
#pragma unroll
  for (int s = 0; s < KS_NUM; ++s) acc[s] = 0.0;
#pragma unroll
  for (int j = 0; j < kPsChunks; ++j) {
    const unsigned long long m_ps = opaque_mask(e + 256 * j < a.n_ps);
#pragma unroll
    for (int s = 0; s < KS_NUM; ++s) acc[s] += keep_masked(f.ks[j][s], m_ps);
  }
#pragma unroll
  for (int s = 0; s < PS_NUM; ++s) acc[KS_NUM + s] = 0.0 + keep_masked(f.ps[s], m_pr);
  double t = 0.0;
#pragma unroll
  for (int j = 0; j < kLsChunks; ++j) t += keep_masked(f.ls[j], opaque_mask(want_ls && e + 256 * j < a.d));
  for (int e = threadIdx.x + 256 * kPsChunks; e < a.n_ps; e += 256) {
#pragma unroll
    for (int s = 0; s < KS_NUM; ++s) acc[s] += cld<COH>(a.pscal + (int64_t)s * a.n_ps + e);
  }
  if (a.prepscal) {
    for (int e = threadIdx.x + 256; e < a.n_prep; e += 256) {
#pragma unroll
      for (int s = 0; s < PS_NUM; ++s) acc[KS_NUM + s] += cld<COH>(a.prepscal + (int64_t)s * a.n_prep + e);
    }
  }
  if (want_ls)
    for (int i = threadIdx.x + 256 * kLsChunks; i < a.d; i += 256) t += cld<COH>(a.theta + a.d + i);
  acc[kNS - 1] = t;
}

// LDS scratch of scalar_reduce: the per-thread values, 16 partial sums of each, the totals
static_assert(16 * kNS <= 256, "scalar_reduce: one thread per (value, 16-entry group)");
struct ScalarShared {
  double v[kNS][256];
  double p[kNS][16];
  double t[kNS];
};

// Block-wide sums of acc[0 .. kNS): every thread stores its values (the caller's own LDS stores ride on the same
// barrier), 16 x kNS threads add 16 entries each, kNS threads add the 16 partials.  Three barriers and ~40 LDS reads
// on the critical path instead of 6 x kNS cross-lane shuffles (1.6 us -> 0.4 us).  *sum_ls gets sum(log sigma).
__device__ __forceinline__ Totals scalar_reduce(const double* acc, ScalarShared* sh, double* sum_ls) {
  const int t = threadIdx.x;
#pragma unroll
  for (int s = 0; s < kNS; ++s) sh->v[s][t] = acc[s];
  __syncthreads();
  if (t < 16 * kNS) {
    const int s = t >> 4, j = t & 15;
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = sh->v[s][j + 16 * i];
    double u = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) u += x[i];
    sh->p[s][j] = u;
  }
  __syncthreads();
  if (t < kNS) {
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = sh->p[t][i];
    double u = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) u += x[i];
    sh->t[t] = u;
  }
  __syncthreads();
  Totals r;
#pragma unroll
  for (int s = 0; s < SF_NUM; ++s) r.v[s] = 0.0;
  r.v[SF_F] = sh->t[KS_F];
  r.v[SF_Q] = sh->t[KS_Q];
  r.v[SF_QE] = sh->t[KS_QE];
  r.v[SF_L1P] = sh->t[KS_L1P];
  r.v[SF_EE] = sh->t[KS_EE];
  r.v[SF_W] = sh->t[KS_NUM + PS_W];
  r.v[SF_FK] = sh->t[KS_NUM + PS_FK];
  r.v[SF_GK] = sh->t[KS_NUM + PS_GK];
  r.v[SF_GEK] = sh->t[KS_NUM + PS_GEK];
  *sum_ls = sh->t[kNS - 1];
  return r;
}

// log of the Student-t density's normaliser
__device__ __forceinline__ double student_log_norm(double df) {
  return lgamma(0.5 * (df + 1.0)) - lgamma(0.5 * df) - 0.5 * log(df * M_PI);
}

// -(lower bound), objectives.py:156-164
__device__ double elbo_value(const EpiArgs& a, const Totals& t, double sum_ls) {
  const int d = a.d;
  const bool student = a.family == VB_FAMILY_MF_STUDENT_T;
  const bool funnel = a.model.id == VB_MODEL_FUNNEL;
  const double W = t.v[SF_W];
  const double invN = 1.0 / a.n_total;
  double F = t.v[SF_F] + W * a.model.c0;
  if (funnel) F += t.v[SF_FK] - 0.5 * t.v[SF_Q];
  if (a.flags & VB_FLAG_PATH_DERIV) {   // -mean(f - log q(theta_stop; z))
    double logq;
    if (student) {
      const double ct = student_log_norm(a.df);
      logq = W * (d * ct - sum_ls) - 0.5 * (a.df + 1.0) * t.v[SF_L1P];
    } else {
      logq = -0.5 * t.v[SF_EE] - W * (0.5 * d * kLog2Pi + sum_ls);
    }
    return -(F - logq) * invN;
  }
  const double H = (student ? 0.0 : 0.5 * d * (1.0 + kLog2Pi)) + sum_ls;   // -(mean f + entropy)
  return -(F * invN + H);
}

// gradient of the plain / path-derivative estimator for one column (what autograd returns for
// objectives.py:154-164)
__device__ __forceinline__ void plain_column(const EpiArgs& a, int i, double g, double ge, double sc,
                                             double sce, double* gmu, double* gls, double ls_i,
                                             double* gmu_i = nullptr, double* gls_i = nullptr) {
  const double sg = exp(ls_i);            // ls_i = a.theta[a.d + i], fetched by the caller ahead of its reduction
  const double invN = 1.0 / a.n_total;
  double m, l;
  if (a.flags & VB_FLAG_PATH_DERIV) {
    m = -(g + sc / sg) * invN;
    l = -(ge * sg + sce) * invN;
  } else {
    m = -g * invN;
    l = -(ge * sg * invN + 1.0);
  }
  gmu[i] = m;
  gls[i] = l;
  if (gmu_i) *gmu_i = m, *gls_i = l;      // for a caller that goes on with them (no store -> load round trip)
}

// Fixed-order sums of the row-block partials of NF fields for column c of the workgroup's run, row blocks
// q, q + 4, q + 8, ... added in that order, into colsum[f][q][c].  NI row blocks of every field are requested at a
// time (NF x NI loads in flight: one memory round trip per pass, and the launcher's geometry -- at most 64 row
// blocks up to 2 workgroups per CU -- makes that one pass for NI = 16); row blocks past the end are read from the
// last one (scalar clamps) and dropped by a mask afterwards.
template <int NF, int NI, bool COH = false>
__device__ __forceinline__ void column_sums(const double* blk_partials, int q, int c, int n_rb,
                                            double (*colsum)[4][64]) {
  const int rb_last = n_rb - 1;
  double t[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) t[f] = 0.0;
  for (int rb0 = q; rb0 <= rb_last; rb0 += 4 * NI) {
    double v[NF][NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const double* prb = blk_partials + min(rb0 + 4 * i, rb_last) * (CF_NUM * 64) + (uint32_t)c;
#pragma unroll
      for (int f = 0; f < NF; ++f) v[f][i] = cld<COH>(prb + f * 64);
    }
    // ---- everything below waits on those loads (and on everything the caller has requested before them) ----
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const unsigned long long m = opaque_mask(rb0 + 4 * i <= rb_last);
#pragma unroll
      for (int f = 0; f < NF; ++f) t[f] += keep_masked(v[f][i], m);
    }
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) colsum[f][q][c] = t[f];
}

// one instantiation per field count; 4 row blocks per thread cover up to 16 (small sample counts), 16 cover the
// launcher's usual 64, 8 at a time for the wide field sets (registers)
template <int NF, bool COH = false>
__device__ __forceinline__ void column_sums_nf(const double* blk_partials, int q, int c, int n_rb,
                                               double (*colsum)[4][64]) {
  if (n_rb > 16)
    column_sums<NF, (NF <= 2 ? 16 : 8), COH>(blk_partials, q, c, n_rb, colsum);
  else
    column_sums<NF, 4, COH>(blk_partials, q, c, n_rb, colsum);
}

// reduce-only mode (the multi-rank path: the sums go to the all-reduce, mf_epilogue_kernel does the rest)
__device__ __forceinline__ void finalize_reduce_only(const EpiArgs& a, const Totals& tot,
                                                               double (*colsum)[4][64], ScalarShared* ssh, int q,
                                                               int c, int col) {
  if (q == 0) {
    for (int f = 0; f < a.nf; ++f)
      a.sums[SF_NUM + (int64_t)f * a.Dp + col] =
          (colsum[f][0][c] + colsum[f][1][c]) + (colsum[f][2][c] + colsum[f][3][c]);
  }
  if (blockIdx.x == 0) {
    // Totals is a register struct: hand it to the SF_NUM writing lanes through LDS, not through a dynamic index
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int s = 0; s < SF_NUM; ++s) ssh->p[0][s] = tot.v[s];
    }
    __syncthreads();
    if (threadIdx.x < SF_NUM) a.sums[threadIdx.x] = ssh->p[0][threadIdx.x];
  }
}

// Row part of the next iteration's prep, block by block in the prep kernel's own order so that the partial sums are
// the same numbers (all 256 threads; thk_next = the coupling column's new (mu, log sigma), published by its owner).
__device__ __forceinline__ void finalize_prep_rows(const EpiArgs& a, const Workspace& ws,
                                                             const double* thk_next, double (*sh3)[PS_NUM]) {
  __syncthreads();
  const double muk = a.next_g.rows ? thk_next[0] : 0.0, lsk = a.next_g.rows ? thk_next[1] : 0.0;
  for (int blk = 0; blk < a.n_prep; ++blk) {
    prep_rows_block(blk, a.n_prep, a.next_wsb, ws, a.next_g, a.model, nullptr, nullptr, muk, lsk, sh3);
    __syncthreads();
  }
}

// grid = Dp / 64 workgroups of 256 threads: thread (c = t & 63, q = t >> 6) sums row blocks
// rb = q, q + 4, ... of column 64 * blockIdx + c, for every field; fixed order => deterministic.
__global__ void __launch_bounds__(256)
mf_finalize_kernel(const EpiArgs a_in, const BatchPtrs bp, const Workspace ws) {
  kernarg_warm<sizeof(EpiArgs) + sizeof(BatchPtrs) + sizeof(Workspace)>();
  EpiArgs a = a_in;   // per-evaluation pointers (blockIdx.y = evaluation index)
  {
    const int b = blockIdx.y;
    double* wsb = ws.base + b * ws.stride;
    a.partials = wsb + ws.off_partials;
    a.pscal = wsb + ws.off_pscal;
    a.prepscal = wsb + ws.off_prepscal;
    a.theta = wsb + ws.off_theta;
    a.sums = ws.sums + b * ws.sum_len;
    a.out = bp.out[b];
  }
  __shared__ double colsum[CF_NUM + 1][4][64];
  __shared__ ScalarShared ssh;
  __shared__ double sh3[4][PS_NUM];
  __shared__ double thk_next[2];
  // q (the wave index) in a scalar register: the row-block offsets and their bounds checks become scalar work
  const int c = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = blockIdx.x * 64 + c;
#ifdef VB_FIN_CLOCK
  // Debug build (tools/build_clk.sh): three s_memrealtime stamps per launch -- entry, exit and ONE midpoint that
  // moves with the iteration number (1500 + i) -- printed for two workgroups.
  const int fc_mid = a.has_step ? (int)a.step.k - 1500 : -1;
  long long fc_t0 = wall_clock64(), fc_tm = 0;
#define VB_FC(i) if (fc_mid == (i)) fc_tm = wall_clock64()
#else
#define VB_FC(i)
#endif
  // This kernel is 16 workgroups of pure latency (profiled with the stamps above, tools/latency_probe.hip and
  // tools/kernarg_probe.hip).  What it costs: (a) dependent global round trips, ~0.7 us each for data another XCD
  // has just written -- so EVERYTHING it reads is requested up front, before anything is used: what the per-column
  // epilogue needs (log sigma of the column and, in the device-resident loop, the optimiser state and the parameter
  // entries of its two steps), the scalar partials, the log-sigma entries of the value, and every row block of
  // every column field in one pass; values read from clamped indices are dropped with opaque masks, not selects,
  // so that the compiler cannot sink a load to its use; (b) the first touch of each 64-byte line of its 1.6 KB
  // argument block, 0.3-0.4 us each when met one by one (kernarg_warm); (c) scratch memory and real function
  // calls: a build with the rare paths in noinline functions (632 B of stack) ran every phase 1.5x slower, so the
  // rare paths (reduce-only mode of the multi-rank path, the row part of the next iteration's prep, the Student-t
  // normaliser) are inlined behind unlikely branches and nothing indexes a register array dynamically.
  double pf_ls = 0.0, pf_s1[2] = {0.0, 0.0}, pf_s2[2] = {0.0, 0.0}, pf_th[2] = {0.0, 0.0};
  if (!a.reduce_only && q == 0 && col < a.d) {
    pf_ls = a.theta[a.d + col];
    if (a.has_step) {
      fit_step_load(a.step, col, &pf_s1[0], &pf_s2[0], &pf_th[0]);
      fit_step_load(a.step, (int64_t)a.d + col, &pf_s1[1], &pf_s2[1], &pf_th[1]);
    }
  }
  VB_FC(10);
  const bool funnel = a.model.id == VB_MODEL_FUNNEL;
  const int k = a.model.k;
  const bool owns_k = funnel && (k / 64 == (int)blockIdx.x);
  const bool need_tot = blockIdx.x == 0 || owns_k;   // workgroup-uniform
  const bool want_ls = blockIdx.x == 0 && !a.reduce_only;
  ScalarFetch sf;
  if (need_tot) scalar_fetch(a, want_ls, &sf);
  VB_FC(11);
  const double* blk_partials = a.partials + (int64_t)blockIdx.x * a.n_rb * (CF_NUM * 64);   // this workgroup's run
  // CF_GE + 1, CF_EK + 1 or CF_NUM fields (see the launcher); the plain ELBO gradient (two fields) falls through
  if (__builtin_expect(a.nf == CF_GE + 1, 1))
    column_sums_nf<CF_GE + 1>(blk_partials, q, c, a.n_rb, colsum);
  else if (a.nf == CF_EK + 1)
    column_sums_nf<CF_EK + 1>(blk_partials, q, c, a.n_rb, colsum);
  else
    column_sums_nf<CF_NUM>(blk_partials, q, c, a.n_rb, colsum);
  VB_FC(1);
  Totals tot;
  double sum_ls = 0.0;
  if (need_tot) {
    double sacc[kNS];
    scalar_accumulate(a, want_ls, sf, sacc);
    VB_FC(2);
    tot = scalar_reduce(sacc, &ssh, &sum_ls);     // its first barrier also publishes colsum
  } else {
    VB_FC(2);
    __syncthreads();
  }
  VB_FC(3);

  if (__builtin_expect(a.reduce_only, 0)) {
    finalize_reduce_only(a, tot, colsum, &ssh, q, c, col);
    return;
  }

  // ---- fused epilogue (no cross-column coupling): plain and path-derivative ELBO gradients ----
  const int d = a.d;
  double* value = a.out;
  double* gmu = a.out + 1;
  double* gls = a.out + 1 + d;
  double new_mu = 0.0, new_ls = 0.0;
  if (q == 0 && col < d) {
    // fields beyond a.nf were not written by column_sums: whatever LDS holds there is read and dropped
    double S[CF_NUM];
#pragma unroll
    for (int f = 0; f < CF_NUM; ++f) {
      const double x = (colsum[f][0][c] + colsum[f][1][c]) + (colsum[f][2][c] + colsum[f][3][c]);
      S[f] = f < a.nf ? x : 0.0;
    }
    double g = S[CF_G], ge = S[CF_GE];
    if (funnel && col == k) {
      g += tot.v[SF_GK] + tot.v[SF_Q];
      ge += tot.v[SF_GEK] + tot.v[SF_QE];
    }
    const bool student = a.family == VB_FAMILY_MF_STUDENT_T;
    double gm, gl;
    plain_column(a, col, g, ge, student ? S[CF_SC] : S[CF_E], student ? S[CF_SCE] : S[CF_EE], gmu, gls, pf_ls,
                 &gm, &gl);
    if (a.has_step) {       // theta_src (the step's theta) is not read by this kernel: it works on the prep copy
      new_mu = fit_step_apply_vals(a.step, col, gm, pf_s1[0], pf_s2[0], pf_th[0]);
      new_ls = fit_step_apply_vals(a.step, (int64_t)d + col, gl, pf_s1[1], pf_s2[1], pf_th[1]);
    }
  }
  VB_FC(4);
  if (a.prep_next) {
    // prep of the next iteration (see mf_prep_kernel) for the columns this workgroup has just stepped ...
    if (q == 0) {
      double mu = 0.0, ls = 0.0;
      if (col < d) {
        mu = new_mu;          // = a.step.theta[col], a.step.theta[d + col] as just stored
        ls = new_ls;
        if (funnel && col == k) thk_next[0] = mu, thk_next[1] = ls;
      }
      prep_column(col, mu, ls, a.next_wsb, ws, a.next_g, a.model);
    }
    // ... and its row part: by the workgroup that owns the funnel's coupling column (the per-row scalars need
    // that column's new parameter only); without row scalars workgroup 0 writes the constant partials.  Not needed
    // when the accumulate kernel computes the row scalars itself (512 samples or more).
    if (__builtin_expect(!a.next_g.inline_rows && (a.next_g.rows ? owns_k : blockIdx.x == 0), 0))
      finalize_prep_rows(a, ws, thk_next, sh3);
  }
  VB_FC(5);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const double val = elbo_value(a, tot, sum_ls);
    value[0] = val;
    if (a.has_step) a.step.values[a.step.k] = val;
  }
  if (a.done) {
    // the blocking caller polls these words instead of synchronising the stream: every thread's result stores are
    // released to the host before the workgroup's word carries the call's sequence number
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_store(a.done + 8 * blockIdx.x, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
#ifdef VB_FIN_CLOCK
  {
    const long long fc_t1 = wall_clock64();
    if (threadIdx.x == 0 && fc_mid >= 0 && fc_mid < 16 && (blockIdx.x == 0 || blockIdx.x == 5))
      printf("finalize block %d  midpoint %2d: entry->mid %4lld  mid->exit %4lld  total %4lld  (10 ns ticks)\n", (int)blockIdx.x,
             fc_mid, fc_tm ? fc_tm - fc_t0 : -1, fc_tm ? fc_t1 - fc_tm : -1, fc_t1 - fc_t0);
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// the whole evaluation in ONE launch (north_star: "... fused into one launch")
// ------------------------------------------------------------------------------------------------
// mf_one_kernel = prep + accumulate + finalize of a single plain ELBO evaluation on one rank:
//   prologue   every workgroup forms what mf_prep_kernel would have left it in memory: the column constants of its 128
//              columns in registers, and (funnel, noise in memory) the row scalars of its own rows -- written to the
//              row-scalar array the streaming loop loads through the scalar cache, by every column block of the row
//              block alike (identical values: no hand-off, no waiting).  theta is read from device memory when it
//              lives there (the device-resident fit loop); a parameter staged in pinned host memory is fetched across
//              PCIe ONCE per column block, by the row-block-0 workgroup, and published through a device copy and an
//              epoch flag the other row blocks poll (bounded; a give-up reads the host copy itself).
//   body       mf_accum_body, unchanged arithmetic; its partial sums leave write-through.
//   tail       tickets instead of a launch boundary: the last workgroup of a column block to arrive sums that
//              block's row-block partials in mf_finalize_kernel's order and writes the gradient (and applies the
//              optimiser step) for its 128 columns; the last workgroup of all adds up the scalar partials, writes
//              the value and finishes the funnel's coupling column, whose gradient needs those totals.
struct OneArgs {
  unsigned* cnt;            // [0]: workgroups that are through; [1 + cb]: accumulate passes of column block cb finished
  unsigned* flag;           // [cb]: epoch of the last publication of column block cb's theta
  double* kbuf;             // the coupling column's sums, parked for the last workgroup: [CF_NUM | ls | s1 s2 theta (x2)]
  unsigned epoch;
  int theta_on_device;      // theta_src is device memory: no publication needed
};

// finalize_kernel's column part for the 64 columns of group `grp` (see mf_finalize_kernel: same sums, same order)
template <bool MOMF, bool TSCF>
__device__ __forceinline__ void one_tail_group(const EpiArgs& a, const OneArgs& o, int grp, double (*colsum)[4][64]) {
  constexpr int NF = TSCF ? CF_NUM : (MOMF ? CF_EK + 1 : CF_GE + 1);
  const int c = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = grp * 64 + c, d = a.d;
  const bool funnel = a.model.id == VB_MODEL_FUNNEL;
  const int k = a.model.k;
  double pf_ls = 0.0, pf_s1[2] = {0.0, 0.0}, pf_s2[2] = {0.0, 0.0}, pf_th[2] = {0.0, 0.0};
  if (q == 0 && col < d) {
    pf_ls = cld<true>(a.theta + d + col);
    if (a.has_step) {
      fit_step_load(a.step, col, &pf_s1[0], &pf_s2[0], &pf_th[0]);
      fit_step_load(a.step, (int64_t)d + col, &pf_s1[1], &pf_s2[1], &pf_th[1]);
    }
  }
  const double* blk_partials = a.partials + (int64_t)grp * a.n_rb * (CF_NUM * 64);
  column_sums_nf<NF, true>(blk_partials, q, c, a.n_rb, colsum);
  __syncthreads();
  if (q == 0 && col < d) {
    double S[CF_NUM];
#pragma unroll
    for (int f = 0; f < CF_NUM; ++f) {
      const double x = (colsum[f][0][c] + colsum[f][1][c]) + (colsum[f][2][c] + colsum[f][3][c]);
      S[f] = f < NF ? x : 0.0;
    }
    if (funnel && col == k) {      // needs the scalar totals: parked for the last workgroup
#pragma unroll
      for (int f = 0; f < CF_NUM; ++f) cst<true>(o.kbuf + f, S[f]);
      cst<true>(o.kbuf + CF_NUM, pf_ls);
      cst<true>(o.kbuf + CF_NUM + 1, pf_s1[0]), cst<true>(o.kbuf + CF_NUM + 2, pf_s2[0]), cst<true>(o.kbuf + CF_NUM + 3, pf_th[0]);
      cst<true>(o.kbuf + CF_NUM + 4, pf_s1[1]), cst<true>(o.kbuf + CF_NUM + 5, pf_s2[1]), cst<true>(o.kbuf + CF_NUM + 6, pf_th[1]);
    } else {
      const bool student = a.family == VB_FAMILY_MF_STUDENT_T;
      double gm, gl;
      plain_column(a, col, S[CF_G], S[CF_GE], student ? S[CF_SC] : S[CF_E], student ? S[CF_SCE] : S[CF_EE], a.out + 1,
                   a.out + 1 + d, pf_ls, &gm, &gl);
      if (a.has_step) {
        fit_step_apply_vals(a.step, col, gm, pf_s1[0], pf_s2[0], pf_th[0]);
        fit_step_apply_vals(a.step, (int64_t)d + col, gl, pf_s1[1], pf_s2[1], pf_th[1]);
      }
    }
  }
  __syncthreads();      // colsum is reused by the next group
}

template <int MODEL, bool MOM, bool TSC, bool GEN>
__global__ void __launch_bounds__(kMfThreads)
mf_one_kernel(const BatchPtrs bp, const Workspace ws, const Geom g, const EpiArgs a_in, const OneArgs o) {
  __shared__ double colsum[CF_NUM + 1][4][64];
  __shared__ ScalarShared ssh;
  __shared__ double thk[2];
  __shared__ int role;
  const int t = threadIdx.x, lane = t & 63;
  int rb, cb;
  mf_wg_coords(g, (int)blockIdx.x, &rb, &cb);
  double* wsb = ws.base;
  const int d = g.d;
  const double* theta_src = bp.theta_src[0];
  double* theta_dev = wsb + ws.off_theta;
  constexpr bool FUN = MODEL == VB_MODEL_FUNNEL;
  const int k = a_in.model.k;

  // ---- prologue: theta of this column block (and of the coupling column) ---------------------------------------------
  const int c0i = cb * kMfCols + 2 * lane;
  double mu[2] = {0.0, 0.0}, ls[2] = {0.0, 0.0};
  if (o.theta_on_device) {
    // (the optimiser's own array: the step of THIS launch rewrites a column only after every workgroup that reads it --
    // the column block's own row blocks, and for the coupling column everybody -- has taken its ticket)
    if (t < 64) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (c0i + j < d) mu[j] = theta_src[c0i + j], ls[j] = theta_src[d + c0i + j];
    }
    if (FUN && t == 64) thk[0] = theta_src[k], thk[1] = theta_src[d + k];
    if (rb == 0 && t < 64) {       // the copy the tail reads (the step overwrites theta_src itself)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (c0i + j < d) cst<true>(theta_dev + c0i + j, mu[j]), cst<true>(theta_dev + d + c0i + j, ls[j]);
    }
  } else if (rb == 0) {
    if (t < 64) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (c0i + j < d) {
          mu[j] = theta_src[c0i + j], ls[j] = theta_src[d + c0i + j];      // across PCIe, once per column block
          cst<true>(theta_dev + c0i + j, mu[j]), cst<true>(theta_dev + d + c0i + j, ls[j]);
        }
    }
    if (FUN && t == 64) thk[0] = theta_src[k], thk[1] = theta_src[d + k];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_store(o.flag + cb, o.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    if (t == 0) {
      int ok = 1, spins = 0;
      const int kcb = FUN ? k / kMfCols : cb;
      while (__hip_atomic_load(o.flag + cb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != o.epoch ||
             __hip_atomic_load(o.flag + kcb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != o.epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 16)) {      // (a publisher that has not been scheduled: take the host copy ourselves)
          ok = 0;
          break;
        }
      }
      role = ok;
    }
    __syncthreads();
    const double* src = role ? (const double*)theta_dev : theta_src;
    if (t < 64) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (c0i + j < d) {
          mu[j] = role ? cld<true>(src + c0i + j) : src[c0i + j];
          ls[j] = role ? cld<true>(src + d + c0i + j) : src[d + c0i + j];
        }
    }
    if (FUN && t == 64) thk[0] = role ? cld<true>(src + k) : src[k], thk[1] = role ? cld<true>(src + d + k) : src[d + k];
  }
  // the four waves of the workgroup own the same 128 columns: wave 0's values to all of them through LDS
  double* xch = &colsum[0][0][0];
  __syncthreads();
  if (t < 64) xch[4 * lane] = mu[0], xch[4 * lane + 1] = mu[1], xch[4 * lane + 2] = ls[0], xch[4 * lane + 3] = ls[1];
  __syncthreads();
  OnePro pro;
  {
    const double m0 = xch[4 * lane], m1 = xch[4 * lane + 1], l0 = xch[4 * lane + 2], l1 = xch[4 * lane + 3];
    const bool in0 = c0i < d, in1 = c0i + 1 < d;
    const double s0 = in0 ? exp(l0) : 0.0, s1 = in1 ? exp(l1) : 0.0;
    if (FUN) {      // the coupling column contributes through the per-row sums, not elementwise (prep_column)
      pro.cp0 = (d2){(in0 && c0i != k) ? m0 : 0.0, (in1 && c0i + 1 != k) ? m1 : 0.0};
      pro.cp1 = (d2){(in0 && c0i != k) ? s0 : 0.0, (in1 && c0i + 1 != k) ? s1 : 0.0};
      pro.cp2 = (d2){0.0, 0.0};
    } else {
      const double* p0 = a_in.model.p0;
      const double* p1 = a_in.model.p1;
      pro.cp0 = (d2){in0 ? m0 - p0[c0i] : 0.0, in1 ? m1 - p0[c0i + 1] : 0.0};
      pro.cp1 = (d2){s0, s1};
      pro.cp2 = (d2){in0 ? p1[c0i] : 0.0, in1 ? p1[c0i + 1] : 0.0};
    }
    pro.muk = FUN ? thk[0] : 0.0;
    pro.lsk = FUN ? thk[1] : 0.0;
  }
  // ---- prologue: row scalars of this row block (funnel, noise in memory) and the coupling column's partial sums ------
  if (FUN && !GEN) {
    const int64_t r0 = (int64_t)rb * g.rows_per_wg;
    const int64_t r1 = (r0 + g.rows_per_wg < g.n) ? r0 + g.rows_per_wg : g.n;
    const double sgk = exp(pro.lsk), muk = pro.muk;
    const double it2 = 1.0 / (a_in.model.tau * a_in.model.tau), dm1 = (double)(d - 1);
    const double* noise = bp.noise[0];
    double W = 0.0, FK = 0.0, GK = 0.0, GEK = 0.0;
    for (int64_t r = r0 + t; r < r1; r += kMfThreads) {
      const double ek = noise[r * g.ld + k];
      const double v = fma(sgk, ek, muk);
      const double gk = fma(-v, it2, -dm1);
      d2* rs = reinterpret_cast<d2*>(wsb + ws.off_rowscal + 4 * r);
      rs[0] = (d2){exp(-2.0 * v), ek};
      rs[1] = (d2){1.0, 0.0};
      W += 1.0;
      FK += v * fma(-0.5 * v, it2, -dm1);
      GK += gk;
      GEK += gk * ek;
    }
    if (cb == 0) {      // (workgroup-uniform)
      W = wave_sum(W), FK = wave_sum(FK), GK = wave_sum(GK), GEK = wave_sum(GEK);
      double(*sh)[PS_NUM] = reinterpret_cast<double(*)[PS_NUM]>(&colsum[1][0][0]);
      if (lane == 0) sh[t >> 6][PS_W] = W, sh[t >> 6][PS_FK] = FK, sh[t >> 6][PS_GK] = GK, sh[t >> 6][PS_GEK] = GEK;
      __syncthreads();
      if (t < PS_NUM)
        cst<true>(wsb + ws.off_prepscal + (int64_t)t * g.n_prep + rb, (sh[0][t] + sh[1][t]) + (sh[2][t] + sh[3][t]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the row scalars are in L2 before the scalar cache asks for them
    __syncthreads();
  } else if (!FUN && rb == 0 && cb == 0 && t < PS_NUM) {
    cst<true>(wsb + ws.off_prepscal + t, t == PS_W ? (double)g.n : 0.0);      // (n_prep = 1)
  }

  // ---- the streaming pass -----------------------------------------------------------------------------------------------
  mf_accum_body<MODEL, MOM, TSC, false, GEN, true>(bp, ws, g, pro);

  // ---- tail: tickets instead of launch boundaries -----------------------------------------------------------------------
  EpiArgs a = a_in;
  a.partials = wsb + ws.off_partials;
  a.pscal = wsb + ws.off_pscal;
  a.prepscal = wsb + ws.off_prepscal;
  a.theta = theta_dev;
  a.out = bp.out[0];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this workgroup's write-through partials have left
  __syncthreads();
  if (t == 0) role = atomicAdd(o.cnt + 1 + cb, 1u) == (unsigned)g.n_rb - 1 ? 1 : 0;
  __syncthreads();
  if (role) {      // the column block is complete: its two groups of 64 columns
    one_tail_group<MOM, TSC>(a, o, 2 * cb, colsum);
    if (2 * cb + 1 < g.Dp / 64) one_tail_group<MOM, TSC>(a, o, 2 * cb + 1, colsum);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (t == 0) role = atomicAdd(o.cnt, 1u) == gridDim.x - 1 ? 1 : 0;
  __syncthreads();
  if (!role) return;
  // the last workgroup of all: scalar totals, the value, the coupling column
  if (t < g.n_cb + 1)                    // (everybody has taken both tickets: ready for the next launch)
    __hip_atomic_store(o.cnt + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ScalarFetch sf;
  scalar_fetch<true>(a, true, &sf);
  double kb[CF_NUM + 7];
  const bool funnel = a.model.id == VB_MODEL_FUNNEL;
  if (funnel && t == 0) {
#pragma unroll
    for (int i = 0; i < CF_NUM + 7; ++i) kb[i] = cld<true>(o.kbuf + i);
  }
  double sacc[kNS], sum_ls = 0.0;
  scalar_accumulate<true>(a, true, sf, sacc);
  const Totals tot = scalar_reduce(sacc, &ssh, &sum_ls);
  if (t == 0) {
    if (funnel) {
      const bool student = a.family == VB_FAMILY_MF_STUDENT_T;
      const double gsum = kb[CF_G] + (tot.v[SF_GK] + tot.v[SF_Q]), ge = kb[CF_GE] + (tot.v[SF_GEK] + tot.v[SF_QE]);
      double gm, gl;
      plain_column(a, k, gsum, ge, student ? kb[CF_SC] : kb[CF_E], student ? kb[CF_SCE] : kb[CF_EE], a.out + 1, a.out + 1 + d,
                   kb[CF_NUM], &gm, &gl);
      if (a.has_step) {
        fit_step_apply_vals(a.step, k, gm, kb[CF_NUM + 1], kb[CF_NUM + 2], kb[CF_NUM + 3]);
        fit_step_apply_vals(a.step, (int64_t)d + k, gl, kb[CF_NUM + 4], kb[CF_NUM + 5], kb[CF_NUM + 6]);
      }
    }
    const double val = elbo_value(a, tot, sum_ls);
    a.out[0] = val;
    if (a.has_step) a.step.values[a.step.k] = val;
  }
}

// one workgroup over the reduced sums: control variates, weighted gradients, post-all-reduce
__global__ void __launch_bounds__(256)
mf_epilogue_kernel(const EpiArgs a_in, const BatchPtrs bp, const Workspace ws) {
  EpiArgs a = a_in;   // per-evaluation pointers (blockIdx.y = evaluation index)
  {
    const int b = blockIdx.y;
    double* wsb = ws.base + b * ws.stride;
    a.partials = wsb + ws.off_partials;
    a.pscal = wsb + ws.off_pscal;
    a.prepscal = wsb + ws.off_prepscal;
    a.theta = wsb + ws.off_theta;
    a.sums = ws.sums + b * ws.sum_len;
    a.out = bp.out[b];
  }
  __shared__ double sh[4];
  const int d = a.d, Dp = a.Dp;
  const double* G = a.sums + SF_NUM + (int64_t)CF_G * Dp;
  const double* GE = a.sums + SF_NUM + (int64_t)CF_GE * Dp;
  const double* E = a.sums + SF_NUM + (int64_t)CF_E * Dp;
  const double* EE = a.sums + SF_NUM + (int64_t)CF_EE * Dp;
  const double* EK = a.sums + SF_NUM + (int64_t)CF_EK * Dp;
  const double* SC = a.sums + SF_NUM + (int64_t)CF_SC * Dp;
  const double* SCE = a.sums + SF_NUM + (int64_t)CF_SCE * Dp;
  const double* S = a.sums;
  Totals tot;
#pragma unroll
  for (int s = 0; s < SF_NUM; ++s) tot.v[s] = S[s];
  const double* mu = a.theta;
  const double* ls = a.theta + d;
  const bool funnel = a.model.id == VB_MODEL_FUNNEL;
  const int k = funnel ? a.model.k : -1;
  const double Wsum = tot.v[SF_W];
  const double gk_add = funnel ? tot.v[SF_GK] + tot.v[SF_Q] : 0.0;
  const double gek_add = funnel ? tot.v[SF_GEK] + tot.v[SF_QE] : 0.0;
  double* value = a.out;
  double* gmu = a.out + 1;
  double* gls = a.out + 1 + d;

  if (a.mode == 2) {   // DIS: -scale * sum_n w_n log q(z_n; theta) and its gradient (objectives.py:405-414)
    const bool st = a.family == VB_FAMILY_MF_STUDENT_T;
    double t_ls = 0.0;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
      const double sg = exp(ls[i]);
      gmu[i] = -a.scale * E[i] / sg;
      gls[i] = -a.scale * (EE[i] - Wsum);
      t_ls += ls[i];
    }
    const double sum_ls = block_sum(t_ls, sh);
    if (threadIdx.x == 0) {
      const double cb = st ? lgamma(0.5 * (a.df + 1.0)) - lgamma(0.5 * a.df) - 0.5 * log(a.df * M_PI)
                           : -0.5 * kLog2Pi;
      const double lterm = st ? -0.5 * (a.df + 1.0) * tot.v[SF_L1P] : -0.5 * tot.v[SF_L1P];
      value[0] = -a.scale * (lterm + Wsum * (d * cb - sum_ls));
    }
    return;
  }

  if (a.mode == 1) {   // weighted gradient only: scale * sum_n w_n [g_n | g_n eps_n sigma + 1]
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
      const double g = G[i] + (i == k ? gk_add : 0.0);
      const double ge = GE[i] + (i == k ? gek_add : 0.0);
      gmu[i] = a.scale * g;
      gls[i] = a.scale * (ge * exp(ls[i]) + Wsum);
    }
    if (threadIdx.x == 0) {
      double F = tot.v[SF_F] + Wsum * a.model.c0;
      if (funnel) F += tot.v[SF_FK] - 0.5 * tot.v[SF_Q];
      value[0] = a.value_src ? a.value_src[0] : F;
    }
    return;
  }

  const double invN = 1.0 / a.n_total;
  const bool student = a.family == VB_FAMILY_MF_STUDENT_T;
  double t_ls = 0.0;
  for (int i = threadIdx.x; i < d; i += blockDim.x) t_ls += ls[i];
  const double sum_ls = block_sum(t_ls, sh);
  if (threadIdx.x == 0) value[0] = elbo_value(a, tot, sum_ls);

  if (a.cv_mode == VB_CV_NONE) {
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
      const double g = G[i] + (i == k ? gk_add : 0.0);
      const double ge = GE[i] + (i == k ? gek_add : 0.0);
      plain_column(a, i, g, ge, student ? SC[i] : E[i], student ? SCE[i] : EE[i], gmu, gls, ls[i]);
    }
    return;
  }

  // ---- RGE control variates, single-pass reduction of objectives.py:200-268 ---------------
  // vbar = s * mean(eps') = sigma * E / N ;  H, g_mu evaluated at m = mu
  const double c2 = student ? (a.df - 2.0) / a.df : 1.0;   // (sigma / s)^2
  double t_m2 = 0.0, t_mv = 0.0, t_mek = 0.0;
  double wk = 0.0, vbar_k = 0.0, sg_k = 0.0;
  if (funnel) {
    wk = exp(-2.0 * mu[k]);
    sg_k = exp(ls[k]);
    vbar_k = sg_k * E[k] * invN;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
      if (i == k) continue;
      const double sg = exp(ls[i]);
      t_m2 += mu[i] * mu[i];
      t_mv += mu[i] * sg * E[i] * invN;
      t_mek += mu[i] * sg * EK[i] * invN;
    }
  }
  const double sum_m2 = block_sum(t_m2, sh);     // sum_{i != k} mu_i^2
  const double sum_mv = block_sum(t_mv, sh);     // sum_{i != k} mu_i vbar_i
  const double sum_mek = block_sum(t_mek, sh);   // sum_{j != k} mu_j sigma_j M2_kj
  const double inv_tau2 = 1.0 / (a.model.tau * a.model.tau);
  const double Hkk = funnel ? (-inv_tau2 - 2.0 * wk * sum_m2) : 0.0;

  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    const double sg = exp(ls[i]);
    const double g = G[i] + (i == k ? gk_add : 0.0);
    const double ge = GE[i] + (i == k ? gek_add : 0.0);
    const double vbar = sg * E[i] * invN;
    double gmu_i, Hii, Hv, Hm2;   // Hm2 = sum_j H_ij sigma_j M2_ij
    if (!funnel) {
      const double iv = a.model.p1[i];
      gmu_i = -(mu[i] - a.model.p0[i]) * iv;
      Hii = -iv;
      Hv = -iv * vbar;
      Hm2 = -iv * sg * EE[i] * invN;
    } else if (i != k) {
      gmu_i = -mu[i] * wk;
      Hii = -wk;
      const double Hik = 2.0 * mu[i] * wk;
      Hv = Hii * vbar + Hik * vbar_k;
      Hm2 = Hii * sg * EE[i] * invN + Hik * sg_k * EK[i] * invN;
    } else {
      gmu_i = -mu[k] * inv_tau2 + wk * sum_m2 - (double)(d - 1);
      Hii = Hkk;
      Hv = 2.0 * wk * sum_mv + Hkk * vbar_k;
      Hm2 = 2.0 * wk * sum_mek + Hkk * sg_k * EE[k] * invN;
    }
    const double mean_block = g * invN - Hv;
    double scale_block = ge * sg * invN + 1.0;
    if (a.cv_mode == VB_CV_LOO_DIAG || a.cv_mode == VB_CV_FULL) scale_block -= gmu_i * vbar;
    if (a.cv_mode == VB_CV_FULL) scale_block += -sg * Hm2 + Hii * sg * sg / c2;
    gmu[i] = -mean_block;
    gls[i] = -scale_block;
  }
}

}  // namespace vb

#include "vb_logistic.h"

namespace vb {

// ---------------------------------------------------------------------------------------------
struct Launch {
  dim3 grid;
  hipStream_t st;
  hipEvent_t ev0, ev1;   // exact kernel begin / end timestamps when profiling (else nullptr)
};

template <int MODEL, bool MOM, bool TSC>
static void launch_accum(bool weighted, const Launch& L, const BatchPtrs& bp, const Workspace& ws,
                         const Geom& g) {
  if (weighted)
    hipExtLaunchKernelGGL((mf_accum_kernel<MODEL, MOM, TSC, true>), L.grid, dim3(kMfThreads), 0, L.st,
                          L.ev0, L.ev1, 0, bp, ws, g);
  else
    hipExtLaunchKernelGGL((mf_accum_kernel<MODEL, MOM, TSC, false>), L.grid, dim3(kMfThreads), 0, L.st,
                          L.ev0, L.ev1, 0, bp, ws, g);
}

template <int MODEL>
static void launch_accum_gen(bool mom, bool tsc, const Launch& L, const BatchPtrs& bp, const Workspace& ws,
                             const Geom& g) {
  if (mom && tsc)
    hipExtLaunchKernelGGL((mf_accum_kernel<MODEL, true, true, false, true>), L.grid, dim3(kMfThreads), 0, L.st,
                          L.ev0, L.ev1, 0, bp, ws, g);
  else if (tsc)
    hipExtLaunchKernelGGL((mf_accum_kernel<MODEL, false, true, false, true>), L.grid, dim3(kMfThreads), 0, L.st,
                          L.ev0, L.ev1, 0, bp, ws, g);
  else if (mom)
    hipExtLaunchKernelGGL((mf_accum_kernel<MODEL, true, false, false, true>), L.grid, dim3(kMfThreads), 0, L.st,
                          L.ev0, L.ev1, 0, bp, ws, g);
  else
    hipExtLaunchKernelGGL((mf_accum_kernel<MODEL, false, false, false, true>), L.grid, dim3(kMfThreads), 0, L.st,
                          L.ev0, L.ev1, 0, bp, ws, g);
}

template <int MODEL>
static void launch_accum_model(bool mom, bool tsc, bool weighted, const Launch& L, const BatchPtrs& bp,
                               const Workspace& ws, const Geom& g) {
  if (g.gen) return launch_accum_gen<MODEL>(mom, tsc, L, bp, ws, g);
  if (mom && tsc) launch_accum<MODEL, true, true>(weighted, L, bp, ws, g);
  else if (mom) launch_accum<MODEL, true, false>(weighted, L, bp, ws, g);
  else if (tsc) launch_accum<MODEL, false, true>(weighted, L, bp, ws, g);
  else launch_accum<MODEL, false, false>(weighted, L, bp, ws, g);
}

template <int MODEL, bool GEN>
static void launch_one_mg(bool mom, bool tsc, const Launch& L, const BatchPtrs& bp, const Workspace& ws, const Geom& g,
                          const EpiArgs& e, const OneArgs& o) {
  if (mom && tsc)
    hipExtLaunchKernelGGL((mf_one_kernel<MODEL, true, true, GEN>), L.grid, dim3(kMfThreads), 0, L.st, L.ev0, L.ev1, 0, bp, ws, g, e, o);
  else if (tsc)
    hipExtLaunchKernelGGL((mf_one_kernel<MODEL, false, true, GEN>), L.grid, dim3(kMfThreads), 0, L.st, L.ev0, L.ev1, 0, bp, ws, g, e, o);
  else if (mom)
    hipExtLaunchKernelGGL((mf_one_kernel<MODEL, true, false, GEN>), L.grid, dim3(kMfThreads), 0, L.st, L.ev0, L.ev1, 0, bp, ws, g, e, o);
  else
    hipExtLaunchKernelGGL((mf_one_kernel<MODEL, false, false, GEN>), L.grid, dim3(kMfThreads), 0, L.st, L.ev0, L.ev1, 0, bp, ws, g, e, o);
}

static void launch_one(bool funnel, bool mom, bool tsc, bool gen, const Launch& L, const BatchPtrs& bp, const Workspace& ws,
                       const Geom& g, const EpiArgs& e, const OneArgs& o) {
  if (funnel && gen) launch_one_mg<VB_MODEL_FUNNEL, true>(mom, tsc, L, bp, ws, g, e, o);
  else if (funnel) launch_one_mg<VB_MODEL_FUNNEL, false>(mom, tsc, L, bp, ws, g, e, o);
  else if (gen) launch_one_mg<VB_MODEL_GAUSS_DIAG, true>(mom, tsc, L, bp, ws, g, e, o);
  else launch_one_mg<VB_MODEL_GAUSS_DIAG, false>(mom, tsc, L, bp, ws, g, e, o);
}

static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

// Logistic-regression target: sample, two MFMA GEMMs, then the explicit-gradient streaming pass.  Fills
// the same workspace (theta copy, column constants, prep scalars, partials) the other models fill.
static int logistic_accumulate(vb_ctx* ctx, hipStream_t st, const ModelDev& m, const NoiseSlot& ns,
                               const BatchPtrs& bp, const Workspace& ws, const Geom& g, bool mom, bool tsc) {
  const int64_t n = g.n, d = g.d, nd = m.n_data;
  const int64_t ldz = round_up(d, 16), ldr = round_up(nd, 16);
  const int n_cu = ctx->prop.multiProcessorCount;
  const int64_t max_blocks = gemm_max_blocks(n, nd);
  int64_t off = 0;
  auto carve = [&off](int64_t doubles) {
    const int64_t o = off;
    off += round_up(doubles, 16);
    return o;
  };
  const int64_t o_z = carve(n * ldz), o_g = carve(n * ldz), o_r = carve(n * ldr), o_part = carve(max_blocks),
                o_fsum = carve(16);
  VB_TRY(ensure(ctx, ctx->lg_work, (size_t)off * sizeof(double)));
  double* base = (double*)ctx->lg_work.ptr;
  double *Z = base + o_z, *G = base + o_g, *Rm = base + o_r, *part = base + o_part, *fsum = base + o_fsum;
  double* wsb = ws.base;
  VB_HIP(ctx, hipMemsetAsync(part, 0, (size_t)max_blocks * sizeof(double), st));
  VB_HIP(ctx, hipMemsetAsync(wsb + ws.off_colp, 0, (size_t)3 * g.Dp * sizeof(double), st));
  hipLaunchKernelGGL(lg_sample_kernel, dim3((unsigned)n, (unsigned)((d + 255) / 256)), dim3(256), 0, st,
                     bp.theta_src[0], wsb + ws.off_theta, wsb + ws.off_colp, g.Dp, (const double*)ns.buf.ptr, ns.ld,
                     Z, ldz, n, (int)d);
  VB_HIP(ctx, hipGetLastError());
  GemmArgs gh;                       // H = Z X'   [n x n_data x d]
  gh.A = Z;
  gh.lda = ldz;
  gh.B = m.p1;
  gh.ldb = m.ldq;
  gh.M = (int)n;
  gh.N = (int)nd;
  gh.K = (int)d;
  gh.tri_mode = 0;
  gemm_f64_launch<true>(st, gh, 1, n_cu, EpiLogit{Rm, ldr, m.p2, part, m.link, m.aux});
  VB_HIP(ctx, hipGetLastError());
  const double ivp = 1.0 / (m.tau * m.tau);
  VB_TRY(glm_grad_enqueue(ctx, st, m, Rm, ldr, Z, G, ldz, n, (int)d));   // G = R X - Z / sd^2
  hipLaunchKernelGGL(lg_scalars_kernel, dim3(1), dim3(256), 0, st, (const double*)part, (int)max_blocks, fsum,
                     wsb + ws.off_prepscal, (double)n);
  VB_HIP(ctx, hipGetLastError());
  const dim3 grid((unsigned)(g.n_rb * g.n_cb));
  if (mom && tsc)
    hipLaunchKernelGGL((lg_accum_kernel<true, true>), grid, dim3(kMfThreads), 0, st, (const double*)ns.buf.ptr, ns.ld,
                       (const double*)G, ldz, ws, g, ivp, (const double*)fsum);
  else if (mom)
    hipLaunchKernelGGL((lg_accum_kernel<true, false>), grid, dim3(kMfThreads), 0, st, (const double*)ns.buf.ptr, ns.ld,
                       (const double*)G, ldz, ws, g, ivp, (const double*)fsum);
  else if (tsc)
    hipLaunchKernelGGL((lg_accum_kernel<false, true>), grid, dim3(kMfThreads), 0, st, (const double*)ns.buf.ptr, ns.ld,
                       (const double*)G, ldz, ws, g, ivp, (const double*)fsum);
  else
    hipLaunchKernelGGL((lg_accum_kernel<false, false>), grid, dim3(kMfThreads), 0, st, (const double*)ns.buf.ptr, ns.ld,
                       (const double*)G, ldz, ws, g, ivp, (const double*)fsum);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// Source model (VB_MODEL_SOURCE, vb_usermodel.hip): sample, the user's row kernel for (f, G), then the same
// explicit-gradient streaming pass (no prior term: ivp = 0, everything is in the user's f).
static int source_accumulate(vb_ctx* ctx, hipStream_t st, const NoiseSlot& ns, const BatchPtrs& bp, const Workspace& ws,
                             const Geom& g, bool mom, bool tsc, const double* roww) {
  const int64_t n = g.n, d = g.d;
  const int64_t ldz = round_up(d, 16);
  int64_t off = 0;
  auto carve = [&off](int64_t doubles) {
    const int64_t o = off;
    off += round_up(doubles, 16);
    return o;
  };
  const int64_t o_z = carve(n * ldz), o_g = carve(n * ldz), o_f = carve(n), o_fsum = carve(16);
  VB_TRY(ensure(ctx, ctx->lg_work, (size_t)off * sizeof(double)));
  double* base = (double*)ctx->lg_work.ptr;
  double *Z = base + o_z, *G = base + o_g, *frow = base + o_f, *fsum = base + o_fsum;
  double* wsb = ws.base;
  VB_HIP(ctx, hipMemsetAsync(wsb + ws.off_colp, 0, (size_t)3 * g.Dp * sizeof(double), st));
  VB_HIP(ctx, hipMemsetAsync(G, 0, (size_t)n * ldz * sizeof(double), st));     // pad columns of G are streamed too
  hipLaunchKernelGGL(lg_sample_kernel, dim3((unsigned)n, (unsigned)((d + 255) / 256)), dim3(256), 0, st,
                     bp.theta_src[0], wsb + ws.off_theta, wsb + ws.off_colp, g.Dp, (const double*)ns.buf.ptr, ns.ld,
                     Z, ldz, n, (int)d);
  VB_HIP(ctx, hipGetLastError());
  VB_TRY(user_rows_enqueue(ctx, st, Z, ldz, n, (int)d, G, ldz, frow));
  if (roww) {      // weighted gradient (AlphaDivergence): sum_n w_n g_n = the plain sums of the row-scaled matrix
    hipLaunchKernelGGL(lg_rowscale_kernel, dim3((unsigned)n, (unsigned)((d + 255) / 256)), dim3(256), 0, st, G, ldz, n,
                       (int)d, roww);
    VB_HIP(ctx, hipGetLastError());
  }
  hipLaunchKernelGGL(lg_scalars_kernel, dim3(1), dim3(256), 0, st, (const double*)frow, (int)n, fsum,
                     wsb + ws.off_prepscal, (double)n, roww, n);
  VB_HIP(ctx, hipGetLastError());
  const dim3 grid((unsigned)(g.n_rb * g.n_cb));
  if (mom && tsc)
    hipLaunchKernelGGL((lg_accum_kernel<true, true>), grid, dim3(kMfThreads), 0, st, (const double*)ns.buf.ptr, ns.ld,
                       (const double*)G, ldz, ws, g, 0.0, (const double*)fsum);
  else if (mom)
    hipLaunchKernelGGL((lg_accum_kernel<true, false>), grid, dim3(kMfThreads), 0, st, (const double*)ns.buf.ptr, ns.ld,
                       (const double*)G, ldz, ws, g, 0.0, (const double*)fsum);
  else if (tsc)
    hipLaunchKernelGGL((lg_accum_kernel<false, true>), grid, dim3(kMfThreads), 0, st, (const double*)ns.buf.ptr, ns.ld,
                       (const double*)G, ldz, ws, g, 0.0, (const double*)fsum);
  else
    hipLaunchKernelGGL((lg_accum_kernel<false, false>), grid, dim3(kMfThreads), 0, st, (const double*)ns.buf.ptr, ns.ld,
                       (const double*)G, ldz, ws, g, 0.0, (const double*)fsum);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

// Enqueue prep -> accumulate -> finalize [-> all-reduce -> epilogue] for a batch of `count`
// independent evaluations.
int mf_enqueue(vb_ctx* ctx, const MfCall& c) {
  const int64_t n = c.n, d = c.d;
  const ModelDev& model = c.model ? *c.model : ctx->model;
  if (c.count < 1 || c.count > kMaxBatch)
    return fail(ctx, VB_ERR_INVALID, "batch size %d outside [1, %d]", c.count, kMaxBatch);
  // `logistic` = "the model's gradient matrix is produced before the streaming pass and loaded by it": the regression
  // targets (two GEMMs) and the source model (the user's row kernel)
  const bool source = model.id == VB_MODEL_SOURCE;
  const bool logistic = model.id == VB_MODEL_LOGISTIC || source;
  // (a source model also takes the weighted-gradient mode of AlphaDivergence: its G is scaled row by row)
  const bool source_weighted = source && c.count == 1 && c.mode == 1 && c.cv_mode == VB_CV_NONE && c.roww[0] != nullptr;
  if (logistic && !source_weighted &&
      (c.count != 1 || c.mode != 0 || c.cv_mode != VB_CV_NONE || c.roww[0] != nullptr))
    return fail(ctx, VB_ERR_UNSUPPORTED,
                "regression and source-model targets support single ExclusiveKL evaluations without control variates");
  if (model.id != VB_MODEL_GAUSS_DIAG && model.id != VB_MODEL_FUNNEL && model.id != kModelLogQ && !logistic)
    return fail(ctx, VB_ERR_UNSUPPORTED,
                "mean-field path supports the gauss_diag and funnel models (model id %d bound)", model.id);
  if (model.dim != d)
    return fail(ctx, VB_ERR_INVALID, "model dimension %d != family dimension %lld", model.dim,
                (long long)d);
  if (c.family != VB_FAMILY_MF_GAUSSIAN && c.family != VB_FAMILY_MF_STUDENT_T)
    return fail(ctx, VB_ERR_INVALID, "family %d is not a mean-field family", c.family);
  if (c.family == VB_FAMILY_MF_STUDENT_T && !(c.df > 2.0))
    return fail(ctx, VB_ERR_INVALID, "df must be greater than 2");
  if (c.cv_mode < VB_CV_NONE || c.cv_mode > VB_CV_LOO_DIRECT)
    return fail(ctx, VB_ERR_INVALID, "unknown control-variate mode %d", c.cv_mode);
  if (n <= 0 || d <= 0) return fail(ctx, VB_ERR_INVALID, "n and d must be positive");
  for (int b = 0; b < c.count; ++b) {
    const NoiseSlot* ns = c.noise[b];
    if (!ns || !ns->buf.ptr || n > ns->n || d != ns->d)
      return fail(ctx, VB_ERR_INVALID, "noise slot of evaluation %d does not hold a %lld x %lld matrix",
                  b, (long long)n, (long long)d);
  }

  const bool pd = (c.flags & VB_FLAG_PATH_DERIV) != 0;
  const bool student = c.family == VB_FAMILY_MF_STUDENT_T;
  const bool mom = (c.mode == 0 && ((pd && !student) || c.cv_mode != VB_CV_NONE)) || c.mode == 2;
  const bool tsc = (c.mode == 0 && pd && student) || (c.mode == 2 && student);
  const int nf = tsc ? CF_NUM : (mom ? CF_EK + 1 : CF_GE + 1);
  const bool weighted = c.roww[0] != nullptr;
  const bool funnel = model.id == VB_MODEL_FUNNEL;
  const bool rows = funnel || weighted;

  Geom g;
  g.ld = c.noise[0]->ld;
  g.n = n;
  g.d = (int)d;
  g.n_cb = (int)((d + kMfCols - 1) / kMfCols);
  g.Dp = g.n_cb * kMfCols;
  // ~2 workgroups per CU over the whole batch; rows per workgroup a multiple of the 4 waves
  const int target_wg = env_int("VB_MF_TARGET_WG", 2 * ctx->prop.multiProcessorCount);
  int n_rb_target = target_wg / g.n_cb / (c.count < 4 ? c.count : 4);
  if (n_rb_target < 8) n_rb_target = 8;
  // narrow families with the noise in memory: the finalize kernel adds one partial per row block and column in a
  // dependent chain (512 of them: 21 us at D = 64, N = 16 384 -- three times the streaming pass over those 8 MB), so a
  // small matrix is cut into at most 128 row blocks.  (Not with the noise generated in registers: that pass is bound by
  // the generator's arithmetic and wants every workgroup it can get -- MFStudentT 67 -> 106 us with the cap.)
  if (!c.gen && n_rb_target > 128 && n * d * 8 < (int64_t)16 << 20) n_rb_target = env_int("VB_MF_MAX_ROW_BLOCKS", 128);
  n_rb_target = (n_rb_target + 7) / 8 * 8;
  int rows_per_wg = (int)((n + n_rb_target - 1) / n_rb_target);
  rows_per_wg = env_int("VB_MF_ROWS_PER_WG", rows_per_wg);
  g.rows_per_wg = (int)round_up(rows_per_wg < kMfWaves ? kMfWaves : rows_per_wg, kMfWaves);
  g.n_rb = (int)((n + g.rows_per_wg - 1) / g.rows_per_wg);
  const int64_t prep_items = rows ? (n > g.Dp ? n : g.Dp) : g.Dp;
  g.n_prep = logistic ? 1 : (int)((prep_items + 255) / 256);
  g.xcd_map = (g.n_rb % 8 == 0) ? env_int("VB_MF_XCD_MAP", 1) : 0;
  g.rows = rows ? 1 : 0;
  g.df = c.df;
  g.gen = 0;
  g.gk0 = g.gk1 = g.gw = 0;
  g.grow0 = 0;
  g.inline_rows = 0;
  g.fk = 0;
  g.ftau = 1.0;
  g.gdf = c.df;
  if (c.gen) {
    if (c.count != 1 || c.mode != 0 || weighted || logistic ||
        (model.id != VB_MODEL_GAUSS_DIAG && model.id != VB_MODEL_FUNNEL))
      return fail(ctx, VB_ERR_UNSUPPORTED,
                  "in-register noise: single mean-field ELBO evaluations on gauss_diag / funnel targets");
    g.gen = student ? 2 : 1;     // the family's own base noise: t_df for MFStudentT
    g.gk0 = (uint32_t)c.gen_seed;
    g.gk1 = (uint32_t)(c.gen_seed >> 32) ^ (uint32_t)(c.gen_stream >> 32);
    g.gw = (uint32_t)c.gen_stream;
    g.grow0 = c.gen_row_offset;
    if (funnel && n >= 512 && env_int("VB_MF_INLINE_ROWS", 1)) {
      // no row part in prep (nor in the previous iteration's finalize, where one workgroup would do it block after
      // block: 17 us at N = 4096): the streaming kernel does it for its own rows, and the device-resident loop stays
      // at two launches per iteration (C1: 35.0 -> 31.4 us; D = 256 / N = 1024: 23.7 -> 21.7 us).  For a handful of
      // rows the fused prep of the finalize kernel is the shorter path (D = 100 / N = 10: 14.1 vs 15.4 us) and is kept
      g.inline_rows = 1;
      g.fk = model.k;
      g.ftau = model.tau;
      g.rows = 0;
      g.n_prep = g.n_rb;          // one entry of coupling-column partials per row block (written by column block 0)
    }
  }
  // The whole evaluation in ONE launch (mf_one_kernel): a single plain ELBO evaluation on one rank, on the main stream.
  // Built as VERDICT r4 item 5 asked, results equal to the launch chain's (tests/test_gpu_one_launch.py) -- and MEASURED
  // SLOWER, so it is off unless VB_MF_ONE=1: C1 blocking call 43.1 against 32.2 us, device-loop iteration 31.4 against
  // 24.3 us; C0 28.9 / 18.5 against 27.3 / 14.3 us (tools/one_launch_bench.py).  Inside one launch a hand-off
  // between workgroups on different XCDs is a write-through store, a ticket and an L2-bypassing load -- a memory round
  // trip each (~2 us), three of them in sequence behind the last streaming workgroup -- where the launch boundary costs
  // ~2 us once and leaves the partial sums in an L2 every workgroup of the next kernel may hit.
  const bool one = env_int("VB_MF_ONE", 0) && c.count == 1 && c.mode == 0 && c.cv_mode == VB_CV_NONE && !ctx->comm &&
                   !logistic && !weighted && !c.pipelined && !c.alternate && !c.skip_prep &&
                   (model.id == VB_MODEL_GAUSS_DIAG || funnel);
  if (one) {
    if (funnel && g.gen) {          // the streaming kernel forms the row scalars of its own rows, whatever n
      g.inline_rows = 1;
      g.fk = model.k;
      g.ftau = model.tau;
      g.rows = 0;
    }
    g.n_prep = funnel ? g.n_rb : 1;      // coupling-column partials per row block (column block 0) / the constant entry
  }
  const bool rows_ws = rows && !g.inline_rows;
  const int prep_grid = g.inline_rows ? (g.Dp + 255) / 256 : g.n_prep;

  // ---- workspace layout (per evaluation) ------------------------------------------------------
  Workspace ws;
  int64_t off = 0;
  auto carve = [&off](int64_t doubles) {
    const int64_t o = off;
    off += round_up(doubles, 16);   // keep every sub-buffer 128-B aligned
    return o;
  };
  ws.off_theta = carve(2 * d);
  ws.off_colp = carve(3 * (int64_t)g.Dp);
  ws.off_rowscal = carve(rows_ws ? 4 * n : 0);
  ws.off_prepscal = carve((int64_t)PS_NUM * g.n_prep);
  ws.off_partials = carve((int64_t)g.n_rb * CF_NUM * g.Dp);
  ws.off_pscal = carve((int64_t)KS_NUM * g.n_rb * g.n_cb);
  ws.stride = off;
  ws.sum_len = round_up((int64_t)SF_NUM + (int64_t)nf * g.Dp, 16);   // only the fields in use are all-reduced
  // three rotating workspace sets: batch i+1 is prepared, and batch i-1 finalised, while batch i streams
  const int set = (int)(ctx->pipe.seq % kPipeSets);
  ctx->pipe.seq++;
  VB_TRY(ensure(ctx, ctx->workspace, (size_t)ws.stride * kMaxBatch * kPipeSets * sizeof(double)));
  VB_TRY(ensure(ctx, ctx->sums, (size_t)round_up((int64_t)SF_NUM + (int64_t)CF_NUM * g.Dp, 16) * kMaxBatch * kPipeSets *
                                    sizeof(double)));
  ws.base = (double*)ctx->workspace.ptr + (size_t)set * ws.stride * kMaxBatch;
  ws.sums = (double*)ctx->sums.ptr + (size_t)set * ws.sum_len * kMaxBatch;

  BatchPtrs bp;
  for (int b = 0; b < kMaxBatch; ++b) {
    const int s = b < c.count ? b : 0;
    bp.noise[b] = (const double*)c.noise[s]->buf.ptr;
    bp.theta_src[b] = c.theta_src[s];
    bp.out[b] = c.out[s];
    bp.roww[b] = c.roww[s];
  }

  // ---- stream plan -----------------------------------------------------------------------------
  // pipelined: prep on `pre`, the streaming kernel on the main stream, finalize (+ all-reduce,
  // epilogue) on `post`, chained by events; otherwise everything in order on the main stream.
  // overlap (sharded jobs, asynchronous batches): prep / streaming / finalize stay in order on the main
  // stream; the all-reduce and the epilogue go to `post` behind ONE event, and nothing on the main stream
  // waits for them except the re-use of this workspace set three batches later -- so RCCL moves batch i's
  // sums while batch i+1 streams.
  Pipeline& P = ctx->pipe;
  hipStream_t st_pre = ctx->stream, st_main = ctx->stream, st_post = ctx->stream;
  const bool overlap = c.overlap_comm && !c.pipelined && ctx->comm != nullptr;
  // alternate: consecutive independent batches go to two streams.  The stream is a function of the workspace
  // set (odd -> `post`), and a set comes round again four batches later, i.e. on the same stream, so no event is
  // needed between batches; `post` only has to be ordered after earlier main-stream work (noise uploads, blocking
  // calls), and later main-stream writers wait for `post` through post_pending as for the other modes.
  const bool alt = c.alternate && !c.pipelined && !overlap && ctx->comm == nullptr;
  const bool alt_b = alt && (set & 1);
  if (overlap) {
    VB_TRY(pipe_init(ctx));
    if (P.fin_valid[set]) VB_HIP(ctx, hipStreamWaitEvent(ctx->stream, P.ev_fin[set], 0));   // set is free again
  } else if (alt) {
    VB_TRY(pipe_init(ctx));
    if (alt_b) {
      if (P.main_dirty) {
        VB_HIP(ctx, hipEventRecord(P.ev_main, ctx->stream));
        VB_HIP(ctx, hipStreamWaitEvent(P.post, P.ev_main, 0));
        P.main_dirty = false;
      }
      st_pre = st_main = st_post = P.post;
    }
  } else if (c.pipelined) {
    VB_TRY(pipe_init(ctx));
    st_pre = P.pre;
    st_post = P.post;
    if (P.main_dirty) {   // earlier main-stream work (noise generation, uploads, ...) precedes prep
      VB_HIP(ctx, hipEventRecord(P.ev_main, ctx->stream));
      VB_HIP(ctx, hipStreamWaitEvent(st_pre, P.ev_main, 0));
      P.main_dirty = false;
    }
    if (P.fin_valid[set]) VB_HIP(ctx, hipStreamWaitEvent(st_pre, P.ev_fin[set], 0));   // set is free again
  } else if (P.post_pending) {   // order this in-order call after everything the pipeline has in flight
    VB_HIP(ctx, hipStreamWaitEvent(ctx->stream, P.ev_fin[P.last_set], 0));
    P.post_pending = false;
  }

  if (one) {
    // counters / flags / the coupling column's parked sums: 64 words + 16 doubles, zeroed once
    if (!ctx->mf_one.ptr) {
      VB_TRY(ensure(ctx, ctx->mf_one, 512));
      VB_HIP(ctx, hipMemsetAsync(ctx->mf_one.ptr, 0, 512, st_main));
    }
    OneArgs o;
    o.cnt = (unsigned*)ctx->mf_one.ptr;                       // [0 .. 32]
    o.flag = (unsigned*)ctx->mf_one.ptr + 40;                 // [40 .. 72)
    o.kbuf = (double*)ctx->mf_one.ptr + 40;                   // bytes [320, 448)
    o.epoch = ++ctx->mf_one_epoch;
    if (o.epoch == 0) o.epoch = ++ctx->mf_one_epoch;
    o.theta_on_device = c.theta_on_device ? 1 : 0;
    if (g.n_cb > 31) return fail(ctx, VB_ERR_UNSUPPORTED, "one-launch evaluation: at most 31 column blocks");
    EpiArgs e1;
    memset(&e1, 0, sizeof e1);
    e1.n_rb = g.n_rb;
    e1.n_ps = g.n_rb * g.n_cb;
    e1.n_prep = g.n_prep;
    e1.nf = nf;
    e1.d = (int)d;
    e1.Dp = g.Dp;
    e1.n_total = (double)c.n_total;
    e1.family = c.family;
    e1.df = c.df;
    e1.flags = c.flags;
    e1.model = model;
    if (c.step && c.step_done) {
      *c.step_done = true;
      e1.has_step = 1;
      e1.step = *c.step;
    }
    if (c.prep_done) *c.prep_done = false;
    Launch L1;
    L1.grid = dim3((unsigned)(g.n_rb * g.n_cb), 1);
    L1.st = st_main;
    prof_events(ctx, &L1.ev0, &L1.ev1, 1);
    launch_one(model.id == VB_MODEL_FUNNEL, mom, tsc, g.gen != 0, L1, bp, ws, g, e1, o);
    VB_HIP(ctx, hipGetLastError());
    P.fin_valid[set] = false;
    P.main_dirty = true;
    ctx->result_stream = st_main;
    return VB_OK;
  }
  if (!logistic && !c.skip_prep) {   // skip_prep: the previous iteration's finalize kernel has done it
    hipLaunchKernelGGL(mf_prep_kernel, dim3((unsigned)prep_grid, (unsigned)c.count), dim3(256), 0, st_pre,
                       bp, ws, g, model);
    VB_HIP(ctx, hipGetLastError());
  }
  if (c.pipelined) {
    VB_HIP(ctx, hipEventRecord(P.ev_prep[set], st_pre));
    VB_HIP(ctx, hipStreamWaitEvent(st_main, P.ev_prep[set], 0));
  }

  Launch L;
  L.grid = dim3((unsigned)(g.n_rb * g.n_cb), (unsigned)c.count);
  L.st = st_main;
  prof_events(ctx, &L.ev0, &L.ev1, c.count);
  if (source)
    VB_TRY(source_accumulate(ctx, st_main, *c.noise[0], bp, ws, g, mom, tsc, c.roww[0]));
  else if (logistic)
    VB_TRY(logistic_accumulate(ctx, st_main, model, *c.noise[0], bp, ws, g, mom, tsc));
  else if (model.id == VB_MODEL_GAUSS_DIAG)
    launch_accum_model<VB_MODEL_GAUSS_DIAG>(mom, tsc, weighted, L, bp, ws, g);
  else if (model.id == VB_MODEL_FUNNEL)
    launch_accum_model<VB_MODEL_FUNNEL>(mom, tsc, weighted, L, bp, ws, g);
  else if (tsc)
    launch_accum<kModelLogQ, true, true>(true, L, bp, ws, g);
  else
    launch_accum<kModelLogQ, true, false>(true, L, bp, ws, g);
  VB_HIP(ctx, hipGetLastError());
  if (c.pipelined) {
    VB_HIP(ctx, hipEventRecord(P.ev_k1[set], st_main));
    VB_HIP(ctx, hipStreamWaitEvent(st_post, P.ev_k1[set], 0));
  }

  EpiArgs e;
  memset(&e, 0, sizeof e);
  e.n_rb = g.n_rb;
  e.n_ps = g.n_rb * g.n_cb;
  e.n_prep = g.n_prep;
  e.nf = nf;
  e.d = (int)d;
  e.Dp = g.Dp;
  e.n_total = (double)c.n_total;
  e.family = c.family;
  e.df = c.df;
  e.flags = c.flags;
  e.cv_mode = c.cv_mode;
  e.mode = c.mode;
  e.scale = c.scale;
  e.value_src = c.value_src;
  e.model = model;
  const bool fused = c.mode == 0 && c.cv_mode == VB_CV_NONE && !ctx->comm;
  e.reduce_only = fused ? 0 : 1;
  if (c.step && c.step_done) {
    *c.step_done = fused && c.count == 1;
    if (*c.step_done) {
      e.has_step = 1;
      e.step = *c.step;
    }
  }
  if (c.prep_done) {
    // the row part (funnel) is done by ONE workgroup, block after block: worth it only while that is shorter than a
    // launch (measured: 16 blocks add 17 us at N = 4096, four or fewer are hidden)
    *c.prep_done = c.prep_next && e.has_step && g.gen && !weighted && !logistic && (!g.rows || g.n_prep <= 4);
    if (*c.prep_done) {
      e.prep_next = 1;
      const int next_set = (int)(ctx->pipe.seq % kPipeSets);   // the set the next mf_enqueue will pick
      e.next_wsb = (double*)ctx->workspace.ptr + (size_t)next_set * ws.stride * kMaxBatch;
      e.next_g = g;
      const uint64_t next_stream = c.gen_stream + 1;
      e.next_g.gk1 = (uint32_t)(c.gen_seed >> 32) ^ (uint32_t)(next_stream >> 32);
      e.next_g.gw = (uint32_t)next_stream;
    }
  }
  if (c.done_groups) *c.done_groups = 0;
  if (c.done_dev && c.done_groups && fused && c.count == 1 && g.Dp / 64 <= 64 && !e.has_step) {
    e.done = c.done_dev;
    e.done_seq = c.done_seq;
    *c.done_groups = g.Dp / 64;
  }
  hipLaunchKernelGGL(mf_finalize_kernel, dim3((unsigned)(g.Dp / 64), (unsigned)c.count), dim3(256), 0,
                     st_post, e, bp, ws);
  VB_HIP(ctx, hipGetLastError());
  if (overlap) {   // hand the reduced sums over to the communication stream
    VB_HIP(ctx, hipEventRecord(P.ev_k1[set], ctx->stream));
    st_post = P.post;
    VB_HIP(ctx, hipStreamWaitEvent(st_post, P.ev_k1[set], 0));
  }
  if (!fused) {
    if (ctx->comm)   // one all-reduce for the whole batch: [count][sum_len] doubles
      VB_TRY(comm_allreduce_sum(ctx, st_post, ws.sums, (size_t)ws.sum_len * c.count));
    hipLaunchKernelGGL(mf_epilogue_kernel, dim3(1, (unsigned)c.count), dim3(256), 0, st_post, e, bp, ws);
    VB_HIP(ctx, hipGetLastError());
  }
  if (c.pipelined || overlap || alt_b) {
    VB_HIP(ctx, hipEventRecord(P.ev_fin[set], st_post));
    P.fin_valid[set] = true;
    P.post_pending = true;
    P.last_set = set;
  } else if (alt) {
    // main-stream batch of an alternating pair: nothing on `post` depends on it
  } else {
    P.fin_valid[set] = false;   // ordered by the main stream itself ...
    P.main_dirty = true;        // ... which a later pipelined prep must wait for
  }
  ctx->result_stream = st_post;
  return VB_OK;
}

int pipe_init(vb_ctx* ctx) {
  Pipeline& P = ctx->pipe;
  if (P.pre) return VB_OK;
  VB_HIP(ctx, hipStreamCreateWithFlags(&P.pre, hipStreamNonBlocking));
  VB_HIP(ctx, hipStreamCreateWithFlags(&P.post, hipStreamNonBlocking));
  VB_HIP(ctx, hipEventCreateWithFlags(&P.ev_main, hipEventDisableTiming));
  for (int i = 0; i < kPipeSets; ++i) {
    VB_HIP(ctx, hipEventCreateWithFlags(&P.ev_prep[i], hipEventDisableTiming));
    VB_HIP(ctx, hipEventCreateWithFlags(&P.ev_k1[i], hipEventDisableTiming));
    VB_HIP(ctx, hipEventCreateWithFlags(&P.ev_fin[i], hipEventDisableTiming));
  }
  return VB_OK;
}

}  // namespace vb
