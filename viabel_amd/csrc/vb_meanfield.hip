// Mean-field ExclusiveKL / weighted-gradient pipeline for gfx950 (MI355X).
//
// Replaces, for MFGaussian / MFStudentT, the whole body of
//   viabel/objectives.py:154-168 (plain + path-derivative ELBO, autograd backward pass) and
//   viabel/objectives.py:170-271 (RGE control variates),
// i.e. sample (approximations.py:212-216) + model log density and gradient + entropy /
// log q + the Monte-Carlo mean, in ONE streaming pass over the noise matrix.
//
// Kernels
//   mf_accum_kernel   HBM-bound.  Grid = row-blocks x column-blocks; a workgroup (4 waves)
//                     owns 128 columns (one 1-KiB, 16-B-per-lane coalesced wave load per
//                     row) and a block of rows; each wave keeps 16 rows (16 KiB) in flight.
//                     Per-column sums live in registers, (mu, sigma, model column
//                     parameters) are read once per lane, row-coupled model scalars
//                     (funnel: exp(-2 v_n)) are computed by 16 lanes and broadcast with
//                     v_readlane, the 4 waves are combined through LDS, and the workgroup
//                     writes one partial per column.  No atomics: deterministic.
//                     blockIdx -> (row-block, column-block) is XCD-aware: the column blocks
//                     of one row block share an XCD (block b runs on XCD b % 8), so the
//                     funnel's broadcast column is fetched into one L2 only.
//   mf_reduce_kernel  sums the row-block partials (fixed order) into the sum vector that a
//                     multi-GPU job all-reduces.
//   mf_epilogue_kernel  O(D): turns sums into (value, grad) for every estimator variant.
//
// Algorithmic HBM bytes per evaluation: N*D*8 (noise) + 4*D*8 (theta, grad) + 8.
#include "vb_common.h"

namespace vb {

typedef double d2 __attribute__((ext_vector_type(2)));

struct MfArgs {
  const double* noise;
  int64_t ld;
  int64_t n;
  int d;
  int Dp;               // n_cb * kMfCols
  const double* theta;  // [mu | log_sigma]
  const double* roww;   // per-row weights (WEIGHTED) or nullptr
  double* partials;     // [n_rb][CF_NUM][Dp]
  double* pscal;        // [n_rb][n_cb][SF_NUM]
  int rows_per_wg;
  int n_rb;
  int n_cb;
  int xcd_map;
  ModelDev model;
  double df;
};

struct ColP {   // per-column constants held in registers
  double mu, sg;
  double m, iv;   // gauss_diag
  bool isK;       // funnel: this is the log-scale column
};
struct ColAcc {
  double G = 0, GE = 0, E = 0, EE = 0, EK = 0, SC = 0, SCE = 0;
};
struct ScalAcc {
  double F = 0, W = 0, Q = 0, QE = 0, L1P = 0;
};
struct RowP {
  double w, ek, wt;
};

__device__ __forceinline__ double bcast(double x, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

template <int MODEL, bool MOM, bool TSC, bool USEWT>
__device__ __forceinline__ void accum(const double e, const ColP& c, const RowP& r,
                                      const double inv_tau2, const double dm1, const double df,
                                      ColAcc& A, ScalAcc& S) {
  const double z = fma(c.sg, e, c.mu);
  double g, fc, q = 0.0;
  if (MODEL == VB_MODEL_GAUSS_DIAG) {
    const double dz = z - c.m;
    g = -dz * c.iv;
    fc = 0.5 * dz * g;
  } else {   // funnel
    if (c.isK) {
      g = fma(-z, inv_tau2, -dm1);
      fc = z * fma(-0.5 * z, inv_tau2, -dm1);
    } else {
      g = -z * r.w;
      q = -z * g;
      fc = -0.5 * q;
    }
  }
  double ew = e;
  if (USEWT) {
    g *= r.wt;
    fc *= r.wt;
    q *= r.wt;
    ew = e * r.wt;
  }
  A.G += g;
  A.GE = fma(g, e, A.GE);
  S.F += fc;
  if (MODEL == VB_MODEL_FUNNEL) {
    S.Q += q;
    S.QE = fma(q, r.ek, S.QE);
  }
  if (MOM) {
    A.E += ew;
    A.EE = fma(ew, e, A.EE);
    if (MODEL == VB_MODEL_FUNNEL) A.EK = fma(ew, r.ek, A.EK);
  }
  if (TSC) {
    const double e2 = e * e;
    double sc = (df + 1.0) * e / (df + e2);
    double l1p = log1p(e2 / df);
    if (USEWT) {
      sc *= r.wt;
      l1p *= r.wt;
    }
    A.SC += sc;
    A.SCE = fma(sc, e, A.SCE);
    S.L1P += l1p;
  }
}

template <int MODEL, bool MOM, bool TSC, bool WEIGHTED>
__global__ void __launch_bounds__(kMfThreads) mf_accum_kernel(const MfArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  int rb, cb;
  {
    const int b = blockIdx.x;
    if (a.xcd_map) {   // all column blocks of a row block on one XCD (block b -> XCD b % 8)
      const int x = b & 7, j = b >> 3;
      cb = j % a.n_cb;
      rb = (j / a.n_cb) * 8 + x;
    } else {
      rb = b / a.n_cb;
      cb = b % a.n_cb;
    }
  }
  const int d = a.d;
  const int c0 = cb * kMfCols + 2 * lane;
  const bool val0 = c0 < d, val1 = c0 + 1 < d;
  const int i0 = val0 ? c0 : 0, i1 = val1 ? c0 + 1 : 0;

  ColP p0, p1;
  p0.mu = a.theta[i0];
  p1.mu = a.theta[i1];
  p0.sg = exp(a.theta[d + i0]);
  p1.sg = exp(a.theta[d + i1]);
  p0.m = p1.m = p0.iv = p1.iv = 0.0;
  p0.isK = p1.isK = false;
  double muk = 0.0, sgk = 0.0, inv_tau2 = 0.0, dm1 = 0.0;
  int kcol = 0;
  if (MODEL == VB_MODEL_GAUSS_DIAG) {
    p0.m = a.model.p0[i0];
    p1.m = a.model.p0[i1];
    p0.iv = a.model.p1[i0];
    p1.iv = a.model.p1[i1];
  } else {
    kcol = a.model.k;
    muk = a.theta[kcol];
    sgk = exp(a.theta[d + kcol]);
    inv_tau2 = 1.0 / (a.model.tau * a.model.tau);
    dm1 = (double)(d - 1);
    p0.isK = (c0 == kcol);
    p1.isK = (c0 + 1 == kcol);
  }
  const double df = a.df;

  ColAcc A0, A1;
  ScalAcc S;

  const int64_t r0 = (int64_t)rb * a.rows_per_wg;
  const int64_t r1 = (r0 + a.rows_per_wg < a.n) ? r0 + a.rows_per_wg : a.n;
  const double* __restrict__ noise = a.noise;
  const int64_t ld = a.ld;

  for (int64_t base = r0 + wave; base < r1; base += (int64_t)kMfWaves * kMfChunk) {
    const bool full = base + (int64_t)kMfWaves * (kMfChunk - 1) < r1;
    // ---- phase A: per-row scalars, one row per lane (lanes 0..15) -------------------------
    RowP rs;
    {
      const int64_t r = base + (int64_t)kMfWaves * (lane & (kMfChunk - 1));
      const bool ok = r < r1;
      rs.wt = ok ? 1.0 : 0.0;
      if (WEIGHTED) rs.wt = ok ? a.roww[r] : 0.0;
      rs.ek = 0.0;
      rs.w = 0.0;
      if (MODEL == VB_MODEL_FUNNEL) {
        rs.ek = ok ? noise[r * ld + kcol] : 0.0;
        rs.w = exp(-2.0 * fma(sgk, rs.ek, muk));
      }
      if (cb == 0 && lane < kMfChunk) S.W += rs.wt;
    }
    // ---- phase B: 16 coalesced 1-KiB row segments in flight per wave ----------------------
    d2 e[kMfChunk];
    if (full) {
      if (val0) {
#pragma unroll
        for (int j = 0; j < kMfChunk; ++j)
          e[j] = __builtin_nontemporal_load(
              reinterpret_cast<const d2*>(noise + (base + (int64_t)kMfWaves * j) * ld + c0));
      }
#pragma unroll
      for (int j = 0; j < kMfChunk; ++j) {
        RowP r;
        r.w = bcast(rs.w, j);
        r.ek = bcast(rs.ek, j);
        r.wt = WEIGHTED ? bcast(rs.wt, j) : 1.0;
        if (val0) accum<MODEL, MOM, TSC, WEIGHTED>(e[j].x, p0, r, inv_tau2, dm1, df, A0, S);
        if (val1) accum<MODEL, MOM, TSC, WEIGHTED>(e[j].y, p1, r, inv_tau2, dm1, df, A1, S);
      }
    } else {
#pragma unroll
      for (int j = 0; j < kMfChunk; ++j) {
        const int64_t r = base + (int64_t)kMfWaves * j;
        e[j] = (d2){0.0, 0.0};
        if (val0 && r < r1)
          e[j] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(noise + r * ld + c0));
      }
#pragma unroll
      for (int j = 0; j < kMfChunk; ++j) {
        RowP r;
        r.w = bcast(rs.w, j);
        r.ek = bcast(rs.ek, j);
        r.wt = bcast(rs.wt, j);
        if (val0) accum<MODEL, MOM, TSC, true>(e[j].x, p0, r, inv_tau2, dm1, df, A0, S);
        if (val1) accum<MODEL, MOM, TSC, true>(e[j].y, p1, r, inv_tau2, dm1, df, A1, S);
      }
    }
  }

  // ---- combine the 4 waves through LDS, one partial per column per workgroup ---------------
  __shared__ d2 red[CF_NUM][kMfWaves][kWave];
  __shared__ double reds[kMfWaves][SF_NUM];
  red[CF_G][wave][lane] = (d2){A0.G, A1.G};
  red[CF_GE][wave][lane] = (d2){A0.GE, A1.GE};
  red[CF_E][wave][lane] = (d2){A0.E, A1.E};
  red[CF_EE][wave][lane] = (d2){A0.EE, A1.EE};
  red[CF_EK][wave][lane] = (d2){A0.EK, A1.EK};
  red[CF_SC][wave][lane] = (d2){A0.SC, A1.SC};
  red[CF_SCE][wave][lane] = (d2){A0.SCE, A1.SCE};
  {
    const double f = wave_sum(S.F), w = wave_sum(S.W), q = wave_sum(S.Q), qe = wave_sum(S.QE),
                 l = wave_sum(S.L1P);
    if (lane == 0) {
      reds[wave][SF_F] = f;
      reds[wave][SF_W] = w;
      reds[wave][SF_Q] = q;
      reds[wave][SF_QE] = qe;
      reds[wave][SF_L1P] = l;
      reds[wave][5] = reds[wave][6] = reds[wave][7] = 0.0;
    }
  }
  __syncthreads();
  for (int f = wave; f < CF_NUM; f += kMfWaves) {
    d2 s = red[f][0][lane];
#pragma unroll
    for (int w = 1; w < kMfWaves; ++w) s += red[f][w][lane];
    *reinterpret_cast<d2*>(a.partials + ((int64_t)rb * CF_NUM + f) * a.Dp + cb * kMfCols + 2 * lane) = s;
  }
  if (threadIdx.x < SF_NUM) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kMfWaves; ++w) s += reds[w][threadIdx.x];
    a.pscal[((int64_t)rb * a.n_cb + cb) * SF_NUM + threadIdx.x] = s;
  }
}

// sums = [SF_NUM scalars | nf x Dp column sums]
__global__ void __launch_bounds__(256) mf_reduce_kernel(const double* __restrict__ partials,
                                                        const double* __restrict__ pscal,
                                                        double* __restrict__ sums, int n_rb,
                                                        int n_cb, int Dp, int nf) {
  const int64_t total = (int64_t)nf * Dp;
  if (blockIdx.x == gridDim.x - 1) {   // scalars
    __shared__ double sh[256][SF_NUM];
    double acc[SF_NUM];
#pragma unroll
    for (int s = 0; s < SF_NUM; ++s) acc[s] = 0.0;
    const int entries = n_rb * n_cb;
    for (int e = threadIdx.x; e < entries; e += 256) {
#pragma unroll
      for (int s = 0; s < SF_NUM; ++s) acc[s] += pscal[(int64_t)e * SF_NUM + s];
    }
#pragma unroll
    for (int s = 0; s < SF_NUM; ++s) sh[threadIdx.x][s] = acc[s];
    __syncthreads();
    for (int stride = 128; stride > 0; stride >>= 1) {
      if ((int)threadIdx.x < stride) {
#pragma unroll
        for (int s = 0; s < SF_NUM; ++s) sh[threadIdx.x][s] += sh[threadIdx.x + stride][s];
      }
      __syncthreads();
    }
    if (threadIdx.x < SF_NUM) sums[threadIdx.x] = sh[0][threadIdx.x];
    return;
  }
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int f = (int)(idx / Dp);
  const int c = (int)(idx % Dp);
  const double* p = partials + (int64_t)f * Dp + c;
  const int64_t stride = (int64_t)CF_NUM * Dp;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int rb = 0;
  for (; rb + 4 <= n_rb; rb += 4) {
    s0 += p[(rb + 0) * stride];
    s1 += p[(rb + 1) * stride];
    s2 += p[(rb + 2) * stride];
    s3 += p[(rb + 3) * stride];
  }
  for (; rb < n_rb; ++rb) s0 += p[rb * stride];
  sums[SF_NUM + idx] = (s0 + s1) + (s2 + s3);
}

struct EpiArgs {
  const double* sums;
  const double* theta;
  double* out;   // [value | grad(2D)]
  int d, Dp;
  double n_total;
  int family;
  double df;
  unsigned flags;
  int cv_mode;
  int mode;       // 0: ELBO (ExclusiveKL); 1: weighted gradient only (alpha / scale given)
  double scale;   // mode 1: grad = scale * [G | GE*sigma + W]
  ModelDev model;
};

__device__ double block_sum(double x, double* sh) {
  x = wave_sum(x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = x;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  return t;
}

constexpr double kLog2Pi = 1.8378770664093454835606594728112;

__global__ void __launch_bounds__(256) mf_epilogue_kernel(const EpiArgs a) {
  __shared__ double sh[8];
  const int d = a.d, Dp = a.Dp;
  const double* S = a.sums;
  const double* G = a.sums + SF_NUM + (int64_t)CF_G * Dp;
  const double* GE = a.sums + SF_NUM + (int64_t)CF_GE * Dp;
  const double* E = a.sums + SF_NUM + (int64_t)CF_E * Dp;
  const double* EE = a.sums + SF_NUM + (int64_t)CF_EE * Dp;
  const double* EK = a.sums + SF_NUM + (int64_t)CF_EK * Dp;
  const double* SC = a.sums + SF_NUM + (int64_t)CF_SC * Dp;
  const double* SCE = a.sums + SF_NUM + (int64_t)CF_SCE * Dp;
  const double* mu = a.theta;
  const double* ls = a.theta + d;
  const bool funnel = a.model.id == VB_MODEL_FUNNEL;
  const int k = a.model.k;
  const double Wsum = S[SF_W];
  const double gk_add = funnel ? S[SF_Q] : 0.0;     // row-coupled part of g_k (see accum)
  const double gek_add = funnel ? S[SF_QE] : 0.0;
  double* value = a.out;
  double* gmu = a.out + 1;
  double* gls = a.out + 1 + d;

  if (a.mode == 1) {   // weighted gradient only: scale * sum_n w_n [g_n | g_n eps_n sigma + 1]
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
      const double g = G[i] + (i == k ? gk_add : 0.0);
      const double ge = GE[i] + (i == k ? gek_add : 0.0);
      gmu[i] = a.scale * g;
      gls[i] = a.scale * (ge * exp(ls[i]) + Wsum);
    }
    if (threadIdx.x == 0) value[0] = S[SF_F] + Wsum * a.model.c0;
    return;
  }

  const double invN = 1.0 / a.n_total;
  const bool student = a.family == VB_FAMILY_MF_STUDENT_T;
  const bool pd = (a.flags & VB_FLAG_PATH_DERIV) != 0;

  // ---- value -----------------------------------------------------------------------------
  double t_ls = 0.0, t_ee = 0.0;
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    t_ls += ls[i];
    if (pd && !student) t_ee += EE[i];
  }
  const double sum_ls = block_sum(t_ls, sh);
  const double sum_ee = block_sum(t_ee, sh);
  const double F = S[SF_F] + Wsum * a.model.c0;
  if (threadIdx.x == 0) {
    double v;
    if (pd) {   // objectives.py:156-159: -mean(f - log q(theta_stop; z))
      double logq;
      if (student) {
        const double ct = lgamma(0.5 * (a.df + 1.0)) - lgamma(0.5 * a.df) - 0.5 * log(a.df * M_PI);
        logq = Wsum * (d * ct - sum_ls) - 0.5 * (a.df + 1.0) * S[SF_L1P];
      } else {
        logq = -0.5 * sum_ee - Wsum * (0.5 * d * kLog2Pi + sum_ls);
      }
      v = -(F - logq) * invN;
    } else {    // objectives.py:160-161: -(mean f + entropy)
      const double H = (student ? 0.0 : 0.5 * d * (1.0 + kLog2Pi)) + sum_ls;
      v = -(F * invN + H);
    }
    value[0] = v;
  }

  // ---- gradient, no control variate (what autograd returns for objectives.py:154-164) -----
  if (a.cv_mode == VB_CV_NONE) {
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
      const double sg = exp(ls[i]);
      const double g = G[i] + (i == k ? gk_add : 0.0);
      const double ge = GE[i] + (i == k ? gek_add : 0.0);
      if (pd) {
        const double sc = student ? SC[i] : E[i];
        const double sce = student ? SCE[i] : EE[i];
        gmu[i] = -(g + sc / sg) * invN;
        gls[i] = -(ge * sg + sce) * invN;
      } else {
        gmu[i] = -g * invN;
        gls[i] = -(ge * sg * invN + 1.0);
      }
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

// ---------------------------------------------------------------------------------------------
template <int MODEL, bool MOM, bool TSC>
static void launch_accum(bool weighted, dim3 grid, hipStream_t st, const MfArgs& a) {
  if (weighted)
    hipLaunchKernelGGL((mf_accum_kernel<MODEL, MOM, TSC, true>), grid, dim3(kMfThreads), 0, st, a);
  else
    hipLaunchKernelGGL((mf_accum_kernel<MODEL, MOM, TSC, false>), grid, dim3(kMfThreads), 0, st, a);
}

template <int MODEL>
static void launch_accum_model(bool mom, bool tsc, bool weighted, dim3 grid, hipStream_t st,
                               const MfArgs& a) {
  if (mom && tsc) launch_accum<MODEL, true, true>(weighted, grid, st, a);
  else if (mom) launch_accum<MODEL, true, false>(weighted, grid, st, a);
  else if (tsc) launch_accum<MODEL, false, true>(weighted, grid, st, a);
  else launch_accum<MODEL, false, false>(weighted, grid, st, a);
}

static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

// Enqueue accumulate -> reduce -> [all-reduce] -> epilogue on ctx->stream.  ctx->theta holds
// the variational parameter; the result lands in ctx->out = [value | grad(2D)].
int mf_elbo_grad_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int64_t n_total,
                         int family, double df, unsigned flags, int cv_mode, const double* roww,
                         int mode, double scale) {
  if (ctx->model.id != VB_MODEL_GAUSS_DIAG && ctx->model.id != VB_MODEL_FUNNEL)
    return fail(ctx, VB_ERR_UNSUPPORTED,
                "mean-field path supports the gauss_diag and funnel models (model id %d bound)",
                ctx->model.id);
  if (ctx->model.dim != d)
    return fail(ctx, VB_ERR_INVALID, "model dimension %d != family dimension %lld", ctx->model.dim,
                (long long)d);
  if (family != VB_FAMILY_MF_GAUSSIAN && family != VB_FAMILY_MF_STUDENT_T)
    return fail(ctx, VB_ERR_INVALID, "family %d is not a mean-field family", family);
  if (family == VB_FAMILY_MF_STUDENT_T && !(df > 2.0))
    return fail(ctx, VB_ERR_INVALID, "df must be greater than 2");
  if (cv_mode < VB_CV_NONE || cv_mode > VB_CV_LOO_DIRECT)
    return fail(ctx, VB_ERR_INVALID, "unknown control-variate mode %d", cv_mode);
  if (n <= 0 || d <= 0 || n > ns.n || d != ns.d)
    return fail(ctx, VB_ERR_INVALID, "noise slot holds %lld x %lld, evaluation asks %lld x %lld",
                (long long)ns.n, (long long)ns.d, (long long)n, (long long)d);

  const bool pd = (flags & VB_FLAG_PATH_DERIV) != 0;
  const bool student = family == VB_FAMILY_MF_STUDENT_T;
  const bool mom = mode == 0 && ((pd && !student) || cv_mode != VB_CV_NONE);
  const bool tsc = mode == 0 && pd && student;
  const int nf = tsc ? CF_NUM : (mom ? CF_EK + 1 : CF_GE + 1);

  const int n_cb = (int)((d + kMfCols - 1) / kMfCols);
  const int Dp = n_cb * kMfCols;
  // ~2 workgroups per CU; rows per workgroup a multiple of the 4 waves
  const int target_wg = env_int("VB_MF_TARGET_WG", 2 * ctx->prop.multiProcessorCount);
  int n_rb_target = target_wg / n_cb;
  if (n_rb_target < 8) n_rb_target = 8;
  n_rb_target = (n_rb_target + 7) / 8 * 8;
  int rows_per_wg = (int)((n + n_rb_target - 1) / n_rb_target);
  rows_per_wg = env_int("VB_MF_ROWS_PER_WG", rows_per_wg);
  rows_per_wg = (int)round_up(rows_per_wg < kMfWaves ? kMfWaves : rows_per_wg, kMfWaves);
  const int n_rb = (int)((n + rows_per_wg - 1) / rows_per_wg);

  VB_TRY(ensure(ctx, ctx->partials,
                ((size_t)n_rb * CF_NUM * Dp + (size_t)n_rb * n_cb * SF_NUM) * sizeof(double)));
  VB_TRY(ensure(ctx, ctx->sums, ((size_t)SF_NUM + (size_t)CF_NUM * Dp) * sizeof(double)));
  VB_TRY(ensure(ctx, ctx->out, (size_t)(1 + 2 * d) * sizeof(double)));

  MfArgs a;
  a.noise = (const double*)ns.buf.ptr;
  a.ld = ns.ld;
  a.n = n;
  a.d = (int)d;
  a.Dp = Dp;
  a.theta = (const double*)ctx->theta.ptr;
  a.roww = roww;
  a.partials = (double*)ctx->partials.ptr;
  a.pscal = a.partials + (size_t)n_rb * CF_NUM * Dp;
  a.rows_per_wg = rows_per_wg;
  a.n_rb = n_rb;
  a.n_cb = n_cb;
  a.xcd_map = (n_rb % 8 == 0) ? env_int("VB_MF_XCD_MAP", 1) : 0;
  a.model = ctx->model;
  a.df = df;

  const dim3 grid((unsigned)(n_rb * n_cb));
  const bool weighted = roww != nullptr;
  prof_begin(ctx);
  if (ctx->model.id == VB_MODEL_GAUSS_DIAG)
    launch_accum_model<VB_MODEL_GAUSS_DIAG>(mom, tsc, weighted, grid, ctx->stream, a);
  else
    launch_accum_model<VB_MODEL_FUNNEL>(mom, tsc, weighted, grid, ctx->stream, a);
  prof_end(ctx);
  VB_HIP(ctx, hipGetLastError());

  const int64_t total = (int64_t)nf * Dp;
  const unsigned rgrid = (unsigned)((total + 255) / 256 + 1);
  hipLaunchKernelGGL(mf_reduce_kernel, dim3(rgrid), dim3(256), 0, ctx->stream,
                     (const double*)a.partials, (const double*)a.pscal, (double*)ctx->sums.ptr,
                     n_rb, n_cb, Dp, nf);
  VB_HIP(ctx, hipGetLastError());

  if (ctx->comm) VB_TRY(comm_allreduce_sum(ctx, (double*)ctx->sums.ptr, (size_t)SF_NUM + (size_t)total));

  EpiArgs e;
  e.sums = (const double*)ctx->sums.ptr;
  e.theta = (const double*)ctx->theta.ptr;
  e.out = (double*)ctx->out.ptr;
  e.d = (int)d;
  e.Dp = Dp;
  e.n_total = (double)n_total;
  e.family = family;
  e.df = df;
  e.flags = flags;
  e.cv_mode = cv_mode;
  e.mode = mode;
  e.scale = scale;
  e.model = ctx->model;
  hipLaunchKernelGGL(mf_epilogue_kernel, dim3(1), dim3(256), 0, ctx->stream, e);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

}  // namespace vb
