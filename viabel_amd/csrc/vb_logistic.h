// Bayesian logistic-regression target for the mean-field ExclusiveKL path (BASELINE configs[4]).
//
// Not in the reference (viabel/stan_models holds no logistic regression); the model is
//   f(z) = sum_i [ y_i eta_i - log(1 + exp(eta_i)) ] + sum_d norm.logpdf(z_d; 0, prior_sd),  eta = X z
// with the prior scale of the reference's Stan test model (viabel/tests/test_models.py:41).  Unlike the
// diagonal Gaussian and the funnel it couples all coordinates through X, so the per-sample gradient needs
// two GEMMs (fp64 MFMA) instead of elementwise work:
//   Z = mu + sigma * E                                  elementwise                     [N x D]
//   H = Z X'            epilogue: R = y - sigmoid(H), sum of log-likelihood terms       [N x n_data x D]
//   G = R X - Z / sd^2                                                                  [N x D x n_data]
// after which the streaming pass only differs from the other models in that g_nd is LOADED (second
// stream) instead of computed; it writes the same partial-sum layout, so the finalize / epilogue
// kernels of the mean-field pipeline are reused unchanged.
// Included by vb_meanfield.hip (needs its Workspace / Geom / EpiArgs definitions).
#pragma once

#include "vb_gemm_f64.h"

namespace vb {

// theta -> device copy, columns [mu | sigma]; Z = mu + sigma * E
__global__ void __launch_bounds__(256) lg_sample_kernel(const double* __restrict__ theta_src, double* theta_dev,
                                                        double* __restrict__ colp, int Dp,
                                                        const double* __restrict__ noise, int64_t ld,
                                                        double* __restrict__ Z, int64_t ldz, int64_t n, int d) {
  const int64_t row = blockIdx.x;            // rows on x: gridDim.y stops at 65 535
  const int col = blockIdx.y * 256 + threadIdx.x;
  if (col >= d) return;
  const double mu = theta_src[col], ls = theta_src[d + col];
  const double sg = exp(ls);
  if (row == 0) {
    theta_dev[col] = mu;
    theta_dev[d + col] = ls;
    colp[col] = mu;
    colp[Dp + col] = sg;
  }
  Z[row * ldz + col] = fma(sg, noise[row * ld + col], mu);
}

struct EpiLogit {           // R = d loglik / d eta (logit link: y - sigmoid(eta)); returns the log-likelihood term
  double* R;
  int64_t ldr;
  const double* y;
  double* part;
  int link;
  double aux;
  __device__ double operator()(int, int row, int col, double eta) const {
    double dl;
    const double ll = glm_term(link, aux, y[col], eta, &dl);
    R[(int64_t)row * ldr + col] = dl;
    return ll;
  }
};

// sum the GEMM's per-workgroup log-likelihood partials; also the prep-kernel scalars (W = n, or the sum of the row
// weights of a weighted evaluation)
__global__ void __launch_bounds__(256) lg_scalars_kernel(const double* __restrict__ part, int n_part,
                                                         double* __restrict__ fsum, double* __restrict__ prepscal,
                                                         double n, const double* __restrict__ roww = nullptr,
                                                         int64_t n_rows = 0) {
  __shared__ double sh[4], shw[4];
  double s = 0.0, w = 0.0;
  for (int i = threadIdx.x; i < n_part; i += 256) s += part[i];
  if (roww)
    for (int64_t i = threadIdx.x; i < n_rows; i += 256) w += roww[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off, 64);
    w += __shfl_down(w, off, 64);
  }
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s, shw[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) fsum[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  const double W = roww ? (shw[0] + shw[1]) + (shw[2] + shw[3]) : n;
  if (threadIdx.x < PS_NUM) prepscal[threadIdx.x] = threadIdx.x == PS_W ? W : 0.0;   // n_prep = 1
}

// G[n][:] *= w_n: the weighted sums of a loaded gradient matrix are the plain sums of the scaled one
__global__ void __launch_bounds__(256) lg_rowscale_kernel(double* __restrict__ G, int64_t ldz, int64_t n, int d,
                                                          const double* __restrict__ roww) {
  const int64_t row = blockIdx.x;
  const int col = blockIdx.y * 256 + threadIdx.x;
  if (col < d) G[row * ldz + col] *= roww[row];
}

// Streaming pass with an explicit gradient matrix: same grid / partial layout as mf_accum_kernel.
// Per element: G += g, GE += g e, (E += e, EE += e^2), F += -1/2 z^2 / sd^2 (prior), z = mu + sigma e.
template <bool MOM, bool TSC>
__global__ void __launch_bounds__(kMfThreads)
lg_accum_kernel(const double* __restrict__ noise, int64_t ld, const double* __restrict__ Gm, int64_t ldz,
                const Workspace ws, const Geom g, double ivp, const double* __restrict__ fsum) {
  typedef double d2x __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rb = blockIdx.x / g.n_cb, cb = blockIdx.x % g.n_cb;
  double* wsb = ws.base;
  const double* colp = wsb + ws.off_colp;
  const int c0i = cb * kMfCols + 2 * lane;
  const bool ok = c0i < ld && c0i < ldz;
  const int64_t r0 = (int64_t)rb * g.rows_per_wg;
  const int64_t r1 = (r0 + g.rows_per_wg < g.n) ? r0 + g.rows_per_wg : g.n;
  d2x mu = (d2x){0.0, 0.0}, sg = (d2x){0.0, 0.0};
  if (c0i < g.Dp) {
    mu = *reinterpret_cast<const d2x*>(colp + c0i);
    sg = *reinterpret_cast<const d2x*>(colp + g.Dp + c0i);
  }
  d2x aG = (d2x){0, 0}, aGE = (d2x){0, 0}, aE = (d2x){0, 0}, aEE = (d2x){0, 0}, aSC = (d2x){0, 0},
      aSCE = (d2x){0, 0};
  double F = 0.0, L1P = 0.0;
  const double df = g.df;
  for (int64_t base = r0 + wave; base < r1; base += (int64_t)kMfWaves * 8) {
    d2x e[8], gg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t r = base + (int64_t)kMfWaves * j;
      e[j] = (d2x){0.0, 0.0};
      gg[j] = (d2x){0.0, 0.0};
      if (r < r1 && ok) {
        e[j] = __builtin_nontemporal_load(reinterpret_cast<const d2x*>(noise + r * ld + c0i));
        gg[j] = __builtin_nontemporal_load(reinterpret_cast<const d2x*>(Gm + r * ldz + c0i));
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (base + (int64_t)kMfWaves * j >= r1) break;
      const d2x z = sg * e[j] + mu;
      aG += gg[j];
      aGE += gg[j] * e[j];
      F -= 0.5 * ivp * (z.x * z.x + z.y * z.y);
      if (MOM) {
        aE += e[j];
        aEE += e[j] * e[j];
      }
      if (TSC) {
        const d2x e2 = e[j] * e[j];
        const d2x sc = (df + 1.0) * e[j] / (df + e2);
        aSC += sc;
        aSCE += sc * e[j];
        L1P += log1p(e2.x / df) + log1p(e2.y / df);
      }
    }
  }
  // columns beyond d hold z = mu = 0 pads (colp pads are zero) and contribute nothing
  constexpr int NF = TSC ? CF_NUM : (MOM ? CF_EK + 1 : CF_GE + 1);
  __shared__ d2x red[NF][kMfWaves][kWave];
  __shared__ double reds[kMfWaves][KS_NUM];
  red[CF_G][wave][lane] = aG;
  red[CF_GE][wave][lane] = aGE;
  if (NF > CF_E) {
    red[CF_E][wave][lane] = aE;
    red[CF_EE][wave][lane] = aEE;
    red[CF_EK][wave][lane] = (d2x){0.0, 0.0};
  }
  if (NF > CF_SC) {
    red[CF_SC][wave][lane] = aSC;
    red[CF_SCE][wave][lane] = aSCE;
  }
  {
    double f = F, l = L1P, ee = aEE.x + aEE.y;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      f += __shfl_down(f, off, 64);
      l += __shfl_down(l, off, 64);
      ee += __shfl_down(ee, off, 64);
    }
    if (lane == 0) {
      reds[wave][KS_F] = f;
      reds[wave][KS_Q] = 0.0;
      reds[wave][KS_QE] = 0.0;
      reds[wave][KS_L1P] = l;
      reds[wave][KS_EE] = ee;
    }
  }
  __syncthreads();
  if (c0i < g.Dp) {
    for (int f = wave; f < NF; f += kMfWaves) {
      d2x s = red[f][0][lane];
#pragma unroll
      for (int w = 1; w < kMfWaves; ++w) s += red[f][w][lane];
      *reinterpret_cast<d2x*>(wsb + ws.off_partials + partial_index(rb, f, c0i, g.n_rb)) = s;
    }
  }
  if (threadIdx.x < KS_NUM) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kMfWaves; ++w) s += reds[w][threadIdx.x];
    if (threadIdx.x == KS_F && blockIdx.x == 0) s += fsum[0];   // the GEMM's log-likelihood total, once
    wsb[ws.off_pscal + (int64_t)threadIdx.x * (g.n_rb * g.n_cb) + (int64_t)rb * g.n_cb + cb] = s;
  }
}

}  // namespace vb
