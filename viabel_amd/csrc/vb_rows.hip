// Per-row (per-sample) quantities: f(x_n) for explicit x.
//
// Model.__call__ (viabel/models.py:27-39) for the device-resident targets; used by the
// diagnostics-side callers (convenience.py:176-179) and by the DIS / alpha objectives for the
// per-sample log weights (objectives.py:394-395, :445).  One wave per row, lanes stride the
// columns (coalesced), DPP/shuffle reduction over the wave.
#include "vb_common.h"

namespace vb {

__device__ __forceinline__ double wave_sum_rows(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

__global__ void __launch_bounds__(256) model_logp_rows_kernel(const double* __restrict__ x,
                                                              int64_t ld, int64_t n, int d,
                                                              ModelDev m, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const double* xr = x + row * ld;
  double acc = 0.0;
  if (m.id == VB_MODEL_GAUSS_DIAG) {
    for (int c = lane; c < d; c += 64) {
      const double dz = xr[c] - m.p0[c];
      acc -= 0.5 * dz * dz * m.p1[c];
    }
  } else {   // funnel
    const double v = xr[m.k];
    const double w = exp(-2.0 * v);
    for (int c = lane; c < d; c += 64) {
      const double z = xr[c];
      if (c == m.k)
        acc += -0.5 * z * z / (m.tau * m.tau) - (double)(d - 1) * z;
      else
        acc -= 0.5 * z * z * w;
    }
  }
  acc = wave_sum_rows(acc);
  if (lane == 0) out[row] = acc + m.c0;
}

int model_logp_rows(vb_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, int64_t d,
                    double* out_dev) {
  if (ctx->model.id != VB_MODEL_GAUSS_DIAG && ctx->model.id != VB_MODEL_FUNNEL)
    return fail(ctx, VB_ERR_UNSUPPORTED, "row log-density implements gauss_diag and funnel");
  const unsigned grid = (unsigned)((n + 3) / 4);
  hipLaunchKernelGGL(model_logp_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream, x_dev, ld, n,
                     (int)d, ctx->model, out_dev);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

}  // namespace vb
