// Device noise generation: counter-based Philox4x32-10 + Box-Muller (fp64).
//
// Throughput-mode replacement for the RandomState draws inside approx.sample
// (viabel/approximations.py:212-216).  Element (global_row, col) is a pure function of
// (seed, stream, global_row, col): sharding the Monte-Carlo axis over GPUs does not change the
// numbers.  The exact numpy legacy stream (parity mode) is drawn on the host and uploaded with
// vb_noise_set_host instead.
#include "vb_common.h"
#include "vb_rng.h"

namespace vb {

// one thread = one pair of columns (2j, 2j+1) of a block of kRngRows rows (blockIdx.y; no 64-bit division): four
// Philox calls, each two Box-Muller pairs
constexpr int kRngRows = 8;
__global__ void __launch_bounds__(256) rng_normal_kernel(double* __restrict__ dst, int64_t ld,
                                                         uint64_t seed, uint64_t stream,
                                                         int64_t row_offset, int64_t n, int64_t d) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int pairs = (int)((d + 1) / 2);
  if (j >= pairs) return;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(stream >> 32);
  const int64_t r0 = (int64_t)blockIdx.y * kRngRows;
  auto store = [&](int64_t r, double va, double vb2) {
    double* p = dst + r * ld + 2 * j;
    if (2 * j + 1 < d) *reinterpret_cast<double2*>(p) = make_double2(va, vb2);
    else p[0] = va;
  };
  if (((row_offset + r0) & 7) == 0) {
    // an aligned block of eight global rows is four quads: rows u and u + 4 share a Philox call (vb_rng.h)
#pragma unroll
    for (int u = 0; u < kRngRows / 2; ++u) {
      const int64_t r = r0 + u;
      if (r >= n) break;
      double q[4];
      philox_normal_quad(k0, k1, philox_quad_id((uint64_t)(row_offset + r)), (uint32_t)j, (uint32_t)stream, q);
      store(r, q[0], q[1]);
      if (r + 4 < n) store(r + 4, q[2], q[3]);
    }
    return;
  }
  for (int u = 0; u < kRngRows; ++u) {        // a shard that starts inside a block: row by row, half a quad each
    const int64_t r = r0 + u;
    if (r >= n) break;
    double va, vb2;
    philox_normal_pair(k0, k1, (uint64_t)(row_offset + r), (uint32_t)j, (uint32_t)stream, &va, &vb2);
    store(r, va, vb2);
  }
}

// Standard Student-t noise (the base draws of MFStudentT.sample, viabel/approximations.py:270-274) by Bailey's polar
// method (Math. Comp. 62 (1994) 779-781): (u, v) uniform on the unit disc, w = u^2 + v^2,
//     t = u sqrt(df (w^(-2/df) - 1) / w)
// is exactly t_df distributed for every df > 0 -- one Philox call per attempt (acceptance pi / 4), one log / expm1,
// one sqrt; no gamma variate.  Attempt a of column e of pair j uses Philox sub-stream 2 a + e, so every value is a
// pure function of (seed, stream, global row, column).
__global__ void __launch_bounds__(256) rng_student_t_kernel(double* __restrict__ dst, int64_t ld, double df,
                                                            uint64_t seed, uint64_t stream, int64_t row_offset,
                                                            int64_t n, int64_t d) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int pairs = (int)((d + 1) / 2);
  if (j >= pairs) return;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(stream >> 32);
  const int64_t r0 = (int64_t)blockIdx.y * kRngRows;
  for (int u = 0; u < kRngRows; ++u) {
    const int64_t r = r0 + u;
    if (r >= n) break;
    const uint64_t grow = (uint64_t)(row_offset + r);
    double* p = dst + r * ld + 2 * j;
    p[0] = student_t_polar(df, grow, (uint32_t)j, (uint32_t)stream, 0, k0, k1);
    if (2 * j + 1 < d) p[1] = student_t_polar(df, grow, (uint32_t)j, (uint32_t)stream, 1, k0, k1);
  }
}

__global__ void __launch_bounds__(256) rng_chisquare_kernel(double* __restrict__ dst, double df, uint64_t seed,
                                                            uint64_t stream, int64_t row_offset, int64_t n) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(stream >> 32);
  dst[r] = philox_chisquare(df, (uint64_t)(row_offset + r), (uint32_t)stream, k0, k1);
}

int rng_chisquare(vb_ctx* ctx, double* dst, double df, uint64_t seed, uint64_t stream, int64_t row_offset, int64_t n) {
  if (!(df > 2.0)) return fail(ctx, VB_ERR_INVALID, "df must be greater than 2");
  hipLaunchKernelGGL(rng_chisquare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dst, df, seed,
                     stream, row_offset, n);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

int rng_fill(vb_ctx* ctx, double* dst, int64_t ld, int kind, double df, uint64_t seed,
             uint64_t stream, int64_t row_offset, int64_t n, int64_t d) {
  const int64_t pairs = (d + 1) / 2;
  if (kind != VB_NOISE_NORMAL && kind != VB_NOISE_STUDENT_T) return fail(ctx, VB_ERR_UNSUPPORTED, "unknown noise kind %d", kind);
  if (kind == VB_NOISE_STUDENT_T && !(df > 2.0)) return fail(ctx, VB_ERR_INVALID, "df must be greater than 2");
  const int64_t chunk = (int64_t)65535 * kRngRows;          // gridDim.y limit
  for (int64_t r0 = 0; r0 < n; r0 += chunk) {
    const int64_t rows = n - r0 < chunk ? n - r0 : chunk;
    const dim3 grid((unsigned)((pairs + 255) / 256), (unsigned)((rows + kRngRows - 1) / kRngRows));
    if (kind == VB_NOISE_NORMAL)
      hipLaunchKernelGGL(rng_normal_kernel, grid, dim3(256), 0, ctx->stream, dst + r0 * ld, ld, seed, stream,
                         row_offset + r0, rows, d);
    else
      hipLaunchKernelGGL(rng_student_t_kernel, grid, dim3(256), 0, ctx->stream, dst + r0 * ld, ld, df, seed, stream,
                         row_offset + r0, rows, d);
    VB_HIP(ctx, hipGetLastError());
  }
  return VB_OK;
}

}  // namespace vb
