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
// norms != nullptr (round 6; only when one workgroup spans the row: gridDim.x == 1): norms[r] = sum_c e_rc^2 of every row the
// workgroup generated, summed in a fixed order (a thread's pair, the wave by shuffles, the waves in order) -- the multivariate
// t's Mahalanobis term is r_n^2 times it, which saves the DIS refresh a pass over the noise matrix (vb_mvt.hip)
constexpr int kRngRows = 8;
__global__ void __launch_bounds__(256) rng_normal_kernel(double* __restrict__ dst, int64_t ld,
                                                         uint64_t seed, uint64_t stream,
                                                         int64_t row_offset, int64_t n, int64_t d,
                                                         double* __restrict__ norms) {
  __shared__ double sh[4][kRngRows];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int pairs = (int)((d + 1) / 2);
  const bool active = j < pairs;
  if (!active && !norms) return;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(stream >> 32);
  const int64_t r0 = (int64_t)blockIdx.y * kRngRows;
  double ss[kRngRows];
#pragma unroll
  for (int u = 0; u < kRngRows; ++u) ss[u] = 0.0;
  auto store = [&](int64_t r, double va, double vb2) {
    double* p = dst + r * ld + 2 * j;
    if (2 * j + 1 < d) *reinterpret_cast<double2*>(p) = make_double2(va, vb2);
    else p[0] = va;
    return 2 * j + 1 < d ? fma(va, va, vb2 * vb2) : va * va;
  };
  if (active) {
    if (((row_offset + r0) & 7) == 0) {
      // an aligned block of eight global rows is four quads: rows u and u + 4 share a Philox call (vb_rng.h)
#pragma unroll
      for (int u = 0; u < kRngRows / 2; ++u) {
        const int64_t r = r0 + u;
        if (r >= n) break;
        double q[4];
        philox_normal_quad(k0, k1, philox_quad_id((uint64_t)(row_offset + r)), (uint32_t)j, (uint32_t)stream, q);
        ss[u] = store(r, q[0], q[1]);
        if (r + 4 < n) ss[u + 4] = store(r + 4, q[2], q[3]);
      }
    } else {
#pragma unroll
      for (int u = 0; u < kRngRows; ++u) {        // a shard that starts inside a block: row by row, half a quad each
        const int64_t r = r0 + u;
        if (r >= n) break;
        double va, vb2;
        philox_normal_pair(k0, k1, (uint64_t)(row_offset + r), (uint32_t)j, (uint32_t)stream, &va, &vb2);
        ss[u] = store(r, va, vb2);
      }
    }
  }
  if (!norms) return;
#pragma unroll
  for (int u = 0; u < kRngRows; ++u) {
    double x = ss[u];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][u] = x;
  }
  __syncthreads();
  if (threadIdx.x < kRngRows && r0 + threadIdx.x < n)
    norms[r0 + threadIdx.x] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}

// The same row norms for a matrix that is already there (noise that came from the host or from the numpy-stream generators, or
// Philox normals generated before anybody asked): the generator's arithmetic and order -- a thread's pair fma(a, a, b b), the
// wave by shuffles, the waves in order -- so a row's norm does not depend on which of the two kernels formed it.
__global__ void __launch_bounds__(256) rng_row_norms_kernel(const double* __restrict__ src, int64_t ld, int64_t n, int64_t d,
                                                            double* __restrict__ norms) {
  __shared__ double sh[4][kRngRows];
  const int j = threadIdx.x;
  const int pairs = (int)((d + 1) / 2);
  const int64_t r0 = (int64_t)blockIdx.x * kRngRows;
  double ss[kRngRows];
#pragma unroll
  for (int u = 0; u < kRngRows; ++u) {
    const int64_t r = r0 + u;
    ss[u] = 0.0;
    if (j < pairs && r < n) {
      const double* p = src + r * ld + 2 * j;
      if (2 * j + 1 < d) {
        const double2 v = *reinterpret_cast<const double2*>(p);
        ss[u] = fma(v.x, v.x, v.y * v.y);
      } else {
        ss[u] = p[0] * p[0];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < kRngRows; ++u) {
    double x = ss[u];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][u] = x;
  }
  __syncthreads();
  if (threadIdx.x < kRngRows && r0 + threadIdx.x < n)
    norms[r0 + threadIdx.x] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}

int rng_row_norms(vb_ctx* ctx, hipStream_t st, const double* src, int64_t ld, int64_t n, int64_t d, double* norms) {
  if (d > 512) return fail(ctx, VB_ERR_UNSUPPORTED, "row norms: at most 512 columns");
  hipLaunchKernelGGL(rng_row_norms_kernel, dim3((unsigned)((n + kRngRows - 1) / kRngRows)), dim3(256), 0, st, src, ld, n, d, norms);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
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
             uint64_t stream, int64_t row_offset, int64_t n, int64_t d, double* norms) {
  const int64_t pairs = (d + 1) / 2;
  if (norms && (kind != VB_NOISE_NORMAL || pairs > 256)) return fail(ctx, VB_ERR_UNSUPPORTED, "row norms: normal noise of at most 512 columns");
  if (kind != VB_NOISE_NORMAL && kind != VB_NOISE_STUDENT_T) return fail(ctx, VB_ERR_UNSUPPORTED, "unknown noise kind %d", kind);
  if (kind == VB_NOISE_STUDENT_T && !(df > 2.0)) return fail(ctx, VB_ERR_INVALID, "df must be greater than 2");
  const int64_t chunk = (int64_t)65535 * kRngRows;          // gridDim.y limit
  for (int64_t r0 = 0; r0 < n; r0 += chunk) {
    const int64_t rows = n - r0 < chunk ? n - r0 : chunk;
    const dim3 grid((unsigned)((pairs + 255) / 256), (unsigned)((rows + kRngRows - 1) / kRngRows));
    if (kind == VB_NOISE_NORMAL)
      hipLaunchKernelGGL(rng_normal_kernel, grid, dim3(256), 0, ctx->stream, dst + r0 * ld, ld, seed, stream,
                         row_offset + r0, rows, d, norms ? norms + r0 : (double*)nullptr);
    else
      hipLaunchKernelGGL(rng_student_t_kernel, grid, dim3(256), 0, ctx->stream, dst + r0 * ld, ld, df, seed, stream,
                         row_offset + r0, rows, d);
    VB_HIP(ctx, hipGetLastError());
  }
  return VB_OK;
}

}  // namespace vb
