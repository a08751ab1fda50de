// Device noise generation: counter-based Philox4x32-10 + Box-Muller (fp64).
//
// Throughput-mode replacement for the RandomState draws inside approx.sample
// (viabel/approximations.py:212-216).  Element (global_row, col) is a pure function of
// (seed, stream, global_row, col): sharding the Monte-Carlo axis over GPUs does not change the
// numbers.  The exact numpy legacy stream (parity mode) is drawn on the host and uploaded with
// vb_noise_set_host instead.
#include "vb_common.h"

namespace vb {

struct Philox4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ Philox4 philox4x32_10(Philox4 c, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
    const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
    Philox4 n;
    n.x = hi1 ^ c.y ^ k0;
    n.y = lo1;
    n.z = hi0 ^ c.w ^ k1;
    n.w = lo0;
    c = n;
    k0 += W0;
    k1 += W1;
  }
  return c;
}

__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo) {
  const uint64_t x = ((uint64_t)hi << 32) | lo;
  return ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0);   // (0, 1)
}

// one thread = one pair of columns (2j, 2j+1) of one row; blockIdx.y = block of kRngRows rows (no 64-bit
// division); Box-Muller with sincospi (no 2 pi range reduction)
constexpr int kRngRows = 8;
__global__ void __launch_bounds__(256) rng_normal_kernel(double* __restrict__ dst, int64_t ld,
                                                         uint64_t seed, uint64_t stream,
                                                         int64_t row_offset, int64_t n, int64_t d) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int pairs = (int)((d + 1) / 2);
  if (j >= pairs) return;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(stream >> 32);
  const int64_t r0 = (int64_t)blockIdx.y * kRngRows;
#pragma unroll
  for (int u = 0; u < kRngRows; ++u) {
    const int64_t r = r0 + u;
    if (r >= n) break;
    const uint64_t grow = (uint64_t)(row_offset + r);
    Philox4 c;
    c.x = (uint32_t)grow;
    c.y = (uint32_t)(grow >> 32);
    c.z = (uint32_t)j;
    c.w = (uint32_t)stream;
    const Philox4 o = philox4x32_10(c, k0, k1);
    const double u1 = u01(o.x, o.y), u2 = u01(o.z, o.w);
    const double rad = sqrt(-2.0 * log(u1));
    double s, co;
    sincospi(2.0 * u2, &s, &co);
    double* p = dst + r * ld + 2 * j;
    if (2 * j + 1 < d) {
      *reinterpret_cast<double2*>(p) = make_double2(rad * co, rad * s);
    } else {
      p[0] = rad * co;
    }
  }
}

int rng_fill(vb_ctx* ctx, double* dst, int64_t ld, int kind, double df, uint64_t seed,
             uint64_t stream, int64_t row_offset, int64_t n, int64_t d) {
  (void)df;
  if (kind != VB_NOISE_NORMAL)
    return fail(ctx, VB_ERR_UNSUPPORTED,
                "device generation implements VB_NOISE_NORMAL; draw other base noise on the host "
                "and upload it with vb_noise_set_host");
  const int64_t pairs = (d + 1) / 2;
  const dim3 grid((unsigned)((pairs + 255) / 256), (unsigned)((n + kRngRows - 1) / kRngRows));
  hipLaunchKernelGGL(rng_normal_kernel, grid, dim3(256), 0, ctx->stream, dst, ld, seed,
                     stream, row_offset, n, d);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

}  // namespace vb
