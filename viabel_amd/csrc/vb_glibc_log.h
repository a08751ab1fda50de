// The host C library's log(double), operation for operation, so that the DEVICE can produce the very bits numpy's
// legacy generator gets from libm (legacy_gauss: f = sqrt(-2 log(r2) / r2); the Marsaglia-Tsang acceptance test
// log(U) < 0.5 X^2 + b (1 - V + log V)).  glibc's log is not correctly rounded (its bound is ~0.52 ulp), so "the
// correctly rounded value" is not what numpy sees on about one argument in a few thousand; only the same sequence of
// IEEE operations on the same table is.
//
// What is restated here is the x86-64 FMA variant of glibc >= 2.28's log (the `__log_fma` ifunc target; the routine
// every FMA-capable host runs): a 128-entry (1/c, log c) table, r = fma(z, 1/c, -1), a degree-5 polynomial, and a
// separate branch for 1 - 2^-4 <= x < 1 + 0x1.09p-4 -- with the multiply-adds fused exactly where that build fuses
// them (read off the routine's instruction sequence: a fused and an unfused evaluation differ in the last bit).  The
// TABLE is not restated: vb_glibc_log_locate() (vb_legacy_rng.cpp) finds `__log_data` in the libm this process has
// loaded and then PROVES the pair (table, sequence) on a few million arguments against the host's own log(); if the
// proof fails (another libm, a host without FMA) the device paths that need it report VB_ERR_UNSUPPORTED and the
// caller draws on the host.  Callers must be compiled with -ffp-contract=off: every fusion below is explicit.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIP__)
#include <hip/hip_runtime.h>
#define VB_GLIBC_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define VB_GLIBC_HD inline
#endif

namespace vb {

struct GlibcLogData {                 // the layout of glibc's `__log_data`
  double ln2hi, ln2lo;
  double A[5];                        // main polynomial
  double B[11];                       // polynomial of the branch near 1
  double tab[256];                    // (1 / c_i, log c_i), i < 128
};
static_assert(sizeof(GlibcLogData) == (18 + 256) * sizeof(double), "layout");

VB_GLIBC_HD double glibc_fma(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __fma_rn(a, b, c);
#else
  return std::fma(a, b, c);
#endif
}

// log(x) for finite normal x > 0 (the callers' arguments: r2 in (0, 1), U in [2^-53, 1), V = (1 + c X)^3 >= 2^-159)
VB_GLIBC_HD double glibc_log(double x, const GlibcLogData& T) {
  uint64_t ix;
  memcpy(&ix, &x, sizeof ix);
  if (ix - 0x3fee000000000000ull < 0x0003090000000000ull) {      // 1 - 2^-4 <= x < 1 + 0x1.09p-4
    if (ix == 0x3ff0000000000000ull) return 0.0;
    const double* B = T.B;
    const double r = x - 1.0;
    double q1 = glibc_fma(r, B[2], B[1]);
    double q2 = glibc_fma(r, B[5], B[4]);
    const double r2 = r * r;
    double q3 = glibc_fma(r, B[8], B[7]);
    q1 = glibc_fma(r2, B[3], q1);
    q2 = glibc_fma(r2, B[6], q2);
    const double r3 = r * r2;
    q3 = glibc_fma(r2, B[9], q3);
    q3 = glibc_fma(r3, B[10], q3);
    double q = glibc_fma(q3, r3, q2);
    q = glibc_fma(q, r3, q1);
    const double t = glibc_fma(r, 0x1p27, r);
    const double rhi = glibc_fma(-0x1p27, r, t);
    const double rhi2 = rhi * rhi;
    const double rlo = r - rhi;
    const double hi = glibc_fma(rhi2, B[0], r);
    const double d = r - hi;
    const double rs = r + rhi;
    double lo = glibc_fma(rhi2, B[0], d);
    lo = glibc_fma(B[0] * rlo, rs, lo);
    const double y = glibc_fma(q, r3, lo);
    return hi + y;
  }
  const uint64_t tmp = ix - 0x3fe6000000000000ull;
  const int i = (int)((tmp >> 45) & 127);
  const int k = (int)((int64_t)tmp >> 52);
  const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
  double z;
  memcpy(&z, &iz, sizeof z);
  const double invc = T.tab[2 * i], logc = T.tab[2 * i + 1];
  const double* A = T.A;
  const double r = glibc_fma(z, invc, -1.0);
  const double kd = (double)k;
  const double w = glibc_fma(kd, T.ln2hi, logc);
  const double p12 = glibc_fma(r, A[2], A[1]);
  const double hi = r + w;
  const double r2 = r * r;
  double lo = w - hi;
  lo = lo + r;
  lo = glibc_fma(kd, T.ln2lo, lo);
  const double r3 = r * r2;
  const double p34 = glibc_fma(r, A[4], A[3]);
  const double s = glibc_fma(r2, A[0], lo);
  const double p = glibc_fma(p34, r2, p12);
  const double y = glibc_fma(r3, p, s);
  return y + hi;
}

// vb_legacy_rng.cpp: the proven table of this process's libm, or nullptr (searched and proven once per process)
const GlibcLogData* vb_glibc_log_locate();

}  // namespace vb
