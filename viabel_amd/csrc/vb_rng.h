// Counter-based Philox4x32-10 + Box-Muller (fp64) device functions shared by the noise generator (vb_rng.hip)
// and by the kernels that generate their noise in registers instead of reading it (vb_meanfield.hip, GEN mode).
// Element (global_row, col) is a pure function of (seed, stream, global_row, col): see vb_rng.hip.
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

namespace vb {

struct Philox4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ Philox4 philox4x32_10(Philox4 c, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32 x 32 -> 64 product per word (v_mad_u64_u32) instead of a mul_hi / mul_lo pair: the quarter-rate
    // integer multiplies are half of the generator's issue cycles
    const uint64_t p0 = (uint64_t)M0 * c.x, p1 = (uint64_t)M1 * c.z;
    Philox4 n;
    n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
    n.w = (uint32_t)p0;
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

// ---- lean fp64 elementary functions for Box-Muller -------------------------------------------------------------
// The generator is VALU-bound (profiles/r02_meanfield_gen_pmc_valu.txt: 130 instructions per normal, most of them
// inside OCML's correctly-rounded log and sincospi with their double-double bookkeeping).  The transform needs the two
// functions on (0, 1) only and to ~1e-16, not to the last bit, so they are written out here: about half the
// instructions, 35 % less time (tools/rng_math_check.hip: log to 3.1e-16 relative, sin / cos to 1.5e-16 absolute against
// 80-bit libm on 4 M arguments incl. tiny ones and ones next to 1).

// log(u), 0 < u < 1 (finite, normal): u = m 2^e with m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(s), s = (m-1)/(m+1)
__device__ __forceinline__ double vb_log_unit(double u) {
  int e;
  double m = frexp(u, &e);                       // m in [1/2, 1)
  if (m < 0.70710678118654752440) {
    m *= 2.0;
    e -= 1;
  }
  const double f = m - 1.0;                      // exact
  const double d = 2.0 + f;
  double y = __builtin_amdgcn_rcp(d);            // 1 / d by two Newton steps on the hardware estimate
  double r = fma(-d, y, 1.0);
  y = fma(y, r, y);
  r = fma(-d, y, 1.0);
  y = fma(y, r, y);
  double sq = f * y;
  sq = fma(fma(-sq, d, f), y, sq);               // s = f / d, correctly rounded up to an ulp
  const double z = sq * sq;                      // z <= 0.0295: 1 / (2 k + 1) z^k below 1e-17 from k = 11
  double p = 1.0 / 23.0;
  p = fma(p, z, 1.0 / 21.0);
  p = fma(p, z, 1.0 / 19.0);
  p = fma(p, z, 1.0 / 17.0);
  p = fma(p, z, 1.0 / 15.0);
  p = fma(p, z, 1.0 / 13.0);
  p = fma(p, z, 1.0 / 11.0);
  p = fma(p, z, 1.0 / 9.0);
  p = fma(p, z, 1.0 / 7.0);
  p = fma(p, z, 1.0 / 5.0);
  p = fma(p, z, 1.0 / 3.0);
  const double lm = fma(2.0 * sq, p * z, 2.0 * sq);      // 2 s (1 + z p)
  const double de = (double)e;
  return fma(de, 6.93147180369123816490e-01, fma(de, 1.90821492927058770002e-10, lm));
}

// sin(2 pi t), cos(2 pi t), 0 <= t < 1: quarter-turn reduction t = k / 4 + f, |f| <= 1/8, Taylor series of the
// remaining angle phi = 2 pi f in [-pi/4, pi/4] (degree 17 / 18), quadrant by selects
__device__ __forceinline__ void vb_sincos_turn(double t, double* sn, double* cs) {
  const double k = rint(4.0 * t);                // 0 .. 4
  const double f = fma(-0.25, k, t);             // exact
  const double phi = 6.28318530717958647692 * f;
  const double z = phi * phi;
  double ps = -1.0 / 355687428096000.0;          // -1 / 17!
  ps = fma(ps, z, 1.0 / 1307674368000.0);        //  1 / 15!
  ps = fma(ps, z, -1.0 / 6227020800.0);
  ps = fma(ps, z, 1.0 / 39916800.0);
  ps = fma(ps, z, -1.0 / 362880.0);
  ps = fma(ps, z, 1.0 / 5040.0);
  ps = fma(ps, z, -1.0 / 120.0);
  ps = fma(ps, z, 1.0 / 6.0);
  const double s0 = fma(-phi * z, ps, phi);      // phi - phi^3 (1/6 - ...)
  double pc = 1.0 / 6402373705728000.0;          //  1 / 18!
  pc = fma(pc, z, -1.0 / 20922789888000.0);      // -1 / 16!
  pc = fma(pc, z, 1.0 / 87178291200.0);
  pc = fma(pc, z, -1.0 / 479001600.0);
  pc = fma(pc, z, 1.0 / 3628800.0);
  pc = fma(pc, z, -1.0 / 40320.0);
  pc = fma(pc, z, 1.0 / 720.0);
  pc = fma(pc, z, -1.0 / 24.0);
  pc = fma(pc, z, 0.5);
  const double c0 = fma(-z, pc, 1.0);            // 1 - phi^2 (1/2 - ...)
  const int q = (int)k & 3;                      // angle = q quarter turns + phi
  const double ss = (q & 1) ? c0 : s0;
  const double cc = (q & 1) ? s0 : c0;
  *sn = (q & 2) ? -ss : ss;
  *cs = (q == 1 || q == 2) ? -cc : cc;
}

// ---- normals: four per Philox call --------------------------------------------------------------------------------
// One Philox4x32-10 call is 160 of the instructions a pair of normals used to cost (two 53-bit uniforms per Box-Muller
// pair: 129 instructions per normal, the generator kernels are VALU-bound).  The four output words now feed TWO
// Box-Muller pairs -- 102 instructions per normal (SQ_INSTS_VALU): word 0 / 2 is the radius uniform of pair A / B, word
// 1 / 3 its angle.
// A 32-bit angle is more than the transform resolves; a 32-bit radius uniform u = (k + 1/2) 2^-32 would cut the tails
// at 6.66 sigma, so the rare small ones (k < 4096, probability 2^-20) take 32 more bits from a second call and keep
// the resolution of a 64-bit uniform where it matters (|z| up to 9.4 sigma).
// The two pairs are the column pair j of global rows g and g ^ 4 (g with bit 2 clear): the streaming kernels walk
// rows in steps of four per wave, so a lane meets both rows of a quad in consecutive steps.  Element (row, col) stays a
// pure function of (seed, stream, row, col): philox_normal_pair returns any row's pair by itself.
__device__ __forceinline__ uint64_t philox_quad_id(uint64_t g) { return ((g >> 3) << 2) | (g & 3); }

__device__ __forceinline__ double u01_32(uint32_t k) { return ((double)k + 0.5) * 2.3283064365386962890625e-10; }

// out[0..1]: columns 2 j, 2 j + 1 of row g (bit 2 clear), out[2..3]: of row g + 4; qid = philox_quad_id(g)
__device__ __forceinline__ void philox_normal_quad(uint32_t k0, uint32_t k1, uint64_t qid, uint32_t j, uint32_t w,
                                                   double* out) {
  Philox4 c;
  c.x = (uint32_t)qid;
  c.y = (uint32_t)(qid >> 32);
  c.z = j;
  c.w = w;
  const Philox4 o = philox4x32_10(c, k0, k1);
  double u1a = u01_32(o.x), u1b = u01_32(o.z);
  if ((o.x < 4096u) | (o.z < 4096u)) {             // a tail radius: 32 more bits (one lane in 2^19 quads)
    Philox4 c2 = c;
    c2.w = w + 0x9E3779B9u * 0xEEu;
    const Philox4 e = philox4x32_10(c2, k0, k1 ^ (0x85EBCA6Bu * 0xEEu));
    if (o.x < 4096u) u1a = ((double)(((uint64_t)o.x << 32) | e.x) + 0.5) * 5.42101086242752217003726400434970855712890625e-20;
    if (o.z < 4096u) u1b = ((double)(((uint64_t)o.z << 32) | e.z) + 0.5) * 5.42101086242752217003726400434970855712890625e-20;
  }
  const double ra = sqrt(-2.0 * vb_log_unit(u1a)), rb = sqrt(-2.0 * vb_log_unit(u1b));
  double s, co;
  vb_sincos_turn(u01_32(o.y), &s, &co);
  out[0] = ra * co;
  out[1] = ra * s;
  vb_sincos_turn(u01_32(o.w), &s, &co);
  out[2] = rb * co;
  out[3] = rb * s;
}

// the two standard normals of column pair j (columns 2 j, 2 j + 1) of global row `grow` (its half of the quad).
// k0 / k1: key words (seed, high stream bits), w: low stream bits
__device__ __forceinline__ void philox_normal_pair(uint32_t k0, uint32_t k1, uint64_t grow, uint32_t j, uint32_t w,
                                                   double* a, double* b) {
  double q[4];
  philox_normal_quad(k0, k1, philox_quad_id(grow), j, w, q);
  const bool hi = (grow >> 2) & 1;
  *a = hi ? q[2] : q[0];
  *b = hi ? q[3] : q[1];
}

// Student-t by Bailey's polar method (see vb_rng.hip)
__device__ __forceinline__ Philox4 philox_sub(uint64_t grow, uint32_t j, uint32_t stream, uint32_t sub, uint32_t k0,
                                              uint32_t k1) {
  Philox4 c;
  c.x = (uint32_t)grow;
  c.y = (uint32_t)(grow >> 32);
  c.z = j;
  c.w = stream + 0x9E3779B9u * sub;
  return philox4x32_10(c, k0, k1 ^ (0x85EBCA6Bu * sub));
}

__device__ __forceinline__ double student_t_polar(double df, uint64_t grow, uint32_t j, uint32_t stream, uint32_t e, uint32_t k0,
                                  uint32_t k1) {
  double u = 0.0, w = 1.0;
  for (uint32_t attempt = 0; attempt < 64; ++attempt) {
    const Philox4 o = philox_sub(grow, j, stream, 2 * attempt + e, k0, k1);
    u = 2.0 * u01(o.x, o.y) - 1.0;
    const double v = 2.0 * u01(o.z, o.w) - 1.0;
    w = fma(u, u, v * v);
    if (w <= 1.0 && w > 0.0) break;
  }
  return u * sqrt(df * expm1(-2.0 / df * log(w)) / w);
}

// chi-square(df) draw number `grow` of a stream (the per-sample radial scale of MultivariateT.sample,
// approximations.py:345: s = sqrt(chisquare(df) / df)), df > 2: twice a Gamma(df / 2) variate by Marsaglia and Tsang's
// squeeze-free method (ACM TOMS 26 (2000) 363-372) -- x standard normal, v = (1 + x / sqrt(9 a - 3))^3, accept when
// log u < x^2 / 2 + d - d v + d log v, d = a - 1/3 (acceptance > 95 % for a > 1).  Attempt t takes its normal from
// Philox sub-stream 2 t and its uniform from sub-stream 2 t + 1 of pseudo column 0xFFFFFFFF, so the value is a pure
// function of (seed, stream, global row) and does not depend on how the sample axis is sharded.
__device__ __forceinline__ double philox_chisquare(double df, uint64_t grow, uint32_t stream, uint32_t k0, uint32_t k1) {
  const double a = 0.5 * df, dd = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * dd);
  double v = 1.0;
  for (uint32_t attempt = 0; attempt < 64; ++attempt) {
    const Philox4 o = philox_sub(grow, 0xFFFFFFFFu, stream, 2 * attempt, k0, k1);
    const double u1 = u01(o.x, o.y), u2 = u01(o.z, o.w);
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    const double x = sqrt(-2.0 * log(u1)) * cs;
    const double t = 1.0 + c * x;
    if (t <= 0.0) continue;
    v = t * t * t;
    const Philox4 q = philox_sub(grow, 0xFFFFFFFFu, stream, 2 * attempt + 1, k0, k1);
    if (log(u01(q.x, q.y)) < 0.5 * x * x + dd - dd * v + dd * log(v)) break;
  }
  return 2.0 * dd * v;
}

}  // namespace vb
