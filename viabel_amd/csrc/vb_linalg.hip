// Symmetric matrix square root on the device -- the `sqrtm(Sigma)` of MultivariateT.sample
// (viabel/approximations.py:348) -- and, optionally, the solution X of the Sylvester equation
// R X + X R = E with R = A^(1/2): the derivative of the root that the reference obtains by differentiating
// `sqrtm` with autograd (ExclusiveKL over a MultivariateT, objectives.py:154-164).
//
// Method: the coupled Newton-Schulz iteration (Higham, Functions of Matrices, eq. 6.35) -- only fp64 MFMA
// GEMMs, no factorisation:
//     Y_0 = M / c,  Z_0 = I;    T = (3 I - Z Y) / 2;    Y <- Y T,   Z <- T Z;     Y -> (M / c)^(1/2)
// with c = ||A||_F so that the spectrum of M / c lies in (0, 1].  For the derivative the iteration runs on the
// block upper-triangular matrix M = [[A, E], [0, A]], whose square root is [[R, X], [0, R]] (the Frechet
// derivative of the root in direction E, i.e. the Sylvester solution).  Three GEMMs per step; the first one's
// epilogue forms T and reduces ||I - Z Y||_F^2, which the host reads to stop when it no longer decreases.
// The caller checks ||R R - A||_F / ||A||_F (returned) and keeps its LAPACK path for matrices this iteration
// cannot resolve (condition numbers beyond ~1e12).
//
// Why not LAPACK on the host as before: OpenBLAS's threaded dsyevd takes 70-95 ms for a 256 x 256 matrix on the
// 256-core GPU host (4 ms on one thread), which made the whole MultivariateT objective call host-bound.
#include "vb_common.h"
#include "vb_gemm_f64.h"

namespace vb {

namespace {

struct EpiNsT {             // T = 1.5 I - 0.5 acc;  returns (delta - acc)^2
  double* T;
  int64_t ld;
  double* part;
  __device__ double operator()(int, int row, int col, double acc) const {
    const double id = row == col ? 1.0 : 0.0;
    T[(int64_t)row * ld + col] = 1.5 * id - 0.5 * acc;
    const double r = id - acc;
    return r * r;
  }
};

struct EpiStore {           // C = acc
  double* C;
  int64_t ld;
  __device__ void operator()(int, int row, int col, double acc) const { C[(int64_t)row * ld + col] = acc; }
};

struct EpiResidual {        // returns (acc - A)^2 over the leading d x d block (R R - A)
  const double* A;
  int64_t ld;
  int d;
  double* part;
  __device__ double operator()(int, int row, int col, double acc) const {
    if (row >= d || col >= d) return 0.0;
    const double r = acc - A[(int64_t)row * ld + col];
    return r * r;
  }
};

GemmArgs square(const double* A, const double* B, int64_t ld, int m) {
  GemmArgs g;
  g.A = A;
  g.lda = ld;
  g.B = B;
  g.ldb = ld;
  g.M = m;
  g.N = m;
  g.K = m;
  g.tri_mode = 0;
  return g;
}

}  // namespace

// host: a (d x d, symmetric positive definite), e (d x d or nullptr) -> root, x (d x d each), info = [iterations,
// final ||I - Z Y||_F, ||R R - A||_F / ||A||_F]
int sym_sqrt(vb_ctx* ctx, const double* a, const double* e, int64_t d, double* root, double* x, double* info,
             double* inv_root) {
  const int m = (int)(e ? 2 * d : d);
  const int64_t ld = round_up(m, 16);
  const int64_t mat = (int64_t)m * ld;
  const int n_cu = ctx->prop.multiProcessorCount;
  const int64_t n_part = gemm_max_blocks(m, m);
  // device: [Y0 | Y1 | Z0 | Z1 | T | A (scaled M)];  pinned host (device-mapped): [staging matrix | GEMM partials].
  // All traffic goes through the pinned buffer: pageable hipMemcpy on this stack occasionally stalls in 10-ms
  // quanta, which would dwarf the ~2 ms of GEMMs; the reducing epilogues write their partials straight into it.
  VB_TRY(ensure(ctx, ctx->scratch, (size_t)(6 * mat) * sizeof(double)));
  VB_TRY(ensure_pinned(ctx, (size_t)(mat + round_up(n_part, 16)) * sizeof(double)));
  double* base = (double*)ctx->scratch.ptr;
  double *Y[2] = {base, base + mat}, *Z[2] = {base + 2 * mat, base + 3 * mat}, *T = base + 4 * mat,
         *M0 = base + 5 * mat;
  double *h = ctx->pin_host, *hp = ctx->pin_host + mat, *part = ctx->pin_dev + mat;
  hipStream_t st = ctx->stream;

  double fro = 0.0, fro_e = 0.0;
  for (int64_t i = 0; i < d * d; ++i) fro += a[i] * a[i];
  fro = sqrt(fro);
  if (!(fro > 0.0) || !std::isfinite(fro)) return fail(ctx, VB_ERR_NUMERIC, "matrix square root: zero or non-finite matrix");
  if (e) {
    for (int64_t i = 0; i < d * d; ++i) fro_e += e[i] * e[i];
    fro_e = sqrt(fro_e);
  }
  // the off-diagonal block is scaled to a tenth of the diagonal blocks' norm (X is linear in E)
  const double e_scale = (e && fro_e > 0.0) ? 0.1 * fro / fro_e : 0.0;
  VB_HIP(ctx, hipMemsetAsync(base, 0, (size_t)(5 * mat) * sizeof(double), st));   // pad columns stay zero
  VB_HIP(ctx, hipStreamSynchronize(st));   // earlier users of the staging buffer are done
  memset(h, 0, (size_t)mat * sizeof(double));
  for (int64_t i = 0; i < m; ++i) h[i * ld + i] = 1.0;
  VB_HIP(ctx, hipMemcpyAsync(Z[0], h, (size_t)mat * sizeof(double), hipMemcpyHostToDevice, st));
  VB_HIP(ctx, hipStreamSynchronize(st));
  for (int64_t i = 0; i < m; ++i) h[i * ld + i] = 0.0;
  for (int64_t i = 0; i < d; ++i)
    for (int64_t j = 0; j < d; ++j) {
      const double v = a[i * d + j] / fro;
      h[i * ld + j] = v;
      if (e) {
        h[(i + d) * ld + (j + d)] = v;
        h[i * ld + (j + d)] = e[i * d + j] * e_scale / fro;
      }
    }
  VB_HIP(ctx, hipMemcpyAsync(Y[0], h, (size_t)mat * sizeof(double), hipMemcpyHostToDevice, st));
  VB_HIP(ctx, hipMemcpyAsync(M0, h, (size_t)mat * sizeof(double), hipMemcpyHostToDevice, st));

  auto zero_part = [&]() {
    for (int64_t i = 0; i < n_part; ++i) hp[i] = 0.0;
  };
  auto read_part = [&](double* out) -> int {
    VB_HIP(ctx, hipStreamSynchronize(st));
    double s = 0.0;
    for (int64_t i = 0; i < n_part; ++i) s += hp[i];
    *out = sqrt(s);
    return VB_OK;
  };

  int cur = 0, it = 0;
  double res = 0.0, prev = 1e300;
  const double floor_tol = 4e-16 * (double)m;
  for (it = 0; it < 100; ++it) {
    zero_part();   // the stream is idle here (read_part synchronised) or has only copies in flight
    gemm_f64_launch<true>(st, square(Z[cur], Y[cur], ld, m), 1, n_cu, EpiNsT{T, ld, part});
    VB_HIP(ctx, hipGetLastError());
    VB_TRY(read_part(&res));
    if (!std::isfinite(res)) return fail(ctx, VB_ERR_NUMERIC, "matrix square root: iteration diverged");
    // converged: the residual is at the rounding floor, or small and no longer contracting
    if (res < floor_tol || (res < 1e-7 && res > 0.5 * prev)) break;
    prev = res;
    gemm_f64_launch<true>(st, square(Y[cur], T, ld, m), 1, n_cu, EpiStore{Y[cur ^ 1], ld});
    gemm_f64_launch<true>(st, square(T, Z[cur], ld, m), 1, n_cu, EpiStore{Z[cur ^ 1], ld});
    VB_HIP(ctx, hipGetLastError());
    cur ^= 1;
  }
  // accuracy of the leading block: ||Y Y - A / c||_F relative to ||A / c||_F = 1
  double acc = 0.0;
  zero_part();
  gemm_f64_launch<true>(st, square(Y[cur], Y[cur], ld, m), 1, n_cu, EpiResidual{M0, ld, (int)d, part});
  VB_HIP(ctx, hipGetLastError());
  VB_TRY(read_part(&acc));

  VB_HIP(ctx, hipMemcpyAsync(h, Y[cur], (size_t)mat * sizeof(double), hipMemcpyDeviceToHost, st));
  VB_HIP(ctx, hipStreamSynchronize(st));
  const double sc = sqrt(fro);
  for (int64_t i = 0; i < d; ++i)
    for (int64_t j = 0; j <= i; ++j) {   // symmetrise the rounding away
      const double v = 0.5 * (h[i * ld + j] + h[j * ld + i]) * sc;
      root[i * d + j] = v;
      root[j * d + i] = v;
    }
  if (e && x) {
    const double xs = e_scale > 0.0 ? sc / e_scale : 0.0;
    for (int64_t i = 0; i < d; ++i)
      for (int64_t j = 0; j < d; ++j) x[i * d + j] = h[i * ld + (j + d)] * xs;
  }
  if (inv_root) {   // Z -> (A / c)^(-1/2): the inverse root comes with the iteration
    VB_HIP(ctx, hipMemcpyAsync(h, Z[cur], (size_t)mat * sizeof(double), hipMemcpyDeviceToHost, st));
    VB_HIP(ctx, hipStreamSynchronize(st));
    for (int64_t i = 0; i < d; ++i)
      for (int64_t j = 0; j <= i; ++j) {
        const double v = 0.5 * (h[i * ld + j] + h[j * ld + i]) / sc;
        inv_root[i * d + j] = v;
        inv_root[j * d + i] = v;
      }
  }
  if (info) {
    info[0] = (double)it;
    info[1] = res;
    info[2] = acc;
  }
  return VB_OK;
}

}  // namespace vb
