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

struct EpiResidual {        // returns (acc - A)^2 over the leading d x dc block (R R - A; dc = 0: d x d)
  const double* A;
  int64_t ld;
  int d;
  double* part;
  int dc = 0;
  __device__ double operator()(int, int row, int col, double acc) const {
    if (row >= d || col >= (dc ? dc : d)) return 0.0;
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
  constexpr int kMaxSteps = 100;
  const int64_t pstride = round_up(n_part, 16);      // residual slots behind the staging matrix: one per step + the check's
  VB_TRY(ensure_pinned(ctx, (size_t)(mat + (int64_t)(kMaxSteps + 2) * pstride) * sizeof(double)));
  double* base = (double*)ctx->scratch.ptr;
  double *Y[2] = {base, base + mat}, *Z[2] = {base + 2 * mat, base + 3 * mat}, *T = base + 4 * mat,
         *M0 = base + 5 * mat;
  double* h = ctx->pin_host;
  hipStream_t st = ctx->stream;

  // scale: the infinity norm max_i sum_j |a_ij| >= lambda_max (round 5; rounds 2-4 used the Frobenius norm, which leaves a
  // well-conditioned spectrum at ~1 / sqrt(d) of the unit interval and costs the iteration five or six linear steps)
  double fro = 0.0, fro_e = 0.0;
  for (int64_t i = 0; i < d; ++i) {
    double r = 0.0;
    for (int64_t j = 0; j < d; ++j) r += fabs(a[i * d + j]);
    if (r > fro || !(r == r)) fro = r;
  }
  if (!(fro > 0.0) || !std::isfinite(fro)) return fail(ctx, VB_ERR_NUMERIC, "matrix square root: zero or non-finite matrix");
  if (e) {
    for (int64_t i = 0; i < d * d; ++i) fro_e += e[i] * e[i];
    fro_e = sqrt(fro_e);
  }
  // the off-diagonal block is scaled to a tenth of the diagonal blocks' norm (X is linear in E)
  const double e_scale = (e && fro_e > 0.0) ? 0.1 * fro / fro_e : 0.0;
  VB_HIP(ctx, hipMemsetAsync(base, 0, (size_t)(5 * mat) * sizeof(double), st));   // pad columns stay zero
  legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
  VB_HIP(ctx, hipStreamSynchronize(st));   // earlier users of the staging buffer are done
  legacy_poll(ctx);
  memset(h, 0, (size_t)mat * sizeof(double));
  for (int64_t i = 0; i < m; ++i) h[i * ld + i] = 1.0;
  VB_HIP(ctx, hipMemcpyAsync(Z[0], h, (size_t)mat * sizeof(double), hipMemcpyHostToDevice, st));
  legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
  VB_HIP(ctx, hipStreamSynchronize(st));
  legacy_poll(ctx);
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

  // residual slots: one per step (the epilogue of step k's first GEMM adds into slot k), read a GROUP of steps at a time
  double* hp_base = ctx->pin_host + mat;
  double* part_base = ctx->pin_dev + mat;
  for (int64_t i = 0; i < (kMaxSteps + 2) * pstride; ++i) hp_base[i] = 0.0;
  auto residual = [&](int slot) {
    double s = 0.0;
    const double* hp = hp_base + (int64_t)slot * pstride;
    for (int64_t i = 0; i < n_part; ++i) s += hp[i];
    return sqrt(s);
  };

  int cur = 0, it = 0;
  double res = 0.0, prev = 1e300;
  const double floor_tol = 4e-16 * (double)m;
  bool converged = false;
  int stop_at = kMaxSteps + 1;      // quadratic convergence: two steps after a residual below 1e-4 the iteration is at its floor
  while (!converged && it < kMaxSteps) {
    const int group = it == 0 ? 4 : (stop_at <= kMaxSteps ? stop_at - it : 1);
    for (int k = 0; k < group; ++k) {
      gemm_f64_launch<true>(st, square(Z[cur], Y[cur], ld, m), 1, n_cu, EpiNsT{T, ld, part_base + (int64_t)(it + k) * pstride});
      gemm_f64_launch<true>(st, square(Y[cur], T, ld, m), 1, n_cu, EpiStore{Y[cur ^ 1], ld});
      gemm_f64_launch<true>(st, square(T, Z[cur], ld, m), 1, n_cu, EpiStore{Z[cur ^ 1], ld});
      cur ^= 1;
    }
    VB_HIP(ctx, hipGetLastError());
    legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
    VB_HIP(ctx, hipStreamSynchronize(st));
    legacy_poll(ctx);
    for (int k = 0; k < group && !converged; ++k) {
      res = residual(it + k);
      if (!std::isfinite(res)) return fail(ctx, VB_ERR_NUMERIC, "matrix square root: iteration diverged");
      if (res < floor_tol || (res < 1e-7 && res > 0.5 * prev)) converged = true;
      if (res < 1e-4 && stop_at > kMaxSteps) stop_at = it + k + 2;
      prev = res;
    }
    it += group;
    if (it >= stop_at) converged = true;
  }
  // accuracy of the leading block: ||Y Y - A / c||_F (||A / c||_2 <= 1)
  double acc = 0.0;
  gemm_f64_launch<true>(st, square(Y[cur], Y[cur], ld, m), 1, n_cu,
                        EpiResidual{M0, ld, (int)d, part_base + (int64_t)kMaxSteps * pstride});
  VB_HIP(ctx, hipGetLastError());
  legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
  VB_HIP(ctx, hipStreamSynchronize(st));
  legacy_poll(ctx);
  acc = residual(kMaxSteps);

  VB_HIP(ctx, hipMemcpyAsync(h, Y[cur], (size_t)mat * sizeof(double), hipMemcpyDeviceToHost, st));
  legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
  VB_HIP(ctx, hipStreamSynchronize(st));
  legacy_poll(ctx);
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
    legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
    VB_HIP(ctx, hipStreamSynchronize(st));
    legacy_poll(ctx);
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

// ---- the same root with the matrix already on the device: Sigma = L L' from the unpacked factor ----------------------
// (the reference-identical MultivariateT + DISInclusiveKL step, rng='numpy', resident on the device: approximations.py:348
// needs sqrtm(Sigma) for the samples and nothing else of order D^3 on the host).  Differences from sym_sqrt above:
//   * the scale c is the infinity norm max_i sum_j |Sigma_ij| >= lambda_max instead of the Frobenius norm -- for the
//     well-conditioned scale matrices an optimiser visits the spectrum of Sigma / c then starts next to 1 and the
//     iteration is quadratic from the first step (4-5 steps at D = 256 where the Frobenius scale, 1 / sqrt(D) of it,
//     takes 10-12);
//   * Y <- Y T and Z <- T Z are ONE batched launch ([Y | T | Z] contiguous);
//   * the host reads the residuals of a GROUP of steps at a time (ring of partial sums in pinned memory) instead of
//     synchronising inside every step.
namespace {

struct EpiStoreBatch {      // C_b = acc, C_b = C + b * stride
  double* C;
  int64_t ld, stride;
  __device__ void operator()(int b, int row, int col, double acc) const { C[b * stride + (int64_t)row * ld + col] = acc; }
};

// c = max_i sum_j |A_ij| into scal[0] (zeroed before): one wave per row, the maximum over rows by an integer atomic on
// the bits of the non-negative sums (order-preserving)
__global__ void __launch_bounds__(256) ns_norm_kernel(const double* __restrict__ A, int d, int64_t ld, double* __restrict__ scal) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= d) return;
  double s = 0.0;
  for (int j = lane; j < d; j += 64) s += fabs(A[(int64_t)i * ld + j]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) atomicMax((unsigned long long*)scal, (unsigned long long)__double_as_longlong(s));
}

// A <- A / c (kept for the accuracy check), Y = A / c, Z = I (pads zero)
__global__ void __launch_bounds__(256) ns_scale_kernel(double* __restrict__ A, int d, int64_t ld, double* __restrict__ Y,
                                                       double* __restrict__ Z, const double* __restrict__ scal) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= (int64_t)d * ld) return;
  const int i = (int)(k / ld), j = (int)(k - (int64_t)i * ld);
  const double v = j < d ? A[k] / scal[0] : 0.0;
  A[k] = v;
  Y[k] = v;
  Z[k] = (i == j) ? 1.0 : 0.0;
}

// root = sqrt(c) (Y + Y') / 2 on the leading d x d block, pads zero  (inverse != 0: Y is the iteration's Z, the result
// (Z + Z') / (2 sqrt(c)) = Sigma^(-1/2))
__global__ void __launch_bounds__(256) ns_finish_kernel(const double* __restrict__ Y, int d, int64_t ld,
                                                        const double* __restrict__ scal, double* __restrict__ root,
                                                        int inverse = 0) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= (int64_t)d * ld) return;
  const int i = (int)(k / ld), j = (int)(k - (int64_t)i * ld);
  const double f = inverse ? 1.0 / sqrt(scal[0]) : sqrt(scal[0]);
  root[k] = j < d ? 0.5 * (Y[k] + Y[(int64_t)j * ld + i]) * f : 0.0;
}

}  // namespace

// The coupled Newton-Schulz iteration on an m x m problem whose start (Y = M / c in set[0], Z = I in set[0] + 2 mat) and
// reference M0 = M / c are on the device (m x ld, ld = round_up(m, 16)).  Returns the set holding the result in *cur_out;
// info = [steps, last residual, ||Y Y - M0||_F over the leading d_check x d_cols block].  VB_ERR_UNSUPPORTED: no
// convergence.  Pinned partial-sum ring: the caller has called ensure_pinned for (kNsMaxSteps + 2) * n_part doubles.
constexpr int kNsMaxSteps = 40;
// What the step-by-step control below would do, given the residuals of the steps applied so far (residual[j]: ||I - Z Y||_F
// of the state BEFORE step j).  Returns the number of steps it ends with, or -1 - k when it needs k more residuals first.
// (The control launches in groups -- four, then what is known to be missing, else one -- and applies every step of a group
// whether or not an earlier one of it met a stopping rule: the count it ends with is part of the result's bits.)
struct NsControl {
  int done = 0, stop_at = kNsMaxSteps + 1;
  bool converged = false, failed = false;
  double res = 0.0, prev = 1e300;
  int next_group() const { return done == 0 ? 4 : (stop_at <= kNsMaxSteps ? stop_at - done : 1); }
  // consume the residuals of the next group (they must be there)
  void take_group(const double* residual, double floor_tol) {
    const int group = next_group();
    for (int k = 0; k < group && !converged; ++k) {
      res = residual[done + k];
      if (!std::isfinite(res)) {
        failed = true;
        return;
      }
      if (res < floor_tol || (res < 1e-7 && res > 0.5 * prev)) converged = true;      // (later steps of the group: harmless)
      if (res < 1e-4 && stop_at > kNsMaxSteps) stop_at = done + k + 2;
      prev = res;
    }
    done += group;
    if (done >= stop_at) converged = true;
  }
  bool finished() const { return failed || converged || done >= kNsMaxSteps; }
};

// hint (round 6): the number of steps the previous root of this size ended with (0: none).  The iteration used to run in
// groups with a stream synchronisation behind each -- the host reads the residuals and decides -- : three wake-ups per root
// at the sizes of the t family's reference-identical mode, ~24 us each in a 0.7-ms call.  With a hint that many steps are
// launched at once and the control above is replayed on their residuals afterwards: when it ends where the hint said, or one
// step earlier (that state is still in the other set), one synchronisation has done; when it wants more, the remaining groups
// run as before; when it would have stopped two or more steps earlier the state is gone: *restart = true and the caller starts
// over without a hint.  The states are the same states whichever way they were launched: the result is bit-identical.
static int ns_run(vb_ctx* ctx, int m, int64_t ld, double* set0, double* set1, const double* M0, int d_check, int* cur_out,
                  double* info, int d_cols = 0, int hint = 0, bool* restart = nullptr) {
  const int64_t mat = (int64_t)m * ld;
  const int n_cu = ctx->prop.multiProcessorCount;
  const int64_t n_part = round_up(gemm_max_blocks(m, m), 16);
  hipStream_t st = ctx->stream;
  double* set[2] = {set0, set1};
  if (restart) *restart = false;
  std::vector<double> residual;      // residual[j], j < applied
  int applied = 0, cur = 0;
  auto launch_steps = [&](int count) {
    for (int k = 0; k < count; ++k) {
      double *Y = set[cur], *T = Y + mat, *Z = Y + 2 * mat;
      // T = (3 I - Z Y) / 2 with ||I - Z Y||_F^2 of the state BEFORE this step into partial slot `applied + k`
      gemm_f64_launch<true>(st, square(Z, Y, ld, m), 1, n_cu, EpiNsT{T, ld, ctx->pin_dev + (int64_t)(applied + k) * n_part});
      GemmArgs g = square(Y, T, ld, m);      // batch 0: Y T, batch 1: T Z
      g.batch = 1;
      g.batch_a = mat, g.batch_b = mat;
      gemm_f64_launch<true>(st, g, 2, n_cu, EpiStoreBatch{set[cur ^ 1], ld, 2 * mat});
      cur ^= 1;
    }
  };
  auto read_residuals = [&](int count) {      // (after a synchronisation)
    for (int k = 0; k < count; ++k) {
      double s = 0.0;
      const double* hp = ctx->pin_host + (int64_t)(applied + k) * n_part;
      for (int64_t i = 0; i < n_part; ++i) s += hp[i];
      residual.push_back(sqrt(s));
    }
    applied += count;
  };
  const double floor_tol = 4e-16 * (double)m;
  NsControl c;
  if (hint >= 4 && hint <= kNsMaxSteps && restart) {
    launch_steps(hint);
    VB_HIP(ctx, hipGetLastError());
    legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
    VB_HIP(ctx, hipStreamSynchronize(st));
    legacy_poll(ctx);
    read_residuals(hint);
    while (!c.finished() && c.done + c.next_group() <= applied) c.take_group(residual.data(), floor_tol);
    if (c.failed) return VB_ERR_UNSUPPORTED;
    if (c.finished() && c.converged && c.done < applied - 1) {      // the control would have stopped two or more steps ago
      *restart = true;
      return VB_OK;
    }
  }
  // the step-by-step control (all of it without a hint; what the hinted launch left otherwise)
  while (!c.finished()) {
    const int group = c.next_group();
    const int missing = c.done + group - applied;      // (> 0 here: the loop above took every complete group)
    if (missing > 0) {
      launch_steps(missing);
      VB_HIP(ctx, hipGetLastError());
      legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
      VB_HIP(ctx, hipStreamSynchronize(st));
      legacy_poll(ctx);
      read_residuals(missing);
    }
    c.take_group(residual.data(), floor_tol);
    if (c.failed) return VB_ERR_UNSUPPORTED;
  }
  if (!c.converged) return VB_ERR_UNSUPPORTED;
  // the state after c.done steps: set[c.done & 1] -- the set the launches ended in, or (hinted launch, one step more than the
  // control wanted) the other one, whose Y and Z the extra step only read
  cur = c.done & 1;
  double* Y = set[cur];
  gemm_f64_launch<true>(st, square(Y, Y, ld, m), 1, n_cu, EpiResidual{M0, ld, d_check, ctx->pin_dev + (int64_t)kNsMaxSteps * n_part, d_cols});
  VB_HIP(ctx, hipGetLastError());
  *cur_out = cur;
  info[0] = (double)c.done, info[1] = c.res, info[2] = -1.0;      // [2]: read by the caller after its own last launch + sync
  return VB_OK;
}

// root (d x ld, device) = (Lfull Lt)^(1/2), Lfull = L (row-major, row stride ld = round_up(d, 16)), Lt = L'.
// info (host) = [steps, last residual ||I - Z Y||_F, ||R R - Sigma||_F / c].  VB_ERR_UNSUPPORTED: not converged to `tol`.
// inv_root != nullptr: also Sigma^(-1/2), the iteration's second limit (checked by the caller where it matters:
// the coupled iteration's residual ||I - Z Y|| bounds both)
int sym_sqrt_dev(vb_ctx* ctx, const double* Lfull, const double* Lt, int64_t d, int64_t ld, double* root, double tol,
                 double* info, double* inv_root) {
  const int m = (int)d;
  if (ld != round_up(d, 16)) return fail(ctx, VB_ERR_INVALID, "sym_sqrt_dev: row stride");
  const int64_t mat = (int64_t)m * ld;
  const int n_cu = ctx->prop.multiProcessorCount;
  const int64_t n_part = round_up(gemm_max_blocks(m, m), 16);
  // device: set p = [Y_p | T_p | Z_p] (p = 0, 1), A / c, one scalar line
  VB_TRY(ensure(ctx, ctx->scratch, (size_t)(7 * mat + 16) * sizeof(double)));
  VB_TRY(ensure_pinned(ctx, (size_t)((kNsMaxSteps + 2) * n_part) * sizeof(double)));
  double* base = (double*)ctx->scratch.ptr;
  double *M0 = base + 6 * mat, *scal = base + 7 * mat;
  hipStream_t st = ctx->stream;
  // the ring of partial sums starts from zero -- cleared by the DEVICE, in stream order (a host memset would have to wait
  // for the stream first: one wake-up per root, ~20 us of a 0.2 ms iteration)
  int cur = 0;
  double loc[3];
  const bool hint_on = !(getenv("VB_NS_HINT") && atoi(getenv("VB_NS_HINT")) == 0);      // (0: the step-by-step control only)
  for (int attempt = 0; attempt < 2; ++attempt) {
    VB_HIP(ctx, hipMemsetAsync(ctx->pin_dev, 0, (size_t)((kNsMaxSteps + 2) * n_part) * sizeof(double), st));
    VB_HIP(ctx, hipMemsetAsync(scal, 0, sizeof(double), st));
    gemm_f64_launch<true>(st, square(Lfull, Lt, ld, m), 1, n_cu, EpiStore{M0, ld});
    hipLaunchKernelGGL(ns_norm_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, (const double*)M0, m, ld, scal);
    hipLaunchKernelGGL(ns_scale_kernel, dim3((unsigned)((mat + 255) / 256)), dim3(256), 0, st, M0, m, ld, base,
                       base + 2 * mat, (const double*)scal);
    VB_HIP(ctx, hipGetLastError());
    // (the previous root of this size says how many steps to launch at once: ns_run; a wrong guess that cannot be used starts over)
    const int hint = (attempt == 0 && hint_on && ctx->ns_hint_m[0] == m) ? ctx->ns_hint_steps[0] : 0;
    bool restart = false;
    VB_TRY(ns_run(ctx, m, ld, base, base + 3 * mat, M0, m, &cur, loc, 0, hint, &restart));
    if (!restart) break;
  }
  ctx->ns_hint_m[0] = m, ctx->ns_hint_steps[0] = (int)loc[0];
  hipLaunchKernelGGL(ns_finish_kernel, dim3((unsigned)((mat + 255) / 256)), dim3(256), 0, st,
                     (const double*)(base + (int64_t)cur * 3 * mat), m, ld, (const double*)scal, root, 0);
  if (inv_root)
    hipLaunchKernelGGL(ns_finish_kernel, dim3((unsigned)((mat + 255) / 256)), dim3(256), 0, st,
                       (const double*)(base + (int64_t)cur * 3 * mat + 2 * mat), m, ld, (const double*)scal, inv_root, 1);
  VB_HIP(ctx, hipGetLastError());
  legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
  VB_HIP(ctx, hipStreamSynchronize(st));
  legacy_poll(ctx);
  double acc = 0.0;
  for (int64_t i = 0; i < n_part; ++i) acc += ctx->pin_host[(int64_t)kNsMaxSteps * n_part + i];
  acc = sqrt(acc);
  if (info) info[0] = loc[0], info[1] = loc[1], info[2] = acc;
  if (!(acc < tol)) return VB_ERR_UNSUPPORTED;
  return VB_OK;
}

// ---- the root's Frechet derivative on the device: X with R X + X R = E, R = (L L')^(1/2) ---------------------------------
// (MultivariateT under ExclusiveKL in the reference-identical mode: the gradient flows through sqrtm, objectives.py:154-164
// over approximations.py:348).  The same iteration on the block matrix [[Sigma, es E], [0, Sigma]] / c, whose root is
// [[R, es X], [0, R]] / sqrt(c); es scales the off-diagonal block to a tenth of the diagonal blocks' norm.
namespace {

// scal[1] += sum of squares of the leading d x d block of E (one workgroup)
__global__ void __launch_bounds__(1024) ns_sumsq_kernel(const double* __restrict__ E, int d, int64_t ld, double* __restrict__ scal) {
  __shared__ double sh[16];
  double s = 0.0;
  for (int64_t k = threadIdx.x; k < (int64_t)d * d; k += 1024) {
    const double v = E[(k / d) * ld + (k % d)];
    s += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += sh[w];
    scal[1] = t;
  }
}

// Y = M0 = [[A, es E], [0, A]] / c, Z = I on the 2 d x ld2 block layout; A = Sigma (d x ld), c = scal[0], es = 0.1 c / ||E||_F
__global__ void __launch_bounds__(256) ns_block_init_kernel(const double* __restrict__ A, const double* __restrict__ E, int d,
                                                            int64_t ld, int64_t ld2, double* __restrict__ Y, double* __restrict__ Z,
                                                            double* __restrict__ M0, double* __restrict__ scal) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int m = 2 * d;
  if (k >= (int64_t)m * ld2) return;
  const int i = (int)(k / ld2), j = (int)(k - (int64_t)i * ld2);
  const double c = scal[0], fe = sqrt(scal[1]);
  const double es = fe > 0.0 ? 0.1 * c / fe : 0.0;
  if (k == 0) scal[2] = es;
  double v = 0.0;
  if (j < m) {
    if (i < d && j < d) v = A[(int64_t)i * ld + j] / c;
    else if (i >= d && j >= d) v = A[(int64_t)(i - d) * ld + (j - d)] / c;
    else if (i < d && j >= d) v = E[(int64_t)i * ld + (j - d)] * es / c;
  }
  Y[k] = v, M0[k] = v;
  Z[k] = (i == j) ? 1.0 : 0.0;
}

// X = Y12 sqrt(c) / es on the d x ld layout (pads zero)
__global__ void __launch_bounds__(256) ns_block_finish_kernel(const double* __restrict__ Y, int d, int64_t ld, int64_t ld2,
                                                              const double* __restrict__ scal, double* __restrict__ X) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= (int64_t)d * ld) return;
  const int i = (int)(k / ld), j = (int)(k - (int64_t)i * ld);
  const double es = scal[2];
  X[k] = (j < d && es > 0.0) ? Y[(int64_t)i * ld2 + (j + d)] * sqrt(scal[0]) / es : 0.0;
}

}  // namespace

int sym_sqrt_frechet_dev(vb_ctx* ctx, const double* Lfull, const double* Lt, const double* E, int64_t d, int64_t ld, double* X,
                         double tol, double* info) {
  const int m = (int)(2 * d);
  if (ld != round_up(d, 16)) return fail(ctx, VB_ERR_INVALID, "sym_sqrt_frechet_dev: row stride");
  const int64_t ld2 = round_up(m, 16), mat = (int64_t)m * ld2, small = d * ld;
  const int n_cu = ctx->prop.multiProcessorCount;
  const int64_t n_part = round_up(gemm_max_blocks(m, m), 16);
  VB_TRY(ensure(ctx, ctx->scratch, (size_t)(7 * mat + small + 16) * sizeof(double)));
  VB_TRY(ensure_pinned(ctx, (size_t)((kNsMaxSteps + 2) * n_part) * sizeof(double)));
  double* base = (double*)ctx->scratch.ptr;
  double *M0 = base + 6 * mat, *A = base + 7 * mat, *scal = A + small;
  hipStream_t st = ctx->stream;
  int cur = 0;
  double loc[3];
  const bool hint_on = !(getenv("VB_NS_HINT") && atoi(getenv("VB_NS_HINT")) == 0);
  for (int attempt = 0; attempt < 2; ++attempt) {
    VB_HIP(ctx, hipMemsetAsync(ctx->pin_dev, 0, (size_t)((kNsMaxSteps + 2) * n_part) * sizeof(double), st));      // (as sym_sqrt_dev)
    VB_HIP(ctx, hipMemsetAsync(scal, 0, 4 * sizeof(double), st));
    gemm_f64_launch<true>(st, square(Lfull, Lt, ld, (int)d), 1, n_cu, EpiStore{A, ld});
    hipLaunchKernelGGL(ns_norm_kernel, dim3((unsigned)((d + 3) / 4)), dim3(256), 0, st, (const double*)A, (int)d, ld, scal);
    hipLaunchKernelGGL(ns_sumsq_kernel, dim3(1), dim3(1024), 0, st, E, (int)d, ld, scal);
    hipLaunchKernelGGL(ns_block_init_kernel, dim3((unsigned)((mat + 255) / 256)), dim3(256), 0, st, (const double*)A, E, (int)d, ld, ld2,
                       base, base + 2 * mat, M0, scal);
    VB_HIP(ctx, hipGetLastError());
    const int hint = (attempt == 0 && hint_on && ctx->ns_hint_m[1] == m) ? ctx->ns_hint_steps[1] : 0;
    bool restart = false;
    // the safety net covers the block that CARRIES the derivative too: rows [0, d) x columns [0, 2 d) of
    // Y Y - M0, i.e. R R - Sigma and (es / c) (R X + X R - E) -- the latter relative to an off-diagonal block scaled to a tenth
    // of the diagonal blocks' norm; an ill-conditioned Sigma whose X has not converged hands the call to the host route
    VB_TRY(ns_run(ctx, m, ld2, base, base + 3 * mat, M0, (int)d, &cur, loc, m, hint, &restart));
    if (!restart) break;
  }
  ctx->ns_hint_m[1] = m, ctx->ns_hint_steps[1] = (int)loc[0];
  hipLaunchKernelGGL(ns_block_finish_kernel, dim3((unsigned)((small + 255) / 256)), dim3(256), 0, st,
                     (const double*)(base + (int64_t)cur * 3 * mat), (int)d, ld, ld2, (const double*)scal, X);
  VB_HIP(ctx, hipGetLastError());
  legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
  VB_HIP(ctx, hipStreamSynchronize(st));
  legacy_poll(ctx);
  double acc = 0.0;
  for (int64_t i = 0; i < n_part; ++i) acc += ctx->pin_host[(int64_t)kNsMaxSteps * n_part + i];
  acc = sqrt(acc);
  if (info) info[0] = loc[0], info[1] = loc[1], info[2] = acc;
  if (!(acc < tol)) return VB_ERR_UNSUPPORTED;
  return VB_OK;
}

// ---- low-rank Gaussian, path derivative (objectives.py:156-159 over approximations.py:610-731) ---------------
// The score Sigma^-1 (x - mu), Sigma = B B' + diag(sigma^2), is linear in the two noise blocks, so everything the
// estimator adds to the entropy-form sums follows from second moments of the noise: with u_n = (sigma W)' eps_n
// (W = D^-2 B, a k-vector per sample) and T = [z | u] (n x 2k),
//     E'T (d x 2k),   T'T (2k x 2k),   sum eps,   sum eps^2 (per column),   sum T
// -- three skinny GEMMs and three column-sum passes; the O(D k^2) Woodbury algebra stays with the caller.
namespace {

__global__ void __launch_bounds__(256) lr_slab_sum_kernel(const double* __restrict__ W, int slabs, int64_t slab,
                                                          double* __restrict__ out, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  double s = 0.0;
  for (int k = 0; k < slabs; ++k) s += W[k * slab + i];   // fixed order
  out[i] = s;
}

struct EpiSlab {            // slab_split = acc
  double* W;
  int64_t ld, slab;
  __device__ void operator()(int split, int row, int col, double acc) const {
    W[split * slab + (int64_t)row * ld + col] = acc;
  }
};

}  // namespace

int lr_path_terms(vb_ctx* ctx, const NoiseSlot& ns, const NoiseSlot& nz, int64_t n, int64_t d, int64_t k,
                  const double* sw_host, double* out_host) {
  if (n <= 0 || n > ns.n || d != ns.d || n > nz.n || k != nz.d || k < 1 || k > 256)
    return fail(ctx, VB_ERR_INVALID, "noise slots must hold n x d and n x k (k <= 256) matrices");
  hipStream_t st = ctx->stream;
  const int n_cu = ctx->prop.multiProcessorCount;
  const int k2 = (int)(2 * k);
  const int64_t ldk = round_up(k, 16), ldt = round_up(2 * k, 16) < 32 ? 32 : round_up(2 * k, 16);   // row strides of sw (d x k), T (n x 2k)
  const int n_rb = (int)((n + 127) / 128);
  int splits = (int)(n / 256);
  if (splits > 32) splits = 32;
  if (splits < 1) splits = 1;
  const int64_t slab_et = d * ldt, slab_tt = (int64_t)k2 * ldt;
  const int64_t out_len = d * k2 + (int64_t)k2 * k2 + 2 * d + k2;
  int64_t off = 0;
  auto carve = [&off](int64_t doubles) {
    const int64_t o = off;
    off += round_up(doubles, 16);
    return o;
  };
  const int64_t o_sw = carve(d * ldk), o_t = carve(n * ldt), o_w1 = carve(splits * slab_et), o_w2 = carve(splits * slab_tt),
                o_et = carve(slab_et), o_tt = carve(slab_tt), o_c1 = carve((int64_t)n_rb * ns.ld),
                o_c2 = carve((int64_t)n_rb * ns.ld), o_c3 = carve((int64_t)n_rb * ldt), o_f = carve((int64_t)n_rb * ((d + 127) / 128 + 1)),
                o_s1 = carve(ns.ld), o_s2 = carve(ns.ld), o_s3 = carve(ldt), o_pack = carve(out_len);
  VB_TRY(ensure(ctx, ctx->glm_work, (size_t)off * sizeof(double)));
  double* base = (double*)ctx->glm_work.ptr;
  double *SW = base + o_sw, *T = base + o_t;
  const double* E = (const double*)ns.buf.ptr;
  VB_HIP(ctx, hipMemsetAsync(SW, 0, (size_t)(d * ldk) * sizeof(double), st));
  VB_HIP(ctx, hipMemsetAsync(T, 0, (size_t)(n * ldt) * sizeof(double), st));
  VB_HIP(ctx, hipMemcpy2DAsync(SW, (size_t)ldk * sizeof(double), sw_host, (size_t)k * sizeof(double),
                               (size_t)k * sizeof(double), (size_t)d, hipMemcpyHostToDevice, st));
  VB_HIP(ctx, hipMemcpy2DAsync(T, (size_t)ldt * sizeof(double), nz.buf.ptr, (size_t)nz.ld * sizeof(double),
                               (size_t)k * sizeof(double), (size_t)n, hipMemcpyDeviceToDevice, st));
  GemmArgs g;                                   // U = E sw  -> columns k .. 2k of T
  g.A = E, g.lda = ns.ld, g.B = SW, g.ldb = ldk;
  g.M = (int)n, g.N = (int)k, g.K = (int)d, g.tri_mode = 0;
  gemm_f64_launch<true>(st, g, 1, n_cu, EpiStore{T + k, ldt});
  VB_HIP(ctx, hipGetLastError());
  g.A = E, g.lda = ns.ld, g.B = T, g.ldb = ldt;  // E'T (d x 2k), contraction over the samples
  g.M = (int)d, g.N = k2, g.K = (int)n;
  gemm_f64_launch<false>(st, g, splits, n_cu, EpiSlab{base + o_w1, ldt, slab_et});
  g.A = T, g.lda = ldt;                          // T'T (2k x 2k)
  g.M = k2;
  gemm_f64_launch<false>(st, g, splits, n_cu, EpiSlab{base + o_w2, ldt, slab_tt});
  VB_HIP(ctx, hipGetLastError());
  auto slab_sum = [&](const double* W, int slabs, int64_t slab, double* out, int64_t count) {
    hipLaunchKernelGGL(lr_slab_sum_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, W, slabs, slab, out,
                       count);
  };
  slab_sum(base + o_w1, splits, slab_et, base + o_et, slab_et);
  slab_sum(base + o_w2, splits, slab_tt, base + o_tt, slab_tt);
  VB_TRY(fr_colsum_enqueue(ctx, E, nullptr, ns.ld, n, (int)d, 0, nullptr, base + o_c1, base + o_f));
  VB_TRY(fr_colsum_enqueue(ctx, E, nullptr, ns.ld, n, (int)d, 0, nullptr, base + o_c2, base + o_f, nullptr, 1));
  VB_TRY(fr_colsum_enqueue(ctx, T, nullptr, ldt, n, k2, 0, nullptr, base + o_c3, base + o_f));
  slab_sum(base + o_c1, n_rb, ns.ld, base + o_s1, ns.ld);
  slab_sum(base + o_c2, n_rb, ns.ld, base + o_s2, ns.ld);
  slab_sum(base + o_c3, n_rb, ldt, base + o_s3, ldt);
  VB_HIP(ctx, hipGetLastError());
  // pack (dense rows) -> one vector to all-reduce and copy out
  double* pack = base + o_pack;
  VB_HIP(ctx, hipMemcpy2DAsync(pack, (size_t)k2 * sizeof(double), base + o_et, (size_t)ldt * sizeof(double),
                               (size_t)k2 * sizeof(double), (size_t)d, hipMemcpyDeviceToDevice, st));
  VB_HIP(ctx, hipMemcpy2DAsync(pack + d * k2, (size_t)k2 * sizeof(double), base + o_tt, (size_t)ldt * sizeof(double),
                               (size_t)k2 * sizeof(double), (size_t)k2, hipMemcpyDeviceToDevice, st));
  double* tail = pack + d * k2 + (int64_t)k2 * k2;
  VB_HIP(ctx, hipMemcpyAsync(tail, base + o_s1, (size_t)d * sizeof(double), hipMemcpyDeviceToDevice, st));
  VB_HIP(ctx, hipMemcpyAsync(tail + d, base + o_s2, (size_t)d * sizeof(double), hipMemcpyDeviceToDevice, st));
  VB_HIP(ctx, hipMemcpyAsync(tail + 2 * d, base + o_s3, (size_t)k2 * sizeof(double), hipMemcpyDeviceToDevice, st));
  if (ctx->comm) VB_TRY(comm_allreduce_sum(ctx, st, pack, (size_t)out_len));
  VB_HIP(ctx, hipMemcpyAsync(out_host, pack, (size_t)out_len * sizeof(double), hipMemcpyDeviceToHost, st));
  legacy_poll(ctx);      // (before the wait: the host enqueues a look-ahead draw instead of idling; after it: what landed meanwhile)
  VB_HIP(ctx, hipStreamSynchronize(st));
  legacy_poll(ctx);
  return VB_OK;
}

}  // namespace vb
