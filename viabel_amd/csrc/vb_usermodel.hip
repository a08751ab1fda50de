// User model as HIP source (VB_MODEL_SOURCE): the adaptor for log densities outside the built-in set.
//
// The reference takes an arbitrary Python callable and differentiates it with autograd (viabel/models.py:17-39,
// convenience.py:75 `bbvi(dim, log_density=...)`).  A GPU engine cannot run a Python callable; what it can take is the
// log density and its gradient written as device code.  The caller hands a source snippet that defines
//
//     __device__ double vb_log_density(const double* z, int d, const double* params, double* grad);
//
// (returns f(z) for one sample z[0..d); writes grad f into grad[0..d) unless grad is NULL; `params` is an array of
// doubles uploaded with the model -- data, hyper-parameters).  It is compiled for this GPU with hiprtc (resolved with
// dlopen at the first use: the library itself keeps linking only libamdhip64 and librccl), wrapped in a row kernel and
// launched wherever the pipelines need (f, G) of a sample matrix: the mean-field ExclusiveKL path streams the noise
// against the G it returns (the "loaded gradient" pass of the regression targets, vb_logistic.h), the dense Gaussian
// family uses it where the funnel's row kernel sits between the sampling GEMM and the gradient GEMM.
#include "vb_common.h"

#include <dlfcn.h>
#include <hip/hiprtc.h>

#include <cstring>
#include <string>

namespace vb {

namespace {

struct Rtc {
  void* handle = nullptr;
  decltype(&hiprtcCreateProgram) create = nullptr;
  decltype(&hiprtcCompileProgram) compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
  decltype(&hiprtcGetProgramLog) log = nullptr;
  decltype(&hiprtcGetCodeSize) code_size = nullptr;
  decltype(&hiprtcGetCode) code = nullptr;
  decltype(&hiprtcDestroyProgram) destroy = nullptr;
};

const Rtc* rtc_load() {
  static Rtc rtc;
  static bool tried = false;
  if (tried) return rtc.handle ? &rtc : nullptr;
  tried = true;
  void* h = dlopen("libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("/opt/rocm/lib/libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) return nullptr;
  rtc.create = (decltype(rtc.create))dlsym(h, "hiprtcCreateProgram");
  rtc.compile = (decltype(rtc.compile))dlsym(h, "hiprtcCompileProgram");
  rtc.log_size = (decltype(rtc.log_size))dlsym(h, "hiprtcGetProgramLogSize");
  rtc.log = (decltype(rtc.log))dlsym(h, "hiprtcGetProgramLog");
  rtc.code_size = (decltype(rtc.code_size))dlsym(h, "hiprtcGetCodeSize");
  rtc.code = (decltype(rtc.code))dlsym(h, "hiprtcGetCode");
  rtc.destroy = (decltype(rtc.destroy))dlsym(h, "hiprtcDestroyProgram");
  if (!rtc.create || !rtc.compile || !rtc.log_size || !rtc.log || !rtc.code_size || !rtc.code || !rtc.destroy) {
    dlclose(h);
    return nullptr;
  }
  rtc.handle = h;
  return &rtc;
}

// one thread per sample: the user function walks its row.  (Rows are ld doubles apart, 128-B aligned.)
// The dimension is known when the source is compiled, so up to kUserDimPrivate the sample and its gradient live in
// private arrays of that fixed size (VB_USER_DIM, defined ahead of the source): the compiler sees constant trip counts
// and two objects that cannot alias the model's data or each other, where a pointer into the global Z / G matrices
// made every `g[j] += ...` of the user's loop a dependent global read-modify-write (robust regression, d = 64,
// 512 observations, 4096 samples: 10.0 ms -> see DESIGN 4.8).  The arithmetic and its order are the user's: same bits.
//
// VB_LOG_DENSITY_PARTS K (a power of two, defined by the source; the dimension must fit the private arrays): K threads
// per sample.  The source then defines
//     __device__ double vb_log_density_part(const double* z, int d, const double* params, double* grad, int part, int n_parts);
// returning the share of f (and writing the share of the gradient) that belongs to `part` -- typically the
// observations i = part, part + K, ..., the prior in part 0 -- and the wrapper adds the K shares with a butterfly of
// shuffles (the same pairs in the same order on every lane: reproducible).  One thread per sample leaves the GPU
// 94 % idle at N = 4096; a model that is a sum over data has this parallelism to give.
// ---- grad = 'auto': forward-mode automatic differentiation of the user's density -----------------------------------
// The reference differentiates an arbitrary callable with autograd (models.py:17-39, objectives.py:191-193).  With
// VB_AUTO_GRAD defined the source gives only the density, generic in its scalar type:
//
//     template <class T> __device__ T vb_log_density(vb::vec<T> z, int d, const double* params);
//
// (z[j] is the j-th coordinate as a T; arithmetic, comparisons and the functions below work on T and mix with double).
// The header in front of it defines vb::dual -- a value and VB_DUAL_K = 8 directional derivatives, all in registers --
// and the wrapper runs ceil(d / 8) threads per sample: thread c seeds the coordinates 8c .. 8c + 7, evaluates the
// density once and stores its eight gradient entries (thread 0 also the value).  Forward mode costs ~(9 / 8) d times
// the density instead of reverse mode's small constant, but it needs no tape, spreads over d / 8 times more threads
// than one-thread-per-sample code -- which is what a GPU with 1024 SIMDs wants from a 4096-sample batch -- and is
// exact to rounding (no step size).  A value-only call (grad == NULL) instantiates the density with T = double.
const char* const kAutoHeader = R"VBSRC(
#define VB_DUAL_K 8
// threads per sample: one per window of VB_DUAL_K coordinates; a density that uses vb::dot (VB_PAD_CHUNKS, set by the host
// when the source mentions it) gets the next power of two -- the surplus threads own empty windows -- so that the
// sample's lanes can share the products through a butterfly
constexpr int vb_chunks_of(int dim) {
  int c = (dim + VB_DUAL_K - 1) / VB_DUAL_K;
#ifdef VB_PAD_CHUNKS
  int p = 1;
  while (p < c) p <<= 1;
  if (p <= 64) c = p;
#endif
  return c;
}
#define VB_CHUNKS vb_chunks_of(VB_USER_DIM_ANY)
namespace vb {
struct dual {
  double v;
  double d[VB_DUAL_K];
  __device__ dual() {}
  __device__ dual(double x) : v(x) {
#pragma unroll
    for (int k = 0; k < VB_DUAL_K; ++k) d[k] = 0.0;
  }
};
// r = value with derivative scale * a.d
__device__ inline dual vb_chain(double value, double scale, const dual& a) {
  dual r;
  r.v = value;
#pragma unroll
  for (int k = 0; k < VB_DUAL_K; ++k) r.d[k] = scale * a.d[k];
  return r;
}
__device__ inline dual operator+(const dual& a, const dual& b) {
  dual r;
  r.v = a.v + b.v;
#pragma unroll
  for (int k = 0; k < VB_DUAL_K; ++k) r.d[k] = a.d[k] + b.d[k];
  return r;
}
__device__ inline dual operator-(const dual& a, const dual& b) {
  dual r;
  r.v = a.v - b.v;
#pragma unroll
  for (int k = 0; k < VB_DUAL_K; ++k) r.d[k] = a.d[k] - b.d[k];
  return r;
}
__device__ inline dual operator*(const dual& a, const dual& b) {
  dual r;
  r.v = a.v * b.v;
#pragma unroll
  for (int k = 0; k < VB_DUAL_K; ++k) r.d[k] = a.d[k] * b.v + a.v * b.d[k];
  return r;
}
__device__ inline dual operator/(const dual& a, const dual& b) {
  const double ib = 1.0 / b.v, q = a.v * ib;
  dual r;
  r.v = q;
#pragma unroll
  for (int k = 0; k < VB_DUAL_K; ++k) r.d[k] = (a.d[k] - q * b.d[k]) * ib;
  return r;
}
__device__ inline dual operator-(const dual& a) { return vb_chain(-a.v, -1.0, a); }
__device__ inline dual operator+(const dual& a) { return a; }
__device__ inline dual operator+(const dual& a, double b) { dual r = a; r.v += b; return r; }
__device__ inline dual operator+(double b, const dual& a) { dual r = a; r.v += b; return r; }
__device__ inline dual operator-(const dual& a, double b) { dual r = a; r.v -= b; return r; }
__device__ inline dual operator-(double b, const dual& a) { return vb_chain(b - a.v, -1.0, a); }
__device__ inline dual operator*(const dual& a, double b) { return vb_chain(a.v * b, b, a); }
__device__ inline dual operator*(double b, const dual& a) { return vb_chain(a.v * b, b, a); }
__device__ inline dual operator/(const dual& a, double b) { return vb_chain(a.v / b, 1.0 / b, a); }
__device__ inline dual operator/(double b, const dual& a) { const double q = b / a.v; return vb_chain(q, -q / a.v, a); }
__device__ inline dual& operator+=(dual& a, const dual& b) { a = a + b; return a; }
__device__ inline dual& operator-=(dual& a, const dual& b) { a = a - b; return a; }
__device__ inline dual& operator*=(dual& a, const dual& b) { a = a * b; return a; }
__device__ inline dual& operator/=(dual& a, const dual& b) { a = a / b; return a; }
__device__ inline dual& operator+=(dual& a, double b) { a.v += b; return a; }
__device__ inline dual& operator-=(dual& a, double b) { a.v -= b; return a; }
__device__ inline dual& operator*=(dual& a, double b) { a = a * b; return a; }
__device__ inline dual& operator/=(dual& a, double b) { a = a / b; return a; }
#define VB_DUAL_CMP(op)                                                              \
  __device__ inline bool operator op(const dual& a, const dual& b) { return a.v op b.v; } \
  __device__ inline bool operator op(const dual& a, double b) { return a.v op b; }        \
  __device__ inline bool operator op(double a, const dual& b) { return a op b.v; }
VB_DUAL_CMP(<) VB_DUAL_CMP(>) VB_DUAL_CMP(<=) VB_DUAL_CMP(>=) VB_DUAL_CMP(==) VB_DUAL_CMP(!=)
#undef VB_DUAL_CMP
__device__ inline double value(double x) { return x; }
__device__ inline double value(const dual& x) { return x.v; }
}  // namespace vb
// elementary functions (global namespace, so that the same call works for T = double through <cmath>)
__device__ inline vb::dual log(const vb::dual& a) { return vb::vb_chain(log(a.v), 1.0 / a.v, a); }
__device__ inline vb::dual log1p(const vb::dual& a) { return vb::vb_chain(log1p(a.v), 1.0 / (1.0 + a.v), a); }
__device__ inline vb::dual exp(const vb::dual& a) { const double e = exp(a.v); return vb::vb_chain(e, e, a); }
__device__ inline vb::dual expm1(const vb::dual& a) { const double e = expm1(a.v); return vb::vb_chain(e, e + 1.0, a); }
__device__ inline vb::dual sqrt(const vb::dual& a) { const double r = sqrt(a.v); return vb::vb_chain(r, 0.5 / r, a); }
__device__ inline vb::dual sin(const vb::dual& a) { return vb::vb_chain(sin(a.v), cos(a.v), a); }
__device__ inline vb::dual cos(const vb::dual& a) { return vb::vb_chain(cos(a.v), -sin(a.v), a); }
__device__ inline vb::dual tanh(const vb::dual& a) { const double t = tanh(a.v); return vb::vb_chain(t, 1.0 - t * t, a); }
__device__ inline vb::dual atan(const vb::dual& a) { return vb::vb_chain(atan(a.v), 1.0 / (1.0 + a.v * a.v), a); }
__device__ inline vb::dual erf(const vb::dual& a) {
  return vb::vb_chain(erf(a.v), 1.1283791670955126 * exp(-a.v * a.v), a);
}
__device__ inline vb::dual fabs(const vb::dual& a) { return a.v < 0.0 ? -a : a; }
__device__ inline vb::dual pow(const vb::dual& a, double p) { return vb::vb_chain(pow(a.v, p), p * pow(a.v, p - 1.0), a); }
__device__ inline vb::dual pow(const vb::dual& a, const vb::dual& p) { return exp(p * log(a)); }
__device__ inline vb::dual fmax(const vb::dual& a, const vb::dual& b) { return a.v >= b.v ? a : b; }
__device__ inline vb::dual fmin(const vb::dual& a, const vb::dual& b) { return a.v <= b.v ? a : b; }
__device__ inline vb::dual fmax(const vb::dual& a, double b) { return a.v >= b ? a : vb::dual(b); }
__device__ inline vb::dual fmin(const vb::dual& a, double b) { return a.v <= b ? a : vb::dual(b); }
// lgamma: derivative = digamma (recurrence up to x >= 6, then the asymptotic series; |error| < 1e-14 for x > 0)
__device__ inline double vb_digamma(double x) {
  double r = 0.0;
  while (x < 6.0) { r -= 1.0 / x; x += 1.0; }
  const double f = 1.0 / (x * x);
  return r + log(x) - 0.5 / x -
         f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132 - f * (691.0 / 32760 - f / 12))))));
}
__device__ inline vb::dual lgamma(const vb::dual& a) { return vb::vb_chain(lgamma(a.v), vb_digamma(a.v), a); }
namespace vb {
// the sample as the density sees it: z[j] of type T
template <class T> struct vec;
template <> struct vec<double> {
  const double* p;
  __device__ double operator[](int j) const { return p[j]; }
};
template <> struct vec<dual> {
  const double* p;
  int lo;              // this thread differentiates with respect to coordinates lo .. lo + VB_DUAL_K - 1
  __device__ dual operator[](int j) const {
    dual r;
    r.v = p[j];
    const int k0 = j - lo;
#pragma unroll
    for (int k = 0; k < VB_DUAL_K; ++k) r.d[k] = k == k0 ? 1.0 : 0.0;
    return r;
  }
};
// sum_j a[j] z[j] over the whole sample (the linear predictor of a regression row): z[j] has derivative e_j, so the
// derivative part of the product is a[lo .. lo + K - 1] itself -- d multiply-adds and K loads per pass instead of the
// (K + 1) d a loop over dual numbers costs
__device__ inline double dot(const double* a, const vec<double>& z, int d) {
  double s = 0.0;
  for (int j = 0; j < d; ++j) s = fma(a[j], z.p[j], s);
  return s;
}
// The threads of one sample (consecutive lanes, see the wrapper) all call this with the same row: when their number is a
// power of two each forms the partial product over ITS window -- the very a[lo .. lo + K - 1] that are its derivative
// part -- and a butterfly over the sample's lanes adds them up (x + y == y + x: every lane gets the same bits), so a
// row costs every thread K loads and K multiply-adds instead of d of each.
__device__ inline dual dot(const double* a, const vec<dual>& z, int d) {
  constexpr int C = VB_CHUNKS;                  // threads per sample
  dual r;
  double s = 0.0;
  if constexpr (C > 1 && C <= 64 && (C & (C - 1)) == 0) {
#pragma unroll
    for (int k = 0; k < VB_DUAL_K; ++k) {
      const int j = z.lo + k;
      const double aj = j < d ? a[j] : 0.0;
      r.d[k] = aj;
      s = fma(aj, j < d ? z.p[j] : 0.0, s);
    }
#pragma unroll
    for (int off = 1; off < C; off <<= 1) s += __shfl_xor(s, off, 64);
    r.v = s;
    return r;
  }
  for (int j = 0; j < d; ++j) s = fma(a[j], z.p[j], s);
  r.v = s;
#pragma unroll
  for (int k = 0; k < VB_DUAL_K; ++k) r.d[k] = z.lo + k < d ? a[z.lo + k] : 0.0;
  return r;
}
}  // namespace vb
#line 1
)VBSRC";

const char* const kAutoWrapper = R"VBSRC(
extern "C" __device__ int vb_user_parts_k = VB_CHUNKS;
extern "C" __global__ void __launch_bounds__(64) vb_user_rows(const double* __restrict__ Z, long long ldz, long long n, int d,
                                        const double* __restrict__ params, double* __restrict__ G, long long ldg,
                                        double* __restrict__ f) {
  constexpr int C = VB_CHUNKS;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long row = t / C;
  const int chunk = (int)(t % C);
  if (row >= n) return;
  if (!G) {                      // value only: one thread per sample, plain doubles
    if (chunk == 0) f[row] = vb_log_density<double>(vb::vec<double>{Z + row * ldz}, d, params);
    return;
  }
  const vb::dual r = vb_log_density<vb::dual>(vb::vec<vb::dual>{Z + row * ldz, chunk * VB_DUAL_K}, d, params);
  if (chunk == 0) f[row] = r.v;
#pragma unroll
  for (int k = 0; k < VB_DUAL_K; ++k) {
    const int j = chunk * VB_DUAL_K + k;
    if (j < d) G[row * ldg + j] = r.d[k];
  }
}
)VBSRC";

const char* const kWrapper = R"VBSRC(
// threads per sample the kernel below was compiled for; the host reads it back from the loaded module
// (hipModuleGetGlobal) instead of guessing it from the source text
#ifdef VB_LOG_DENSITY_PARTS
extern "C" __device__ int vb_user_parts_k = VB_LOG_DENSITY_PARTS;
#else
extern "C" __device__ int vb_user_parts_k = 1;
#endif
#ifdef VB_LOG_DENSITY_PARTS
extern "C" __global__ void __launch_bounds__(64) vb_user_rows(const double* __restrict__ Z, long long ldz, long long n, int d,
                                        const double* __restrict__ params, double* __restrict__ G, long long ldg,
                                        double* __restrict__ f) {
  constexpr int K = VB_LOG_DENSITY_PARTS;
  static_assert(K >= 2 && K <= 64 && (K & (K - 1)) == 0, "VB_LOG_DENSITY_PARTS must be a power of two between 2 and 64");
  static_assert(VB_USER_DIM > 0, "VB_LOG_DENSITY_PARTS needs a model dimension of at most 128");
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long row = t / K;
  const int part = (int)(t % K);
  const bool live = row < n;
  if (!live) row = n - 1;                  // the whole wave takes part in the shuffles
  double zl[VB_USER_DIM > 0 ? VB_USER_DIM : 1], gl[VB_USER_DIM > 0 ? VB_USER_DIM : 1];
  for (int j = 0; j < VB_USER_DIM; ++j) zl[j] = Z[row * ldz + j], gl[j] = 0.0;
  double v = vb_log_density_part(zl, VB_USER_DIM, params, G ? gl : (double*)0, part, K);
  for (int off = 1; off < K; off <<= 1) {
    v += __shfl_xor(v, off, 64);
    if (G)
      for (int j = 0; j < VB_USER_DIM; ++j) gl[j] += __shfl_xor(gl[j], off, 64);
  }
  if (!live) return;
  if (part == 0) f[row] = v;
  if (G)
    for (int j = 0; j < VB_USER_DIM; ++j)
      if ((j & (K - 1)) == part) G[row * ldg + j] = gl[j];
}
#else
extern "C" __global__ void __launch_bounds__(64) vb_user_rows(const double* __restrict__ Z, long long ldz, long long n, int d,
                                        const double* __restrict__ params, double* __restrict__ G, long long ldg,
                                        double* __restrict__ f) {
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
#if VB_USER_DIM > 0
  double zl[VB_USER_DIM], gl[VB_USER_DIM];
  for (int j = 0; j < VB_USER_DIM; ++j) zl[j] = Z[row * ldz + j], gl[j] = 0.0;
  f[row] = vb_log_density(zl, VB_USER_DIM, params, G ? gl : (double*)0);
  if (G)
    for (int j = 0; j < VB_USER_DIM; ++j) G[row * ldg + j] = gl[j];
#else
  f[row] = vb_log_density(Z + row * ldz, d, params, G ? G + row * ldg : (double*)0);
#endif
}
#endif
)VBSRC";
constexpr int64_t kUserDimPrivate = 128;      // 2 x 8 x 128 B of private memory per sample at most

}  // namespace

int user_model_bind(vb_ctx* ctx, int64_t dim, const double* params, size_t n_params);

void user_model_release(vb_ctx* ctx) {      // vb_destroy: unload everything this context compiled
  for (auto& m : ctx->user_modules) (void)hipModuleUnload(m.module);
  ctx->user_modules.clear();
  ctx->user_module = nullptr;
  ctx->user_fn = nullptr;
}

int user_model_set(vb_ctx* ctx, int64_t dim, const char* source, const double* params, size_t n_params) {
  if (!source || !*source) return fail(ctx, VB_ERR_INVALID, "empty model source");
  if (dim <= 0) return fail(ctx, VB_ERR_INVALID, "model dimension must be positive");
  if (n_params > 0 && !params) return fail(ctx, VB_ERR_INVALID, "NULL params");
  uint64_t hash = 1469598103934665603ull;      // FNV-1a of the source text
  for (const char* c = source; *c; ++c) hash = (hash ^ (uint64_t)(unsigned char)*c) * 1099511628211ull;
  const int64_t priv_dim = dim <= kUserDimPrivate ? dim : 0;      // compiled into the wrapper: part of the module's key
  // `#define VB_AUTO_GRAD` as the FIRST line of the source selects the automatic-differentiation wrapper (the density
  // alone, generic in its scalar type); anything else is the explicit-gradient interface
  const bool auto_grad = strncmp(source, "#define VB_AUTO_GRAD", 20) == 0;
  hash = (hash ^ (uint64_t)(auto_grad ? dim : priv_dim)) * 1099511628211ull;
  const vb_ctx::UserModule* cached = nullptr;
  for (const auto& m : ctx->user_modules)
    if (m.hash == hash) cached = &m;
  const Rtc* rtc = cached ? nullptr : rtc_load();
  if (!cached && !rtc)
    return fail(ctx, VB_ERR_UNSUPPORTED, "libhiprtc.so not found: a source model needs the HIP runtime compiler");
  if (cached) {
    VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->user_module = cached->module;
    ctx->user_fn = cached->fn;
    ctx->user_parts = cached->parts;
    return user_model_bind(ctx, dim, params, n_params);
  }
  const std::string full =
      auto_grad ? "#define VB_USER_DIM_ANY " + std::to_string(dim) + "\n" +
                      (std::string(source).find("vb::dot") != std::string::npos ? "#define VB_PAD_CHUNKS 1\n" : "") + kAutoHeader +
                      std::string(source) + "\n" + kAutoWrapper
                : "#define VB_USER_DIM " + std::to_string(priv_dim) + "\n#line 1\n" + std::string(source) + "\n" + kWrapper;
  hiprtcProgram prog = nullptr;
  if (rtc->create(&prog, full.c_str(), "vb_user_model.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
    return fail(ctx, VB_ERR_HIP, "hiprtcCreateProgram failed");
  const std::string arch = std::string("--offload-arch=") + ctx->prop.gcnArchName;
  const char* opts[] = {arch.c_str(), "-O3", "-std=c++17", "-ffp-contract=off"};
  const hiprtcResult rc = rtc->compile(prog, 4, opts);
  if (rc != HIPRTC_SUCCESS) {
    size_t ls = 0;
    std::string log;
    if (rtc->log_size(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
      log.resize(ls);
      (void)rtc->log(prog, &log[0]);
    }
    (void)rtc->destroy(&prog);
    if (log.find("VB_LOG_DENSITY_PARTS needs a model dimension") != std::string::npos)      // the wrapper's static_assert
      return fail(ctx, VB_ERR_UNSUPPORTED, "VB_LOG_DENSITY_PARTS needs a model dimension of at most %lld",
                  (long long)kUserDimPrivate);
    if (log.size() > 1500) log.resize(1500);
    return fail(ctx, VB_ERR_INVALID, "model source does not compile: %s", log.c_str());
  }
  size_t cs = 0;
  std::string code;
  if (rtc->code_size(prog, &cs) != HIPRTC_SUCCESS || cs == 0) {
    (void)rtc->destroy(&prog);
    return fail(ctx, VB_ERR_HIP, "hiprtcGetCodeSize failed");
  }
  code.resize(cs);
  const hiprtcResult rg = rtc->code(prog, &code[0]);
  (void)rtc->destroy(&prog);
  if (rg != HIPRTC_SUCCESS) return fail(ctx, VB_ERR_HIP, "hiprtcGetCode failed");
  VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  VB_HIP(ctx, hipModuleLoadData(&mod, code.data()));
  if (hipModuleGetFunction(&fn, mod, "vb_user_rows") != hipSuccess) {
    (void)hipModuleUnload(mod);
    return fail(ctx, VB_ERR_HIP, "compiled model has no vb_user_rows kernel");
  }
  // K threads per sample, as compiled (a text scan of the source can be fooled by `# define`, macros or #if 0)
  int parts = 0;
  hipDeviceptr_t kptr = nullptr;
  size_t kbytes = 0;
  if (hipModuleGetGlobal(&kptr, &kbytes, mod, "vb_user_parts_k") != hipSuccess || kbytes != sizeof(int) ||
      hipMemcpy(&parts, kptr, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess || parts < 1 ||
      (!auto_grad && (parts > 64 || (parts & (parts - 1)) != 0))) {
    (void)hipModuleUnload(mod);
    return fail(ctx, VB_ERR_HIP, "compiled model does not report its threads per sample (vb_user_parts_k = %d)", parts);
  }
  // bounded cache: drop the least recently compiled module that is not the bound one (work in flight was drained above)
  constexpr size_t kMaxUserModules = 16;
  while (ctx->user_modules.size() >= kMaxUserModules) {
    size_t victim = 0;
    while (victim < ctx->user_modules.size() && ctx->user_modules[victim].module == ctx->user_module) ++victim;
    if (victim == ctx->user_modules.size()) break;
    (void)hipModuleUnload(ctx->user_modules[victim].module);
    ctx->user_modules.erase(ctx->user_modules.begin() + (long)victim);
  }
  ctx->user_modules.push_back({hash, mod, fn, parts});
  ctx->user_module = mod;
  ctx->user_fn = fn;
  ctx->user_parts = parts;
  return user_model_bind(ctx, dim, params, n_params);
}

// upload the parameter array and make the source model the context's model (the previous parameter buffer may still
// be read by work in flight: the callers synchronised the stream)
int user_model_bind(vb_ctx* ctx, int64_t dim, const double* params, size_t n_params) {
  VB_TRY(ensure(ctx, ctx->user_params, (n_params > 0 ? n_params : 1) * sizeof(double)));
  if (n_params > 0) {
    VB_HIP(ctx, hipMemcpyAsync(ctx->user_params.ptr, params, n_params * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  ModelDev m;
  m.id = VB_MODEL_SOURCE;
  m.dim = (int)dim;
  m.c0 = 0.0;
  m.p0 = (const double*)ctx->user_params.ptr;
  ctx->model = m;
  ctx->user_host_fn = nullptr;      // a compiled source replaces a host callback
  ctx->user_host_arg = nullptr;
  return VB_OK;
}

// A host callable with its gradient (vb_set_model_callback; models.py:80-104): to every pipeline it is a source model
// whose row kernel happens to be a round trip through pinned host memory (user_rows_enqueue below).
int user_model_set_callback(vb_ctx* ctx, int64_t dim, vb_model_callback fn, void* user) {
  if (dim <= 0) return fail(ctx, VB_ERR_INVALID, "model dimension must be positive");
  VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  VB_TRY(user_model_bind(ctx, dim, nullptr, 0));
  ctx->user_host_fn = fn;
  ctx->user_host_arg = user;
  return VB_OK;
}

static int user_rows_host(vb_ctx* ctx, hipStream_t st, const double* Z, int64_t ldz, int64_t n, int d, double* G,
                          int64_t ldg, double* f) {
  const size_t nd = (size_t)n * (size_t)d, need = 2 * nd + (size_t)n;
  if (ctx->user_host_pin_doubles < need) {
    if (ctx->user_host_pin) {
      VB_HIP(ctx, hipStreamSynchronize(st));
      VB_HIP(ctx, hipHostFree(ctx->user_host_pin));
      ctx->user_host_pin = nullptr;
      ctx->user_host_pin_doubles = 0;
    }
    VB_HIP(ctx, hipHostMalloc((void**)&ctx->user_host_pin, need * sizeof(double), hipHostMallocDefault));
    ctx->user_host_pin_doubles = need;
  }
  double *zh = ctx->user_host_pin, *fh = zh + nd, *gh = fh + n;
  const size_t row = (size_t)d * sizeof(double);
  VB_HIP(ctx, hipMemcpy2DAsync(zh, row, Z, (size_t)ldz * sizeof(double), row, (size_t)n, hipMemcpyDeviceToHost, st));
  VB_HIP(ctx, hipStreamSynchronize(st));
  const int rc = ctx->user_host_fn(ctx->user_host_arg, zh, n, (int64_t)d, fh, G ? gh : nullptr);
  if (rc != 0) return fail(ctx, VB_ERR_CALLBACK, "model callback returned %d", rc);
  VB_HIP(ctx, hipMemcpyAsync(f, fh, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
  if (G)
    VB_HIP(ctx, hipMemcpy2DAsync(G, (size_t)ldg * sizeof(double), gh, row, row, (size_t)n, hipMemcpyHostToDevice, st));
  // the pinned buffer is reused by the next call: its copies must have left before that call's D2H lands in it, which
  // stream order guarantees (the next D2H is enqueued behind these H2D copies on the same stream)
  return VB_OK;
}

// f[row] = f(Z[row]), G[row] = grad f(Z[row]) (G may be NULL) for the bound source model
int user_rows_enqueue(vb_ctx* ctx, hipStream_t st, const double* Z, int64_t ldz, int64_t n, int d, double* G,
                      int64_t ldg, double* f) {
  if (ctx->model.id == VB_MODEL_SOURCE && ctx->user_host_fn) return user_rows_host(ctx, st, Z, ldz, n, d, G, ldg, f);
  if (!ctx->user_fn || ctx->model.id != VB_MODEL_SOURCE) return fail(ctx, VB_ERR_STATE, "no source model bound");
  long long ldz_ = ldz, n_ = n, ldg_ = ldg;
  const double* params = (const double*)ctx->user_params.ptr;
  void* args[] = {(void*)&Z, (void*)&ldz_, (void*)&n_, (void*)&d, (void*)&params, (void*)&G, (void*)&ldg_, (void*)&f};
  const int64_t threads = n * ctx->user_parts;
  VB_HIP(ctx, hipModuleLaunchKernel(ctx->user_fn, (unsigned)((threads + 63) / 64), 1, 1, 64, 1, 1, 0, st, args, nullptr));
  return VB_OK;
}

}  // namespace vb
