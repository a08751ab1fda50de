// Internal definitions shared by the HIP translation units of libviabel_hip.so.
// Not part of the C ABI (see include/viabel_hip.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/viabel_hip.h"

namespace vb {

// ---- geometry of the mean-field accumulation kernel ------------------------------------
constexpr int kWave = 64;          // CDNA wavefront
constexpr int kMfThreads = 256;    // 4 waves per workgroup
constexpr int kMfWaves = kMfThreads / kWave;
constexpr int kMfCols = 128;       // columns per workgroup: 64 lanes x 2 doubles = one 1-KiB wave load
constexpr int kMfChunk = 16;       // rows a wave keeps in flight (16 x 16-B loads per lane)

// per-column partial sums (fields) and per-workgroup scalars written by the accumulation kernel
enum ColField { CF_G = 0, CF_GE, CF_E, CF_EE, CF_EK, CF_SC, CF_SCE, CF_NUM };
enum ScalField { SF_F = 0, SF_W, SF_Q, SF_QE, SF_L1P, SF_NUM = 8 };

struct ModelDev {
  int id = -1;
  int dim = 0;
  int k = 0;           // funnel: index of the log-scale coordinate
  double tau = 1.0;    // funnel: log_sigma_stdev
  double c0 = 0.0;     // additive constant of f per sample
  const double* p0 = nullptr;   // gauss_diag: mean[D]       gauss_full: mean[D]
  const double* p1 = nullptr;   // gauss_diag: 1/sd^2 [D]    gauss_full: P [D x D]
};

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
};

struct NoiseSlot {
  DeviceBuffer buf;
  int64_t n = 0, d = 0, ld = 0;   // ld: row stride in doubles (multiple of 16)
};

struct ResultSlot {
  double* host = nullptr;   // pinned [1 + p]
  int64_t p = 0;
  bool pending = false;
};

}  // namespace vb

struct vb_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipDeviceProp_t prop;
  std::string last_error;

  vb::NoiseSlot noise[VB_MAX_SLOTS];
  vb::ResultSlot results[VB_MAX_SLOTS];
  vb::ResultSlot sync_result;           // used by the synchronous entry points

  vb::ModelDev model;
  vb::DeviceBuffer model_params;        // device copy of the model's double parameters
  std::vector<double> model_host;       // host copy (epilogue constants)

  vb::DeviceBuffer theta;               // device copy of the variational parameter
  vb::DeviceBuffer partials;            // per-workgroup partial sums
  vb::DeviceBuffer sums;                // reduced sums (the vector that is all-reduced)
  vb::DeviceBuffer out;                 // [value | grad] on the device
  vb::DeviceBuffer scratch;             // generic device scratch (x upload, ...)
  vb::DeviceBuffer scratch2;            // per-row outputs
  vb::DeviceBuffer rowvec;              // per-row weights

  void* comm = nullptr;                 // ncclComm_t when a communicator is attached
  int n_ranks = 1, rank = 0;

  bool profile = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
};

namespace vb {

int fail(vb_ctx* ctx, int code, const char* fmt, ...);
int ensure(vb_ctx* ctx, DeviceBuffer& b, size_t bytes);

#define VB_HIP(ctx, expr)                                                              \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess)                                                             \
      return vb::fail((ctx), VB_ERR_HIP, "%s failed: %s (%s:%d)", #expr,               \
                      hipGetErrorString(e__), __FILE__, __LINE__);                     \
  } while (0)

#define VB_TRY(expr)                   \
  do {                                 \
    int rc__ = (expr);                 \
    if (rc__ != VB_OK) return rc__;    \
  } while (0)

// mean-field ExclusiveKL pipeline (vb_meanfield.hip)
int mf_elbo_grad_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int64_t n_total,
                         int family, double df, unsigned flags, int cv_mode,
                         const double* roww /*device or null*/, int mode /*0 elbo*/,
                         double alpha_or_scale);

// Philox noise generation (vb_rng.hip)
int rng_fill(vb_ctx* ctx, double* dst, int64_t ld, int kind, double df, uint64_t seed,
             uint64_t stream, int64_t row_offset, int64_t n, int64_t d);

// model log density for explicit x (vb_rows.hip)
int model_logp_rows(vb_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, int64_t d, double* out_dev);

// RCCL (vb_comm.hip)
int comm_allreduce_sum(vb_ctx* ctx, double* buf, size_t count);

// profiling helpers
void prof_begin(vb_ctx* ctx);
void prof_end(vb_ctx* ctx);

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

}  // namespace vb
