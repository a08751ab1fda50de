// Internal definitions shared by the HIP translation units of libviabel_hip.so.
// Not part of the C ABI (see include/viabel_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

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
constexpr int kMaxBatch = 32;      // independent evaluations per launch (blockIdx.y)
constexpr int kModelLogQ = 99;     // internal pseudo model: weighted log q(z; theta) statistics (DIS)

// Touch every 64-byte line of the kernel-argument segment at the top of a kernel.  The argument block of a launch
// is fresh device memory: the first scalar load of each line takes 0.3-0.4 us, later ones hit the scalar cache
// (tools/kernarg_probe.hip), and the compiler fetches arguments where they are first needed, so a latency-bound
// kernel with a large argument block (mf_finalize: 1.6 KB, 26 lines) meets those misses one after another.  One
// batch of loads here brings all lines in for the price of one miss.
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
#if defined(__HIP_DEVICE_COMPILE__)
  // All loads target ONE scalar register that stays reserved (in/out operand) until the wait below, so none of them
  // can land in a register the compiler has given to something else; their values are never used.
  typedef const uint32_t __attribute__((address_space(4))) * KernargPtr;
  KernargPtr p = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
  uint32_t t = 0;
#pragma unroll
  for (int off = 0; off < BYTES; off += 64) asm volatile("s_load_dword %0, %1, %2" : "+s"(t) : "s"(p), "s"(off));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t));
#endif
}

// per-column partial sums (fields) and per-workgroup scalars written by the accumulation kernel
enum ColField { CF_G = 0, CF_GE, CF_E, CF_EE, CF_EK, CF_SC, CF_SCE, CF_NUM };

// Where the partial sum of (row block rb, field f, column col) lives: [col / 64][rb][f][col % 64], i.e. everything
// one finalize workgroup (64 columns) reads is one contiguous run of n_rb * CF_NUM * 512 bytes (57 KB for 4096
// samples) instead of n_rb * CF_NUM separate 512-byte pieces spread over the whole array.
__host__ __device__ __forceinline__ int64_t partial_index(int64_t rb, int f, int64_t col, int64_t n_rb) {
  return (((col >> 6) * n_rb + rb) * CF_NUM + f) * 64 + (col & 63);
}
enum ScalField { SF_F = 0, SF_W, SF_Q, SF_QE, SF_L1P, SF_EE, SF_FK, SF_GK, SF_GEK, SF_NUM = 12 };
// scalars written per workgroup by the accumulation kernel / by the prep kernel (SoA: [s][entry])
enum KScal { KS_F = 0, KS_Q, KS_QE, KS_L1P, KS_EE, KS_NUM };
enum PScal { PS_W = 0, PS_FK, PS_GK, PS_GEK, PS_NUM };

struct ModelDev {
  int id = -1;
  int dim = 0;
  int k = 0;           // funnel: index of the log-scale coordinate
  double tau = 1.0;    // funnel: log_sigma_stdev
  double c0 = 0.0;     // additive constant of f per sample
  const double* p0 = nullptr;   // gauss_diag: mean[D]       gauss_full: mean[D]
  const double* p1 = nullptr;   // gauss_diag: 1/sd^2 [D]    gauss_full: P [D x ldp]
  int64_t ldp = 0;              // gauss_full: row stride of P; logistic: row stride of X (multiples of 16)
  const double* p2 = nullptr;   // logistic: y [n_data]   (p0 = X [n_data x ldp], p1 = X' [D x ldq])
  int64_t ldq = 0;              // logistic: row stride of X'
  int64_t n_data = 0;           // logistic: observations
  int link = 0;                 // regression target: VB_GLM_* likelihood
  double aux = 1.0;             // VB_GLM_GAUSSIAN: observation noise stdev
};

// per-observation log-likelihood term (without constants) and its derivative with respect to eta = x' b
__device__ __forceinline__ double glm_term(int link, double aux, double y, double eta, double* dl) {
  if (link == VB_GLM_POISSON) {           // y eta - exp(eta)
    const double mu = exp(eta);
    *dl = y - mu;
    return y * eta - mu;
  }
  if (link == VB_GLM_GAUSSIAN) {          // -(y - eta)^2 / (2 s^2)
    const double r = (y - eta) / (aux * aux);
    *dl = r;
    return -0.5 * r * (y - eta);
  }
  const double t = exp(-fabs(eta));       // Bernoulli-logit: y eta - log(1 + exp(eta)), overflow-safe
  const double p = eta >= 0.0 ? 1.0 / (1.0 + t) : t / (1.0 + t);
  *dl = y - p;
  return y * eta - (fmax(eta, 0.0) + log1p(t));
}

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
};

// a Philox generation request (vb_noise_generate / vb_chisq_generate): what a buffer holds
struct NoiseReq {
  int kind = 0;
  double df = 0.0;
  uint64_t seed = 0, stream = 0;
  int64_t row_offset = 0, n = 0, d = 0;
  bool valid = false;
};
// Look-ahead generation (vb_api.hip, noise_prefetch): a blocking call draws fresh Philox noise, runs its kernels, and
// leaves the GPU idle while the host turns around.  When the last requests for a buffer walked the stream index in equal
// steps (two confirmations), the NEXT request's values are generated into `shadow` behind the call's last kernel, while
// the host waits and returns; the next vb_noise_generate with exactly those arguments adopts the shadow (a pointer swap)
// instead of launching.  Counter-based streams: the values are the ones the request would have generated.
// A caller whose next request is not a walk of the stream index (AlphaDivergence seeds every call from the host's
// generator, objectives.py:455) can name it instead: vb_noise_hint_seed.  A wrong prediction is never adopted.
struct NoiseAhead {
  DeviceBuffer shadow;
  DeviceBuffer shadow_norms;             // row norms of the shadow (NoiseSlot::norms), generated with it when pre_norms
  bool pre_norms = false;
  int64_t shadow_d = 0, shadow_ld = 0;   // geometry the shadow's pad columns were zeroed for
  NoiseReq last, pre;                    // what the live buffer holds; what the shadow holds (pre.valid)
  NoiseReq hint;                         // the caller's own prediction of the next request (vb_noise_hint_seed), used once
  int64_t delta = 0;
  int streak = 0;
};
struct NoiseSlot {
  DeviceBuffer buf;
  int64_t n = 0, d = 0, ld = 0;   // ld: row stride in doubles (multiple of 16)
  NoiseAhead ahead;
  // sum_c e_rc^2 of every row (rng_normal_kernel), formed while the Philox normals are generated once a reader has asked for
  // them (want_norms: noise_row_norms); they describe the buffer's contents only while norms_req equals ahead.last
  DeviceBuffer norms;
  NoiseReq norms_req;
  bool want_norms = false;
};
// the row norms of the slot's CURRENT contents (at most 512 columns: else nullptr): the ones formed with the contents, or --
// contents of another origin, or generated before anybody asked -- formed now by a pass over the matrix on `st`
// (rng_row_norms_kernel: the same bits); from now on the slot's Philox fills bring theirs along
const double* noise_row_norms(vb_ctx* ctx, NoiseSlot& s, hipStream_t st);
int rng_row_norms(vb_ctx* ctx, hipStream_t st, const double* src, int64_t ld, int64_t n, int64_t d, double* norms);

// Software pipeline over three HIP streams (prep | streaming kernel | finalize + collectives), used
// by the asynchronous batch entry points so that consecutive batches overlap.
constexpr int kPipeSets = 4;
struct Pipeline {
  hipStream_t pre = nullptr, post = nullptr;
  hipEvent_t ev_main = nullptr;
  hipEvent_t ev_prep[kPipeSets] = {}, ev_k1[kPipeSets] = {}, ev_fin[kPipeSets] = {};
  bool fin_valid[kPipeSets] = {};
  uint64_t seq = 0;
  int last_set = 0;
  bool main_dirty = true;     // main-stream work was enqueued that `pre` has not been ordered after
  bool post_pending = false;  // `post` holds work the main stream has not been ordered after
};

struct ResultSlot {
  double* host = nullptr;   // pinned, device-mapped: [theta staging (p) | value | grad (p)]
  double* dev = nullptr;    // device address of `host`
  int64_t p = 0;
  bool pending = false;
  uint64_t batch_id = 0;    // enqueue ticket of the evaluation that last used this slot (0: none)
};

}  // namespace vb

namespace vb { struct LegacySpec; }

struct vb_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipDeviceProp_t prop;
  std::string last_error;

  vb::NoiseSlot noise[VB_MAX_SLOTS];
  vb::NoiseSlot gen_geom;               // geometry of the matrix an in-register (GEN) evaluation stands for
  vb::ResultSlot results[VB_MAX_SLOTS];
  vb::ResultSlot sync_result;           // used by the synchronous entry points
  // completion words of the blocking mean-field call (pinned, device-mapped): finalize workgroup g stores the call's
  // sequence number into done_host[8 g] behind its results, the host polls them instead of waking up through
  // hipStreamSynchronize (VB_MF_FLAGSYNC=0 turns this off)
  unsigned long long* done_host = nullptr;
  unsigned long long* done_dev = nullptr;
  unsigned long long done_seq = 0;
  int done_groups = 0;                  // workgroups that signal for the call in flight (0: wait on the stream)

  vb::ModelDev model;
  vb::DeviceBuffer model_params;        // device copy of the model's double parameters
  std::vector<double> model_host;       // host copy (epilogue constants)

  vb::DeviceBuffer theta;               // device copy of the variational parameter
  vb::DeviceBuffer workspace;           // per-evaluation work buffers of the mean-field pipeline
  vb::DeviceBuffer sums;                // reduced sums (the vector that is all-reduced)
  vb::DeviceBuffer mf_one;              // one-launch mean-field evaluation: tickets, publication flags, parked sums
  unsigned mf_one_epoch = 0;
  vb::DeviceBuffer out;                 // [value | grad] on the device
  vb::DeviceBuffer scratch;             // generic device scratch (x upload, ...)
  vb::DeviceBuffer scratch2;            // per-row outputs
  vb::DeviceBuffer rowvec;              // per-row weights
  vb::DeviceBuffer fr_work;             // full-rank pipeline work buffers
  vb::DeviceBuffer lg_work;             // logistic-regression target: Z, R, G, partials
  hipModule_t user_module = nullptr;    // VB_MODEL_SOURCE: the compiled user model in use (vb_usermodel.hip)
  hipFunction_t user_fn = nullptr;
  struct UserModule {                   // every source compiled by this context, by content hash: binding a model
    uint64_t hash;                      // again (two objectives taking turns) costs no second hiprtc run
    hipModule_t module;
    hipFunction_t fn;
    int parts;                          // threads per sample (VB_LOG_DENSITY_PARTS of the source, 1 without)
  };
  int user_parts = 1;                   // ... of the module in use
  // a host callable with its gradient (vb_set_model_callback): the row "kernel" is a round trip through pinned memory
  vb_model_callback user_host_fn = nullptr;
  void* user_host_arg = nullptr;
  double* user_host_pin = nullptr;      // [z (n x d) | f (n) | g (n x d)]
  size_t user_host_pin_doubles = 0;
  std::vector<UserModule> user_modules;
  vb::DeviceBuffer user_params;
  vb::DeviceBuffer glm_work;            // regression targets: split-K slabs of the gradient GEMM
  vb::DeviceBuffer mvt_state;           // multivariate-t DIS: state samples X, scratch
  int64_t mvt_n = 0, mvt_d = 0, mvt_n_total = 0;
  int64_t mvt_lq_off = 0;               // where this rank's log q of the residual pass start inside o_lq (the refresh writes them
                                        // at the shard's offset of the gathered vector, a gradient at another parameter at 0)
  vb::DeviceBuffer mvt_ekl_state;       // the resident ExclusiveKL / AlphaDivergence of the t family: factor algebra + sums (NOT the
                                        // DIS state's buffer: an ELBO monitor beside a DIS fit must leave its samples alone)
  std::vector<double> mvt_theta;        // parameter the device-side residuals of the DIS state belong to
  bool mvt_dev_factors = false;         // ... and its factors (L, L', L^-1) were formed on the device
  const double* mvt_e_noise = nullptr;  // the residuals E' of that parameter are NOT stored: E'_n = noise_n / s_n (this matrix,
  int64_t mvt_e_noise_ld = 0;           // row stride mvt_e_noise_ld; see mvt_residuals)
  std::vector<double> mvt_stage;        // host staging of the factor uploads (one synchronisation per pass)
  double* mvt_pin = nullptr;            // pinned staging of the throughput mode's parameter upload (no synchronisation)
  size_t mvt_pin_doubles = 0;
  int mvt_pin_slot = 0;
  hipEvent_t mvt_pin_ev[2] = {nullptr, nullptr};   // recorded behind each slot's staged copy; waited for before the slot is rewritten
  void* push_host = nullptr;            // two mapped staging slots of push_small (vb_api.hip) and their device address
  void* push_dev = nullptr;
  size_t push_bytes = 0;                // ... per slot
  int push_slot = 0;
  hipEvent_t push_ev[2] = {nullptr, nullptr};
  void* fetch_host = nullptr;           // mapped host memory of fetch_blocking (vb_api.hip): segments | completion word,
  void* fetch_dev = nullptr;            // and its device address
  size_t fetch_bytes = 0;
  unsigned long long fetch_seq = 0;
  vb::DeviceBuffer fetch_ticket;        // the copy kernel's workgroup ticket (zero between launches)
  hipStream_t mvt_side = nullptr;       // side stream of the deferred triangular inverse (mvt_factors_device)
  hipEvent_t mvt_ev_fork = nullptr, mvt_ev_join = nullptr;
  hipEvent_t done_ev = nullptr;      // wait_then_prefetch
  uint64_t ahead_generated = 0, ahead_adopted = 0;      // look-ahead buffers generated / adopted (vb_noise_ahead_stats)
  bool mvt_inv_pending = false;         // the main stream has not yet waited for the side stream's inverse
  bool mvt_inv_queued = false;          // ... which has been enqueued already (else mvt_inv_args describes it)
  struct {
    double* base = nullptr;
    int64_t o_theta = 0, o_lt = 0, o_wt = 0, o_tscr = 0, o_mu = 0, o_li = 0, o_lfull = 0, o_c = 0, ld = 0;
    int d = 0;
    bool clean = false;
    bool lfull_here = false;            // the side stream's prep also forms L = (L')' (the main stream had no prep launch)
  } mvt_inv_args;
  double* mvt_pin_dev = nullptr;        // device address of mvt_pin (mapped: the unpack reads the parameter in place)
  std::vector<double> mvt_prior;        // tempering-prior parameter the device copy was made from
  int64_t mvt_inv_key[4] = {0, 0, 0, 0};   // (state buffer, n, n_total, d) for which the inverse's zero triangle is known clean
  vb::DeviceBuffer dis_state;           // DIS: [cols of the refresh theta | log p | base b | log prior | w]
  int64_t dis_n = 0, dis_d = 0;         // shape of the DIS state (0: none)
  int64_t dis_n_total = 0;              // whole-job sample count of the DIS state
  vb::DeviceBuffer chi_dev;             // device-generated chi-square draws (vb_chisq_generate)
  vb::DeviceBuffer mvt_invs;            // 1 / s_n of the throughput-mode t ExclusiveKL (vb_elbo_grad_mvt_chol)
  vb::NoiseAhead chi_ahead;             // look-ahead generation of the next chi-square draws (noise_prefetch)
  int64_t chi_n = 0;                    // how many of them are valid (0: none)
  double chi_df = 0.0;
  vb::DeviceBuffer bisect_work;         // DIS tempering bisection: interval / ESS tables of the look-ahead rounds
  unsigned long long bisect_bar_base = 0;   // the resident bisection kernel's barrier counter before the next launch,
  size_t bisect_bar_words = 0;              // where in bisect_work it lives (doubles) and for which allocation it was zeroed
  void* bisect_bar_ptr = nullptr;
  vb::DeviceBuffer mvt_elbo;            // multivariate-t ExclusiveKL: root, mean, row scales
  vb::DeviceBuffer lr_work;             // low-rank Gaussian family: workspace of the streaming pipeline
  vb::DeviceBuffer lr_obj;              // low-rank Gaussian under DIS / alpha: samples, residuals, Woodbury vectors
  int64_t lr_n = 0, lr_d = 0, lr_k = 0, lr_n_total = 0;   // shape of the low-rank DIS state (0: none)
  struct TemperPrior {                  // vb_dis_set_temper_prior: a tempering prior other than the refresh's own argument
    int kind = 0;                       // VB_PRIOR_*
    int64_t d = 0, ld = 0;
    double df = 0.0, c0 = 0.0;          // additive constant of the log density
    vb::DeviceBuffer buf;               // diag t: [loc ld | 1/sigma ld];  dense: [W = L^-T (d x ld) | c = L^-1 loc (ld)]
    vb::DeviceBuffer work;              // dense: U (n x ld), maha / c_n (n each)
  } temper;
  uint64_t dis_gen[3] = {0, 0, 0};      // refresh counters of the DIS states: mean-field, dense (t / Gaussian), low-rank
  vb::DeviceBuffer rows_work;           // Model.__call__ for the dense targets: GEMM output rows
  vb::DeviceBuffer psis_lw;             // PSIS: log importance weights (+ 16 result scalars)
  int64_t psis_n = 0;                   // number of device-resident log weights (0: none)
  vb::DeviceBuffer psis_work;           // the multi-workgroup smoothing's exchange area (barrier counter, histograms, tail lists)
  unsigned long long psis_bar_base = 0; // value of that counter before the next launch
  double* pin_host = nullptr;           // pinned, device-mapped staging (vb_linalg.hip): host / device address
  double* pin_dev = nullptr;
  size_t pin_bytes = 0;
  vb::DeviceBuffer fit_work;            // device-resident fit: [theta | out | state | value history | iterates]
  int64_t fit_out_off = 0;
  uint64_t mvt_epi_rows_calls = 0, mvt_chain_fetch_calls = 0;      // vb_mvt_route_stats
  int ns_hint_m[2] = {0, 0}, ns_hint_steps[2] = {0, 0};      // steps the last Newton-Schulz root ([0]) / Frechet iteration ([1]) of size m ended with
  int64_t fit_hist_off = 0, fit_hist_len = 0, fit_hist_p = 0;      // where the last fit's kept iterates sit in fit_work (len 0: none)
  // vb_fit's per-iteration rows (iterates, directions, gradients) leave while the next iterations run: a copy stream, a
  // ring of pinned slots, one event pair per slot (vb_api.hip, FitRowStream)
  static constexpr int kFitRing = 4;
  hipStream_t fit_copy_st = nullptr;
  double* fit_ring = nullptr;
  size_t fit_ring_doubles = 0;          // capacity of ONE slot
  hipEvent_t fit_ev_step[kFitRing] = {}, fit_ev_copy[kFitRing] = {};
  vb::DeviceBuffer tri_map;             // XCD-aware tile list of the lower-triangular gradient GEMM (int pairs)
  int tri_map_key[3] = {0, 0, 0};       // (d, tile rows, tile columns) the list was built for
  int tri_map_blocks = 0;
  vb::DeviceBuffer fr_lt;               // full-rank: unpacked parameter [mu (ldz) | L' (d x ldl)]
  int64_t fr_lt_d = 0;                  // dimension the unpacked copy of fr_theta was made for (0: stale)
  const double* fr_lt_owner = nullptr;  // a parameter other than fr_theta whose unpacked copy fr_lt currently holds (vb_fit's)
  vb::DeviceBuffer fr_theta;            // full-rank: resident flat parameter
  vb::DeviceBuffer fr_out;              // full-rank: [value | grad] on the device
  int64_t fr_p = 0;                     // length of the resident full-rank parameter
  // the blocking call's pipelined parameter upload (vb_elbo_grad_fullrank, round 6): the flat parameter crosses PCIe in
  // row chunks of L, HEAVIEST rows first, on `up_stream`; behind each chunk its columns of L' are unpacked and an event is
  // recorded; the sampling product of a chunk's column blocks starts behind its event (column block b of Z = E L' + mu needs
  // rows [64 b, 64 b + 64) of L only) while the lighter rows are still in flight.
  struct FrUpload {
    int n_chunks = 0;
    int bn_begin[4] = {0, 0, 0, 0}, bn_count[4] = {0, 0, 0, 0};      // column blocks of 64 per chunk, chunk 0 = the last rows
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};           // behind chunk c's copy + unpack
    bool consumed = false;               // the pipeline started the sampling product chunk by chunk (else it waits for ev[last])
  } fr_up;
  bool fr_up_active = false;            // fr_up describes THIS evaluation's parameter
  bool fr_busy = false;                 // an asynchronous evaluation (vb_elbo_grad_fullrank_enqueue) may still read fr_theta / fr_lt
  hipStream_t up_stream = nullptr;      // copies + unpacks of the chunks
  hipStream_t up_side[2] = {nullptr, nullptr};   // sampling products of chunks 1, 2 (chunk 0's runs on the main stream)
  hipEvent_t up_ev_main = nullptr, up_ev_join[2] = {nullptr, nullptr};
  uint64_t fr_up_calls = 0;             // evaluations that took the pipelined route (observability: tests)
  // numpy's legacy normal stream on the device (vb_legacy_dev.hip): scratch; where the jump polynomials were uploaded
  vb::DeviceBuffer alpha_g;             // AlphaDivergence, correlated-Gaussian target: G of the samples (see FrWeighted::g_ready)
  vb::DeviceBuffer legacy_work;
  const void* legacy_poly_at = nullptr;
  size_t legacy_poly_bytes = 0;
  bool legacy_table_ready = false;      // the double-double log's table is in this device's constant memory
  vb::LegacySpec* legacy_spec = nullptr;   // look-ahead generation of the next call's numpy-stream draws (created on first use)
  double* legacy_pin = nullptr;         // pinned staging of the device draw's results
  size_t legacy_pin_doubles = 0;
  // fused full-rank evaluation (vb_fullrank_fused.h): ticket counter, error word and tile flags; the work list
  vb::DeviceBuffer fz_words, fz_items;
  int64_t fz_key[5] = {0, 0, 0, 0, 0};  // (n, d, splits, phases, tile_blocks) the list was built for
  int fz_n_items = 0;
  unsigned fz_epoch = 0;
  int fr_fused_mode = -1;               // -1: VB_FR_FUSED from the environment; 0 off, 2 / 3 phases fused
  uint64_t fr_seq = 0;                  // sharded full-rank evaluations enqueued (selects the sum set)

  void* comm = nullptr;                 // ncclComm_t when a communicator is attached (the context itself under a
                                        // host-staged transport: non-NULL means "this is a sharded job" everywhere)
  int n_ranks = 1, rank = 0;
  vb_host_collective_fn host_fn = nullptr;   // vb_comm_init_host: the caller's collective over host memory
  void* host_user = nullptr;
  double* host_stage = nullptr;         // pinned staging buffer of the host-staged transport
  size_t host_stage_cap = 0;            // ... in doubles
  // xGMI-native transport (vb_comm_init_ipc): every rank's window [3 flag words | data cap | result cap], the peers'
  // windows mapped through IPC handles; ipc_seq counts the collectives (the flags carry it)
  struct IpcComm {
    double* win[16] = {};               // [rank] window base (own allocation at [rank], the others hipIpcOpenMemHandle'd)
    size_t cap = 0;                     // doubles per data / result area
    unsigned long long seq = 0;
    unsigned* ticket = nullptr;         // device: last-block tickets of the three phases
    unsigned* err_host = nullptr;       // pinned, device-mapped: a poll gave up (checked by comm_check)
    unsigned* err_dev = nullptr;
    int poll_log2 = 32;                 // optional poll-count bound of the device-side waits (VB_IPC_POLL_LOG2; 32: none)
    double timeout_s = 20.0;            // their wall-time bound (VB_IPC_TIMEOUT_S)
    bool on = false;
  } ipc;

  bool profile = false;
  struct ProfLog {                      // one per profiled kernel id (VB_PROF_*)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t used = 0;
    int64_t evals = 0;                  // evaluations covered by the recorded launches
  } prof[VB_PROF_NUM];

  // completion tickets: enqueue k records batch_events[k % size] after its last kernel, so a
  // result slot is never re-staged while the evaluation that used it is still in flight
  std::vector<hipEvent_t> batch_events;
  uint64_t batch_id = 0, batch_done = 0;

  vb::Pipeline pipe;
  hipStream_t result_stream = nullptr;  // stream the last enqueue's results are produced on
};

namespace vb {

int fail(vb_ctx* ctx, int code, const char* fmt, ...);
int ensure(vb_ctx* ctx, DeviceBuffer& b, size_t bytes);
int ensure_pinned(vb_ctx* ctx, size_t bytes);   // ctx->pin_host / pin_dev hold at least `bytes`

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

// mean-field ExclusiveKL pipeline (vb_meanfield.hip): a batch of independent evaluations
struct MfCall {
  int count = 0;
  const NoiseSlot* noise[kMaxBatch] = {};
  const double* theta_src[kMaxBatch] = {};   // device-visible [mu | log_sigma]
  double* out[kMaxBatch] = {};               // device-visible [value | grad(2D)]
  const double* roww[kMaxBatch] = {};        // device per-row weights or nullptr
  int64_t n = 0, d = 0, n_total = 0;
  int family = 0;
  double df = 0.0;
  unsigned flags = 0;
  int cv_mode = 0;
  int mode = 0;        // 0: ELBO (ExclusiveKL); 1: weighted gradient only
  bool pipelined = false;   // spread prep / stream / finalize over the three pipeline streams
  bool alternate = false;      // single GPU, independent asynchronous batches: odd workspace sets run on the
                               // `post` stream, even ones on the main stream, so the small prep / finalize kernels of
                               // one batch overlap the streaming kernel of the other (no events between them)
  bool overlap_comm = false;   // sharded job: all-reduce + epilogue on the `post` stream, so the next batch's
                               // kernels run on the main stream while RCCL moves this batch's sums
  double scale = 0.0;  // mode 1
  const double* value_src = nullptr;   // mode 1: device scalar reported as the objective value
  const ModelDev* model = nullptr;     // overrides ctx->model (mode 2: the log-q pseudo model)
  // gen != 0 (single ELBO evaluation, Gaussian base noise, gauss_diag / funnel target): the streaming kernel
  // generates its noise in registers -- element (gen_row_offset + n, col) of Philox stream (gen_seed, gen_stream),
  // the values vb_noise_generate would have written -- instead of reading the noise slot
  int gen = 0;
  uint64_t gen_seed = 0, gen_stream = 0;
  int64_t gen_row_offset = 0;
  // device-resident fit: optimiser step to apply to (value, grad); the finalize kernel does it itself when it
  // also computes the gradient (single GPU, no control variate) and sets *step_done, otherwise the caller launches
  // fit_step_kernel
  const struct FitStep* step = nullptr;
  bool* step_done = nullptr;
  // ... and, with in-register noise, the prep of the NEXT iteration (its theta is final once the step is applied):
  // prep_next asks for it, *prep_done reports it, skip_prep tells the next call that its prep has been done
  bool prep_next = false, skip_prep = false;
  bool* prep_done = nullptr;
  bool theta_on_device = false;   // theta_src[0] is device memory (the fit loop's iterate), not a pinned staging copy
  // blocking call: ask the finalize kernel to signal completion through these words (see vb_ctx::done_host); *done_groups
  // reports how many workgroups will (0: this evaluation does not end in the fused finalize -- wait on the stream)
  unsigned long long* done_dev = nullptr;
  unsigned long long done_seq = 0;
  int* done_groups = nullptr;
};
int mf_enqueue(vb_ctx* ctx, const MfCall& call);
// user model given as HIP source (vb_usermodel.hip)
int user_model_set(vb_ctx* ctx, int64_t dim, const char* source, const double* params, size_t n_params);
int user_rows_enqueue(vb_ctx* ctx, hipStream_t st, const double* Z, int64_t ldz, int64_t n, int d, double* G,
                      int64_t ldg, double* f);
void user_model_release(vb_ctx* ctx);
struct LegacyFinish;
// (defer != nullptr: the kernels are enqueued, the state is left untouched and *defer describes the finish -- exact path only,
// VB_ERR_UNSUPPORTED otherwise)
int legacy_dev_randn(vb_ctx* ctx, uint32_t key[624], int* pos, int* has_gauss, double* gauss, const NoiseSlot& ns,
                     int64_t n_total, int64_t d, int64_t row_begin, int64_t rows, LegacyFinish* defer = nullptr);
// numpy's legacy word stream on the device (vb_legacy_dev.hip) for the draws built on it (vb_legacy_gamma.hip)
struct FetchSeg {                  // `bytes` (a multiple of 8) from device address `src` (8-byte aligned) to host address `dst`
  const void* src;
  size_t bytes;
  void* dst;
};
struct LegacyWords {
  const uint32_t* words = nullptr;   // untempered output words, words[0] = the word at the generator's position
  int64_t* scal = nullptr;           // 8 zeroed 64-bit scalars
  uint32_t* extra = nullptr;         // the caller's scratch
  const void* logtab = nullptr;      // GlibcLogData on the device, or NULL when the host's log could not be restated
  int64_t pre = 0, n_words = 0;      // words left in the generator's current block; words generated
  uint32_t* key_io = nullptr;        // 624 words: the uploaded key; legacy_mt_finish_fetch gathers the end block here
  int64_t* meta = nullptr;           // 2 scalars of that gather: [status, new position]
};
// What the host still has to do when a device draw's kernels have run (legacy_dev_randn / legacy_dev_gamma with `defer`):
// find the generator's end block, bring the draw's scalars and that block to the host, decide whether the draw stands
// (word budget), and form the generator's new state.  The blocking draws do all of it at once (legacy_mt_finish_fetch);
// the look-ahead draws of round 6 (LegacySpec, vb_api.hip) launch it behind the draw on their own stream
// (legacy_finish_launch), poll for it from the main thread's waits and complete it when the next call asks for the values.
struct LegacyFinish {
  LegacyWords lw;
  const int64_t* src_dev = nullptr;      // consumed words w_star = mult * (src_dev[0] + add)
  int64_t mult = 0, add = 0;
  const void* extra_src[2] = {nullptr, nullptr};      // up to two segments of at most 8 int64 each
  int extra_words[2] = {0, 0};
  int kind = 0;                          // 0: randn (extras: 4 scalars | accepted pairs), 1: gamma-based (extras: 4 end scalars)
  int64_t pairs = 0, n_vals = 0;         // randn
  int pos_in = 0;
};
constexpr int kLegacyFinishWords = 352;      // uint64 words of the host landing area: [extras 16 | block 312 | meta 2 | w | ... | done]
int legacy_finish_launch(vb_ctx* ctx, hipStream_t st, const LegacyFinish& f, unsigned long long* landing_dev, unsigned long long seq);
// VB_OK: the draw stands, (key, pos, has_gauss, gauss) are the generator's state behind it; VB_ERR_UNSUPPORTED: declined
int legacy_finish_complete(vb_ctx* ctx, const LegacyFinish& f, const unsigned long long* landing, uint32_t key[624], int* pos,
                           int* has_gauss, double* gauss);
int legacy_mt_words(vb_ctx* ctx, const uint32_t key[624], int pos, int64_t n_words, size_t extra_u32, LegacyWords* out);
int legacy_mt_finish(vb_ctx* ctx, const LegacyWords& lw, int64_t w_star, uint32_t key[624], int* pos);
// The same with the end position still on the device: w_star = mult * (src_dev[0] + add).  A one-workgroup kernel finds
// the end block and gathers its words; ONE fetch_blocking brings the caller's result segments (extra), the block and the
// new position.  accept(extra results) decides on the host whether the draw stands (false: nothing is changed, returns
// VB_ERR_UNSUPPORTED).
int legacy_mt_finish_fetch(vb_ctx* ctx, const LegacyWords& lw, const int64_t* src_dev, int64_t mult, int64_t add,
                           const FetchSeg* extra, int n_extra, bool (*accept)(void*), void* accept_arg, uint32_t key[624],
                           int* pos);
// chisquare (prog 0) / standard_t (prog 1) draws, values o_first ... n - 1 of the request, into rows of a noise-slot-like
// array (vb_legacy_gamma.hip); the generator must hold no cached normal
int legacy_dev_gamma(vb_ctx* ctx, int prog, double df, uint32_t key[624], int* pos, int* has_gauss, double* gauss,
                     double* dst, int64_t ld, int64_t o_first, int64_t n_total, int64_t d, int64_t row_begin, int64_t rows,
                     LegacyFinish* defer = nullptr);
void vb_legacy_finish_pairs(const double* list, int64_t n, double* fixed);      // vb_legacy_rng.cpp (host libm)

// ---- look-ahead generation of numpy's legacy streams (round 6; vb_api.hip: legacy_spec_*) --------------------------------
// A family in the reference-identical mode asks for the same draws call after call (randn(N, D); standard_t(df, (N, D));
// chisquare(df, N) then randn(N, D)) from one persistent RandomState.  The draws' host decisions need the generator's state
// BEFORE them, and that is known the moment the previous call's draws are complete: at the end of a call's draws
// (vb_legacy_round_end) the NEXT call's draws are started from that state on a stream of their own, into shadow buffers,
// with scratch of their own -- beside the objective's kernels, which the reference-identical mode used to run strictly
// behind 0.3-1.1 ms of generation.  A draw request that finds the generator where the speculation started (all 624 words,
// position, cached normal) and asks for exactly what was speculated adopts the shadow by a pointer swap and sets the
// generator to the speculated end state; anything else (a host draw in between, another shape, a reseed) discards the
// speculation and draws as before -- values and states are numpy's either way.  No thread: the second request of a round
// is enqueued from the main thread's waits (fetch_blocking polls legacy_spec_poll) once the first one's finish has landed.
struct LegacyReq {
  int prog = -2;            // -1: randn, 0: chisquare, 1: standard_t
  double df = 0.0;
  int64_t n_total = 0, d = 0, row_begin = 0, rows = 0;
  int slot = -1;
  bool same(const LegacyReq& o) const {
    return prog == o.prog && df == o.df && n_total == o.n_total && d == o.d && row_begin == o.row_begin && rows == o.rows &&
           slot == o.slot;
  }
};
struct LegacyGenState {
  uint32_t key[624];
  int pos = 0, has_gauss = 0;
  double gauss = 0.0;
};
struct LegacySpec {
  static constexpr int kMaxReqs = 3;
  LegacyReq round[kMaxReqs], prev_round[kMaxReqs];      // the draws since the last round end; of the round before
  int n_round = 0, n_prev = -1;
  const void* round_rng = nullptr;
  uint64_t round_uid = 0;                   // ... and its uid: a new generator may live at a destroyed one's address
  bool round_overflow = false;
  bool active = false, failed = false;
  LegacyReq reqs[kMaxReqs];
  int n_reqs = 0, n_adopted = 0, n_finished = 0, in_flight = -1;
  LegacyGenState state[kMaxReqs + 1];      // state[i]: the generator before request i
  LegacyGenState head_state;               // ... and behind the host-drawn head of the request in flight
  LegacyFinish fin;
  DeviceBuffer shadow[kMaxReqs];
  int64_t shadow_d[kMaxReqs] = {0, 0, 0}, shadow_ld[kMaxReqs] = {0, 0, 0};      // layout the shadow's pad columns are zero for
  DeviceBuffer work;                   // the speculation's own scratch (ctx->legacy_work's twin) and what goes with it
  const void* poly_at = nullptr;
  size_t poly_bytes = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_main = nullptr;
  unsigned long long* land_host = nullptr;   // mapped landing area of the deferred finish (kLegacyFinishWords)
  unsigned long long* land_dev = nullptr;
  unsigned long long seq = 0;
  ::vb_legacy_rng* clone = nullptr;
  uint64_t launched = 0, adopted = 0, discarded = 0;
  // a caller whose rounds look alike but whose generator moves in between (host draws: a chi-square vector below the device
  // gate, sample() calls) would pay a wasted generation and a wait for it every call: two jobs discarded in a row stop the
  // speculation for the next 64 rounds of this generator
  int discard_streak = 0, cooldown = 0;
};


// log density of the installed tempering prior (ctx->temper.kind != 0) at the rows of X
int mvt_dis_clip_enqueue(vb_ctx* ctx, int64_t n_total, double threshold);
int mvt_dis_scalars_get(vb_ctx* ctx, double out[4]);
int mvt_elbo_symroot(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t n_total, int64_t d, double df,
                     const double* theta_host, double* value_grad_host, double* info, bool path_deriv = false);
int mvt_alpha_symroot(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t n_total, int64_t d, double df, double alpha,
                      const double* theta_host, double* value_grad_host, double* info);
int sym_sqrt_dev(vb_ctx* ctx, const double* Lfull, const double* Lt, int64_t d, int64_t ld, double* root, double tol,
                 double* info, double* inv_root = nullptr);      // vb_linalg.hip
int sym_sqrt_frechet_dev(vb_ctx* ctx, const double* Lfull, const double* Lt, const double* E, int64_t d, int64_t ld, double* X,
                         double tol, double* info);
int temper_prior_rows(vb_ctx* ctx, const double* X, int64_t ld, int64_t n, int64_t d, double* out);
int temper_prior_set(vb_ctx* ctx, int kind, int64_t d, double df, const double* loc, const double* scale, double log_det_l);
int user_model_set_callback(vb_ctx* ctx, int64_t dim, vb_model_callback fn, void* user);
int pipe_init(vb_ctx* ctx);

// per-row log weights and AlphaDivergence (vb_rowstats.hip)
int rowstats_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, const double* theta_src,
                     const ModelDev& model, int student, double df, double* cols, double* scal,
                     double* out_f, double* out_b, const ModelDev* prior = nullptr,
                     double* out_prior = nullptr);
int dis_refresh_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t n_total, int64_t d, int family, double df,
                        const double* theta_src, const double* prior_host, double eps_prev, double ess_target,
                        int max_its, double* eps_out, double* ess_out, int* status_out, double* w_host,
                        double* logp_host, double* logq_host);
int dis_grad_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int family, double df,
                     const double* theta_src, const double* w_host, double scale, double* out);
int mvt_elbo_sums(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int64_t n_total, const double* mu_host,
                  const double* root_host, const double* inv_s_host, double* f_sum, double* g_sum, double* c_full);

int mvt_alpha_chol_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int64_t n_total, double df, double alpha,
                           const double* theta_dev, double sum_log_diag, double* out_dev);
int mvt_alpha_sums(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int64_t n_total, double df, double alpha,
                   const double* mu_host, const double* root_host, const double* inv_s_host, double sum_log_diag,
                   double* value, double* w_sum, double* g_sum, double* c_full);
int alpha_fullrank_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t n_total, int64_t d, double alpha,
                           const double* theta_dev, double sum_log_diag, double* out, double df = 0.0,
                           const double* inv_s = nullptr);

// vb_lowrank.hip
int lr_elbo_grad_enqueue(vb_ctx* ctx, const NoiseSlot& ns, const NoiseSlot& nz, int64_t n, int64_t d, int64_t k,
                         int64_t n_total, const double* theta_src, double* out);

// vb_psis.hip
int log_weights_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int family, double df,
                        const double* theta_src);
int psis_enqueue(vb_ctx* ctx, int64_t n, double reff, const double* weights_in = nullptr, double* weights_out = nullptr,
                 double* khat_out = nullptr, bool* fused_out = nullptr);
int mvt_dis_psis_enqueue(vb_ctx* ctx, int64_t n_total, double reff);
int psis_tail_size(int64_t n, double reff);

int alpha_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t n_total, int64_t d, int family, double df,
                  double alpha, const double* theta_src, double* out);
int sync_streams(vb_ctx* ctx);   // main + pipeline streams
// blocking device -> host fetch of small results: one gathering kernel into mapped memory + a polled completion word
// (vb_api.hip); plain copies + hipStreamSynchronize above 1 MB
int fetch_blocking(vb_ctx* ctx, hipStream_t st, const FetchSeg* segs, int n_segs);
// A look-ahead draw of numpy's streams may have finished (its end state landed in pinned memory): take note and enqueue the
// job's NEXT request, which starts from that state.  Cheap (one pinned word read); called wherever the host has just waited
// for the device anyway -- a job of two draws (the t family: chi-square, then normals) used to get its second draw started
// only at the call's final wait, 0.45 ms after the first had finished (round 6, profiles/r06_c3_parity_timeline.txt).
void legacy_poll(vb_ctx* ctx);
// the same in two halves around a producer kernel that writes the mapped buffer itself (vb_api.hip)
struct FetchPlan {
  bool ok = false;
  int n = 0;
  long long first[9] = {0};            // first word of segment k in the mapped buffer (64-byte aligned); [n] = total
  unsigned long long* host = nullptr;  // the mapped buffer, host / device address
  unsigned long long* dev = nullptr;
  size_t o_done = 0;                   // word index of the completion word
  unsigned long long* done_dev = nullptr;
  unsigned* ticket = nullptr;          // device counter, zero between launches
  unsigned long long seq = 0;
};
int fetch_plan(vb_ctx* ctx, const FetchSeg* segs, int n_segs, FetchPlan* plan);
int fetch_wait(vb_ctx* ctx, hipStream_t st, const FetchPlan& plan, const FetchSeg* segs);
// host -> device copy of a small caller-owned array without a synchronisation (mapped staging slots + a copy kernel)
int push_small(vb_ctx* ctx, hipStream_t st, const void* host_src, size_t bytes, void* dev_dst, size_t row_bytes = 0,
               size_t dst_stride_bytes = 0);      // row_bytes != 0: rows of row_bytes land dst_stride_bytes apart
int wait_then_prefetch(vb_ctx* ctx);   // host waits for the work enqueued so far; the look-ahead noise goes behind it
void noise_prefetch(vb_ctx* ctx);   // look-ahead Philox generation; call right before a blocking call starts to wait (vb_api.hip)
int comm_check(vb_ctx* ctx);     // VB_ERR_COMM when a device-side wait of the IPC transport has given up (vb_comm.hip)

// full-rank Gaussian ExclusiveKL (vb_fullrank.hip)
struct FrSums;
// weighted sums of the dense-Gaussian pipeline (AlphaDivergence): rows of G scaled by roww[n]; result
// scale * [sum s g | tril(sum s g eps'), free diagonal x L_ii + wsum[0]], value = value[0] (device scalars)
struct FrWeighted {
  const double* roww;
  double scale;
  const double* wsum;
  const double* value;
  // the samples Z = mu + (E root) [/ s] of this very noise and parameter, already formed by the caller (row stride
  // round_up(d, 16); the alpha-divergence weights needed them): the pipeline does not repeat the sampling product for
  // the targets that read plain samples (funnel, regression, source models)
  const double* z_ready = nullptr;
  // ... and, for the correlated-Gaussian target, the model's gradient rows G of those samples (the weights needed f,
  // which costs the same N x D x D product): the pipeline then skips its sampling and model products
  const double* g_ready = nullptr;
};
int fr_sample_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, const double* theta_dev, double* Z,
                      const double* mu_dev = nullptr, const double* root_dev = nullptr,
                      const double* row_scale = nullptr);
int alpha_mvt_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t n_total, int64_t d, double df, double alpha,
                      const double* mu_dev, const double* root_dev, const double* invs_dev, double sum_log_diag,
                      FrSums* sums, const double** value_wsum);
int fr_pipeline_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int64_t n_total,
                        const double* theta_dev, double* out_dev, const double* mu_dev, const double* root_dev,
                        const double* row_scale, FrSums* sums_out, unsigned flags = 0,
                        const FrWeighted* weighted = nullptr);
int fr_elbo_grad_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int64_t n_total,
                         const double* theta_dev, double* out_dev, unsigned flags = 0,
                         const FrWeighted* weighted = nullptr);

// sum vector of the dense paths: [F | column sums (ldz) | C (d x ldl)]
struct FrSums {
  double* sums;
  int64_t off_col, off_c, len;
};
struct FitStep;
// optimiser step of the dense family fused with the unpack of the stepped parameter (vb_fit): theta <- step(theta, grad)
// and mu, L' of the NEW theta into fr_lt in one kernel; the next evaluation of `theta_dev` skips its unpack
int fr_step_unpack_enqueue(vb_ctx* ctx, const FitStep& a, int64_t d);
int fr_upload_begin(vb_ctx* ctx, const double* theta_host, int64_t d);
int fr_unpack_enqueue(vb_ctx* ctx, hipStream_t st, const double* theta_dev, int D, int64_t ldl, double* Lt, double* mu,
                      double* theta_copy = nullptr);
int fr_tri_inverse_enqueue(vb_ctx* ctx, hipStream_t st, const double* theta_dev, const double* Lt, int D, int64_t ldl,
                           double* Xa, double* T, bool clean = false);
int fr_colsum_enqueue(vb_ctx* ctx, const double* G, const double* Zc, int64_t ldz, int64_t n, int d, int fmode,
                      const double* ivar, double* colpart, double* fpart, const double* roww = nullptr,
                      int square = 0);   // square: column sums of the squared entries
int fr_reduce_enqueue(vb_ctx* ctx, const double* Cpart, int splits, int64_t slab, int d, int64_t ldl,
                      const double* colpart, int n_rb, int64_t ldz, const double* fpart, int n_fpart, FrSums S, bool mirror = false,
                      const double* wpart = nullptr);
// lower triangle of C = A' B for k-major A, B (n x d, row stride ld), split over n into `splits` slabs
int lr_elbo_sums_any_rank(vb_ctx* ctx, const NoiseSlot& ns, const NoiseSlot& nz, int64_t n, int64_t d, int64_t k,
                          const double* theta_host, double* out_host);
int mvt_elbo_chol_enqueue(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int64_t n_total, double df,
                          const double* theta_dev, double* out_dev);
int noise_moments(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, double* colsum_host, double* gram_host);
int gram_lower_enqueue(vb_ctx* ctx, const double* A, const double* B, int64_t ld, int d, int64_t n, int splits,
                       double* Cpart, int64_t ldc, int64_t slab);
int gram_lower_colsum_enqueue(vb_ctx* ctx, const double* A, const double* B, int64_t ld, int d, int64_t n, int splits,
                              double* Cpart, int64_t ldc, int64_t slab, double* colsum, int64_t colsum_ld,
                              int colsum_rows, bool* fused);
int gram_splits(vb_ctx* ctx, int d, int64_t n);
// ESS bisection of DISInclusiveKL (vb_rowstats.hip); lq = b - scal_in[0]
int dis_bisect_enqueue(vb_ctx* ctx, const double* lp, const double* b, const double* lprior, const double* scal_in,
                       int64_t n, double eps_prev, double ess_target, int max_its, double* w, double* lq_out,
                       double* scal_out);

// multivariate-t DIS (vb_mvt.hip)
int mvt_dis_refresh(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t n_total, int64_t d, double df,
                    const double* theta_host,
                    const double* chi_host, const double* root_host, const double* linv_host,
                    const double* prior_host, double eps_prev, double ess_target, int max_its, double* eps_out,
                    double* ess_out, double* w_host, double* logp_host, double* logq_host, bool sym_root = false,
                    double* root_info = nullptr);
int mvt_dis_state_get(vb_ctx* ctx, double* logp_host, double* logq_host, int64_t n_total);
int dis_state_get(vb_ctx* ctx, double* logp_host, double* logq_host, int64_t n_total);
int mvt_dis_grad(vb_ctx* ctx, int64_t n, int64_t d, double df, const double* theta_host, const double* linv_host,
                 const double* w_host, double* wsum_out, double* wlogq_out, double* dmu_out, double* gram_out,
                 double scale = 0.0, double* packed_out = nullptr, int64_t resample_m = 0, uint64_t seed = 0,
                 uint64_t stream = 0, double* res_out = nullptr, double* grad_direct = nullptr);
int mvt_dis_weights_get(vb_ctx* ctx, double* w_host, int64_t n_total, int resampled);

// symmetric square root / its derivative by coupled Newton-Schulz GEMM iterations (vb_linalg.hip)
int sym_sqrt(vb_ctx* ctx, const double* a, const double* e, int64_t d, double* root, double* x, double* info,
             double* inv_root = nullptr);
// low-rank Gaussian, path derivative: noise-only sums (vb_linalg.hip); out = [E'T (d x 2k) | T'T (2k x 2k) |
// sum eps (d) | sum eps^2 (d) | sum T (2k)], T = [z | u], u_n = sw' eps_n
int lr_path_terms(vb_ctx* ctx, const NoiseSlot& ns, const NoiseSlot& nz, int64_t n, int64_t d, int64_t k,
                  const double* sw_host, double* out_host);
// multivariate t, path derivative: noise-only sums (vb_mvt.hip)
int mvt_path_terms(vb_ctx* ctx, const NoiseSlot& ns, int64_t n, int64_t d, int64_t n_total, double df,
                   const double* inv_s_host, double* m_w, double* e_w, double* log1p_sum);

// one optimiser step on the device (vb_fit.hip): (value, grad) at `out` -> state, theta, histories
struct FitStep {
  int kind = 0;                 // VB_OPT_*
  int first = 0;                // the optimiser has no state yet (first descent_direction call)
  int64_t p = 0, k = 0;         // parameter length, iteration index
  double lr = 0.0, jitter = 0.0;
  double beta1 = 0.0, one_minus_beta1 = 0.0, beta2 = 0.0, one_minus_beta2 = 0.0;
  const double* out = nullptr;  // [value | grad (p)]
  double* theta = nullptr;      // updated in place
  double *s1 = nullptr, *s2 = nullptr;   // second-moment state, momentum
  double* values = nullptr;     // values[k] = value
  double* hist = nullptr;       // iterate k >= hist_first lands in row k - hist_first (nullptr: no history)
  int64_t hist_first = 0;
  double* dirs = nullptr;       // dirs[k * p + i] = descent direction (nullptr: not logged)
  double* grads = nullptr;      // grads[k * p + i] = gradient (nullptr: not logged)
};
int fit_step_enqueue(vb_ctx* ctx, const FitStep& step);

// Philox noise generation (vb_rng.hip)
int rng_fill(vb_ctx* ctx, double* dst, int64_t ld, int kind, double df, uint64_t seed,
             uint64_t stream, int64_t row_offset, int64_t n, int64_t d, double* norms = nullptr);
int rng_chisquare(vb_ctx* ctx, double* dst, double df, uint64_t seed, uint64_t stream, int64_t row_offset, int64_t n);

// regression targets: G = R X - Z / prior_sd^2 (n x d, contraction over the n_data observations), split over the
// observations when the output alone cannot fill the chip (vb_rows.hip)
int glm_grad_enqueue(vb_ctx* ctx, hipStream_t st, const ModelDev& m, const double* R, int64_t ldr, const double* Z,
                     double* G, int64_t ldz, int64_t n, int d);

// model log density for explicit x (vb_rows.hip)
int model_logp_rows(vb_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, int64_t d, double* out_dev);
int model_prior_maha_rows(vb_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, int64_t d, double* out_dev,
                          const double* prior_mean, const double* prior_ivar, double prior_c0, double* prior_out,
                          const double* E, int64_t lde, const double* rs, double df, double lq_const, double* maha, double* lq,
                          double* cn);
int model_and_prior_logp_rows(vb_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, int64_t d, double* out_dev,
                              const double* prior_mean, const double* prior_ivar, double prior_c0, double* prior_out);
// G[row] = grad f(x[row]) (row stride ld, pad columns zero), f[row] = f(x[row]) for the bound model
int model_grad_rows(vb_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, int64_t d, double* g_dev, double* f_dev);

// RCCL (vb_comm.hip)
int comm_allreduce_sum(vb_ctx* ctx, hipStream_t stream, double* buf, size_t count);
int comm_allreduce_max(vb_ctx* ctx, hipStream_t stream, double* buf, size_t count);
int comm_allgather(vb_ctx* ctx, hipStream_t stream, const double* send, double* recv, size_t count);
// Monte-Carlo-axis shard of this rank: rows [begin, begin + count) of n_total, the contiguous blocks of
// viabel_amd.objectives.shard_rows (the first n_total % n_ranks ranks hold one row more).  Fails when `n` is not
// this rank's count.
int comm_shard_begin(vb_ctx* ctx, int64_t n, int64_t n_total, int64_t* begin);
// vec[0 .. n_total) on every rank from the ranks' own blocks vec[begin .. begin + n): an in-place all-gather when the
// shards are equal, otherwise zero fill outside the own block + sum all-reduce (x + 0 is exact, so the gathered
// values are the owners' bits either way).  Without a communicator: nothing to do.
int comm_gather_rows(vb_ctx* ctx, hipStream_t stream, double* vec, int64_t begin, int64_t n, int64_t n_total);
int comm_gather_rows3(vb_ctx* ctx, hipStream_t stream, double* v0, double* v1, double* v2, int64_t begin, int64_t n,
                      int64_t n_total);

// profiling: event pair for the next launch of the dominant kernel (nullptrs when disabled)
void prof_events(vb_ctx* ctx, hipEvent_t* ev0, hipEvent_t* ev1, int evals, int kernel_id = VB_PROF_MF_ACCUM);

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

}  // namespace vb
