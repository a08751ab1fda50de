// C ABI of libviabel_hip.so: contexts, noise slots, model binding, synchronous / asynchronous
// evaluators.  See include/viabel_hip.h for the contract and the reference seam it replaces.
#include "vb_common.h"

#include <mutex>

static void legacy_spec_poll(vb_ctx* ctx);      // (look-ahead draws of numpy's streams: defined with the legacy entry points)

namespace vb {

static std::string g_last_error;
static std::mutex g_err_mutex;

int fail(vb_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) {
    ctx->last_error = buf;
  } else {
    std::lock_guard<std::mutex> lk(g_err_mutex);
    g_last_error = buf;
  }
  return code;
}

int ensure(vb_ctx* ctx, DeviceBuffer& b, size_t bytes) {
  if (b.bytes >= bytes && b.ptr) return VB_OK;
  if (b.ptr) {
    // buffers may still be referenced by work in flight on any of the context's streams
    VB_TRY(sync_streams(ctx));
    VB_HIP(ctx, hipFree(b.ptr));
    b.ptr = nullptr;
    b.bytes = 0;
    if (&b == &ctx->mvt_state) {      // caches keyed on the state buffer: a new allocation may return the old address
      ctx->mvt_prior.clear();
      ctx->mvt_inv_key[0] = 0;
      // a deferred inverse that was never enqueued (an error between the factor call and the refresh's end) describes
      // the buffer that is about to go: drop it (what WAS enqueued has finished: sync_streams above)
      ctx->mvt_inv_pending = false;
      ctx->mvt_inv_queued = false;
    }
  }
  size_t cap = bytes < 256 ? 256 : bytes;
  VB_HIP(ctx, hipMalloc(&b.ptr, cap));
  VB_HIP(ctx, hipMemsetAsync(b.ptr, 0, cap, ctx->stream));
  b.bytes = cap;
  return VB_OK;
}

int ensure_pinned(vb_ctx* ctx, size_t bytes) {
  if (ctx->pin_host && ctx->pin_bytes >= bytes) return VB_OK;
  if (ctx->pin_host) {
    VB_TRY(sync_streams(ctx));
    VB_HIP(ctx, hipHostFree(ctx->pin_host));
    ctx->pin_host = nullptr;
    ctx->pin_bytes = 0;
  }
  VB_HIP(ctx, hipHostMalloc((void**)&ctx->pin_host, bytes, hipHostMallocMapped));
  VB_HIP(ctx, hipHostGetDevicePointer((void**)&ctx->pin_dev, ctx->pin_host, 0));
  ctx->pin_bytes = bytes;
  return VB_OK;
}

// ---- small results to the host without a copy engine and without a wake-up ------------------------------------------------
// A blocking call that ends in hipMemcpyAsync(pageable destination) + hipStreamSynchronize pays a staging kernel, a second
// wait and an interrupt-driven wake-up: 40-50 us behind a 0.3 ms step (measured on the multivariate-t DIS step, C3 shape:
// 0.32 -> 0.27 ms).  fetch_blocking() gathers up to eight device segments into MAPPED host memory with one kernel
// (contiguous full-width stores: whole lines on the bus), whose last workgroup -- a ticket counter, zero between launches --
// stores a sequence number into a completion word behind a system-scope fence; the host polls the word and copies the
// segments out.  Above kFetchMaxBytes (the host-side copy out of the mapped buffer costs more than the runtime's pinned
// DMA then) and with VB_FETCH_FLAGSYNC=0 it is the plain copies + synchronisation.
constexpr size_t kFetchMaxBytes = (size_t)1 << 20;
constexpr int kFetchMaxSegs = 8;
struct FetchArgs {
  const unsigned long long* src[kFetchMaxSegs];
  long long first[kFetchMaxSegs + 1];      // first word of segment k in the mapped buffer (64-byte aligned); [n] = total
  long long words[kFetchMaxSegs];
  int n;
};

__global__ void __launch_bounds__(256) fetch_copy_kernel(FetchArgs a, unsigned long long* __restrict__ dst,
                                                         unsigned* __restrict__ ticket, unsigned long long* __restrict__ done,
                                                         unsigned long long seq) {
  for (int k = 0; k < a.n; ++k) {
    const unsigned long long* __restrict__ src = a.src[k];
    unsigned long long* __restrict__ out = dst + a.first[k];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.words[k]; i += (long long)gridDim.x * 256)
      out[i] = src[i];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// fetch_blocking in two halves, for a producer kernel that stores its results into the mapped buffer ITSELF (round 6: the
// chain-rule kernel of the multivariate t's DIS step -- a gathering launch of its own cost 10.6 us behind a 14-us producer):
// fetch_plan lays the segments out and hands the device addresses over (plan.ok == false: the segments do not qualify,
// nothing was changed -- the caller takes fetch_blocking's route); the producer's workgroups store their share of segment k
// from word plan.first[k] on, then each does what fetch_copy_kernel's do (system-scope fence, ticket, the last one stores
// plan.seq into plan.done_dev); fetch_wait polls and copies the segments out.
int fetch_plan(vb_ctx* ctx, const FetchSeg* segs, int n_segs, FetchPlan* plan) {
  plan->ok = false;
  size_t total = 0;
  bool ok = n_segs <= kFetchMaxSegs;
  for (int k = 0; k < n_segs; ++k) {
    ok = ok && segs[k].bytes % 8 == 0 && ((uintptr_t)segs[k].src & 7) == 0;
    total += (segs[k].bytes + 63) / 64 * 64;
  }
  const char* e = getenv("VB_FETCH_FLAGSYNC");
  if (!ok || total > kFetchMaxBytes || (e && atoi(e) == 0)) return VB_OK;
  const size_t need = total + 64;      // ... | completion word (a line of its own)
  if (ctx->fetch_bytes < need) {
    if (ctx->fetch_host) {
      VB_TRY(sync_streams(ctx));
      VB_HIP(ctx, hipHostFree(ctx->fetch_host));
      ctx->fetch_host = nullptr;
      ctx->fetch_bytes = 0;
    }
    const size_t cap = need < (size_t)1 << 16 ? (size_t)1 << 16 : need;
    VB_HIP(ctx, hipHostMalloc((void**)&ctx->fetch_host, cap, hipHostMallocMapped));
    memset(ctx->fetch_host, 0, cap);
    VB_HIP(ctx, hipHostGetDevicePointer((void**)&ctx->fetch_dev, ctx->fetch_host, 0));
    ctx->fetch_bytes = cap;
  }
  VB_TRY(ensure(ctx, ctx->fetch_ticket, 64));      // (zero-filled when new; the kernels leave it zero)
  long long at = 0;
  for (int k = 0; k < n_segs; ++k) {
    plan->first[k] = at;
    at += (long long)((segs[k].bytes + 63) / 64 * 8);
  }
  plan->first[n_segs] = at;
  plan->n = n_segs;
  plan->host = (unsigned long long*)ctx->fetch_host;
  plan->dev = (unsigned long long*)ctx->fetch_dev;
  plan->o_done = ctx->fetch_bytes / 8 - 8;
  plan->done_dev = plan->dev + plan->o_done;
  plan->ticket = (unsigned*)ctx->fetch_ticket.ptr;
  plan->seq = ++ctx->fetch_seq;
  plan->ok = true;
  return VB_OK;
}

int fetch_wait(vb_ctx* ctx, hipStream_t st, const FetchPlan& plan, const FetchSeg* segs) {
  if (st == ctx->stream) noise_prefetch(ctx);      // the next call's noise behind the producing kernel, while the host polls
  volatile unsigned long long* word = plan.host + plan.o_done;
  bool seen = false;
  ::legacy_spec_poll(ctx);      // (a look-ahead draw whose start was deferred to the caller's first wait: this may be it)
  for (unsigned spins = 0; spins < 2000000u && !seen; ++spins) {      // ~10 ms, then the stream
    seen = *word == plan.seq;
    if (!seen) {
      __builtin_ia32_pause();
      if ((spins & 255u) == 255u) ::legacy_spec_poll(ctx);      // (a look-ahead draw's finish may have landed: start the next one)
    }
  }
  if (!seen) VB_HIP(ctx, hipStreamSynchronize(st));
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  for (int k = 0; k < plan.n; ++k)
    if (segs[k].bytes) memcpy(segs[k].dst, plan.host + plan.first[k], segs[k].bytes);
  return comm_check(ctx);
}

int fetch_blocking(vb_ctx* ctx, hipStream_t st, const FetchSeg* segs, int n_segs) {
  FetchPlan plan;
  VB_TRY(fetch_plan(ctx, segs, n_segs, &plan));
  if (!plan.ok) {
    for (int k = 0; k < n_segs; ++k)
      if (segs[k].bytes)
        VB_HIP(ctx, hipMemcpyAsync(segs[k].dst, segs[k].src, segs[k].bytes, hipMemcpyDeviceToHost, st));
    if (st == ctx->stream) noise_prefetch(ctx);      // the next call's noise behind this call's copies, while the host waits
    ::legacy_spec_poll(ctx);      // (in front of the wait: the host enqueues a look-ahead draw instead of idling)
    VB_HIP(ctx, hipStreamSynchronize(st));
    ::legacy_spec_poll(ctx);
    return comm_check(ctx);
  }
  FetchArgs a;
  a.n = n_segs;
  for (int k = 0; k < n_segs; ++k) {
    a.src[k] = (const unsigned long long*)segs[k].src;
    a.first[k] = plan.first[k];
    a.words[k] = (long long)(segs[k].bytes / 8);
  }
  a.first[n_segs] = plan.first[n_segs];
  long long blocks = (plan.first[n_segs] + 1023) / 1024;      // ~four words per thread
  if (blocks < 1) blocks = 1;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(fetch_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, plan.dev, plan.ticket, plan.done_dev, plan.seq);
  VB_HIP(ctx, hipGetLastError());
  return fetch_wait(ctx, st, plan, segs);
}

// The other direction: `bytes` of a caller-owned (pageable) host array into device memory without a synchronisation -- the
// bytes are copied into one of two mapped staging slots (an event behind each slot's reader guards its reuse) and a
// kernel reads them across the bus.  The runtime's own pageable path stages too, but then the call has to wait for the
// stream (the caller may rewrite its array as soon as we return).  Above kFetchMaxBytes: copy + synchronisation.
// (row_words / dst_stride: the source's rows of row_words words land row_stride words apart -- a dense host matrix into a
// padded device one; row_words == 0: a flat copy)
__global__ void __launch_bounds__(256) push_copy_kernel(const unsigned long long* __restrict__ src,
                                                        unsigned long long* __restrict__ dst, long long words,
                                                        long long row_words, long long dst_stride) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < words; i += (long long)gridDim.x * 256)
    dst[row_words ? (i / row_words) * dst_stride + i % row_words : i] = src[i];
}

int push_small(vb_ctx* ctx, hipStream_t st, const void* host_src, size_t bytes, void* dev_dst, size_t row_bytes,
               size_t dst_stride_bytes) {
  const char* e = getenv("VB_FETCH_FLAGSYNC");
  if (bytes % 8 != 0 || bytes > kFetchMaxBytes || ((uintptr_t)dev_dst & 7) != 0 || row_bytes % 8 != 0 ||
      dst_stride_bytes % 8 != 0 || (e && atoi(e) == 0)) {
    if (row_bytes)
      VB_HIP(ctx, hipMemcpy2DAsync(dev_dst, dst_stride_bytes, host_src, row_bytes, row_bytes, bytes / row_bytes,
                                   hipMemcpyHostToDevice, st));
    else
      VB_HIP(ctx, hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, st));
    ::legacy_spec_poll(ctx);      // (in front of the wait, as in fetch_blocking)
    VB_HIP(ctx, hipStreamSynchronize(st));      // caller keeps ownership of `host_src`
    return VB_OK;
  }
  if (ctx->push_bytes < bytes) {
    if (ctx->push_host) {
      VB_TRY(sync_streams(ctx));
      VB_HIP(ctx, hipHostFree(ctx->push_host));
      ctx->push_host = nullptr;
      ctx->push_bytes = 0;
    }
    const size_t cap = bytes < (size_t)1 << 16 ? (size_t)1 << 16 : (bytes + 63) / 64 * 64;
    VB_HIP(ctx, hipHostMalloc((void**)&ctx->push_host, 2 * cap, hipHostMallocMapped));
    VB_HIP(ctx, hipHostGetDevicePointer((void**)&ctx->push_dev, ctx->push_host, 0));
    ctx->push_bytes = cap;
  }
  ctx->push_slot ^= 1;
  hipEvent_t& ev = ctx->push_ev[ctx->push_slot];
  if (!ev) VB_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  else VB_HIP(ctx, hipEventSynchronize(ev));
  char* slot = (char*)ctx->push_host + (size_t)ctx->push_slot * ctx->push_bytes;
  memcpy(slot, host_src, bytes);
  const long long words = (long long)(bytes / 8);
  long long blocks = (words + 1023) / 1024;
  if (blocks < 1) blocks = 1;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(push_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                     (const unsigned long long*)((char*)ctx->push_dev + (size_t)ctx->push_slot * ctx->push_bytes),
                     (unsigned long long*)dev_dst, words, (long long)(row_bytes / 8), (long long)(dst_stride_bytes / 8));
  VB_HIP(ctx, hipGetLastError());
  VB_HIP(ctx, hipEventRecord(ev, st));
  return VB_OK;
}

int sync_streams(vb_ctx* ctx) {
  VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->pipe.pre) {
    VB_HIP(ctx, hipStreamSynchronize(ctx->pipe.pre));
    VB_HIP(ctx, hipStreamSynchronize(ctx->pipe.post));
  }
  if (ctx->mvt_side) VB_HIP(ctx, hipStreamSynchronize(ctx->mvt_side));
  if (ctx->fit_copy_st) VB_HIP(ctx, hipStreamSynchronize(ctx->fit_copy_st));
  if (ctx->up_stream) {
    VB_HIP(ctx, hipStreamSynchronize(ctx->up_stream));
    VB_HIP(ctx, hipStreamSynchronize(ctx->up_side[0]));
    VB_HIP(ctx, hipStreamSynchronize(ctx->up_side[1]));
  }
  if (ctx->legacy_spec && ctx->legacy_spec->stream) VB_HIP(ctx, hipStreamSynchronize(ctx->legacy_spec->stream));
  ctx->fr_busy = false;
  ctx->pipe.post_pending = false;
  return comm_check(ctx);
}

// Main-stream work that writes buffers the pipeline may still be reading (noise, model parameters)
// is ordered after everything the pipeline has in flight, and the next prep is ordered after it.
static int main_stream_write(vb_ctx* ctx) {
  Pipeline& P = ctx->pipe;
  if (P.post_pending) {
    VB_HIP(ctx, hipStreamWaitEvent(ctx->stream, P.ev_fin[P.last_set], 0));
    P.post_pending = false;
  }
  P.main_dirty = true;
  return VB_OK;
}

// ---- look-ahead generation (see NoiseAhead, vb_common.h) -------------------------------------------------------------------
static bool req_same_but_stream(const NoiseReq& a, const NoiseReq& b) {
  return a.kind == b.kind && a.df == b.df && a.seed == b.seed && a.row_offset == b.row_offset && a.n == b.n && a.d == b.d;
}
static void req_observe(NoiseAhead& h, const NoiseReq& r) {
  if (h.last.valid && req_same_but_stream(h.last, r)) {
    const int64_t step = (int64_t)(r.stream - h.last.stream);
    if (step != 0 && step == h.delta) {
      if (h.streak < 1000) ++h.streak;
    } else {
      h.delta = step;
      h.streak = step != 0 ? 1 : 0;
    }
  } else {
    h.delta = 0;
    h.streak = 0;
  }
  h.last = r;
  h.last.valid = true;
}
static bool noise_ahead_on() {
  const char* e = getenv("VB_NOISE_AHEAD");
  return !(e && atoi(e) == 0);
}

const double* noise_row_norms(vb_ctx* ctx, NoiseSlot& s, hipStream_t st) {
  if (!s.buf.ptr || s.d > 512 || s.n <= 0) return nullptr;
  const NoiseReq &a = s.ahead.last, &b = s.norms_req;
  if (a.valid && b.valid && req_same_but_stream(a, b) && a.stream == b.stream && s.norms.ptr && a.n == s.n && a.d == s.d &&
      s.norms.bytes >= (size_t)s.n * sizeof(double))
    return (const double*)s.norms.ptr;
  s.want_norms = true;
  s.norms_req.valid = false;
  if (ensure(ctx, s.norms, (size_t)s.n * sizeof(double)) != VB_OK) return nullptr;
  if (rng_row_norms(ctx, st, (const double*)s.buf.ptr, s.ld, s.n, s.d, (double*)s.norms.ptr) != VB_OK) return nullptr;
  if (a.valid && a.n == s.n && a.d == s.d) s.norms_req = a;      // (Philox contents the slot keeps track of: good until they change)
  return (const double*)s.norms.ptr;
}

void noise_prefetch(vb_ctx* ctx) {
  if (!ctx || !noise_ahead_on()) return;
  bool ordered = false;      // main_stream_write once, and only when something is generated
  for (int slot = 0; slot < VB_MAX_SLOTS; ++slot) {
    NoiseSlot& s = ctx->noise[slot];
    NoiseAhead& h = s.ahead;
    const bool hinted = h.hint.valid;
    h.hint.valid = false;      // (a hint speaks for the call it was given in)
    if ((h.streak < 2 && !hinted) || !h.last.valid || h.pre.valid || !s.buf.ptr || s.n != h.last.n || s.d != h.last.d) continue;
    const size_t bytes = (size_t)s.n * s.ld * sizeof(double);
    const bool fresh = !h.shadow.ptr || h.shadow.bytes < bytes;
    if (ensure(ctx, h.shadow, bytes) != VB_OK) continue;      // (zero-filled when new)
    if (!ordered && main_stream_write(ctx) != VB_OK) return;
    ordered = true;
    if (!fresh && (h.shadow_d != s.d || h.shadow_ld != s.ld) &&
        hipMemsetAsync(h.shadow.ptr, 0, h.shadow.bytes, ctx->stream) != hipSuccess)      // pad columns [d, ld) must be zero
      continue;
    h.shadow_d = s.d, h.shadow_ld = s.ld;
    NoiseReq nx = h.last;
    if (hinted) nx.seed = h.hint.seed, nx.stream = h.hint.stream;
    else nx.stream = h.last.stream + (uint64_t)h.delta;
    // (row norms ride along once a reader has asked for them: NoiseSlot::want_norms)
    const bool with_norms = s.want_norms && nx.kind == VB_NOISE_NORMAL && nx.d <= 512 &&
                            ensure(ctx, h.shadow_norms, (size_t)nx.n * sizeof(double)) == VB_OK;
    if (rng_fill(ctx, (double*)h.shadow.ptr, s.ld, nx.kind, nx.df, nx.seed, nx.stream, nx.row_offset, nx.n, nx.d,
                 with_norms ? (double*)h.shadow_norms.ptr : (double*)nullptr) != VB_OK)
      continue;
    h.pre_norms = with_norms;
    h.pre = nx;
    ++ctx->ahead_generated;
  }
  NoiseAhead& c = ctx->chi_ahead;
  const bool chi_hinted = c.hint.valid;
  c.hint.valid = false;
  if ((c.streak >= 2 || chi_hinted) && c.last.valid && !c.pre.valid && ctx->chi_n == c.last.n && ctx->chi_dev.ptr) {
    if (ensure(ctx, c.shadow, (size_t)c.last.n * sizeof(double)) != VB_OK) return;
    if (!ordered && main_stream_write(ctx) != VB_OK) return;
    NoiseReq nx = c.last;
    if (chi_hinted) nx.seed = c.hint.seed, nx.stream = c.hint.stream;
    else nx.stream = c.last.stream + (uint64_t)c.delta;
    if (rng_chisquare(ctx, (double*)c.shadow.ptr, nx.df, nx.seed, nx.stream, nx.row_offset, nx.n) == VB_OK) {
      c.pre = nx;
      ++ctx->ahead_generated;
    }
  }
}


// A blocking call whose results reach mapped host memory from its last kernel: the event marks that kernel, the next
// call's noise (NoiseAhead) is enqueued behind it, and the host waits for the event only.
int wait_then_prefetch(vb_ctx* ctx) {
  if (!ctx->done_ev) VB_HIP(ctx, hipEventCreateWithFlags(&ctx->done_ev, hipEventDisableTiming));
  VB_HIP(ctx, hipEventRecord(ctx->done_ev, ctx->stream));
  noise_prefetch(ctx);
  VB_HIP(ctx, hipEventSynchronize(ctx->done_ev));
  return VB_OK;
}

void prof_events(vb_ctx* ctx, hipEvent_t* ev0, hipEvent_t* ev1, int evals, int kernel_id) {
  *ev0 = *ev1 = nullptr;
  if (!ctx->profile || kernel_id < 0 || kernel_id >= VB_PROF_NUM) return;
  vb_ctx::ProfLog& log = ctx->prof[kernel_id];
  if (log.used == log.events.size()) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    log.events.push_back({a, b});
  }
  *ev0 = log.events[log.used].first;
  *ev1 = log.events[log.used].second;
  log.used++;
  log.evals += evals;
}

static int check_slot(vb_ctx* ctx, int slot) {
  if (slot < 0 || slot >= VB_MAX_SLOTS)
    return fail(ctx, VB_ERR_INVALID, "slot %d out of range [0, %d)", slot, VB_MAX_SLOTS);
  return VB_OK;
}

static int noise_alloc(vb_ctx* ctx, int slot, int64_t n, int64_t d, bool keep_history = false) {
  VB_TRY(check_slot(ctx, slot));
  if (n <= 0 || d <= 0) return fail(ctx, VB_ERR_INVALID, "noise shape must be positive");
  NoiseSlot& s = ctx->noise[slot];
  if (!keep_history) {      // a writer other than vb_noise_generate: the look-ahead's history ends here
    s.ahead.last.valid = s.ahead.pre.valid = s.ahead.hint.valid = false;
    s.ahead.streak = 0;
  }
  // The t family's DIS state may be reading its residuals straight out of this slot (mvt_e_noise, vb_mvt.hip): new
  // contents or a new buffer end that -- a gradient that comes without a refresh in between forms its residuals from
  // the state samples again
  if (ctx->mvt_e_noise && ctx->mvt_e_noise == (const double*)s.buf.ptr) {
    ctx->mvt_e_noise = nullptr;
    ctx->mvt_theta.clear();
  }
  // 128-B aligned rows; a row stride that is a multiple of 4 KiB would put one column of every row on the
  // same HBM channel (the funnel's coupling column is read down the rows), so such strides get one more
  // 128-B pad
  int64_t ld = round_up(d, 16);
  if (ld % 512 == 0) ld += 16;
  const bool had = s.buf.ptr != nullptr && s.buf.bytes >= (size_t)n * ld * sizeof(double);
  VB_TRY(ensure(ctx, s.buf, (size_t)n * ld * sizeof(double)));
  // invariant relied on by the streaming kernel: the pad columns [d, ld) hold zeros
  if (had && (s.d != d || s.ld != ld))
    VB_HIP(ctx, hipMemsetAsync(s.buf.ptr, 0, s.buf.bytes, ctx->stream));
  s.n = n;
  s.d = d;
  s.ld = ld;
  return VB_OK;
}

// Block until the enqueue with completion ticket `id` has finished.
static int wait_ticket(vb_ctx* ctx, uint64_t id) {
  if (id == 0) return VB_OK;
  if (ctx->batch_id - id >= ctx->batch_events.size()) return sync_streams(ctx);   // event slot recycled
  VB_HIP(ctx, hipEventSynchronize(ctx->batch_events[id % ctx->batch_events.size()]));
  return VB_OK;
}

// Stage theta in the result slot's pinned, device-mapped buffer: the prep kernel reads it from
// there and the finalize / epilogue kernel writes [value | grad] back into the same buffer, so an
// evaluation needs no separate copy commands on the stream.
static int stage_theta(vb_ctx* ctx, ResultSlot& rs, const double* theta, int64_t p) {
  const size_t need = (size_t)(1 + 2 * p) * sizeof(double);   // [theta staging | value | grad]
  if (rs.p < p || !rs.host) {
    if (rs.host) {
      VB_TRY(sync_streams(ctx));
      VB_HIP(ctx, hipHostFree(rs.host));
      rs.host = nullptr;
    }
    VB_HIP(ctx, hipHostMalloc((void**)&rs.host, need, hipHostMallocMapped));
    VB_HIP(ctx, hipHostGetDevicePointer((void**)&rs.dev, rs.host, 0));
    rs.p = p;
  }
  VB_TRY(wait_ticket(ctx, rs.batch_id));   // the evaluation that last used this slot may still be running
  memcpy(rs.host, theta, (size_t)p * sizeof(double));
  rs.pending = true;
  return VB_OK;
}

// record the completion ticket of the enqueue that just used `rs[0..count)`
static int ticket(vb_ctx* ctx, ResultSlot** rs, int count) {
  if (ctx->batch_events.empty()) {
    ctx->batch_events.resize(32);
    for (auto& e : ctx->batch_events) VB_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const uint64_t id = ++ctx->batch_id;
  VB_HIP(ctx, hipEventRecord(ctx->batch_events[id % ctx->batch_events.size()],
                             ctx->result_stream ? ctx->result_stream : ctx->stream));
  for (int b = 0; b < count; ++b) rs[b]->batch_id = id;
  return VB_OK;
}

}  // namespace vb

using namespace vb;

extern "C" {

const char* vb_version(void) { return "viabel_hip 0.1.0 (gfx950)"; }

int vb_device_count(int* count) {
  if (!count) return fail(nullptr, VB_ERR_INVALID, "count is NULL");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    return fail(nullptr, VB_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  }
  return VB_OK;
}

int vb_create(int device_id, vb_ctx** out) {
  if (!out) return fail(nullptr, VB_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return fail(nullptr, VB_ERR_HIP, "no HIP device available (%s)",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= count)
    return fail(nullptr, VB_ERR_INVALID, "device %d out of range [0, %d)", device_id, count);
  vb_ctx* ctx = new vb_ctx();
  ctx->device = device_id;
  e = hipSetDevice(device_id);
  if (e == hipSuccess) e = hipGetDeviceProperties(&ctx->prop, device_id);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    fail(nullptr, VB_ERR_HIP, "context creation failed: %s", hipGetErrorString(e));
    delete ctx;
    return VB_ERR_HIP;
  }
  *out = ctx;
  return VB_OK;
}

int vb_destroy(vb_ctx* ctx) {
  if (!ctx) return VB_OK;
  (void)hipSetDevice(ctx->device);
  (void)sync_streams(ctx);
  user_model_release(ctx);
  vb_comm_destroy(ctx);
  if (ctx->pipe.pre) {
    (void)hipStreamDestroy(ctx->pipe.pre);
    (void)hipStreamDestroy(ctx->pipe.post);
    (void)hipEventDestroy(ctx->pipe.ev_main);
    for (int i = 0; i < kPipeSets; ++i) {
      (void)hipEventDestroy(ctx->pipe.ev_prep[i]);
      (void)hipEventDestroy(ctx->pipe.ev_k1[i]);
      (void)hipEventDestroy(ctx->pipe.ev_fin[i]);
    }
  }
  for (auto& s : ctx->noise) {
    if (s.buf.ptr) (void)hipFree(s.buf.ptr);
    if (s.ahead.shadow.ptr) (void)hipFree(s.ahead.shadow.ptr);
    if (s.norms.ptr) (void)hipFree(s.norms.ptr);
    if (s.ahead.shadow_norms.ptr) (void)hipFree(s.ahead.shadow_norms.ptr);
  }
  if (ctx->chi_ahead.shadow.ptr) (void)hipFree(ctx->chi_ahead.shadow.ptr);
  for (auto& r : ctx->results)
    if (r.host) (void)hipHostFree(r.host);
  if (ctx->sync_result.host) (void)hipHostFree(ctx->sync_result.host);
  if (ctx->done_host) (void)hipHostFree(ctx->done_host);
  if (ctx->pin_host) (void)hipHostFree(ctx->pin_host);
  for (DeviceBuffer* b : {&ctx->model_params, &ctx->theta, &ctx->workspace, &ctx->sums, &ctx->out,
                          &ctx->scratch, &ctx->scratch2, &ctx->rowvec, &ctx->fr_work, &ctx->fr_theta,
                          &ctx->fr_out, &ctx->dis_state, &ctx->mvt_state, &ctx->lg_work, &ctx->user_params, &ctx->psis_lw, &ctx->psis_work, &ctx->rows_work,
                          &ctx->lr_work, &ctx->mvt_elbo, &ctx->fit_work, &ctx->glm_work, &ctx->fr_lt, &ctx->bisect_work, &ctx->chi_dev, &ctx->lr_obj, &ctx->gen_geom.buf, &ctx->tri_map, &ctx->mvt_invs, &ctx->temper.buf, &ctx->temper.work, &ctx->fz_words, &ctx->fz_items, &ctx->legacy_work, &ctx->alpha_g, &ctx->mf_one, &ctx->fetch_ticket})
    if (b->ptr) (void)hipFree(b->ptr);
  if (ctx->mvt_pin) (void)hipHostFree(ctx->mvt_pin);
  if (ctx->fetch_host) (void)hipHostFree(ctx->fetch_host);
  if (ctx->push_host) (void)hipHostFree(ctx->push_host);
  for (hipEvent_t e : ctx->push_ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->user_host_pin) (void)hipHostFree(ctx->user_host_pin);
  if (ctx->legacy_pin) (void)hipHostFree(ctx->legacy_pin);
  for (hipEvent_t e : ctx->mvt_pin_ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->mvt_side) {
    (void)hipStreamSynchronize(ctx->mvt_side);
    (void)hipStreamDestroy(ctx->mvt_side);
    (void)hipEventDestroy(ctx->mvt_ev_join);
  }
  if (ctx->done_ev) (void)hipEventDestroy(ctx->done_ev);
  if (ctx->legacy_spec) {
    LegacySpec* S = ctx->legacy_spec;
    if (S->stream) {
      (void)hipStreamSynchronize(S->stream);
      (void)hipStreamDestroy(S->stream);
    }
    if (S->ev_main) (void)hipEventDestroy(S->ev_main);
    if (S->land_host) (void)hipHostFree(S->land_host);
    for (DeviceBuffer& b : S->shadow)
      if (b.ptr) (void)hipFree(b.ptr);
    if (S->work.ptr) (void)hipFree(S->work.ptr);
    if (S->clone) vb_legacy_rng_destroy(S->clone);
    delete S;
    ctx->legacy_spec = nullptr;
  }
  if (ctx->up_stream) {
    (void)hipStreamSynchronize(ctx->up_stream);
    (void)hipStreamDestroy(ctx->up_stream);
    for (int i = 0; i < 2; ++i) {
      (void)hipStreamSynchronize(ctx->up_side[i]);
      (void)hipStreamDestroy(ctx->up_side[i]);
      (void)hipEventDestroy(ctx->up_ev_join[i]);
    }
    (void)hipEventDestroy(ctx->up_ev_main);
    for (hipEvent_t e : ctx->fr_up.ev)
      if (e) (void)hipEventDestroy(e);
  }
  if (ctx->fit_copy_st) {
    (void)hipStreamSynchronize(ctx->fit_copy_st);
    (void)hipStreamDestroy(ctx->fit_copy_st);
  }
  if (ctx->fit_ring) (void)hipHostFree(ctx->fit_ring);
  for (int i = 0; i < vb_ctx::kFitRing; ++i) {
    if (ctx->fit_ev_step[i]) (void)hipEventDestroy(ctx->fit_ev_step[i]);
    if (ctx->fit_ev_copy[i]) (void)hipEventDestroy(ctx->fit_ev_copy[i]);
  }
  for (auto& e : ctx->batch_events) (void)hipEventDestroy(e);
  for (auto& log : ctx->prof)
    for (auto& ev : log.events) {
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return VB_OK;
}

const char* vb_last_error(vb_ctx* ctx) {
  if (ctx) return ctx->last_error.c_str();
  std::lock_guard<std::mutex> lk(g_err_mutex);
  return g_last_error.c_str();
}

int vb_device_info(vb_ctx* ctx, char* name, size_t name_len, int* n_cu, uint64_t* hbm_bytes) {
  if (!ctx) return VB_ERR_INVALID;
  if (name && name_len) snprintf(name, name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
  if (n_cu) *n_cu = ctx->prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (uint64_t)ctx->prop.totalGlobalMem;
  return VB_OK;
}

int vb_sync(vb_ctx* ctx) {
  if (!ctx) return VB_ERR_INVALID;
  VB_HIP(ctx, hipSetDevice(ctx->device));
  return sync_streams(ctx);
}

// ---- noise ---------------------------------------------------------------------------------
int vb_noise_set_host(vb_ctx* ctx, int slot, const double* host, int64_t n, int64_t d) {
  if (!ctx || !host) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  VB_TRY(noise_alloc(ctx, slot, n, d));
  NoiseSlot& s = ctx->noise[slot];
  // (caller keeps ownership of `host`: small matrices are staged and scattered by a kernel, no wait; large ones are
  // copied and waited for)
  return push_small(ctx, ctx->stream, host, (size_t)(n * d) * sizeof(double), s.buf.ptr, (size_t)d * sizeof(double),
                    (size_t)s.ld * sizeof(double));
}

int vb_noise_generate(vb_ctx* ctx, int slot, int kind, double df, uint64_t seed, uint64_t stream,
                      int64_t row_offset, int64_t n, int64_t d) {
  if (!ctx) return VB_ERR_INVALID;
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  VB_TRY(check_slot(ctx, slot));
  NoiseReq r;
  r.kind = kind, r.df = df, r.seed = seed, r.stream = stream, r.row_offset = row_offset, r.n = n, r.d = d, r.valid = true;
  {
    NoiseSlot& s = ctx->noise[slot];
    NoiseAhead& h = s.ahead;
    if (h.pre.valid && h.shadow.ptr && s.buf.ptr && s.n == n && s.d == d && req_same_but_stream(h.pre, r) &&
        h.pre.stream == stream && h.shadow.bytes >= (size_t)n * s.ld * sizeof(double) && noise_ahead_on()) {
      // the look-ahead generated exactly this request: adopt its buffer (same geometry: n, d, ld stay)
      if (ctx->mvt_e_noise && ctx->mvt_e_noise == (const double*)s.buf.ptr) {      // (as noise_alloc: new contents)
        ctx->mvt_e_noise = nullptr;
        ctx->mvt_theta.clear();
      }
      std::swap(s.buf, h.shadow);
      s.norms_req.valid = false;
      if (h.pre_norms) {
        std::swap(s.norms, h.shadow_norms);
        s.norms_req = r;
      }
      h.pre_norms = false;
      h.pre.valid = false;
      ++ctx->ahead_adopted;
      req_observe(h, r);
      return VB_OK;
    }
  }
  VB_TRY(noise_alloc(ctx, slot, n, d, true));
  NoiseSlot& s = ctx->noise[slot];
  s.ahead.pre.valid = false;      // (a shadow that was not asked for is dropped)
  s.norms_req.valid = false;
  const bool with_norms = s.want_norms && kind == VB_NOISE_NORMAL && d <= 512;
  if (with_norms) VB_TRY(ensure(ctx, s.norms, (size_t)n * sizeof(double)));
  VB_TRY(rng_fill(ctx, (double*)s.buf.ptr, s.ld, kind, df, seed, stream, row_offset, n, d,
                  with_norms ? (double*)s.norms.ptr : (double*)nullptr));
  if (with_norms) s.norms_req = r;
  req_observe(s.ahead, r);
  return VB_OK;
}

int vb_noise_ahead_stats(vb_ctx* ctx, uint64_t* generated, uint64_t* adopted) {
  if (!ctx || !generated || !adopted) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  *generated = ctx->ahead_generated;
  *adopted = ctx->ahead_adopted;
  return VB_OK;
}

int vb_noise_hint_seed(vb_ctx* ctx, uint64_t slot_mask, int with_chi, uint64_t seed) {
  if (!ctx) return VB_ERR_INVALID;
  static_assert(VB_MAX_SLOTS <= 64, "slot_mask is one bit per noise slot");
  for (int slot = 0; slot < VB_MAX_SLOTS; ++slot) {
    NoiseAhead& h = ctx->noise[slot].ahead;
    if (!((slot_mask >> slot) & 1ull) || !h.last.valid) continue;
    h.hint = h.last;
    h.hint.seed = seed;
  }
  NoiseAhead& c = ctx->chi_ahead;
  if (with_chi && c.last.valid) {
    c.hint = c.last;
    c.hint.seed = seed;
  }
  return VB_OK;
}

int vb_chisq_generate(vb_ctx* ctx, double df, uint64_t seed, uint64_t stream, int64_t row_offset, int64_t n) {
  if (!ctx) return VB_ERR_INVALID;
  if (n <= 0) return fail(ctx, VB_ERR_INVALID, "n must be positive");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  NoiseReq r;
  r.kind = -1, r.df = df, r.seed = seed, r.stream = stream, r.row_offset = row_offset, r.n = n, r.d = 1, r.valid = true;
  NoiseAhead& c = ctx->chi_ahead;
  if (c.pre.valid && c.shadow.ptr && ctx->chi_dev.ptr && req_same_but_stream(c.pre, r) && c.pre.stream == stream &&
      c.shadow.bytes >= (size_t)n * sizeof(double) && ctx->chi_dev.bytes >= (size_t)n * sizeof(double) && noise_ahead_on()) {
    std::swap(ctx->chi_dev, c.shadow);
    c.pre.valid = false;
    ++ctx->ahead_adopted;
    ctx->chi_n = n;
    ctx->chi_df = df;
    req_observe(c, r);
    return VB_OK;
  }
  c.pre.valid = false;
  VB_TRY(ensure(ctx, ctx->chi_dev, (size_t)n * sizeof(double)));
  ctx->chi_n = 0;
  VB_TRY(rng_chisquare(ctx, (double*)ctx->chi_dev.ptr, df, seed, stream, row_offset, n));
  ctx->chi_n = n;
  ctx->chi_df = df;
  req_observe(c, r);
  return VB_OK;
}

int vb_chisq_get_host(vb_ctx* ctx, double* host, int64_t n) {
  if (!ctx || !host) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (n <= 0 || n != ctx->chi_n) return fail(ctx, VB_ERR_STATE, "no %lld device chi-square draws", (long long)n);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_HIP(ctx, hipMemcpyAsync(host, ctx->chi_dev.ptr, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VB_OK;
}

int vb_noise_get_host(vb_ctx* ctx, int slot, double* host, int64_t n, int64_t d) {
  if (!ctx || !host) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(check_slot(ctx, slot));
  NoiseSlot& s = ctx->noise[slot];
  if (!s.buf.ptr || n > s.n || d != s.d)
    return fail(ctx, VB_ERR_STATE, "noise slot %d does not hold a %lld x %lld matrix", slot,
                (long long)n, (long long)d);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_HIP(ctx, hipMemcpy2DAsync(host, (size_t)d * sizeof(double), s.buf.ptr,
                               (size_t)s.ld * sizeof(double), (size_t)d * sizeof(double), (size_t)n,
                               hipMemcpyDeviceToHost, ctx->stream));
  VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VB_OK;
}

// ---- model ---------------------------------------------------------------------------------
int vb_set_model(vb_ctx* ctx, int model_id, int64_t dim, const double* dparams, size_t n_dparams,
                 const int64_t* iparams, size_t n_iparams) {
  if (!ctx) return VB_ERR_INVALID;
  if (dim <= 0) return fail(ctx, VB_ERR_INVALID, "model dimension must be positive");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  const double kLog2Pi = 1.8378770664093454835606594728112;
  ModelDev m;
  m.id = model_id;
  m.dim = (int)dim;
  std::vector<double> dev;   // what is uploaded
  if (model_id == VB_MODEL_GAUSS_DIAG) {
    if (n_dparams != (size_t)(2 * dim) || !dparams)
      return fail(ctx, VB_ERR_INVALID, "gauss_diag expects dparams = [mean(D) | stdev(D)]");
    dev.resize(2 * dim);
    double c0 = -0.5 * dim * kLog2Pi;
    for (int64_t i = 0; i < dim; ++i) {
      const double sd = dparams[dim + i];
      if (!(sd > 0.0)) return fail(ctx, VB_ERR_INVALID, "gauss_diag stdev must be positive");
      dev[i] = dparams[i];
      dev[dim + i] = 1.0 / (sd * sd);
      c0 -= log(sd);
    }
    m.c0 = c0;
  } else if (model_id == VB_MODEL_FUNNEL) {
    if (n_dparams != 1 || n_iparams != 1 || !dparams || !iparams)
      return fail(ctx, VB_ERR_INVALID, "funnel expects dparams = [log_sigma_stdev], iparams = [scale_index]");
    if (iparams[0] < 0 || iparams[0] >= dim || !(dparams[0] > 0.0))
      return fail(ctx, VB_ERR_INVALID, "funnel scale_index / log_sigma_stdev out of range");
    m.k = (int)iparams[0];
    m.tau = dparams[0];
    m.c0 = -log(m.tau) - 0.5 * kLog2Pi - 0.5 * (double)(dim - 1) * kLog2Pi;
  } else if (model_id == VB_MODEL_GAUSS_FULL) {
    if (n_dparams != (size_t)(dim + dim * dim + 1) || !dparams)
      return fail(ctx, VB_ERR_INVALID, "gauss_full expects dparams = [mean(D) | P(DxD) | logdet P]");
    m.c0 = 0.5 * dparams[dim + dim * dim] - 0.5 * dim * kLog2Pi;
    m.ldp = round_up(dim, 16);
    // [mean (ldp) | P (dim x ldp)], rows padded so every GEMM operand row is 16-B aligned
    dev.assign((size_t)m.ldp * (dim + 1), 0.0);
    for (int64_t i = 0; i < dim; ++i) dev[i] = dparams[i];
    for (int64_t i = 0; i < dim; ++i)
      for (int64_t j = 0; j < dim; ++j) dev[(size_t)m.ldp * (i + 1) + j] = dparams[dim + i * dim + j];
  } else if (model_id == VB_MODEL_LOGISTIC) {
    const int link = (n_iparams == 2 && iparams) ? (int)iparams[1] : VB_GLM_BERNOULLI_LOGIT;
    const size_t extra = link == VB_GLM_GAUSSIAN ? 2 : 1;
    if ((n_iparams != 1 && n_iparams != 2) || !iparams || iparams[0] <= 0 || !dparams ||
        link < VB_GLM_BERNOULLI_LOGIT || link > VB_GLM_GAUSSIAN ||
        n_dparams != (size_t)(iparams[0] * dim + iparams[0]) + extra)
      return fail(ctx, VB_ERR_INVALID,
                  "regression target expects dparams = [X(n_data x D) | y(n_data) | prior_sd (| noise_sd)], "
                  "iparams = [n_data (, link)]");
    const int64_t nd = iparams[0];
    const double sd = dparams[nd * dim + nd];
    if (!(sd > 0.0)) return fail(ctx, VB_ERR_INVALID, "logistic prior_sd must be positive");
    m.n_data = nd;
    m.ldp = round_up(dim, 16);
    m.ldq = round_up(nd, 16);
    m.tau = sd;
    m.link = link;
    m.c0 = -(double)dim * (log(sd) + 0.5 * kLog2Pi);
    if (link == VB_GLM_POISSON) {           // - sum_i log(y_i !)
      for (int64_t i = 0; i < nd; ++i) {
        const double yi = dparams[nd * dim + i];
        if (!(yi >= 0.0)) return fail(ctx, VB_ERR_INVALID, "Poisson counts must be non-negative");
        m.c0 -= lgamma(yi + 1.0);
      }
    } else if (link == VB_GLM_GAUSSIAN) {
      m.aux = dparams[nd * dim + nd + 1];
      if (!(m.aux > 0.0)) return fail(ctx, VB_ERR_INVALID, "noise_sd must be positive");
      m.c0 -= (double)nd * (log(m.aux) + 0.5 * kLog2Pi);
    }
    // [X (nd x ldp) | X' (dim x ldq) | y (ldq)], rows padded for the GEMM operand loads
    dev.assign((size_t)nd * m.ldp + (size_t)dim * m.ldq + (size_t)m.ldq, 0.0);
    for (int64_t i = 0; i < nd; ++i)
      for (int64_t j = 0; j < dim; ++j) {
        const double v = dparams[i * dim + j];
        dev[(size_t)i * m.ldp + j] = v;
        dev[(size_t)nd * m.ldp + (size_t)j * m.ldq + i] = v;
      }
    for (int64_t i = 0; i < nd; ++i) dev[(size_t)nd * m.ldp + (size_t)dim * m.ldq + i] = dparams[nd * dim + i];
  } else {
    return fail(ctx, VB_ERR_INVALID, "unknown model id %d", model_id);
  }
  if (!dev.empty()) {
    VB_TRY(ensure(ctx, ctx->model_params, dev.size() * sizeof(double)));
    VB_HIP(ctx, hipMemcpyAsync(ctx->model_params.ptr, dev.data(), dev.size() * sizeof(double),
                               hipMemcpyHostToDevice, ctx->stream));
    VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    m.p0 = (const double*)ctx->model_params.ptr;
    m.p1 = m.p0 + (model_id == VB_MODEL_GAUSS_FULL ? m.ldp : dim);
    if (model_id == VB_MODEL_LOGISTIC) {
      m.p1 = m.p0 + (size_t)m.n_data * m.ldp;
      m.p2 = m.p1 + (size_t)dim * m.ldq;
    }
  }
  ctx->model = m;
  return VB_OK;
}

int vb_set_model_source(vb_ctx* ctx, int64_t dim, const char* source, const double* params, size_t n_params) {
  if (!ctx) return VB_ERR_INVALID;
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  return user_model_set(ctx, dim, source, params, n_params);
}

int vb_dis_set_temper_prior(vb_ctx* ctx, int kind, int64_t d, double df, const double* loc, const double* scale,
                            double log_det_l) {
  if (!ctx) return VB_ERR_INVALID;
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  return temper_prior_set(ctx, kind, d, df, loc, scale, log_det_l);
}

// ---- look-ahead generation of numpy's legacy streams (LegacySpec, vb_common.h) --------------------------------------------
static int legacy_gamma_device(vb_ctx* ctx, vb_legacy_rng* rng, int prog, double df, double* dst, int64_t ld, int64_t n_total,
                               int64_t d, int64_t row_begin, int64_t rows, LegacyFinish* defer);

static bool legacy_spec_on() {
  const char* e = getenv("VB_LEGACY_AHEAD");      // (read per call: the tests switch it)
  return !(e && atoi(e) == 0);
}

static void gen_state_get(const vb_legacy_rng* g, LegacyGenState* s) {
  (void)vb_legacy_rng_get_state(g, s->key, &s->pos, &s->has_gauss, &s->gauss);
}
static bool gen_state_same(const LegacyGenState& a, const LegacyGenState& b) {
  return a.pos == b.pos && a.has_gauss == b.has_gauss && memcmp(a.key, b.key, sizeof a.key) == 0 &&
         (!a.has_gauss || memcmp(&a.gauss, &b.gauss, sizeof(double)) == 0);
}

// main <-> speculation: the draw functions take their stream and their scratch from the context
struct LegacySpecGuard {
  vb_ctx* ctx;
  explicit LegacySpecGuard(vb_ctx* c) : ctx(c) { swap(); }
  ~LegacySpecGuard() { swap(); }
  void swap() {
    LegacySpec& S = *ctx->legacy_spec;
    std::swap(ctx->stream, S.stream);
    std::swap(ctx->legacy_work, S.work);
    std::swap(ctx->legacy_poly_at, S.poly_at);
    std::swap(ctx->legacy_poly_bytes, S.poly_bytes);
  }
};

static int64_t noise_row_stride(int64_t d) {
  int64_t ld = round_up(d, 16);
  if (ld % 512 == 0) ld += 16;
  return ld;
}

static int legacy_spec_create(vb_ctx* ctx) {
  if (ctx->legacy_spec) return VB_OK;
  LegacySpec* S = new (std::nothrow) LegacySpec;
  if (!S) return fail(ctx, VB_ERR_HIP, "out of host memory");
  ctx->legacy_spec = S;
  VB_HIP(ctx, hipStreamCreateWithFlags(&S->stream, hipStreamNonBlocking));
  VB_HIP(ctx, hipEventCreateWithFlags(&S->ev_main, hipEventDisableTiming));
  VB_HIP(ctx, hipHostMalloc((void**)&S->land_host, kLegacyFinishWords * sizeof(unsigned long long), hipHostMallocMapped));
  memset(S->land_host, 0, kLegacyFinishWords * sizeof(unsigned long long));
  VB_HIP(ctx, hipHostGetDevicePointer((void**)&S->land_dev, S->land_host, 0));
  VB_TRY(vb_legacy_rng_create(0, &S->clone));
  return VB_OK;
}

static void legacy_spec_cancel(vb_ctx* ctx) {
  LegacySpec* S = ctx->legacy_spec;
  if (!S || !S->active) return;
  (void)hipStreamSynchronize(S->stream);      // its kernels write the shadows and its scratch: quiescent before anything reuses them
  S->active = false;
  S->in_flight = -1;
  ++S->discarded;
  if (++S->discard_streak >= 2) S->cooldown = 64;
}

// enqueue the next unfinished request of the job (nothing in flight): draw kernels + the deferred finish on the speculation's stream
static void legacy_spec_enqueue_next(vb_ctx* ctx) {
  LegacySpec& S = *ctx->legacy_spec;
  if (!S.active || S.failed || S.in_flight >= 0 || S.n_finished >= S.n_reqs) return;
  const int i = S.n_finished;
  const LegacyReq& r = S.reqs[i];
  const LegacyGenState& g = S.state[i];
  if (vb_legacy_rng_set_state(S.clone, g.key, g.pos, g.has_gauss, g.gauss) != VB_OK) {
    S.failed = true;
    return;
  }
  int rc = VB_OK;
  {
    LegacySpecGuard guard(ctx);
    const int64_t ld = r.prog == 0 ? 1 : noise_row_stride(r.d);
    rc = ensure(ctx, S.shadow[i], (size_t)r.rows * ld * sizeof(double));
    // (the noise slots' invariant -- pad columns [d, ld) hold zeros -- for a buffer that last held another layout)
    if (rc == VB_OK && r.prog != 0 && (S.shadow_d[i] != r.d || S.shadow_ld[i] != ld)) {
      if (hipMemsetAsync(S.shadow[i].ptr, 0, S.shadow[i].bytes, ctx->stream) != hipSuccess) rc = VB_ERR_HIP;
      S.shadow_d[i] = r.d, S.shadow_ld[i] = ld;
    }
    if (rc == VB_OK && r.prog == -1) {
      NoiseSlot view;
      view.buf = S.shadow[i], view.n = r.rows, view.d = r.d, view.ld = ld;
      LegacyGenState h = g;
      rc = legacy_dev_randn(ctx, h.key, &h.pos, &h.has_gauss, &h.gauss, view, r.n_total, r.d, r.row_begin, r.rows, &S.fin);
      S.head_state = g;      // (no host-drawn head: the finish starts from the request's own start state)
    } else if (rc == VB_OK) {
      rc = legacy_gamma_device(ctx, S.clone, r.prog, r.df, (double*)S.shadow[i].ptr, ld, r.n_total, r.d, r.row_begin, r.rows, &S.fin);
      if (rc == VB_OK) gen_state_get(S.clone, &S.head_state);
    }
    if (rc == VB_OK) rc = legacy_finish_launch(ctx, ctx->stream, S.fin, S.land_dev, ++S.seq);
  }
  if (rc != VB_OK) {
    S.failed = true;      // (declined or failed: nothing of this request is ever adopted; the caller draws as before)
    return;
  }
  S.in_flight = i;
}

// has the request in flight landed?  Then complete it on the host and start the next one.  Cheap when nothing is pending;
// called from the draw entry points and from fetch_blocking's wait.
static void legacy_spec_poll(vb_ctx* ctx) {
  LegacySpec* Sp = ctx->legacy_spec;
  if (Sp && Sp->active && !Sp->failed && Sp->in_flight < 0 && Sp->n_finished < Sp->n_reqs) {
    legacy_spec_enqueue_next(ctx);      // (a job whose start was deferred to the caller's first wait: vb_legacy_round_end)
    return;
  }
  if (!Sp || !Sp->active || Sp->in_flight < 0) return;
  LegacySpec& S = *Sp;
  if (*(volatile unsigned long long*)(S.land_host + 344) != S.seq) return;
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  LegacyGenState e = S.head_state;
  const int rc = legacy_finish_complete(ctx, S.fin, S.land_host, e.key, &e.pos, &e.has_gauss, &e.gauss);
  const int i = S.in_flight;
  S.in_flight = -1;
  if (rc != VB_OK) {
    S.failed = true;
    return;
  }
  S.state[i + 1] = e;
  S.n_finished = i + 1;
  legacy_spec_enqueue_next(ctx);
}

// The draw `r` of generator `rng` is being asked for: is it the job's next request, from the state the job assumed?  Then wait
// for it (it has usually landed), and report the shadow to adopt and the generator's end state.  false: not ours (the job, if
// any, is discarded) -- the caller draws.
static bool legacy_spec_take(vb_ctx* ctx, vb_legacy_rng* rng, const LegacyReq& r, int* index) {
  LegacySpec* Sp = ctx->legacy_spec;
  if (!Sp || !Sp->active) return false;
  LegacySpec& S = *Sp;
  const int i = S.n_adopted;
  LegacyGenState now;
  gen_state_get(rng, &now);
  if (!legacy_spec_on() || i >= S.n_reqs || !r.same(S.reqs[i]) || (i <= S.n_finished && !gen_state_same(now, S.state[i])) ||
      i > S.n_finished) {
    // (i > n_finished: request i - 1 was adopted, so i <= n_finished always -- kept as a guard)
    legacy_spec_cancel(ctx);
    return false;
  }
  unsigned spins = 0;
  while (S.n_finished <= i && !S.failed) {
    if (S.in_flight < 0) legacy_spec_enqueue_next(ctx);
    legacy_spec_poll(ctx);
    if (S.n_finished > i || S.failed) break;
    if (++spins > 200000u) {      // ~1 ms of polling: the draw is long -- sleep on its stream instead
      (void)hipStreamSynchronize(S.stream);
      spins = 0;
    } else {
      __builtin_ia32_pause();
    }
  }
  if (S.failed) {
    legacy_spec_cancel(ctx);
    return false;
  }
  *index = i;
  return true;
}

static void legacy_spec_adopted(vb_ctx* ctx, vb_legacy_rng* rng, int i) {
  LegacySpec& S = *ctx->legacy_spec;
  const LegacyGenState& e = S.state[i + 1];
  (void)vb_legacy_rng_set_state(rng, e.key, e.pos, e.has_gauss, e.gauss);
  ++S.adopted;
  S.discard_streak = 0;
  S.n_adopted = i + 1;
  if (S.n_adopted >= S.n_reqs) S.active = false;      // the job is used up
}

static void legacy_round_note(vb_ctx* ctx, const vb_legacy_rng* rng, const LegacyReq& r) {
  if (legacy_spec_create(ctx) != VB_OK) return;
  LegacySpec& S = *ctx->legacy_spec;
  if (S.round_rng != rng || S.round_uid != vb_legacy_rng_uid(rng)) {
    S.round_rng = rng;
    S.round_uid = vb_legacy_rng_uid(rng);
    S.n_round = 0;
    S.round_overflow = false;
    S.n_prev = -1;
    S.discard_streak = S.cooldown = 0;
  }
  if (S.n_round < LegacySpec::kMaxReqs) S.round[S.n_round++] = r;
  else S.round_overflow = true;
}

int vb_legacy_rng_randn_device(vb_ctx* ctx, vb_legacy_rng* rng, int slot, int64_t n_total, int64_t d, int64_t row_begin,
                               int64_t rows) {
  if (!ctx || !rng) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (n_total <= 0 || d <= 0 || row_begin < 0 || rows <= 0 || row_begin + rows > n_total)
    return fail(ctx, VB_ERR_INVALID, "rows [%lld, %lld) of a %lld x %lld draw", (long long)row_begin,
                (long long)(row_begin + rows), (long long)n_total, (long long)d);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  VB_TRY(noise_alloc(ctx, slot, rows, d));
  LegacyReq req;
  req.prog = -1, req.n_total = n_total, req.d = d, req.row_begin = row_begin, req.rows = rows, req.slot = slot;
  int si = 0;
  legacy_spec_poll(ctx);
  if (legacy_spec_take(ctx, rng, req, &si)) {      // the look-ahead drew exactly this: its buffer becomes the slot's
    LegacySpec& S = *ctx->legacy_spec;
    if (S.shadow[si].bytes >= (size_t)rows * ctx->noise[slot].ld * sizeof(double)) {
      std::swap(ctx->noise[slot].buf, S.shadow[si]);
      legacy_spec_adopted(ctx, rng, si);
      legacy_round_note(ctx, rng, req);
      return VB_OK;
    }
    legacy_spec_cancel(ctx);
  }
  uint32_t key[624];
  int pos = 0, has_gauss = 0;
  double gauss = 0.0;
  VB_TRY(vb_legacy_rng_get_state(rng, key, &pos, &has_gauss, &gauss));
  const int rc = legacy_dev_randn(ctx, key, &pos, &has_gauss, &gauss, ctx->noise[slot], n_total, d, row_begin, rows);
  if (rc == VB_ERR_UNSUPPORTED) return fail(ctx, VB_ERR_UNSUPPORTED, "device draw not available for this request: draw on the host");
  VB_TRY(rc);
  legacy_round_note(ctx, rng, req);
  return vb_legacy_rng_set_state(rng, key, pos, has_gauss, gauss);
}

// chisquare (prog 0) / standard_t (prog 1) on the device (vb_legacy_gamma.hip).  The device path starts from a generator
// without a cached normal: while one is cached, the leading values are drawn one at a time by the host generator (each
// such draw leaves the cache set with probability ~1/2) and copied to their places.
static int legacy_gamma_device(vb_ctx* ctx, vb_legacy_rng* rng, int prog, double df, double* dst, int64_t ld, int64_t n_total,
                               int64_t d, int64_t row_begin, int64_t rows, LegacyFinish* defer = nullptr) {
  uint32_t key[624];
  int pos = 0, has_gauss = 0;
  double gauss = 0.0;
  VB_TRY(vb_legacy_rng_get_state(rng, key, &pos, &has_gauss, &gauss));
  uint32_t key0[624];
  memcpy(key0, key, sizeof key0);
  const int pos0 = pos, has0 = has_gauss;
  const double gauss0 = gauss;
  auto rewind = [&]() { return vb_legacy_rng_set_state(rng, key0, pos0, has0, gauss0); };
  const int64_t n_out = n_total * d;
  int64_t o = 0;
  // The device path starts from a generator without a cached normal: while there is one, values are drawn on the host
  // (a couple on average, now and then dozens: each value flips the cache an unpredictable number of times).  They are
  // collected and uploaded in row runs through push_small -- one staged copy per run, no wait; a copy + stream wait PER
  // VALUE made a 0.17 ms draw take 0.5-3 ms every other call.
  // (declined or failed means generator untouched: every failure below rewinds before it returns -- ADVICE r5)
  std::vector<double> head;
  while (has_gauss && o < n_out) {
    double v = 0.0;
    int rc = prog == 1 ? vb_legacy_rng_standard_t(rng, df, &v, 1) : vb_legacy_rng_chisquare(rng, df, &v, 1);
    if (rc == VB_OK) rc = vb_legacy_rng_get_state(rng, key, &pos, &has_gauss, &gauss);
    if (rc != VB_OK) {
      (void)rewind();
      return rc;
    }
    head.push_back(v);
    ++o;
  }
  for (int64_t i = 0; i < o;) {      // value i is entry (i / d, i % d) of the request
    const int64_t row = i / d, col = i - row * d;
    int64_t run = d - col < o - i ? d - col : o - i;
    if (row >= row_begin && row < row_begin + rows) {
      const int rc = push_small(ctx, ctx->stream, head.data() + i, (size_t)run * sizeof(double), dst + (row - row_begin) * ld + col);
      if (rc != VB_OK) {
        (void)rewind();
        return rc;
      }
    }
    i += run;
  }
  if (o == n_out) {
    if (defer) {      // (nothing left for the device: a look-ahead draw needs a finish to poll -- not this path's case)
      (void)rewind();
      return VB_ERR_UNSUPPORTED;
    }
    return VB_OK;
  }
  const int rc = legacy_dev_gamma(ctx, prog, df, key, &pos, &has_gauss, &gauss, dst, ld, o, n_total, d, row_begin, rows, defer);
  if (defer && rc == VB_OK) return VB_OK;      // (the generator stands behind the host-drawn head; the caller completes the draw)
  if (rc != VB_OK) {
    (void)rewind();
    if (rc == VB_ERR_UNSUPPORTED) return fail(ctx, VB_ERR_UNSUPPORTED, "device draw not available for this request: draw on the host");
    return rc;
  }
  return vb_legacy_rng_set_state(rng, key, pos, has_gauss, gauss);
}

int vb_legacy_rng_standard_t_device(vb_ctx* ctx, vb_legacy_rng* rng, double df, int slot, int64_t n_total, int64_t d,
                                    int64_t row_begin, int64_t rows) {
  if (!ctx || !rng) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (n_total <= 0 || d <= 0 || row_begin < 0 || rows <= 0 || row_begin + rows > n_total || !(df > 0.0))
    return fail(ctx, VB_ERR_INVALID, "rows [%lld, %lld) of a %lld x %lld draw, df %g", (long long)row_begin,
                (long long)(row_begin + rows), (long long)n_total, (long long)d, df);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  VB_TRY(noise_alloc(ctx, slot, rows, d));
  LegacyReq req;
  req.prog = 1, req.df = df, req.n_total = n_total, req.d = d, req.row_begin = row_begin, req.rows = rows, req.slot = slot;
  int si = 0;
  legacy_spec_poll(ctx);
  if (legacy_spec_take(ctx, rng, req, &si)) {
    LegacySpec& S = *ctx->legacy_spec;
    if (S.shadow[si].bytes >= (size_t)rows * ctx->noise[slot].ld * sizeof(double)) {
      std::swap(ctx->noise[slot].buf, S.shadow[si]);
      legacy_spec_adopted(ctx, rng, si);
      legacy_round_note(ctx, rng, req);
      return VB_OK;
    }
    legacy_spec_cancel(ctx);
  }
  NoiseSlot& ns = ctx->noise[slot];
  VB_TRY(legacy_gamma_device(ctx, rng, 1, df, (double*)ns.buf.ptr, ns.ld, n_total, d, row_begin, rows));
  legacy_round_note(ctx, rng, req);
  return VB_OK;
}

int vb_legacy_rng_chisquare_device(vb_ctx* ctx, vb_legacy_rng* rng, double df, int64_t n, double* host_out) {
  if (!ctx || !rng) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (n <= 0 || !(df > 0.0)) return fail(ctx, VB_ERR_INVALID, "%lld chi-square draws, df %g", (long long)n, df);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  VB_TRY(ensure(ctx, ctx->chi_dev, (size_t)n * sizeof(double)));
  ctx->chi_n = 0;
  ctx->chi_ahead.last.valid = ctx->chi_ahead.pre.valid = false;      // (not a Philox request: the look-ahead's history ends)
  ctx->chi_ahead.streak = 0;
  LegacyReq req;
  req.prog = 0, req.df = df, req.n_total = n, req.d = 1, req.row_begin = 0, req.rows = n, req.slot = -1;
  int si = 0;
  bool adopted = false;
  legacy_spec_poll(ctx);
  if (legacy_spec_take(ctx, rng, req, &si)) {
    LegacySpec& S = *ctx->legacy_spec;
    if (S.shadow[si].bytes >= (size_t)n * sizeof(double)) {
      std::swap(ctx->chi_dev, S.shadow[si]);
      legacy_spec_adopted(ctx, rng, si);
      adopted = true;
    } else {
      legacy_spec_cancel(ctx);
    }
  }
  if (!adopted) VB_TRY(legacy_gamma_device(ctx, rng, 0, df, (double*)ctx->chi_dev.ptr, 1, n, 1, 0, n));
  legacy_round_note(ctx, rng, req);
  ctx->chi_n = n;
  ctx->chi_df = df;
  if (host_out) {
    VB_HIP(ctx, hipMemcpyAsync(host_out, ctx->chi_dev.ptr, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return VB_OK;
}

int vb_legacy_round_end(vb_ctx* ctx, vb_legacy_rng* rng) {
  if (!ctx || !rng) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  LegacySpec* Sp = ctx->legacy_spec;
  if (!Sp || Sp->round_rng != rng || Sp->round_uid != vb_legacy_rng_uid(rng))
    return VB_OK;      // no device draw of this generator since the last round end
  LegacySpec& S = *Sp;
  bool same = !S.round_overflow && S.n_round > 0 && S.n_round == S.n_prev;
  for (int i = 0; same && i < S.n_round; ++i) same = S.round[i].same(S.prev_round[i]);
  S.n_prev = S.round_overflow ? -1 : S.n_round;
  for (int i = 0; i < S.n_round; ++i) S.prev_round[i] = S.round[i];
  S.n_round = 0;
  S.round_overflow = false;
  if (!legacy_spec_on() || !vb_legacy_rng_log_proven()) return VB_OK;
  if (S.active) {      // (a job whose requests were not all asked for: the rounds changed)
    legacy_spec_cancel(ctx);
    return VB_OK;
  }
  if (!same) return VB_OK;
  if (S.cooldown > 0) {
    if (--S.cooldown == 0) S.discard_streak = 0;
    return VB_OK;
  }
  VB_HIP(ctx, hipSetDevice(ctx->device));
  // the next round's draws, from where the generator stands now.  The shadows were the slots' buffers until the last adoption:
  // whatever is queued on the main stream may still read them
  VB_HIP(ctx, hipEventRecord(S.ev_main, ctx->stream));
  VB_HIP(ctx, hipStreamWaitEvent(S.stream, S.ev_main, 0));
  S.n_reqs = S.n_prev;
  for (int i = 0; i < S.n_reqs; ++i) S.reqs[i] = S.prev_round[i];
  gen_state_get(rng, &S.state[0]);
  S.n_adopted = S.n_finished = 0;
  S.in_flight = -1;
  S.failed = false;
  S.active = true;
  ++S.launched;
  // Round 6: the multivariate t's round (chi-square, then normals) belongs to a long call -- 0.6 ms of root, products and
  // bisection with stream synchronisations inside -- and enqueuing its first draw here costs the host ~0.15 ms (twenty-odd
  // launches) in front of the call's own first launch.  Its start waits for the caller's first wait instead (legacy_poll: in
  // front of the root's synchronisation, where the host would idle), or for the request itself.  Every other round starts at
  // once.  VB_LEGACY_DEFER=0: always at once.
  // Measured and NOT done: deferring every multi-draw round (the low-rank family's two normal draws: 377-405 -> 430-449 us -- its
  // call has no wait before the final one), or any round whose last job nobody had to wait for (the dense Gaussian family:
  // 630 -> 680-700 us -- its first wait, the pageable parameter upload, is a blocking call, not a wait the host could use).
  const char* de = getenv("VB_LEGACY_DEFER");
  const bool t_round = S.n_reqs >= 2 && S.reqs[0].prog == 0;      // chi-square first: the multivariate t
  if (!t_round || (de && atoi(de) == 0)) legacy_spec_enqueue_next(ctx);
  return VB_OK;
}

int vb_legacy_ahead_stats(vb_ctx* ctx, uint64_t* launched, uint64_t* adopted, uint64_t* discarded) {
  if (!ctx || !launched || !adopted || !discarded) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  const LegacySpec* S = ctx->legacy_spec;
  *launched = S ? S->launched : 0;
  *adopted = S ? S->adopted : 0;
  *discarded = S ? S->discarded : 0;
  return VB_OK;
}

int vb_set_model_callback(vb_ctx* ctx, int64_t dim, vb_model_callback fn, void* user) {
  if (!ctx) return VB_ERR_INVALID;
  if (!fn) return fail(ctx, VB_ERR_INVALID, "NULL callback");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  return user_model_set_callback(ctx, dim, fn, user);
}

int vb_model_logp(vb_ctx* ctx, const double* x_host, int64_t n, int64_t d, double* out_host) {
  if (!ctx || !x_host || !out_host) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (d != ctx->model.dim)
    return fail(ctx, VB_ERR_INVALID, "x has %lld columns, model dimension is %d", (long long)d,
                ctx->model.dim);
  if (n <= 0) return fail(ctx, VB_ERR_INVALID, "n must be positive");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t ld = round_up(d, 16);
  VB_TRY(ensure(ctx, ctx->scratch, (size_t)n * ld * sizeof(double)));
  VB_TRY(ensure(ctx, ctx->scratch2, (size_t)n * sizeof(double)));
  double* xd = (double*)ctx->scratch.ptr;
  double* od = (double*)ctx->scratch2.ptr;
  VB_HIP(ctx, hipMemcpy2DAsync(xd, (size_t)ld * sizeof(double), x_host, (size_t)d * sizeof(double),
                               (size_t)d * sizeof(double), (size_t)n, hipMemcpyHostToDevice,
                               ctx->stream));
  VB_TRY(model_logp_rows(ctx, xd, ld, n, d, od));
  VB_HIP(ctx, hipMemcpyAsync(out_host, od, (size_t)n * sizeof(double), hipMemcpyDeviceToHost,
                             ctx->stream));
  VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VB_OK;
}

int vb_noise_moments(vb_ctx* ctx, int slot, int64_t n, int64_t d, double* colsum, double* gram) {
  if (!ctx || !colsum) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (slot < 0 || slot >= VB_MAX_SLOTS || !ctx->noise[slot].buf.ptr)
    return fail(ctx, VB_ERR_INVALID, "noise slot %d is empty", slot);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  return noise_moments(ctx, ctx->noise[slot], n, d, colsum, gram);
}

int vb_model_grad(vb_ctx* ctx, const double* x_host, int64_t n, int64_t d, double* f_host, double* g_host) {
  if (!ctx || !x_host || !g_host) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (d != ctx->model.dim)
    return fail(ctx, VB_ERR_INVALID, "x has %lld columns, model dimension is %d", (long long)d,
                ctx->model.dim);
  if (n <= 0) return fail(ctx, VB_ERR_INVALID, "n must be positive");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t ld = round_up(d, 16);
  VB_TRY(ensure(ctx, ctx->scratch, (size_t)2 * n * ld * sizeof(double)));
  VB_TRY(ensure(ctx, ctx->scratch2, (size_t)n * sizeof(double)));
  double* xd = (double*)ctx->scratch.ptr;
  double* gd = xd + n * ld;
  double* fd = (double*)ctx->scratch2.ptr;
  VB_HIP(ctx, hipMemsetAsync(xd, 0, (size_t)n * ld * sizeof(double), ctx->stream));
  VB_HIP(ctx, hipMemcpy2DAsync(xd, (size_t)ld * sizeof(double), x_host, (size_t)d * sizeof(double),
                               (size_t)d * sizeof(double), (size_t)n, hipMemcpyHostToDevice,
                               ctx->stream));
  VB_TRY(model_grad_rows(ctx, xd, ld, n, d, gd, fd));
  if (f_host)
    VB_HIP(ctx, hipMemcpyAsync(f_host, fd, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  VB_HIP(ctx, hipMemcpy2DAsync(g_host, (size_t)d * sizeof(double), gd, (size_t)ld * sizeof(double),
                               (size_t)d * sizeof(double), (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VB_OK;
}

// ---- ExclusiveKL, mean field ------------------------------------------------------------------
// Enqueue `count` independent evaluations: evaluation b streams noise slot slots[b] with parameter
// thetas[b * 2d ...] and lands in result slot *rs[b].
struct GenNoise {            // in-register noise of a single evaluation (MfCall::gen)
  uint64_t seed, stream;
  int64_t row_offset;
};

static int mf_call(vb_ctx* ctx, int count, const int* slots, int64_t n, int64_t d, int64_t n_total,
                   int family, double df, const double* thetas, unsigned flags, int cv_mode,
                   ResultSlot** rs, bool pipelined, bool overlap_comm = false, bool alternate = false,
                   const GenNoise* gen = nullptr, bool blocking = false) {
  if (!ctx || !thetas || !slots) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (count < 1) return fail(ctx, VB_ERR_INVALID, "count must be positive");
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (n_total < n) return fail(ctx, VB_ERR_INVALID, "n_total must be >= n");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  for (int b0 = 0; b0 < count; b0 += kMaxBatch) {
    MfCall c;
    c.count = count - b0 < kMaxBatch ? count - b0 : kMaxBatch;
    for (int b = 0; b < c.count; ++b) {
      const int slot = slots[b0 + b];
      VB_TRY(check_slot(ctx, slot));
      if (!gen && !ctx->noise[slot].buf.ptr) return fail(ctx, VB_ERR_STATE, "noise slot %d is empty", slot);
      ResultSlot& r = *rs[b0 + b];
      VB_TRY(stage_theta(ctx, r, thetas + (size_t)(b0 + b) * 2 * d, 2 * d));
      c.noise[b] = gen ? &ctx->gen_geom : &ctx->noise[slot];
      c.theta_src[b] = r.dev;
      c.out[b] = r.dev + r.p;
    }
    c.n = n;
    c.d = d;
    c.n_total = n_total;
    c.family = family;
    c.df = df;
    c.flags = flags;
    c.cv_mode = cv_mode;
    c.pipelined = pipelined;
    c.overlap_comm = overlap_comm;
    c.alternate = alternate;
    if (gen) {
      c.gen = 1;
      c.gen_seed = gen->seed;
      c.gen_stream = gen->stream;
      c.gen_row_offset = gen->row_offset;
    }
    ctx->done_groups = 0;
    static const bool flagsync = !(getenv("VB_MF_FLAGSYNC") && atoi(getenv("VB_MF_FLAGSYNC")) == 0);
    if (blocking && flagsync && count == 1) {
      if (!ctx->done_host) {
        VB_HIP(ctx, hipHostMalloc((void**)&ctx->done_host, 64 * 8 * sizeof(unsigned long long), hipHostMallocMapped));
        memset(ctx->done_host, 0, 64 * 8 * sizeof(unsigned long long));
        VB_HIP(ctx, hipHostGetDevicePointer((void**)&ctx->done_dev, ctx->done_host, 0));
      }
      c.done_dev = ctx->done_dev;
      c.done_seq = ++ctx->done_seq;
      c.done_groups = &ctx->done_groups;
    }
    VB_TRY(mf_enqueue(ctx, c));
    VB_TRY(ticket(ctx, rs + b0, c.count));
  }
  return VB_OK;
}

// Wait for the blocking mean-field evaluation just enqueued: on the finalize kernel's completion words when it signals
// (mf_call(..., blocking)), else -- or after 2 ms of polling -- on the stream.
static int mf_wait_blocking(vb_ctx* ctx) {
  const int groups = ctx->done_groups;
  if (groups > 0) {
    const unsigned long long seq = ctx->done_seq;
    volatile unsigned long long* w = ctx->done_host;
    for (unsigned spins = 0; spins < 400000u; ++spins) {
      int g = 0;
      while (g < groups && w[8 * g] == seq) ++g;
      if (g == groups) {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        return comm_check(ctx);
      }
      __builtin_ia32_pause();
    }
  }
  VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VB_OK;
}

int vb_elbo_grad_meanfield(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total,
                           int family, double df, const double* theta, unsigned flags,
                           int cv_mode, double* value, double* grad) {
  if (!ctx || !value || !grad) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  ResultSlot* rs = &ctx->sync_result;
  VB_TRY(mf_call(ctx, 1, &slot, n, d, n_total, family, df, theta, flags, cv_mode, &rs, false, false, false, nullptr, true));
  VB_TRY(mf_wait_blocking(ctx));
  rs->pending = false;
  *value = rs->host[rs->p];
  memcpy(grad, rs->host + rs->p + 1, (size_t)(2 * d) * sizeof(double));
  return VB_OK;
}

int vb_elbo_grad_meanfield_philox(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, int64_t row_offset,
                                  int family, double df, const double* theta, unsigned flags, int cv_mode,
                                  uint64_t seed, uint64_t stream, double* value, double* grad) {
  if (!ctx || !value || !grad) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  if (n <= 0 || d <= 0) return fail(ctx, VB_ERR_INVALID, "noise shape must be positive");
  // the noise stays in registers: the kernels only need the GEOMETRY of the matrix that vb_noise_generate would
  // have written (row stride as noise_alloc pads it).  No slot is allocated or claimed (this used to
  // allocate an n x ld buffer that was never read and left the slot describing stale data).
  {
    NoiseSlot& gslot = ctx->gen_geom;
    int64_t ld = round_up(d, 16);
    if (ld % 512 == 0) ld += 16;
    VB_TRY(ensure(ctx, gslot.buf, 256));     // a valid address; never dereferenced in GEN mode
    gslot.n = n;
    gslot.d = d;
    gslot.ld = ld;
  }
  ResultSlot* rs = &ctx->sync_result;
  const GenNoise gen{seed, stream, row_offset};
  VB_TRY(mf_call(ctx, 1, &slot, n, d, n_total, family, df, theta, flags, cv_mode, &rs, false, false, false, &gen, true));
  VB_TRY(mf_wait_blocking(ctx));
  rs->pending = false;
  *value = rs->host[rs->p];
  memcpy(grad, rs->host + rs->p + 1, (size_t)(2 * d) * sizeof(double));
  return VB_OK;
}

int vb_elbo_grad_meanfield_async(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total,
                                 int family, double df, const double* theta, unsigned flags,
                                 int cv_mode, int rslot) {
  if (!ctx) return VB_ERR_INVALID;
  VB_TRY(check_slot(ctx, rslot));
  ResultSlot* rs = &ctx->results[rslot];
  return mf_call(ctx, 1, &slot, n, d, n_total, family, df, theta, flags, cv_mode, &rs, false);
}

int vb_elbo_grad_meanfield_batch_async(vb_ctx* ctx, int count, const int* slots, int64_t n, int64_t d,
                                       int64_t n_total, int family, double df, const double* thetas,
                                       unsigned flags, int cv_mode, const int* rslots) {
  if (!ctx || !rslots) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (count < 1 || count > VB_MAX_SLOTS)
    return fail(ctx, VB_ERR_INVALID, "count %d outside [1, %d]", count, VB_MAX_SLOTS);
  ResultSlot* rs[VB_MAX_SLOTS];
  for (int b = 0; b < count; ++b) {
    VB_TRY(check_slot(ctx, rslots[b]));
    for (int a = 0; a < b; ++a)
      if (rslots[a] == rslots[b]) return fail(ctx, VB_ERR_INVALID, "result slot %d used twice", rslots[b]);
    rs[b] = &ctx->results[rslots[b]];
  }
  // VB_PIPELINE=1 spreads prep / stream / finalize over three event-chained HIP streams.  Measured
  // on MI355X (ROCm 7.2) the cross-stream event waits cost more than the overlap buys, so the
  // default keeps a batch in order on the main stream; independent contexts overlap instead.
  static const bool pipelined = getenv("VB_PIPELINE") && atoi(getenv("VB_PIPELINE")) != 0;
  // sharded jobs overlap the all-reduce of one batch with the kernels of the next (VB_COMM_OVERLAP=0 disables)
  static const bool overlap = !(getenv("VB_COMM_OVERLAP") && atoi(getenv("VB_COMM_OVERLAP")) == 0);
  // VB_MF_ALT=1 (single GPU): consecutive batches alternate between two streams.  Measured +3 % throughput
  // (157 vs 153 k evaluations/s at C1): the streaming kernel already saturates HBM, so only the small prep /
  // finalize kernels overlap -- and two concurrent streaming kernels make per-kernel timings meaningless.  Off.
  static const bool alt = getenv("VB_MF_ALT") && atoi(getenv("VB_MF_ALT")) != 0;
  return mf_call(ctx, count, slots, n, d, n_total, family, df, thetas, flags, cv_mode, rs, pipelined,
                 overlap && ctx->comm != nullptr, alt && ctx->comm == nullptr);
}

int vb_result_get(vb_ctx* ctx, int rslot, double* value, double* grad, int64_t p) {
  if (!ctx || !value || !grad) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(check_slot(ctx, rslot));
  ResultSlot& rs = ctx->results[rslot];
  if (!rs.host || !rs.pending || p > rs.p)
    return fail(ctx, VB_ERR_STATE, "result slot %d holds no pending result of size %lld", rslot,
                (long long)p);
  VB_TRY(wait_ticket(ctx, rs.batch_id));
  *value = rs.host[rs.p];
  memcpy(grad, rs.host + rs.p + 1, (size_t)p * sizeof(double));
  return VB_OK;
}

// ---- ExclusiveKL, low-rank Gaussian family (approximations.py:610-731) ------------------------------
int vb_elbo_grad_lowrank(vb_ctx* ctx, int slot_eps, int slot_z, int64_t n, int64_t d, int64_t k, int64_t n_total,
                         const double* theta, unsigned flags, double* value, double* grad) {
  if (!ctx || !theta || !value || !grad) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(check_slot(ctx, slot_eps));
  VB_TRY(check_slot(ctx, slot_z));
  if (slot_eps == slot_z) return fail(ctx, VB_ERR_INVALID, "eps and z need different noise slots");
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (!ctx->noise[slot_eps].buf.ptr || !ctx->noise[slot_z].buf.ptr)
    return fail(ctx, VB_ERR_STATE, "noise slot is empty");
  if (flags != 0) return fail(ctx, VB_ERR_UNSUPPORTED, "low-rank family: only the entropy-form estimator is implemented");
  if (d <= 0 || k <= 0) return fail(ctx, VB_ERR_INVALID, "d and k must be positive");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t p = 2 * d + d * k;
  ResultSlot& rs = ctx->sync_result;
  VB_TRY(stage_theta(ctx, rs, theta, p));
  VB_TRY(lr_elbo_grad_enqueue(ctx, ctx->noise[slot_eps], ctx->noise[slot_z], n, d, k, n_total, rs.dev, rs.dev + rs.p));
  VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  rs.pending = false;
  *value = rs.host[rs.p];
  memcpy(grad, rs.host + rs.p + 1, (size_t)p * sizeof(double));
  return VB_OK;
}

int vb_elbo_sums_lowrank(vb_ctx* ctx, int slot_eps, int slot_z, int64_t n, int64_t d, int64_t k, const double* theta,
                         double* out) {
  if (!ctx || !theta || !out) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(check_slot(ctx, slot_eps));
  VB_TRY(check_slot(ctx, slot_z));
  if (slot_eps == slot_z) return fail(ctx, VB_ERR_INVALID, "eps and z need different noise slots");
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (!ctx->noise[slot_eps].buf.ptr || !ctx->noise[slot_z].buf.ptr)
    return fail(ctx, VB_ERR_STATE, "noise slot is empty");
  if (d <= 0 || k <= 0) return fail(ctx, VB_ERR_INVALID, "d and k must be positive");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  return lr_elbo_sums_any_rank(ctx, ctx->noise[slot_eps], ctx->noise[slot_z], n, d, k, theta, out);
}

// ---- importance weights + PSIS (convenience.py:166-179, _psis.py:113-209) ---------------------------
int vb_log_weights_meanfield(vb_ctx* ctx, int slot, int64_t n, int64_t d, int family, double df,
                             const double* theta, double* lw) {
  if (!ctx || !theta) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(check_slot(ctx, slot));
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (!ctx->noise[slot].buf.ptr) return fail(ctx, VB_ERR_STATE, "noise slot %d is empty", slot);
  if (family != VB_FAMILY_MF_GAUSSIAN && family != VB_FAMILY_MF_STUDENT_T)
    return fail(ctx, VB_ERR_INVALID, "family %d is not a mean-field family", family);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  ResultSlot& rs = ctx->sync_result;
  VB_TRY(stage_theta(ctx, rs, theta, 2 * d));
  VB_TRY(log_weights_enqueue(ctx, ctx->noise[slot], n, d, family, df, rs.dev));
  if (lw)
    VB_HIP(ctx, hipMemcpyAsync(lw, ctx->psis_lw.ptr, (size_t)n * sizeof(double), hipMemcpyDeviceToHost,
                               ctx->stream));
  VB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VB_OK;
}

int vb_psis_smooth(vb_ctx* ctx, const double* lw_in, int64_t n, double reff, double* lw_out, double* khat) {
  if (!ctx || !lw_out || !khat) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (n <= 1) return fail(ctx, VB_ERR_INVALID, "More than one log-weight needed.");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  if (lw_in) {
    VB_TRY(ensure(ctx, ctx->psis_lw, (size_t)(round_up(n, 16) + 16) * sizeof(double)));
    VB_HIP(ctx, hipMemcpyAsync(ctx->psis_lw.ptr, lw_in, (size_t)n * sizeof(double), hipMemcpyHostToDevice,
                               ctx->stream));
    ctx->psis_n = n;
  } else if (ctx->psis_n != n) {
    return fail(ctx, VB_ERR_STATE, "no device-resident log weights of length %lld (vb_log_weights_*)",
                (long long)n);
  }
  VB_TRY(psis_enqueue(ctx, n, reff));
  double res[4];
  double* lw = (double*)ctx->psis_lw.ptr;
  const FetchSeg psis_segs[2] = {{lw, (size_t)n * sizeof(double), lw_out}, {lw + round_up(n, 16), sizeof res, res}};
  VB_TRY(fetch_blocking(ctx, ctx->stream, psis_segs, 2));
  // (the multi-workgroup kernel poisons its results with NaN when a workgroup did not reach a grid barrier within the
  // poll bound: k-hat and the tail count both NaN)
  if (res[0] != res[0] && res[1] != res[1])
    return fail(ctx, VB_ERR_STATE, "PSIS: a workgroup of the smoothing kernel did not arrive at a grid barrier (results invalid); "
                                   "VB_PSIS_GRID=0 selects the single-workgroup kernel");
#ifdef VB_PSIS_CLOCK
  {
    double dbg[16];
    (void)hipMemcpy(dbg, lw + round_up(n, 16), sizeof dbg, hipMemcpyDeviceToHost);
    fprintf(stderr, "psis phases (us):");
    for (int i = 5; i < 13; ++i) fprintf(stderr, " %.1f", (dbg[i] - dbg[i - 1]) / 100.0);
    fprintf(stderr, "\n");
  }
#endif
  if (getenv("VB_PSIS_TRACE")) {
    double dbg[16];
    (void)hipMemcpy(dbg, lw + round_up(n, 16), sizeof dbg, hipMemcpyDeviceToHost);
    fprintf(stderr, "[psis] k %.10g tail %.0f xcutoff %.17g sigma %.10g |", dbg[0], dbg[1], dbg[2], dbg[3]);
    for (int i = 4; i < 12; ++i) fprintf(stderr, " %.17g", dbg[i]);
    fprintf(stderr, "\n");
  }
  ctx->psis_n = 0;        // the resident weights have been smoothed in place
  *khat = res[0];
  return VB_OK;
}

// ---- AlphaDivergence, mean field (objectives.py:443-463) -------------------------------------------
int vb_alpha_grad_meanfield(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, int family, double df,
                            const double* theta, double alpha, double* value, double* grad) {
  if (!ctx || !theta || !value || !grad) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(check_slot(ctx, slot));
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (!ctx->noise[slot].buf.ptr) return fail(ctx, VB_ERR_STATE, "noise slot %d is empty", slot);
  if (family != VB_FAMILY_MF_GAUSSIAN && family != VB_FAMILY_MF_STUDENT_T)
    return fail(ctx, VB_ERR_INVALID, "family %d is not a mean-field family", family);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  ResultSlot& rs = ctx->sync_result;
  VB_TRY(stage_theta(ctx, rs, theta, 2 * d));
  VB_TRY(alpha_enqueue(ctx, ctx->noise[slot], n, n_total, d, family, df, alpha, rs.dev, rs.dev + rs.p));
  VB_TRY(wait_then_prefetch(ctx));
  rs.pending = false;
  *value = rs.host[rs.p];
  memcpy(grad, rs.host + rs.p + 1, (size_t)(2 * d) * sizeof(double));
  return VB_OK;
}

int vb_alpha_grad_fullrank(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, const double* theta,
                           double alpha, double* value, double* grad) {
  if (!ctx || !theta || !value || !grad) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(check_slot(ctx, slot));
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (!ctx->noise[slot].buf.ptr) return fail(ctx, VB_ERR_STATE, "noise slot %d is empty", slot);
  VB_TRY(vb_fullrank_set_theta(ctx, theta, d));
  double sum_log_diag = 0.0;
  for (int64_t i = 0; i < d; ++i) sum_log_diag += theta[d + i * (i + 1) / 2 + i];   // free diagonal = log L_ii
  VB_TRY(alpha_fullrank_enqueue(ctx, ctx->noise[slot], n, n_total, d, alpha, (const double*)ctx->fr_theta.ptr,
                                sum_log_diag, (double*)ctx->fr_out.ptr));
  return vb_fullrank_get(ctx, value, grad, d + d * (d + 1) / 2);
}

int vb_alpha_grad_mvt_chol(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df, const double* theta,
                           double alpha, double* value, double* grad) {
  if (!ctx || !theta || !value || !grad) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(check_slot(ctx, slot));
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (!ctx->noise[slot].buf.ptr) return fail(ctx, VB_ERR_STATE, "noise slot %d is empty", slot);
  if (n_total < n) return fail(ctx, VB_ERR_INVALID, "n_total must be >= n");
  if (!(df > 2.0)) return fail(ctx, VB_ERR_INVALID, "df must be greater than 2");
  if (ctx->chi_n != n || ctx->chi_df != df || !ctx->chi_dev.ptr)
    return fail(ctx, VB_ERR_STATE, "needs %lld device chi-square(%g) draws (vb_chisq_generate)", (long long)n, df);
  VB_TRY(vb_fullrank_set_theta(ctx, theta, d));
  double sum_log_diag = 0.0;
  for (int64_t i = 0; i < d; ++i) sum_log_diag += theta[d + i * (i + 1) / 2 + i];   // free diagonal = log L_ii
  VB_TRY(mvt_alpha_chol_enqueue(ctx, ctx->noise[slot], n, d, n_total, df, alpha, (const double*)ctx->fr_theta.ptr,
                                sum_log_diag, (double*)ctx->fr_out.ptr));
  return vb_fullrank_get(ctx, value, grad, d + d * (d + 1) / 2);
}

// ---- DISInclusiveKL, mean field (objectives.py:283-416) ---------------------------------------------
int vb_dis_refresh_meanfield(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, int family, double df,
                             const double* theta, const double* prior_theta, double eps_prev,
                             double ess_target, int max_bisection_its, double* eps, double* ess,
                             double* w, double* log_p, double* log_q) {
  if (!ctx || !theta || !prior_theta || !eps || !ess || !w) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(check_slot(ctx, slot));
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (!ctx->noise[slot].buf.ptr) return fail(ctx, VB_ERR_STATE, "noise slot %d is empty", slot);
  if (family != VB_FAMILY_MF_GAUSSIAN && family != VB_FAMILY_MF_STUDENT_T)
    return fail(ctx, VB_ERR_INVALID, "family %d is not a mean-field family", family);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  ResultSlot& rs = ctx->sync_result;
  VB_TRY(stage_theta(ctx, rs, theta, 2 * d));
  int status = 0;
  return dis_refresh_enqueue(ctx, ctx->noise[slot], n, n_total, d, family, df, rs.dev, prior_theta, eps_prev, ess_target,
                             max_bisection_its, eps, ess, &status, w, log_p, log_q);
}

int vb_dis_grad_meanfield(vb_ctx* ctx, int slot, int64_t n, int64_t d, int family, double df,
                          const double* theta, const double* weights, double scale, double* value,
                          double* grad) {
  if (!ctx || !theta || !weights || !value || !grad) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(check_slot(ctx, slot));
  if (!ctx->noise[slot].buf.ptr) return fail(ctx, VB_ERR_STATE, "noise slot %d is empty", slot);
  if (family != VB_FAMILY_MF_GAUSSIAN && family != VB_FAMILY_MF_STUDENT_T)
    return fail(ctx, VB_ERR_INVALID, "family %d is not a mean-field family", family);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  ResultSlot& rs = ctx->sync_result;
  VB_TRY(stage_theta(ctx, rs, theta, 2 * d));
  VB_TRY(dis_grad_enqueue(ctx, ctx->noise[slot], n, d, family, df, rs.dev, weights, scale, rs.dev + rs.p));
  VB_TRY(wait_then_prefetch(ctx));
  rs.pending = false;
  *value = rs.host[rs.p];
  memcpy(grad, rs.host + rs.p + 1, (size_t)(2 * d) * sizeof(double));
  return VB_OK;
}

// ---- DISInclusiveKL, MultivariateT (approximations.py:322-382, objectives.py:391-414) ----------------
int vb_dis_refresh_mvt(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df, const double* theta,
                       const double* chi, const double* sqrt_sigma, const double* l_inv,
                       const double* prior_theta, double eps_prev, double ess_target, int max_bisection_its,
                       double* eps, double* ess, double* w, double* log_p, double* log_q) {
  if (!ctx || !theta || !prior_theta) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  // chi may be NULL: device draws (vb_chisq_generate); sqrt_sigma and l_inv both NULL: factors on the device;
  // w NULL (throughput mode): device-resident step -- nothing is copied back and nothing waited for here,
  // vb_dis_step_mvt_packed returns eps / ess with the gradient (sharded jobs: the three per-sample vectors are gathered
  // on the device, the weights are formed redundantly on every rank)
  if (!w && (chi || sqrt_sigma || l_inv))
    return fail(ctx, VB_ERR_INVALID, "w == NULL needs the throughput mode (chi, sqrt_sigma, l_inv NULL)");
  if (w && (!eps || !ess)) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(check_slot(ctx, slot));
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (!ctx->noise[slot].buf.ptr) return fail(ctx, VB_ERR_STATE, "noise slot %d is empty", slot);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  return mvt_dis_refresh(ctx, ctx->noise[slot], n, n_total, d, df, theta, chi, sqrt_sigma, l_inv, prior_theta, eps_prev,
                         ess_target, max_bisection_its, eps, ess, w, log_p, log_q);
}

int vb_dis_refresh_mvt_symroot(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df, const double* theta,
                               const double* prior_theta, double eps_prev, double ess_target, int max_bisection_its,
                               double* root_info) {
  if (!ctx || !theta || !prior_theta) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (n_total < n) return fail(ctx, VB_ERR_INVALID, "n_total must be >= n");
  if (!(df > 2.0)) return fail(ctx, VB_ERR_INVALID, "df must be greater than 2");
  VB_TRY(check_slot(ctx, slot));
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (!ctx->noise[slot].buf.ptr) return fail(ctx, VB_ERR_STATE, "noise slot %d is empty", slot);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  const int rc = mvt_dis_refresh(ctx, ctx->noise[slot], n, n_total, d, df, theta, nullptr, nullptr, nullptr, prior_theta, eps_prev,
                                 ess_target, max_bisection_its, nullptr, nullptr, nullptr, nullptr, nullptr, true, root_info);
  if (rc == VB_ERR_UNSUPPORTED) return fail(ctx, VB_ERR_UNSUPPORTED, "matrix square root: not resolved on the device");
  return rc;
}

static int elbo_grad_mvt_symroot(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df, const double* theta,
                                 double* value, double* grad, double* info, bool path_deriv) {
  if (!ctx || !theta || !value || !grad) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (!(df > 2.0)) return fail(ctx, VB_ERR_INVALID, "df must be greater than 2");
  if (n_total < n) return fail(ctx, VB_ERR_INVALID, "n_total must be >= n");
  VB_TRY(check_slot(ctx, slot));
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (!ctx->noise[slot].buf.ptr) return fail(ctx, VB_ERR_STATE, "noise slot %d is empty", slot);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  const size_t p = (size_t)(d + d * (d + 1) / 2);
  std::vector<double>& vg = ctx->mvt_stage;
  vg.resize(1 + p);
  const int rc = mvt_elbo_symroot(ctx, ctx->noise[slot], n, n_total, d, df, theta, vg.data(), info, path_deriv);
  if (rc == VB_ERR_UNSUPPORTED) return fail(ctx, VB_ERR_UNSUPPORTED, "matrix square root: not resolved on the device");
  VB_TRY(rc);
  *value = vg[0];
  memcpy(grad, vg.data() + 1, p * sizeof(double));
  return VB_OK;
}

int vb_elbo_grad_mvt_symroot(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df, const double* theta,
                             double* value, double* grad, double* info) {
  return elbo_grad_mvt_symroot(ctx, slot, n, d, n_total, df, theta, value, grad, info, false);
}

int vb_alpha_grad_mvt_symroot(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df, double alpha,
                              const double* theta, double* value, double* grad, double* info) {
  if (!ctx || !theta || !value || !grad) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (!(df > 2.0)) return fail(ctx, VB_ERR_INVALID, "df must be greater than 2");
  if (n_total < n) return fail(ctx, VB_ERR_INVALID, "n_total must be >= n");
  VB_TRY(check_slot(ctx, slot));
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (!ctx->noise[slot].buf.ptr) return fail(ctx, VB_ERR_STATE, "noise slot %d is empty", slot);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  const size_t p = (size_t)(d + d * (d + 1) / 2);
  std::vector<double>& vg = ctx->mvt_stage;
  vg.resize(1 + p);
  const int rc = mvt_alpha_symroot(ctx, ctx->noise[slot], n, n_total, d, df, alpha, theta, vg.data(), info);
  if (rc == VB_ERR_UNSUPPORTED) return fail(ctx, VB_ERR_UNSUPPORTED, "matrix square root: not resolved on the device");
  VB_TRY(rc);
  *value = vg[0];
  memcpy(grad, vg.data() + 1, p * sizeof(double));
  return VB_OK;
}

int vb_elbo_grad_mvt_symroot_path(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df,
                                  const double* theta, double* value, double* grad, double* info) {
  return elbo_grad_mvt_symroot(ctx, slot, n, d, n_total, df, theta, value, grad, info, true);
}

int vb_fullrank_upload_stats(vb_ctx* ctx, uint64_t* pipelined_calls) {
  if (!ctx || !pipelined_calls) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  *pipelined_calls = ctx->fr_up_calls;
  return VB_OK;
}

// Page-locked host memory for result arrays (round 6): a 4.2-MB gradient copied into pageable memory is staged by the runtime
// (118 us at D = 1024: 36 GB/s); into a pinned block the DMA engine writes it directly.  The Python binding hands such
// blocks out as the numpy arrays it returns and takes them back when the arrays die (viabel_amd/_lib.py: PinnedPool).
int vb_host_alloc(size_t bytes, void** ptr) {
  if (!ptr || bytes == 0) return fail(nullptr, VB_ERR_INVALID, "vb_host_alloc: NULL pointer or zero bytes");
  *ptr = nullptr;
  const hipError_t e = hipHostMalloc(ptr, bytes, hipHostMallocDefault);
  if (e != hipSuccess) return fail(nullptr, VB_ERR_HIP, "hipHostMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
  return VB_OK;
}

int vb_host_free(void* ptr) {
  if (!ptr) return VB_OK;
  const hipError_t e = hipHostFree(ptr);
  if (e != hipSuccess) return fail(nullptr, VB_ERR_HIP, "hipHostFree: %s", hipGetErrorString(e));
  return VB_OK;
}

// ---- parking a DIS state (round 6) ---------------------------------------------------------------------
// The reference keeps the state of a DISInclusiveKL per OBJECT (objectives.py:391-403): two objectives with
// num_resampling_batches > 1 may take turns.  Here the state samples live in the context, one set per family kind; an
// objective about to refresh over another one's kept state parks that state first -- the buffers are DETACHED (pointer
// moves, no copies: the context allocates afresh for the newcomer) together with the shapes, the parameter the residuals
// belong to, the generation counter and the noise slot the state reads -- and the owner re-installs it (parking whoever
// holds the context then) before its next kept-weights step.
namespace {
struct DisParked {
  int kind = -1, slot = -1;
  DeviceBuffer state, noise;
  int64_t noise_n = 0, noise_d = 0, noise_ld = 0;
  int64_t n = 0, d = 0, n_total = 0, k = 0, lq_off = 0;
  std::vector<double> theta;
  bool dev_factors = false, e_noise = false, has_noise = false;
  uint64_t gen = 0;
};

void dis_detach(vb_ctx* ctx, int kind, int slot, DisParked* P) {
  P->kind = kind, P->slot = slot, P->gen = ctx->dis_gen[kind];
  if (kind == 0) {
    std::swap(P->state, ctx->dis_state);
    P->n = ctx->dis_n, P->d = ctx->dis_d, P->n_total = ctx->dis_n_total;
    ctx->dis_n = ctx->dis_d = ctx->dis_n_total = 0;
  } else if (kind == 1) {
    std::swap(P->state, ctx->mvt_state);
    P->n = ctx->mvt_n, P->d = ctx->mvt_d, P->n_total = ctx->mvt_n_total, P->lq_off = ctx->mvt_lq_off;
    P->theta.swap(ctx->mvt_theta);
    P->dev_factors = ctx->mvt_dev_factors;
    P->e_noise = ctx->mvt_e_noise != nullptr;
    ctx->mvt_n = ctx->mvt_d = ctx->mvt_n_total = 0;
    ctx->mvt_e_noise = nullptr;
    ctx->mvt_prior.clear();      // (caches keyed on the state buffer)
    ctx->mvt_inv_key[0] = 0;
  } else {
    std::swap(P->state, ctx->lr_obj);
    P->n = ctx->lr_n, P->d = ctx->lr_d, P->k = ctx->lr_k, P->n_total = ctx->lr_n_total;
    ctx->lr_n = ctx->lr_d = ctx->lr_k = ctx->lr_n_total = 0;
  }
  if (slot >= 0 && slot < VB_MAX_SLOTS && ctx->noise[slot].buf.ptr) {
    NoiseSlot& s = ctx->noise[slot];
    std::swap(P->noise, s.buf);
    P->noise_n = s.n, P->noise_d = s.d, P->noise_ld = s.ld;
    s.n = s.d = s.ld = 0;
    s.ahead.last.valid = s.ahead.pre.valid = s.ahead.hint.valid = false;
    s.ahead.streak = 0;
    P->has_noise = true;
  }
}

void dis_attach(vb_ctx* ctx, DisParked* P) {
  const int kind = P->kind;
  ctx->dis_gen[kind] = P->gen;
  if (kind == 0) {
    std::swap(P->state, ctx->dis_state);
    ctx->dis_n = P->n, ctx->dis_d = P->d, ctx->dis_n_total = P->n_total;
  } else if (kind == 1) {
    std::swap(P->state, ctx->mvt_state);
    ctx->mvt_n = P->n, ctx->mvt_d = P->d, ctx->mvt_n_total = P->n_total, ctx->mvt_lq_off = P->lq_off;
    ctx->mvt_theta.swap(P->theta);
    ctx->mvt_dev_factors = P->dev_factors;
    ctx->mvt_prior.clear();
    ctx->mvt_inv_key[0] = 0;
  } else {
    std::swap(P->state, ctx->lr_obj);
    ctx->lr_n = P->n, ctx->lr_d = P->d, ctx->lr_k = P->k, ctx->lr_n_total = P->n_total;
  }
  if (P->has_noise) {
    NoiseSlot& s = ctx->noise[P->slot];
    std::swap(P->noise, s.buf);
    s.n = P->noise_n, s.d = P->noise_d, s.ld = P->noise_ld;
    s.ahead.last.valid = s.ahead.pre.valid = s.ahead.hint.valid = false;
    s.ahead.streak = 0;
    if (kind == 1 && P->e_noise) {
      ctx->mvt_e_noise = (const double*)s.buf.ptr;
      ctx->mvt_e_noise_ld = s.ld;
    }
  }
}

void dis_parked_free(DisParked* P) {
  if (P->state.ptr) (void)hipFree(P->state.ptr);
  if (P->noise.ptr) (void)hipFree(P->noise.ptr);
  delete P;
}
}  // namespace

int vb_dis_state_park(vb_ctx* ctx, int kind, int slot, void** handle) {
  if (!ctx || !handle) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (kind < 0 || kind > 2) return fail(ctx, VB_ERR_INVALID, "DIS state kind %d outside [0, 2]", kind);
  if (slot >= VB_MAX_SLOTS) return fail(ctx, VB_ERR_INVALID, "noise slot %d out of range", slot);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  legacy_spec_cancel(ctx);            // (a look-ahead draw may be aimed at the slot's shadow)
  VB_TRY(sync_streams(ctx));          // nothing in flight reads what is being moved
  ctx->mvt_inv_pending = ctx->mvt_inv_queued = false;
  DisParked* P = new (std::nothrow) DisParked;
  if (!P) return fail(ctx, VB_ERR_HIP, "out of host memory");
  dis_detach(ctx, kind, slot, P);
  *handle = P;
  return VB_OK;
}

int vb_dis_state_unpark(vb_ctx* ctx, void* handle) {
  if (!ctx || !handle) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  DisParked* P = (DisParked*)handle;
  VB_HIP(ctx, hipSetDevice(ctx->device));
  legacy_spec_cancel(ctx);
  VB_TRY(sync_streams(ctx));
  ctx->mvt_inv_pending = ctx->mvt_inv_queued = false;
  // what the context holds now (the caller parked it first if anybody still needs it) goes away with the handle
  dis_attach(ctx, P);
  dis_parked_free(P);      // (after the swaps: the handle owns what the context held)
  return VB_OK;
}

int vb_dis_state_drop(void* handle) {
  if (handle) dis_parked_free((DisParked*)handle);
  return VB_OK;
}

int vb_dis_generation(vb_ctx* ctx, int kind, uint64_t* generation) {
  if (!ctx || !generation) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (kind < 0 || kind > 2) return fail(ctx, VB_ERR_INVALID, "DIS state kind %d outside [0, 2]", kind);
  *generation = ctx->dis_gen[kind];
  return VB_OK;
}

int vb_dis_grad_mvt_packed(vb_ctx* ctx, int64_t n, int64_t d, double df, const double* theta, const double* weights,
                           double scale, double* value, double* grad) {
  if (!ctx || !theta || !weights || !value || !grad) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  const size_t p = (size_t)(d + d * (d + 1) / 2);
  (void)p;
  VB_TRY(mvt_dis_grad(ctx, n, d, df, theta, nullptr, weights, nullptr, nullptr, nullptr, nullptr, scale, value, 0, 0, 0,
                      nullptr, grad));
  return VB_OK;
}

int vb_dis_psis_mvt(vb_ctx* ctx, int64_t n_total, double reff) {
  if (!ctx) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (!(reff > 0.0)) return fail(ctx, VB_ERR_INVALID, "Reff must be positive");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  return mvt_dis_psis_enqueue(ctx, n_total, reff);
}

int vb_dis_clip_mvt(vb_ctx* ctx, int64_t n_total, double threshold) {
  if (!ctx) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  return mvt_dis_clip_enqueue(ctx, n_total, threshold);
}

int vb_dis_step_mvt_packed(vb_ctx* ctx, int64_t n, int64_t d, double df, const double* theta, int64_t resample_m,
                           uint64_t seed, uint64_t stream, double scale, double* eps, double* ess, double* khat,
                           double* value, double* grad) {
  if (!ctx || !theta || !eps || !ess || !value || !grad) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (resample_m < 0) return fail(ctx, VB_ERR_INVALID, "resample_m must be >= 0");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  const size_t p = (size_t)(d + d * (d + 1) / 2);
  (void)p;
  double res[4] = {0.0, 0.0, 0.0, 0.0};
  VB_TRY(mvt_dis_grad(ctx, n, d, df, theta, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, scale, value,
                      resample_m, seed, stream, res, grad));
  *eps = res[0];
  *ess = res[1];
  if (khat) *khat = res[3];
  if ((int)(res[2]) == 3) return fail(ctx, VB_ERR_STATE, "tempering bisection: a workgroup of the resident kernel did not arrive at a grid barrier (results invalid); VB_DIS_RESIDENT=0 selects the launch chain");
  if ((int)res[2] == 1)
    return fail(ctx, VB_ERR_NUMERIC, "All weights zero! Suggests overflow in importance density.");
  return VB_OK;
}

int vb_dis_weights_get(vb_ctx* ctx, double* w, int64_t n_total, int resampled) {
  if (!ctx || !w) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  return mvt_dis_weights_get(ctx, w, n_total, resampled);
}

int vb_dis_scalars_get(vb_ctx* ctx, double out[4]) {
  if (!ctx || !out) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(mvt_dis_scalars_get(ctx, out));
  if ((int)(out[2]) == 3) return fail(ctx, VB_ERR_STATE, "tempering bisection: a workgroup of the resident kernel did not arrive at a grid barrier (results invalid); VB_DIS_RESIDENT=0 selects the launch chain");
  if ((int)out[2] == 1) return fail(ctx, VB_ERR_NUMERIC, "All weights zero! Suggests overflow in importance density.");
  return VB_OK;
}

int vb_dis_state_get(vb_ctx* ctx, int dense, double* log_p, double* log_q, int64_t n_total) {
  if (!ctx) return VB_ERR_INVALID;
  VB_HIP(ctx, hipSetDevice(ctx->device));
  return dense ? mvt_dis_state_get(ctx, log_p, log_q, n_total) : dis_state_get(ctx, log_p, log_q, n_total);
}

int vb_dis_grad_mvt(vb_ctx* ctx, int64_t n, int64_t d, double df, const double* theta, const double* l_inv,
                    const double* weights, double* w_sum, double* w_logq, double* d_mu, double* gram) {
  if (!ctx || !theta || !l_inv || !weights || !w_sum || !w_logq || !d_mu || !gram)
    return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  return mvt_dis_grad(ctx, n, d, df, theta, l_inv, weights, w_sum, w_logq, d_mu, gram);
}

int vb_elbo_sums_mvt(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, const double* mu,
                     const double* sqrt_sigma, const double* inv_s, double* f_sum, double* g_sum, double* c_full) {
  if (!ctx || !mu || !sqrt_sigma || !inv_s || !f_sum || !g_sum || !c_full)
    return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(check_slot(ctx, slot));
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (!ctx->noise[slot].buf.ptr) return fail(ctx, VB_ERR_STATE, "noise slot %d is empty", slot);
  if (n_total < n) return fail(ctx, VB_ERR_INVALID, "n_total must be >= n");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  return mvt_elbo_sums(ctx, ctx->noise[slot], n, d, n_total, mu, sqrt_sigma, inv_s, f_sum, g_sum, c_full);
}

// ---- ExclusiveKL, full-rank Gaussian ---------------------------------------------------------------
int vb_fullrank_set_theta(vb_ctx* ctx, const double* theta, int64_t d) {
  if (!ctx || !theta || d <= 0) return fail(ctx, VB_ERR_INVALID, "bad argument");
  const int64_t p = d + d * (d + 1) / 2;
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(ensure(ctx, ctx->fr_theta, (size_t)p * sizeof(double)));
  VB_TRY(ensure(ctx, ctx->fr_out, (size_t)(1 + p) * sizeof(double)));
  VB_TRY(main_stream_write(ctx));   // epilogues still in flight on `post` read the old parameter
  VB_TRY(push_small(ctx, ctx->stream, theta, (size_t)p * sizeof(double), ctx->fr_theta.ptr));   // caller keeps `theta`
  ctx->fr_p = p;
  ctx->fr_lt_d = 0;                                 // the unpacked copy (mu, L') is stale
  return VB_OK;
}

int vb_elbo_grad_fullrank_enqueue(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total,
                                  unsigned flags) {
  if (!ctx) return VB_ERR_INVALID;
  VB_TRY(check_slot(ctx, slot));
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (!ctx->noise[slot].buf.ptr) return fail(ctx, VB_ERR_STATE, "noise slot %d is empty", slot);
  if (ctx->fr_p != d + d * (d + 1) / 2)
    return fail(ctx, VB_ERR_STATE, "no resident parameter of dimension %lld (vb_fullrank_set_theta)",
                (long long)d);
  if (n_total < n) return fail(ctx, VB_ERR_INVALID, "n_total must be >= n");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  ctx->fr_busy = true;      // (asynchronous: the next parameter upload must stay behind this evaluation's reads)
  return fr_elbo_grad_enqueue(ctx, ctx->noise[slot], n, d, n_total, (const double*)ctx->fr_theta.ptr,
                              (double*)ctx->fr_out.ptr, flags);
}

int vb_fullrank_get(vb_ctx* ctx, double* value, double* grad, int64_t p) {
  if (!ctx || !value || !grad) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (p != ctx->fr_p || !ctx->fr_out.ptr) return fail(ctx, VB_ERR_STATE, "no full-rank result of length %lld", (long long)p);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));   // a sharded evaluation finishes on `post`
  unsigned fz_pair[2] = {0, 0};       // fused evaluation (vb_fullrank_fused.h): [1] = a dependency poll that gave up
  const FetchSeg segs[3] = {{ctx->fr_out.ptr, sizeof(double), value},
                            {(const double*)ctx->fr_out.ptr + 1, (size_t)p * sizeof(double), grad},
                            {ctx->fz_words.ptr, sizeof fz_pair, fz_pair}};
  VB_TRY(fetch_blocking(ctx, ctx->stream, segs, ctx->fz_words.ptr ? 3 : 2));      // (plain copies above 1 MB; comm_check inside)
  const unsigned fz_err = fz_pair[1];
  if (fz_err) {
    VB_HIP(ctx, hipMemsetAsync(ctx->fz_words.ptr, 0, 2 * sizeof(unsigned), ctx->stream));
    return fail(ctx, VB_ERR_STATE, "fused full-rank evaluation: a tile gave up waiting for its input (results invalid)");
  }
  return VB_OK;
}

int vb_elbo_grad_fullrank(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total,
                          const double* theta, unsigned flags, double* value, double* grad) {
  if (!ctx || !theta || d <= 0) return fail(ctx, VB_ERR_INVALID, "bad argument");
  const int64_t p = d + d * (d + 1) / 2;
  // Round 6, built and MEASURED SLOWER -- off unless VB_FR_UPLOAD_PIPE=1: a parameter too large for the
  // staged small-copy path (4.2 MB at D = 1024) crosses PCIe in row chunks, heaviest rows of L first, and the sampling
  // product of a chunk's column blocks starts behind its copy while the rest is in flight (fr_upload_begin); the noise
  // generated for this call (already queued on the main stream) runs beside the first chunk.  Results are bit-identical
  // (tests/test_gpu_fetch_routes.py).  tools/r6_api_call_probe.py, D = 1024 / N = 4096, blocking call with a pinned
  // gradient array: one copy + synchronisation in front 539 us; this route with 1 / 2 / 3 / 4 chunks 561 / 612 / 636 /
  // 708 us from a pageable parameter and 556 / 572 / 586 / 605 us from a pinned one.  Every hipMemcpyAsync costs 15-35 us
  // whatever its size (a pageable source is pinned and unpinned per call), every cross-stream event wait ~10 us: the ~70 us
  // of overlap the chunks buy are spent twice over.  What did pay: the gradient lands in page-locked memory (vb_host_alloc,
  // PinnedPool in the Python binding): 549 -> 539 us.
  const char* pe = getenv("VB_FR_UPLOAD_PIPE");      // (read per call: the tests switch it)
  const bool pipe_env = pe && atoi(pe) != 0;
  const bool pipelined = pipe_env && !ctx->comm && (size_t)p * sizeof(double) > kFetchMaxBytes && d % 16 == 0 &&
                         !(flags & VB_FLAG_PATH_DERIV);
  if (pipelined) {
    VB_HIP(ctx, hipSetDevice(ctx->device));
    VB_TRY(ensure(ctx, ctx->fr_theta, (size_t)p * sizeof(double)));
    VB_TRY(ensure(ctx, ctx->fr_out, (size_t)(1 + p) * sizeof(double)));
    VB_TRY(main_stream_write(ctx));
    ctx->fr_p = p;
    VB_TRY(fr_upload_begin(ctx, theta, d));
    ++ctx->fr_up_calls;
    const int rc = vb_elbo_grad_fullrank_enqueue(ctx, slot, n, d, n_total, flags);
    ctx->fr_busy = false;
    if (rc != VB_OK) {
      ctx->fr_up_active = false;
      (void)hipStreamSynchronize(ctx->up_stream);      // the caller's array must not be read after we return
      return rc;
    }
    return vb_fullrank_get(ctx, value, grad, p);
  }
  VB_TRY(vb_fullrank_set_theta(ctx, theta, d));
  VB_TRY(vb_elbo_grad_fullrank_enqueue(ctx, slot, n, d, n_total, flags));
  ctx->fr_busy = false;
  return vb_fullrank_get(ctx, value, grad, d + d * (d + 1) / 2);
}

int vb_elbo_grad_mvt_chol(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df, const double* theta,
                          double* value, double* grad) {
  if (!ctx || !theta || !value || !grad) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(check_slot(ctx, slot));
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (!ctx->noise[slot].buf.ptr) return fail(ctx, VB_ERR_STATE, "noise slot %d is empty", slot);
  if (n_total < n) return fail(ctx, VB_ERR_INVALID, "n_total must be >= n");
  VB_TRY(vb_fullrank_set_theta(ctx, theta, d));
  VB_TRY(mvt_elbo_chol_enqueue(ctx, ctx->noise[slot], n, d, n_total, df, (const double*)ctx->fr_theta.ptr,
                               (double*)ctx->fr_out.ptr));
  return vb_fullrank_get(ctx, value, grad, d + d * (d + 1) / 2);
}

// ---- symmetric square root (approximations.py:348) -------------------------------------------------------
int vb_sym_sqrt(vb_ctx* ctx, const double* a, const double* e, int64_t d, double* root, double* x, double* info) {
  if (!ctx || !a || !root) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (d <= 0 || d > 8192) return fail(ctx, VB_ERR_INVALID, "matrix dimension %lld outside [1, 8192]", (long long)d);
  if ((e == nullptr) != (x == nullptr)) return fail(ctx, VB_ERR_INVALID, "e and x go together");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  return sym_sqrt(ctx, a, e, d, root, x, info);
}

int vb_sym_sqrt_inv(vb_ctx* ctx, const double* a, const double* e, int64_t d, double* root, double* x, double* info,
                    double* inv_root) {
  if (!ctx || !a || !root) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (d <= 0 || d > 8192) return fail(ctx, VB_ERR_INVALID, "matrix dimension %lld outside [1, 8192]", (long long)d);
  if ((e == nullptr) != (x == nullptr)) return fail(ctx, VB_ERR_INVALID, "e and x go together");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  return sym_sqrt(ctx, a, e, d, root, x, info, inv_root);
}

int vb_lowrank_path_terms(vb_ctx* ctx, int slot_eps, int slot_z, int64_t n, int64_t d, int64_t k, int64_t n_total,
                          const double* sw, double* out) {
  if (!ctx || !sw || !out) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(check_slot(ctx, slot_eps));
  VB_TRY(check_slot(ctx, slot_z));
  if (!ctx->noise[slot_eps].buf.ptr || !ctx->noise[slot_z].buf.ptr)
    return fail(ctx, VB_ERR_STATE, "noise slots %d / %d are empty", slot_eps, slot_z);
  if (n_total < n) return fail(ctx, VB_ERR_INVALID, "n_total must be >= n");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  return lr_path_terms(ctx, ctx->noise[slot_eps], ctx->noise[slot_z], n, d, k, sw, out);
}

int vb_alpha_sums_mvt(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df, double alpha,
                      const double* mu, const double* sqrt_sigma, const double* inv_s, double sum_log_diag,
                      double* value, double* w_sum, double* g_sum, double* c_full) {
  if (!ctx || !mu || !sqrt_sigma || !value || !w_sum || !g_sum || !c_full)      // (inv_s may be NULL: device draws)
    return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(check_slot(ctx, slot));
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (!ctx->noise[slot].buf.ptr) return fail(ctx, VB_ERR_STATE, "noise slot %d is empty", slot);
  if (n_total < n) return fail(ctx, VB_ERR_INVALID, "n_total must be >= n");
  if (!(df > 0.0)) return fail(ctx, VB_ERR_INVALID, "df must be positive");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  return mvt_alpha_sums(ctx, ctx->noise[slot], n, d, n_total, df, alpha, mu, sqrt_sigma, inv_s, sum_log_diag, value,
                        w_sum, g_sum, c_full);
}

int vb_mvt_path_terms(vb_ctx* ctx, int slot, int64_t n, int64_t d, int64_t n_total, double df, const double* inv_s,
                      double* m_w, double* e_w, double* log1p_sum) {
  if (!ctx || !inv_s || !m_w || !e_w || !log1p_sum) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  VB_TRY(check_slot(ctx, slot));
  if (!ctx->noise[slot].buf.ptr) return fail(ctx, VB_ERR_STATE, "noise slot %d is empty", slot);
  if (!(df > 2.0)) return fail(ctx, VB_ERR_INVALID, "df must be greater than 2");
  if (n_total < n) return fail(ctx, VB_ERR_INVALID, "n_total must be >= n");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  return mvt_path_terms(ctx, ctx->noise[slot], n, d, n_total, df, inv_s, m_w, e_w, log1p_sum);
}

// ---- device-resident fit (optimization.py:83-127) ----------------------------------------------------
// The rows vb_fit logs per iteration (iterate, descent direction, gradient: optimization.py:83-127 returns every iterate,
// :541 FASO's gradient history) used to leave in one pageable copy after the last step: at p = 525 312 (D = 1024 dense)
// 4.2 MB per row at ~13 GB/s, as long again as the iteration that produced it.  With long rows each iteration's rows go,
// behind an event, through a copy stream into a ring of pinned slots while the following iterations run; the enqueuing
// thread, which is R iterations ahead of the GPU at most, moves a slot to the caller's arrays before it reuses it.
namespace {
struct FitRowStream {
  vb_ctx* ctx = nullptr;
  bool on = false;
  int64_t p = 0, n_iters = 0, hist_first = 0;
  const double *d_hist = nullptr, *d_dirs = nullptr, *d_grads = nullptr;     // device rows
  double *h_hist = nullptr, *h_dirs = nullptr, *h_grads = nullptr;           // caller's arrays
  int64_t drained = 0;          // iterations whose rows have reached the caller

  static size_t min_row_bytes() {
    const char* e = getenv("VB_FIT_STREAM_MIN_BYTES");
    return e ? (size_t)atoll(e) : (size_t)1 << 18;
  }
  int begin() {
    const char* e = getenv("VB_FIT_STREAM_ROWS");
    on = (h_hist || h_dirs || h_grads) && !(e && atoi(e) == 0) && (size_t)p * sizeof(double) >= min_row_bytes();
    if (!on) return VB_OK;
    if (!ctx->fit_copy_st) VB_HIP(ctx, hipStreamCreateWithFlags(&ctx->fit_copy_st, hipStreamNonBlocking));
    const size_t slot = (size_t)round_up(3 * p, 16);
    if (ctx->fit_ring_doubles < slot) {
      VB_HIP(ctx, hipStreamSynchronize(ctx->fit_copy_st));
      if (ctx->fit_ring) VB_HIP(ctx, hipHostFree(ctx->fit_ring));
      ctx->fit_ring = nullptr;
      ctx->fit_ring_doubles = 0;
      VB_HIP(ctx, hipHostMalloc((void**)&ctx->fit_ring, vb_ctx::kFitRing * slot * sizeof(double), hipHostMallocDefault));
      ctx->fit_ring_doubles = slot;
    }
    for (int i = 0; i < vb_ctx::kFitRing; ++i) {
      if (!ctx->fit_ev_step[i]) VB_HIP(ctx, hipEventCreateWithFlags(&ctx->fit_ev_step[i], hipEventDisableTiming));
      if (!ctx->fit_ev_copy[i]) VB_HIP(ctx, hipEventCreateWithFlags(&ctx->fit_ev_copy[i], hipEventDisableTiming));
    }
    return VB_OK;
  }
  int drain_one() {           // iteration `drained`: wait for its copies, hand the rows over
    const int64_t k = drained;
    const int slot = (int)(k % vb_ctx::kFitRing);
    VB_HIP(ctx, hipEventSynchronize(ctx->fit_ev_copy[slot]));
    const double* src = ctx->fit_ring + (size_t)slot * ctx->fit_ring_doubles;
    const size_t row = (size_t)p * sizeof(double);
    if (h_hist && k >= hist_first) memcpy(h_hist + (k - hist_first) * p, src, row);
    if (h_dirs) memcpy(h_dirs + k * p, src + p, row);
    if (h_grads) memcpy(h_grads + k * p, src + 2 * p, row);
    ++drained;
    return VB_OK;
  }
  int after_step(int64_t k) {      // iteration k's kernels (its step included) are enqueued on the main stream
    if (!on) return VB_OK;
    if (k >= vb_ctx::kFitRing) VB_TRY(drain_one());      // the slot's previous tenant: iteration k - R
    const int slot = (int)(k % vb_ctx::kFitRing);
    hipStream_t cs = ctx->fit_copy_st;
    VB_HIP(ctx, hipEventRecord(ctx->fit_ev_step[slot], ctx->stream));
    VB_HIP(ctx, hipStreamWaitEvent(cs, ctx->fit_ev_step[slot], 0));
    double* dst = ctx->fit_ring + (size_t)slot * ctx->fit_ring_doubles;
    const size_t row = (size_t)p * sizeof(double);
    if (h_hist && k >= hist_first)
      VB_HIP(ctx, hipMemcpyAsync(dst, d_hist + (k - hist_first) * p, row, hipMemcpyDeviceToHost, cs));
    if (h_dirs) VB_HIP(ctx, hipMemcpyAsync(dst + p, d_dirs + k * p, row, hipMemcpyDeviceToHost, cs));
    if (h_grads) VB_HIP(ctx, hipMemcpyAsync(dst + 2 * p, d_grads + k * p, row, hipMemcpyDeviceToHost, cs));
    VB_HIP(ctx, hipEventRecord(ctx->fit_ev_copy[slot], cs));
    return VB_OK;
  }
  int finish() {
    if (!on) return VB_OK;
    while (drained < n_iters) VB_TRY(drain_one());
    return VB_OK;
  }
};
}  // namespace

int vb_fit(vb_ctx* ctx, int slot, int slot_aux, int64_t n, int64_t d, int64_t n_total, int64_t row_offset, int family,
           double df, unsigned flags, int cv_mode, int noise_kind, double noise_df, uint64_t seed,
           uint64_t first_stream, int opt_kind, const double hyper[4], int64_t n_iters, double* theta, int64_t p,
           double* state, int has_state, double* values, double* history, int64_t hist_len, double* directions,
           double* gradients) {
  if (!ctx || !hyper || !theta || !values) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (ctx->model.id < 0) return fail(ctx, VB_ERR_STATE, "no model bound (vb_set_model)");
  if (n <= 0 || d <= 0 || n_iters <= 0) return fail(ctx, VB_ERR_INVALID, "n, d and n_iters must be positive");
  if (n_total < n) return fail(ctx, VB_ERR_INVALID, "n_total must be >= n");
  if (opt_kind < VB_OPT_SGD || opt_kind > VB_OPT_ADAGRAD)
    return fail(ctx, VB_ERR_INVALID, "unknown optimiser kind %d", opt_kind);
  if (hist_len < 0 || hist_len > n_iters || (hist_len > 0 && !history))
    return fail(ctx, VB_ERR_INVALID, "hist_len must be in [0, n_iters] with a history buffer");
  if (has_state && !state) return fail(ctx, VB_ERR_INVALID, "has_state set without a state buffer");
  const bool meanfield = family == VB_FAMILY_MF_GAUSSIAN || family == VB_FAMILY_MF_STUDENT_T;
  const bool fullrank = family == VB_FAMILY_FULLRANK_GAUSSIAN;
  const bool lowrank = family == VB_FAMILY_LOWRANK_GAUSSIAN;
  if (!meanfield && !fullrank && !lowrank)
    return fail(ctx, VB_ERR_UNSUPPORTED, "device-resident fit: family %d is not supported", family);
  const int64_t lr_k = lowrank ? (p - 2 * d) / d : 0;
  if (lowrank) {
    if (lr_k < 1 || lr_k > 16 || p != 2 * d + d * lr_k)
      return fail(ctx, VB_ERR_INVALID, "low-rank family: parameter length %lld is not 2 d + d k with 1 <= k <= 16",
                  (long long)p);
    if (cv_mode != VB_CV_NONE || (flags & VB_FLAG_PATH_DERIV))
      return fail(ctx, VB_ERR_UNSUPPORTED, "low-rank family: entropy-form estimator only");
    if (slot_aux == slot) return fail(ctx, VB_ERR_INVALID, "the two noise blocks need different slots");
  } else if (p != (meanfield ? 2 * d : d + d * (d + 1) / 2)) {
    return fail(ctx, VB_ERR_INVALID, "parameter length %lld does not match the family", (long long)p);
  }
  if (fullrank && cv_mode != VB_CV_NONE)
    return fail(ctx, VB_ERR_UNSUPPORTED, "full-rank family: the RGE control variates do not apply");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  VB_TRY(noise_alloc(ctx, slot, n, d));
  NoiseSlot& ns = ctx->noise[slot];
  if (lowrank) VB_TRY(noise_alloc(ctx, slot_aux, n, lr_k));

  // device state: [theta (p) | out (1 + p) | s1 (p) | s2 (p) | values (n_iters) | iterates (hist_len x p)]
  int64_t off = 0;
  auto carve = [&off](int64_t doubles) {
    const int64_t o = off;
    off += round_up(doubles, 16);
    return o;
  };
  const int64_t o_theta = carve(p), o_out = carve(1 + p), o_s1 = carve(p), o_s2 = carve(p),
                o_val = carve(n_iters), o_hist = carve(hist_len * p), o_dirs = carve(directions ? n_iters * p : 0),
                o_grads = carve(gradients ? n_iters * p : 0);
  ctx->fit_hist_len = 0;      // (the kept iterates of an earlier fit are about to be overwritten)
  VB_TRY(ensure(ctx, ctx->fit_work, (size_t)off * sizeof(double)));
  double* base = (double*)ctx->fit_work.ptr;
  double* theta_dev = base + o_theta;
  double* out_dev = base + o_out;
  hipStream_t st = ctx->stream;
  VB_HIP(ctx, hipMemcpyAsync(theta_dev, theta, (size_t)p * sizeof(double), hipMemcpyHostToDevice, st));
  if (has_state) {
    VB_HIP(ctx, hipMemcpyAsync(base + o_s1, state, (size_t)p * sizeof(double), hipMemcpyHostToDevice, st));
    VB_HIP(ctx, hipMemcpyAsync(base + o_s2, state + p, (size_t)p * sizeof(double), hipMemcpyHostToDevice, st));
  } else {   // the part of the state an optimiser does not use is returned as zeros, not as stale workspace
    VB_HIP(ctx, hipMemsetAsync(base + o_s1, 0, (size_t)(o_val - o_s1) * sizeof(double), st));
  }
  VB_HIP(ctx, hipStreamSynchronize(st));   // the caller's buffers are pageable: copies above are staged

  FitStep step;
  step.kind = opt_kind;
  step.p = p;
  step.lr = hyper[0];
  step.beta1 = hyper[1];
  step.one_minus_beta1 = 1.0 - hyper[1];
  step.beta2 = hyper[2];
  step.one_minus_beta2 = 1.0 - hyper[2];
  step.jitter = hyper[3];
  step.out = out_dev;
  step.theta = theta_dev;
  step.s1 = base + o_s1;
  step.s2 = base + o_s2;
  step.values = base + o_val;
  step.hist = hist_len > 0 ? base + o_hist : nullptr;
  step.hist_first = n_iters - hist_len;
  step.dirs = directions ? base + o_dirs : nullptr;
  step.grads = gradients ? base + o_grads : nullptr;
  FitRowStream rows;
  rows.ctx = ctx, rows.p = p, rows.n_iters = n_iters, rows.hist_first = n_iters - hist_len;
  rows.d_hist = step.hist, rows.d_dirs = step.dirs, rows.d_grads = step.grads;
  rows.h_hist = hist_len > 0 ? history : nullptr, rows.h_dirs = directions, rows.h_grads = gradients;
  VB_TRY(rows.begin());

  MfCall c;
  if (meanfield) {
    c.count = 1;
    c.noise[0] = &ns;
    c.theta_src[0] = theta_dev;
    c.theta_on_device = true;
    c.out[0] = out_dev;
    c.n = n;
    c.d = d;
    c.n_total = n_total;
    c.family = family;
    c.df = df;
    c.flags = flags;
    c.cv_mode = cv_mode;
  }
  static const bool gen_env = !(getenv("VB_FIT_GEN") && atoi(getenv("VB_FIT_GEN")) == 0);
  const bool gen_in_kernel =
      gen_env && (ctx->model.id == VB_MODEL_GAUSS_DIAG || ctx->model.id == VB_MODEL_FUNNEL) &&
      ((family == VB_FAMILY_MF_GAUSSIAN && noise_kind == VB_NOISE_NORMAL) ||
       (family == VB_FAMILY_MF_STUDENT_T && noise_kind == VB_NOISE_STUDENT_T && noise_df == df));
  bool step_done = false, prep_done = false;
  // the dense family's fused step leaves mu / L' of the FIT's iterate in fr_lt (fr_step_unpack_enqueue): on every way out
  // of this function -- error returns included -- the copy is declared stale for the resident parameter too, or a later
  // set_theta-once / enqueue-many caller with the same d would be evaluated at the fit's parameter
  struct LtReset {
    vb_ctx* c;
    bool on;
    ~LtReset() {
      if (on) {
        c->fr_lt_owner = nullptr;
        c->fr_lt_d = 0;
      }
    }
  } lt_reset{ctx, fullrank};
  if (meanfield) {
    c.step = &step;
    c.step_done = &step_done;
    c.prep_done = &prep_done;
  }
  for (int64_t k = 0; k < n_iters; ++k) {
    step.k = k;
    step.first = (k == 0 && !has_state) ? 1 : 0;
    step_done = false;
    if (lowrank) {
      NoiseSlot& nz = ctx->noise[slot_aux];
      const uint64_t s2 = 2 * (first_stream + (uint64_t)k);
      VB_TRY(rng_fill(ctx, (double*)ns.buf.ptr, ns.ld, VB_NOISE_NORMAL, 0.0, seed, s2, row_offset, n, d));
      VB_TRY(rng_fill(ctx, (double*)nz.buf.ptr, nz.ld, VB_NOISE_NORMAL, 0.0, seed, s2 + 1, row_offset, n, lr_k));
      VB_TRY(lr_elbo_grad_enqueue(ctx, ns, nz, n, d, lr_k, n_total, theta_dev, out_dev));
    } else if (gen_in_kernel) {
      // single-use Gaussian noise never touches HBM: the streaming kernel generates it in registers
      c.skip_prep = prep_done;                 // done by the previous iteration's finalize kernel
      c.prep_next = k + 1 < n_iters;
      prep_done = false;
      c.gen = 1;
      c.gen_seed = seed;
      c.gen_stream = first_stream + (uint64_t)k;
      c.gen_row_offset = row_offset;
      VB_TRY(mf_enqueue(ctx, c));
    } else {
      VB_TRY(rng_fill(ctx, (double*)ns.buf.ptr, ns.ld, noise_kind, noise_df, seed, first_stream + (uint64_t)k,
                      row_offset, n, d));
      if (meanfield)
        VB_TRY(mf_enqueue(ctx, c));
      else
        VB_TRY(fr_elbo_grad_enqueue(ctx, ns, n, d, n_total, theta_dev, out_dev, flags));
    }
    if (fullrank && ctx->pipe.post_pending) {   // sharded full-rank evaluations finish on the communication stream
      VB_HIP(ctx, hipStreamWaitEvent(st, ctx->pipe.ev_fin[ctx->pipe.last_set], 0));
      ctx->pipe.post_pending = false;
    }
    static const bool fuse_env = !(getenv("VB_FIT_STEP_UNPACK") && atoi(getenv("VB_FIT_STEP_UNPACK")) == 0);
    if (fullrank && fuse_env && !ctx->comm) {
      // dense family: the step writes mu and L' of the stepped parameter itself, the next evaluation skips its unpack
      VB_TRY(fr_step_unpack_enqueue(ctx, step, d));
      step_done = true;
    }
    if (!step_done) VB_TRY(fit_step_enqueue(ctx, step));
    VB_TRY(rows.after_step(k));
  }
  VB_HIP(ctx, hipMemcpyAsync(theta, theta_dev, (size_t)p * sizeof(double), hipMemcpyDeviceToHost, st));
  VB_HIP(ctx, hipMemcpyAsync(values, base + o_val, (size_t)n_iters * sizeof(double), hipMemcpyDeviceToHost, st));
  if (state) {
    VB_HIP(ctx, hipMemcpyAsync(state, base + o_s1, (size_t)p * sizeof(double), hipMemcpyDeviceToHost, st));
    VB_HIP(ctx, hipMemcpyAsync(state + p, base + o_s2, (size_t)p * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  if (hist_len > 0 && !rows.on)
    VB_HIP(ctx, hipMemcpyAsync(history, base + o_hist, (size_t)(hist_len * p) * sizeof(double),
                               hipMemcpyDeviceToHost, st));
  if (directions && !rows.on)
    VB_HIP(ctx, hipMemcpyAsync(directions, base + o_dirs, (size_t)(n_iters * p) * sizeof(double),
                               hipMemcpyDeviceToHost, st));
  if (gradients && !rows.on)
    VB_HIP(ctx, hipMemcpyAsync(gradients, base + o_grads, (size_t)(n_iters * p) * sizeof(double),
                               hipMemcpyDeviceToHost, st));
  VB_TRY(rows.finish());
  VB_HIP(ctx, hipStreamSynchronize(st));
  ctx->fit_hist_off = o_hist, ctx->fit_hist_len = hist_len, ctx->fit_hist_p = p, ctx->fit_out_off = o_out;      // (vb_fit_history_mean)
  return VB_OK;
}

namespace {
// out[j] = (h[0][j] + h[1][j] + ... in row order) / rows: numpy's add.reduce over the leading axis followed by true_divide
__global__ void __launch_bounds__(256) fit_history_mean_kernel(const double* __restrict__ h, int64_t rows, int64_t p,
                                                               double* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= p) return;
  double s = h[j];
  for (int64_t r0 = 1; r0 < rows; r0 += 8) {      // eight rows' loads in flight, added in row order
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = h[(r0 + u < rows ? r0 + u : rows - 1) * p + j];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (r0 + u < rows) s += v[u];
  }
  out[j] = s / (double)rows;
}
}  // namespace

int vb_mvt_route_stats(vb_ctx* ctx, uint64_t* epilogue_rows, uint64_t* chain_fetch) {
  if (!ctx || !epilogue_rows || !chain_fetch) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  *epilogue_rows = ctx->mvt_epi_rows_calls;
  *chain_fetch = ctx->mvt_chain_fetch_calls;
  return VB_OK;
}

int vb_fit_history_mean(vb_ctx* ctx, int64_t rows, int64_t p, double* mean) {
  if (!ctx || !mean || rows <= 0 || p <= 0) return fail(ctx, VB_ERR_INVALID, "bad argument");
  if (!ctx->fit_work.ptr || ctx->fit_hist_len < rows || ctx->fit_hist_p != p)
    return fail(ctx, VB_ERR_STATE, "no resident iterate history of %lld rows x %lld (the last fit kept %lld x %lld)", (long long)rows,
                (long long)p, (long long)ctx->fit_hist_len, (long long)ctx->fit_hist_p);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(main_stream_write(ctx));
  const double* h = (const double*)ctx->fit_work.ptr + ctx->fit_hist_off + (ctx->fit_hist_len - rows) * p;
  double* out = (double*)ctx->fit_work.ptr + ctx->fit_out_off;      // (the fit's own [value | gradient] area: free once it has returned)
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(fit_history_mean_kernel, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, st, h, rows, p, out);
  VB_HIP(ctx, hipGetLastError());
  const FetchSeg seg[1] = {{out, (size_t)p * sizeof(double), mean}};
  return fetch_blocking(ctx, st, seg, 1);
}

// ---- measurement --------------------------------------------------------------------------------
int vb_profile_enable(vb_ctx* ctx, int on) {
  if (!ctx) return VB_ERR_INVALID;
  ctx->profile = on != 0;
  return VB_OK;
}

int vb_profile_read_kernel(vb_ctx* ctx, int kernel_id, int64_t* launches, int64_t* evals, double* total_ms,
                           int reset) {
  if (!ctx || !launches || !total_ms) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (kernel_id < 0 || kernel_id >= VB_PROF_NUM)
    return fail(ctx, VB_ERR_INVALID, "profile kernel id %d out of range [0, %d)", kernel_id, VB_PROF_NUM);
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(sync_streams(ctx));
  vb_ctx::ProfLog& log = ctx->prof[kernel_id];
  double ms = 0.0;
  for (size_t i = 0; i < log.used; ++i) {
    float t = 0.f;
    VB_HIP(ctx, hipEventElapsedTime(&t, log.events[i].first, log.events[i].second));
    ms += t;
  }
  *launches = (int64_t)log.used;
  *total_ms = ms;
  if (evals) *evals = log.evals;
  if (reset) {
    log.used = 0;
    log.evals = 0;
  }
  return VB_OK;
}

int vb_profile_read(vb_ctx* ctx, int64_t* launches, int64_t* evals, double* total_ms, int reset) {
  return vb_profile_read_kernel(ctx, VB_PROF_MF_ACCUM, launches, evals, total_ms, reset);
}

}  // extern "C"

namespace vb {
void legacy_poll(vb_ctx* ctx) {
  if (ctx) ::legacy_spec_poll(ctx);
}
}  // namespace vb
