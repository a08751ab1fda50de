// Multi-GPU: the Monte-Carlo axis is sharded over ranks (one process per GPU); the only exchange
// is one sum all-reduce of the fp64 partial-sum vector per objective call, over RCCL / xGMI.
// The reference has no distributed code at all; this is the natural sharding of
// `mean_n phi(eps_n; theta)` (objectives.py:161, :216, :233, :255, :268).
#include "vb_common.h"

#include <rccl/rccl.h>

#include <algorithm>

namespace vb {

// host-staged transport (vb_comm_init_host): device -> pinned host, the caller's collective, host -> device; the
// stream is drained on both sides, so the call is ordered like the RCCL kernel it stands in for
static int host_collective(vb_ctx* ctx, hipStream_t stream, double* buf, size_t count, int op) {
  if (count == 0) return VB_OK;
  if (ctx->host_stage_cap < count) {
    if (ctx->host_stage) VB_HIP(ctx, hipHostFree(ctx->host_stage));
    ctx->host_stage = nullptr;
    ctx->host_stage_cap = 0;
    const size_t cap = count + count / 4 + 1024;
    VB_HIP(ctx, hipHostMalloc((void**)&ctx->host_stage, cap * sizeof(double), hipHostMallocDefault));
    ctx->host_stage_cap = cap;
  }
  VB_HIP(ctx, hipMemcpyAsync(ctx->host_stage, buf, count * sizeof(double), hipMemcpyDeviceToHost, stream));
  VB_HIP(ctx, hipStreamSynchronize(stream));
  const int rc = ctx->host_fn(ctx->host_user, ctx->host_stage, count, op);
  if (rc != 0) return fail(ctx, VB_ERR_COMM, "host collective (%s of %zu doubles) failed with code %d",
                           op == VB_HOST_MAX ? "max" : "sum", count, rc);
  VB_HIP(ctx, hipMemcpyAsync(buf, ctx->host_stage, count * sizeof(double), hipMemcpyHostToDevice, stream));
  VB_HIP(ctx, hipStreamSynchronize(stream));     // the staging buffer is free again when this returns
  return VB_OK;
}

// ---- xGMI-native transport (vb_comm_init_ipc) -------------------------------------------------------------------------
// All-reduce without a ring: every rank copies its vector into its own window, then reduces ITS 1/G slice by reading the
// G windows in RANK ORDER -- every element is summed once, by one rank, in a fixed order, so all ranks end with the same
// bits whatever the timing (a ring's order depends on the slice) -- and finally reads the reduced slices back.  On an
// 8-GPU xGMI node that is two passes of (G - 1) / G of the vector over seven point-to-point links in parallel (4.2 MB:
// ~2 x 3.7 MB / (7 x ~50 GB/s achievable) ~ 20 us plus three launches) against the ring's 2 (G - 1) dependent steps;
// UNMEASURED here (one GPU per box): tests/test_gpu_two_ranks.py runs it between two processes on one GPU.
//
// Synchronisation is on the device: window word 0 / 1 / 2 = sequence number of the last collective whose data /
// reduced slice / read-back this rank has finished (system-scope atomics behind a system-scope fence; the last block
// of a phase stores it).  A phase's blocks poll the peers' words before they touch peer memory:
//   publish  waits done >= seq - 1 on every peer (nobody still reads the previous collective out of my window)
//   reduce   waits data >= seq, gather waits reduced >= seq.          Polls are bounded; a give-up poisons the result with NaN.
constexpr int kIpcFlagDoubles = 16;      // 128 B in front of the data area

// The poll bound is WALL TIME (2^27 polls kept the spinning kernels resident and the host blocked for minutes
// behind a dead peer): `max_ticks` of the 100 MHz constant clock -- VB_IPC_TIMEOUT_S seconds, default 20 (a rank that
// compiles a source model or runs a host callable is late by seconds, not by that; slow peers opt in to more) -- read every
// 64 polls; `max_spins` (2^VB_IPC_POLL_LOG2, only when that variable is set) bounds the poll count as before.  A give-up is
// RECORDED: `err` is a word in pinned host memory that the host reads at the next collective, at every synchronising entry
// point and through vb_comm_check -- the poisoned result never passes for data.
__device__ __forceinline__ bool ipc_wait(const double* const* win, int n_ranks, int self, int word, unsigned long long want,
                                         unsigned max_spins, unsigned* err, unsigned long long max_ticks) {
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();
    for (int p = 0; p < n_ranks; ++p) {
      // (the own word too for the publish phase: a collective issued on another stream of this context must not
      // overwrite the window before this rank's previous read-back has finished -- the peers could otherwise move on
      // and rewrite the result areas it is still reading)
      if (p == self && word != 2) continue;
      const unsigned long long* f = reinterpret_cast<const unsigned long long*>(win[p]) + word;
      unsigned spins = 0;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > max_spins || ((spins & 63u) == 0 && wall_clock64() - t0 > max_ticks)) {
          ok = false;
          __hip_atomic_store(err, 1u + (unsigned)word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // system scope: the peers' data behind their flags
  }
  return __syncthreads_and((int)ok) != 0;
}

__device__ __forceinline__ void ipc_signal(double* own, int word, unsigned long long seq, unsigned* ticket, int phase) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // this block's stores, system scope
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = __hip_atomic_fetch_add(ticket + phase, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done == gridDim.x - 1) {                       // the phase's last block: every block's stores are released
      __hip_atomic_store(ticket + phase, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(own) + word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

struct IpcArgs {
  const double* win[16];
  int n_ranks, rank;
  size_t cap;
  unsigned long long seq;
  unsigned* ticket;
  unsigned* err;            // pinned host word: a poll gave up (1 + the flag word it waited for)
  unsigned max_spins;
  unsigned long long max_ticks;
};

__global__ void __launch_bounds__(256) ipc_publish_kernel(IpcArgs a, const double* __restrict__ buf, size_t count) {
  double* own = const_cast<double*>(a.win[a.rank]);
  const bool ok = ipc_wait(a.win, a.n_ranks, a.rank, 2, a.seq - 1, a.max_spins, a.err, a.max_ticks);
  double* data = own + kIpcFlagDoubles;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
    data[i] = ok ? buf[i] : NAN;
  ipc_signal(own, 0, a.seq, a.ticket, 0);
}

// slice of rank r: [r * per, min(count, (r + 1) * per)), per = ceil(count / G)
__global__ void __launch_bounds__(256) ipc_reduce_kernel(IpcArgs a, size_t count, int op) {
  double* own = const_cast<double*>(a.win[a.rank]);
  const bool ok = ipc_wait(a.win, a.n_ranks, a.rank, 0, a.seq, a.max_spins, a.err, a.max_ticks);
  const size_t per = (count + a.n_ranks - 1) / a.n_ranks, lo = (size_t)a.rank * per, hi = lo + per < count ? lo + per : count;
  double* res = own + kIpcFlagDoubles + a.cap;
  for (size_t i = lo + (size_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += (size_t)gridDim.x * 256) {
    double s = a.win[0][kIpcFlagDoubles + i];
    for (int p = 1; p < a.n_ranks; ++p) {              // rank order: the same sum on every run and for every timing
      const double v = a.win[p][kIpcFlagDoubles + i];
      s = op == VB_HOST_MAX ? fmax(s, v) : s + v;
    }
    res[i] = ok ? s : NAN;
  }
  ipc_signal(own, 1, a.seq, a.ticket, 1);
}

__global__ void __launch_bounds__(256) ipc_gather_kernel(IpcArgs a, double* __restrict__ buf, size_t count) {
  double* own = const_cast<double*>(a.win[a.rank]);
  const bool ok = ipc_wait(a.win, a.n_ranks, a.rank, 1, a.seq, a.max_spins, a.err, a.max_ticks);
  const size_t per = (count + a.n_ranks - 1) / a.n_ranks;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    const int owner = (int)(i / per);
    buf[i] = ok ? a.win[owner][kIpcFlagDoubles + a.cap + i] : NAN;
  }
  ipc_signal(own, 2, a.seq, a.ticket, 2);
}

// A poll of an earlier collective gave up: its result (and everything computed from it) is NaN by construction; say so.
int comm_check(vb_ctx* ctx) {
  vb_ctx::IpcComm& c = ctx->ipc;
  if (!c.on || !c.err_host || *(volatile unsigned*)c.err_host == 0) return VB_OK;
  const unsigned word = *(volatile unsigned*)c.err_host - 1;
  // where everybody stood: [data, reduced, done] sequence numbers of every rank's window against this rank's own count
  char where[512];
  int at = snprintf(where, sizeof where, "issued %llu;", (unsigned long long)c.seq);
  for (int p = 0; p < ctx->n_ranks && at < (int)sizeof where - 64; ++p) {
    unsigned long long f[3] = {0, 0, 0};
    if (c.win[p] && hipMemcpy(f, c.win[p], sizeof f, hipMemcpyDeviceToHost) == hipSuccess)
      at += snprintf(where + at, sizeof where - at, " rank %d [%llu %llu %llu]", p, f[0], f[1], f[2]);
  }
  return fail(ctx, VB_ERR_COMM, "IPC transport: a peer did not reach collective phase %u within the poll bound (%g s, "
                               "VB_IPC_TIMEOUT_S); results since then are invalid -- the communicator must be rebuilt (%s)",
              word, c.timeout_s, where);
}

static int ipc_collective(vb_ctx* ctx, hipStream_t stream, double* buf, size_t count, int op) {
  if (count == 0) return VB_OK;
  vb_ctx::IpcComm& c = ctx->ipc;
  VB_TRY(comm_check(ctx));
  if (count > c.cap)
    return fail(ctx, VB_ERR_COMM, "IPC window holds %zu doubles, the collective has %zu (vb_comm_ipc_window)", c.cap, count);
  IpcArgs a;
  for (int p = 0; p < 16; ++p) a.win[p] = c.win[p];
  a.n_ranks = ctx->n_ranks, a.rank = ctx->rank, a.cap = c.cap, a.seq = ++c.seq, a.ticket = c.ticket;
  a.err = c.err_dev, a.max_spins = c.poll_log2 >= 32 ? 0xffffffffu : (1u << c.poll_log2);
  a.max_ticks = (unsigned long long)(c.timeout_s * 1e8);      // wall_clock64: 100 MHz
  const unsigned blocks = (unsigned)std::min<size_t>(256, (count + 255) / 256);
  hipLaunchKernelGGL(ipc_publish_kernel, dim3(blocks), dim3(256), 0, stream, a, (const double*)buf, count);
  hipLaunchKernelGGL(ipc_reduce_kernel, dim3(blocks), dim3(256), 0, stream, a, count, op);
  hipLaunchKernelGGL(ipc_gather_kernel, dim3(blocks), dim3(256), 0, stream, a, buf, count);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

int comm_allreduce_sum(vb_ctx* ctx, hipStream_t stream, double* buf, size_t count) {
  if (!ctx->comm) return VB_OK;
  if (ctx->ipc.on) return ipc_collective(ctx, stream, buf, count, VB_HOST_SUM);
  if (ctx->host_fn) return host_collective(ctx, stream, buf, count, VB_HOST_SUM);
  ncclResult_t r = ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, (ncclComm_t)ctx->comm,
                                 stream);
  if (r != ncclSuccess)
    return fail(ctx, VB_ERR_COMM, "ncclAllReduce failed: %s", ncclGetErrorString(r));
  return VB_OK;
}

int comm_allreduce_max(vb_ctx* ctx, hipStream_t stream, double* buf, size_t count) {
  if (!ctx->comm) return VB_OK;
  if (ctx->ipc.on) return ipc_collective(ctx, stream, buf, count, VB_HOST_MAX);
  if (ctx->host_fn) return host_collective(ctx, stream, buf, count, VB_HOST_MAX);
  ncclResult_t r = ncclAllReduce(buf, buf, count, ncclDouble, ncclMax, (ncclComm_t)ctx->comm, stream);
  if (r != ncclSuccess)
    return fail(ctx, VB_ERR_COMM, "ncclAllReduce(max) failed: %s", ncclGetErrorString(r));
  return VB_OK;
}

// recv[r * count .. (r + 1) * count) = rank r's send[0 .. count); without a communicator: a copy
int comm_allgather(vb_ctx* ctx, hipStream_t stream, const double* send, double* recv, size_t count) {
  if (!ctx->comm) {
    if (send != recv)
      VB_HIP(ctx, hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, stream));
    return VB_OK;
  }
  if (ctx->host_fn || ctx->ipc.on) {    // zero everybody else's chunk, then a sum (x + 0 + ... + 0 is exact)
    const size_t r = (size_t)ctx->rank, g = (size_t)ctx->n_ranks;
    if (send != recv + r * count)
      VB_HIP(ctx, hipMemcpyAsync(recv + r * count, send, count * sizeof(double), hipMemcpyDeviceToDevice, stream));
    if (r > 0) VB_HIP(ctx, hipMemsetAsync(recv, 0, r * count * sizeof(double), stream));
    if (r + 1 < g) VB_HIP(ctx, hipMemsetAsync(recv + (r + 1) * count, 0, (g - r - 1) * count * sizeof(double), stream));
    if (ctx->ipc.on) return ipc_collective(ctx, stream, recv, g * count, VB_HOST_SUM);
    return host_collective(ctx, stream, recv, g * count, VB_HOST_SUM);
  }
  ncclResult_t r = ncclAllGather(send, recv, count, ncclDouble, (ncclComm_t)ctx->comm, stream);
  if (r != ncclSuccess)
    return fail(ctx, VB_ERR_COMM, "ncclAllGather failed: %s", ncclGetErrorString(r));
  return VB_OK;
}

int comm_shard_begin(vb_ctx* ctx, int64_t n, int64_t n_total, int64_t* begin) {
  const int64_t g = ctx->n_ranks, r = ctx->rank;
  const int64_t base = n_total / g, extra = n_total % g;
  *begin = r * base + (r < extra ? r : extra);
  const int64_t count = base + (r < extra ? 1 : 0);
  if (n != count)
    return fail(ctx, VB_ERR_INVALID, "rank %d of %d holds %lld of %lld samples, its shard is %lld (shard_rows)", (int)r,
                (int)g, (long long)n, (long long)n_total, (long long)count);
  return VB_OK;
}

int comm_gather_rows(vb_ctx* ctx, hipStream_t stream, double* vec, int64_t begin, int64_t n, int64_t n_total) {
  if (!ctx->comm) return VB_OK;
  if (n * (int64_t)ctx->n_ranks == n_total) return comm_allgather(ctx, stream, vec + begin, vec, (size_t)n);
  if (begin > 0) VB_HIP(ctx, hipMemsetAsync(vec, 0, (size_t)begin * sizeof(double), stream));
  if (begin + n < n_total)
    VB_HIP(ctx, hipMemsetAsync(vec + begin + n, 0, (size_t)(n_total - begin - n) * sizeof(double), stream));
  return comm_allreduce_sum(ctx, stream, vec, (size_t)n_total);
}

// The three per-sample vectors of a DIS refresh ([log p | log q | log prior] in one order or another: v0 < v1 < v2, `stride`
// doubles apart, each holding this rank's rows at [begin, begin + n)) in ONE collective (round 6; they were three): a kernel
// zeroes everything that is not this rank's block -- the other ranks' rows and the pads up to `stride` -- and one sum
// all-reduce over the 3 x stride doubles puts every rank's rows in place (x + 0 + ... + 0 is exact).  On RCCL an all-reduce
// of 3 N doubles (393 KB at BASELINE configs[3]) moves twice what three all-gathers would, but at this size each collective is its
// launch + handshake latency, not its bytes: two fewer of them per objective call.
__global__ void __launch_bounds__(256) gather3_zero_kernel(double* __restrict__ v, int64_t stride, int64_t begin, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= 3 * stride) return;
  const int64_t j = i % stride;
  if (j < begin || j >= begin + n) v[i] = 0.0;
}

int comm_gather_rows3(vb_ctx* ctx, hipStream_t stream, double* v0, double* v1, double* v2, int64_t begin, int64_t n,
                      int64_t n_total) {
  if (!ctx->comm) return VB_OK;
  const int64_t stride = v1 - v0;
  if (stride < n_total || v2 - v1 != stride) {      // (not one block: vector by vector)
    VB_TRY(comm_gather_rows(ctx, stream, v0, begin, n, n_total));
    VB_TRY(comm_gather_rows(ctx, stream, v1, begin, n, n_total));
    return comm_gather_rows(ctx, stream, v2, begin, n, n_total);
  }
  hipLaunchKernelGGL(gather3_zero_kernel, dim3((unsigned)((3 * stride + 255) / 256)), dim3(256), 0, stream, v0, stride, begin, n);
  VB_HIP(ctx, hipGetLastError());
  return comm_allreduce_sum(ctx, stream, v0, (size_t)(3 * stride));
}

}  // namespace vb

using namespace vb;

extern "C" {

int vb_comm_unique_id(char id[VB_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) <= VB_COMM_ID_BYTES, "ncclUniqueId larger than VB_COMM_ID_BYTES");
  if (!id) return fail(nullptr, VB_ERR_INVALID, "id is NULL");
  ncclUniqueId u;
  ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess)
    return fail(nullptr, VB_ERR_COMM, "ncclGetUniqueId failed: %s", ncclGetErrorString(r));
  memset(id, 0, VB_COMM_ID_BYTES);
  memcpy(id, &u, sizeof u);
  return VB_OK;
}

int vb_comm_init(vb_ctx* ctx, const char id[VB_COMM_ID_BYTES], int n_ranks, int rank) {
  if (!ctx || !id) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks)
    return fail(ctx, VB_ERR_INVALID, "rank %d / n_ranks %d invalid", rank, n_ranks);
  if (ctx->comm) return fail(ctx, VB_ERR_STATE, "communicator already attached");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  ncclComm_t c;
  ncclResult_t r = ncclCommInitRank(&c, n_ranks, u, rank);
  if (r != ncclSuccess)
    return fail(ctx, VB_ERR_COMM, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
  ctx->comm = c;
  ctx->n_ranks = n_ranks;
  ctx->rank = rank;
  return VB_OK;
}

int vb_comm_init_host(vb_ctx* ctx, vb_host_collective_fn fn, void* user, int n_ranks, int rank) {
  if (!ctx || !fn) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks)
    return fail(ctx, VB_ERR_INVALID, "rank %d / n_ranks %d invalid", rank, n_ranks);
  if (ctx->comm) return fail(ctx, VB_ERR_STATE, "communicator already attached");
  ctx->host_fn = fn;
  ctx->host_user = user;
  ctx->comm = ctx;
  ctx->n_ranks = n_ranks;
  ctx->rank = rank;
  return VB_OK;
}

int vb_comm_ipc_window(vb_ctx* ctx, size_t cap_doubles, char handle[VB_IPC_HANDLE_BYTES]) {
  static_assert(sizeof(hipIpcMemHandle_t) <= VB_IPC_HANDLE_BYTES, "hipIpcMemHandle_t larger than VB_IPC_HANDLE_BYTES");
  if (!ctx || !handle || cap_doubles == 0) return fail(ctx, VB_ERR_INVALID, "NULL argument or empty window");
  if (ctx->comm || ctx->ipc.win[0] || ctx->ipc.cap) return fail(ctx, VB_ERR_STATE, "communicator or window already present");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  const size_t bytes = (kIpcFlagDoubles + 2 * cap_doubles) * sizeof(double);
  double* w = nullptr;
  // The window -- flag words and data -- is polled and read by PEER GPUs while kernels run on both sides: fine-grained
  // (system-coherent) device memory, as RCCL allocates its own flag and buffer areas; coarse-grained hipMalloc memory is
  // only guaranteed visible across devices at kernel boundaries.  VB_IPC_COARSE=1 keeps the plain allocation (debugging).
  static const bool coarse = getenv("VB_IPC_COARSE") && atoi(getenv("VB_IPC_COARSE")) != 0;
  hipError_t e = coarse ? hipMalloc((void**)&w, bytes) : hipExtMallocWithFlags((void**)&w, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) return fail(ctx, VB_ERR_HIP, "IPC window (%zu bytes, %s): %s", bytes, coarse ? "coarse" : "fine-grained",
                                   hipGetErrorString(e));
  unsigned* ticket = nullptr;
  unsigned* err_host = nullptr;
  void* err_dev = nullptr;
  auto undo = [&]() {
    (void)hipFree(w);
    if (ticket) (void)hipFree(ticket);
    if (err_host) (void)hipHostFree(err_host);
  };
  hipIpcMemHandle_t h;
  if ((e = hipMemset(w, 0, bytes)) != hipSuccess || (e = hipMalloc((void**)&ticket, 4 * sizeof(unsigned))) != hipSuccess ||
      (e = hipMemset(ticket, 0, 4 * sizeof(unsigned))) != hipSuccess ||
      (e = hipHostMalloc((void**)&err_host, 64, hipHostMallocMapped)) != hipSuccess ||
      (e = hipHostGetDevicePointer(&err_dev, err_host, 0)) != hipSuccess ||
      (e = hipDeviceSynchronize()) != hipSuccess ||      // the zeroed flags are in place before any peer can map the window
      (e = hipIpcGetMemHandle(&h, w)) != hipSuccess) {
    undo();
    return fail(ctx, VB_ERR_HIP, "IPC window set-up: %s", hipGetErrorString(e));
  }
  memset(err_host, 0, 64);
  memset(handle, 0, VB_IPC_HANDLE_BYTES);
  memcpy(handle, &h, sizeof h);
  ctx->ipc.cap = cap_doubles;
  ctx->ipc.ticket = ticket;
  ctx->ipc.err_host = err_host;
  ctx->ipc.err_dev = (unsigned*)err_dev;
  const char* pl = getenv("VB_IPC_POLL_LOG2");      // (an explicit poll-count bound on top of the wall-time one)
  ctx->ipc.poll_log2 = pl ? std::max(8, std::min(32, atoi(pl))) : 32;
  const char* ts = getenv("VB_IPC_TIMEOUT_S");
  ctx->ipc.timeout_s = ts && atof(ts) > 0.0 ? atof(ts) : 20.0;
  ctx->ipc.win[15] = w;                // parked until vb_comm_init_ipc knows the rank
  return VB_OK;
}

int vb_comm_init_ipc(vb_ctx* ctx, const char* handles, int n_ranks, int rank) {
  if (!ctx || !handles) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  if (n_ranks < 1 || n_ranks > 15 || rank < 0 || rank >= n_ranks)
    return fail(ctx, VB_ERR_INVALID, "rank %d / n_ranks %d invalid (at most 15 ranks)", rank, n_ranks);
  if (ctx->comm) return fail(ctx, VB_ERR_STATE, "communicator already attached");
  if (!ctx->ipc.win[15]) return fail(ctx, VB_ERR_STATE, "no window (vb_comm_ipc_window first)");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  double* own = ctx->ipc.win[15];
  ctx->ipc.win[15] = nullptr;
  for (int p = 0; p < n_ranks; ++p) {
    if (p == rank) {
      ctx->ipc.win[p] = own;
      continue;
    }
    hipIpcMemHandle_t h;
    memcpy(&h, handles + (size_t)p * VB_IPC_HANDLE_BYTES, sizeof h);
    void* ptr = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      for (int q = 0; q < p; ++q) {      // close what was opened: a retry maps everything afresh
        if (q != rank && ctx->ipc.win[q]) (void)hipIpcCloseMemHandle(ctx->ipc.win[q]);
        ctx->ipc.win[q] = nullptr;
      }
      ctx->ipc.win[15] = own;
      return fail(ctx, VB_ERR_COMM, "hipIpcOpenMemHandle of rank %d's window failed: %s", p, hipGetErrorString(e));
    }
    ctx->ipc.win[p] = (double*)ptr;
  }
  ctx->ipc.on = true;
  ctx->ipc.seq = 0;
  ctx->comm = ctx;
  ctx->n_ranks = n_ranks;
  ctx->rank = rank;
  return VB_OK;
}

int vb_comm_info(vb_ctx* ctx, int* n_ranks, int* rank) {
  if (!ctx || !n_ranks || !rank) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  *n_ranks = 1;
  *rank = 0;
  if (!ctx->comm) return VB_OK;      // no communicator: a one-rank job
  if (ctx->host_fn || ctx->ipc.on) {
    *n_ranks = ctx->n_ranks;
    *rank = ctx->rank;
    return VB_OK;
  }
  ncclResult_t r = ncclCommCount((ncclComm_t)ctx->comm, n_ranks);
  if (r == ncclSuccess) r = ncclCommUserRank((ncclComm_t)ctx->comm, rank);
  if (r != ncclSuccess) return fail(ctx, VB_ERR_COMM, "ncclCommCount failed: %s", ncclGetErrorString(r));
  return VB_OK;
}

// The collective ALONE (bench.py --gpus N: `allreduce_us`): `reps` back-to-back sum all-reduces of `count` doubles on the
// context's stream between two HIP events, after `warm` untimed ones; every rank must call it with the same arguments.
int vb_comm_allreduce_time(vb_ctx* ctx, size_t count, int warm, int reps, double* us_per_collective) {
  if (!ctx || !us_per_collective || count == 0 || reps < 1 || warm < 0) return fail(ctx, VB_ERR_INVALID, "invalid argument");
  VB_HIP(ctx, hipSetDevice(ctx->device));
  VB_TRY(ensure(ctx, ctx->scratch, count * sizeof(double)));
  double* buf = (double*)ctx->scratch.ptr;
  hipStream_t st = ctx->stream;
  VB_HIP(ctx, hipMemsetAsync(buf, 0, count * sizeof(double), st));
  for (int i = 0; i < warm; ++i) VB_TRY(comm_allreduce_sum(ctx, st, buf, count));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float ms = 0.f;
  int rc = VB_OK;
  // (every exit destroys both events: ADVICE r5)
  hipError_t he = hipEventCreate(&e0);
  if (he == hipSuccess) he = hipEventCreate(&e1);
  if (he == hipSuccess) he = hipStreamSynchronize(st);
  if (he == hipSuccess) he = hipEventRecord(e0, st);
  for (int i = 0; he == hipSuccess && i < reps && rc == VB_OK; ++i) rc = comm_allreduce_sum(ctx, st, buf, count);
  if (he == hipSuccess) he = hipEventRecord(e1, st);
  if (he == hipSuccess) he = hipStreamSynchronize(st);
  if (he == hipSuccess && rc == VB_OK) he = hipEventElapsedTime(&ms, e0, e1);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (he != hipSuccess) return fail(ctx, VB_ERR_HIP, "vb_comm_allreduce_time: %s", hipGetErrorString(he));
  VB_TRY(rc);
  *us_per_collective = 1e3 * (double)ms / reps;
  return comm_check(ctx);
}

int vb_comm_check(vb_ctx* ctx) {
  if (!ctx) return VB_ERR_INVALID;
  return comm_check(ctx);
}

int vb_comm_destroy(vb_ctx* ctx) {
  if (!ctx) return VB_OK;
  if (!ctx->comm) {
    // a window that never became a communicator (vb_comm_ipc_window without vb_comm_init_ipc, or a failed init)
    if (ctx->ipc.win[15]) (void)hipFree(ctx->ipc.win[15]);
    if (ctx->ipc.ticket) (void)hipFree(ctx->ipc.ticket);
    if (ctx->ipc.err_host) (void)hipHostFree(ctx->ipc.err_host);
    ctx->ipc = vb_ctx::IpcComm();
    return VB_OK;
  }
  if (ctx->ipc.on) {
    (void)hipStreamSynchronize(ctx->stream);
    for (int p = 0; p < ctx->n_ranks; ++p) {
      if (!ctx->ipc.win[p]) continue;
      if (p == ctx->rank) (void)hipFree(ctx->ipc.win[p]);
      else (void)hipIpcCloseMemHandle(ctx->ipc.win[p]);
      ctx->ipc.win[p] = nullptr;
    }
    if (ctx->ipc.ticket) (void)hipFree(ctx->ipc.ticket);
    if (ctx->ipc.err_host) (void)hipHostFree(ctx->ipc.err_host);
    ctx->ipc = vb_ctx::IpcComm();
  } else if (ctx->host_fn) {
    ctx->host_fn = nullptr;
    ctx->host_user = nullptr;
    if (ctx->host_stage) (void)hipHostFree(ctx->host_stage);
    ctx->host_stage = nullptr;
    ctx->host_stage_cap = 0;
  } else {
    ncclCommDestroy((ncclComm_t)ctx->comm);
  }
  ctx->comm = nullptr;
  ctx->n_ranks = 1;
  ctx->rank = 0;
  return VB_OK;
}

}  // extern "C"
