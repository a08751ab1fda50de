// Multi-GPU: the Monte-Carlo axis is sharded over ranks (one process per GPU); the only exchange
// is one sum all-reduce of the fp64 partial-sum vector per objective call, over RCCL / xGMI.
// The reference has no distributed code at all; this is the natural sharding of
// `mean_n phi(eps_n; theta)` (objectives.py:161, :216, :233, :255, :268).
#include "vb_common.h"

#include <rccl/rccl.h>

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

int comm_allreduce_sum(vb_ctx* ctx, hipStream_t stream, double* buf, size_t count) {
  if (!ctx->comm) return VB_OK;
  if (ctx->host_fn) return host_collective(ctx, stream, buf, count, VB_HOST_SUM);
  ncclResult_t r = ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, (ncclComm_t)ctx->comm,
                                 stream);
  if (r != ncclSuccess)
    return fail(ctx, VB_ERR_COMM, "ncclAllReduce failed: %s", ncclGetErrorString(r));
  return VB_OK;
}

int comm_allreduce_max(vb_ctx* ctx, hipStream_t stream, double* buf, size_t count) {
  if (!ctx->comm) return VB_OK;
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
  if (ctx->host_fn) {    // zero everybody else's chunk, then a sum (x + 0 + ... + 0 is exact)
    const size_t r = (size_t)ctx->rank, g = (size_t)ctx->n_ranks;
    if (send != recv + r * count)
      VB_HIP(ctx, hipMemcpyAsync(recv + r * count, send, count * sizeof(double), hipMemcpyDeviceToDevice, stream));
    if (r > 0) VB_HIP(ctx, hipMemsetAsync(recv, 0, r * count * sizeof(double), stream));
    if (r + 1 < g) VB_HIP(ctx, hipMemsetAsync(recv + (r + 1) * count, 0, (g - r - 1) * count * sizeof(double), stream));
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

int vb_comm_info(vb_ctx* ctx, int* n_ranks, int* rank) {
  if (!ctx || !n_ranks || !rank) return fail(ctx, VB_ERR_INVALID, "NULL argument");
  *n_ranks = 1;
  *rank = 0;
  if (!ctx->comm) return VB_OK;      // no communicator: a one-rank job
  if (ctx->host_fn) {
    *n_ranks = ctx->n_ranks;
    *rank = ctx->rank;
    return VB_OK;
  }
  ncclResult_t r = ncclCommCount((ncclComm_t)ctx->comm, n_ranks);
  if (r == ncclSuccess) r = ncclCommUserRank((ncclComm_t)ctx->comm, rank);
  if (r != ncclSuccess) return fail(ctx, VB_ERR_COMM, "ncclCommCount failed: %s", ncclGetErrorString(r));
  return VB_OK;
}

int vb_comm_destroy(vb_ctx* ctx) {
  if (!ctx || !ctx->comm) return VB_OK;
  if (ctx->host_fn) {
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
