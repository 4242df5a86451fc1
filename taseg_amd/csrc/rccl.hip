// SyncBatchNorm with the statistics all-reduce issued by the library itself, on the compute stream.
//
// torch.distributed runs every collective on the process group's own stream: per all-reduce the compute stream
// records an event, the collective stream waits for it, and the compute stream waits for the collective again - two
// cross-stream hand-overs and ~30 us of c10d / Python dispatch for a 1 - 6 KB message, 126 times per training step
// (one per BatchNorm and direction).  Here RCCL is bound at run time (dlopen of the librccl.so torch already loaded;
// no link-time dependency), the library owns one communicator per SyncBatchNorm process group, and
//   ts_bn_sync_forward   = sliced sums -> ncclAllReduce([2C + 1] doubles) -> statistics -> elementwise pass
//   ts_bn_sync_backward  = sliced sums -> ncclAllReduce([2C] doubles)     -> elementwise pass
// are ONE host call each, every kernel and the collective in order on the caller's stream.
// Reference semantics: nn.SyncBatchNorm as every TASeg config uses it (IF_DIST: True; R/pcseg/model/segmentor/voxel/
// minkunet/minkunet.py:23-25), statistics over all voxels of all ranks.
#include <dlfcn.h>
#include <string.h>

#include "common.h"

namespace {
typedef struct {
  char internal[128];
} ts_nccl_id;                       // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void *ts_nccl_comm;
enum { TS_NCCL_SUM = 0, TS_NCCL_FLOAT64 = 8 };   // ncclSum, ncclFloat64 (rccl.h)

struct RcclApi {
  void *handle = nullptr;
  int (*GetUniqueId)(ts_nccl_id *) = nullptr;
  int (*CommInitRank)(ts_nccl_comm *, int, ts_nccl_id, int) = nullptr;
  int (*CommDestroy)(ts_nccl_comm) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, ts_nccl_comm, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
} g_rccl;

int rccl_fail(const char *what, int rc) {
  ts_set_error("%s: RCCL error %d (%s)", what, rc, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
  return TS_ERR_LAUNCH_FAILED;
}
}  // namespace

// Bind the RCCL entry points.  `path` = the librccl.so of the running PyTorch (already mapped into the process, so
// this returns the same image and the same RCCL state torch uses); NULL tries the default search path.
extern "C" int ts_rccl_load(const char *path) {
  if (g_rccl.handle) return TS_OK;
  void *h = dlopen(path ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
  TS_REQUIRE(h, TS_ERR_UNSUPPORTED, "ts_rccl_load: %s", dlerror());
  RcclApi api;
  api.handle = h;
  api.GetUniqueId = (int (*)(ts_nccl_id *))dlsym(h, "ncclGetUniqueId");
  api.CommInitRank = (int (*)(ts_nccl_comm *, int, ts_nccl_id, int))dlsym(h, "ncclCommInitRank");
  api.CommDestroy = (int (*)(ts_nccl_comm))dlsym(h, "ncclCommDestroy");
  api.AllReduce = (int (*)(const void *, void *, size_t, int, int, ts_nccl_comm, hipStream_t))dlsym(h, "ncclAllReduce");
  api.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
  TS_REQUIRE(api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce, TS_ERR_UNSUPPORTED,
             "ts_rccl_load: librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce");
  g_rccl = api;
  return TS_OK;
}

extern "C" int ts_rccl_unique_id(void *id128) {
  TS_REQUIRE(g_rccl.handle, TS_ERR_UNSUPPORTED, "ts_rccl_unique_id: call ts_rccl_load first");
  TS_REQUIRE(id128, TS_ERR_INVALID_ARGUMENT, "ts_rccl_unique_id: null pointer");
  const int rc = g_rccl.GetUniqueId((ts_nccl_id *)id128);
  return rc == 0 ? TS_OK : rccl_fail("ncclGetUniqueId", rc);
}

// Collective over the `nranks` processes that hold the same id; the calling thread's current HIP device is the rank's.
extern "C" int ts_rccl_comm_init(const void *id128, int32_t nranks, int32_t rank, void **comm) {
  TS_REQUIRE(g_rccl.handle, TS_ERR_UNSUPPORTED, "ts_rccl_comm_init: call ts_rccl_load first");
  TS_REQUIRE(id128 && comm && nranks >= 1 && rank >= 0 && rank < nranks, TS_ERR_INVALID_ARGUMENT,
             "ts_rccl_comm_init: bad arguments");
  ts_nccl_id id;
  memcpy(&id, id128, sizeof id);
  ts_nccl_comm c = nullptr;
  const int rc = g_rccl.CommInitRank(&c, nranks, id, rank);
  if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
  *comm = c;
  return TS_OK;
}

extern "C" int ts_rccl_comm_destroy(void *comm) {
  if (!comm || !g_rccl.handle) return TS_OK;
  const int rc = g_rccl.CommDestroy((ts_nccl_comm)comm);
  return rc == 0 ? TS_OK : rccl_fail("ncclCommDestroy", rc);
}

// in-place sum of `count` doubles over the communicator, on `stream`
extern "C" int ts_rccl_allreduce_f64(void *comm, double *buf, int64_t count, ts_stream_t stream) {
  TS_REQUIRE(g_rccl.handle && comm, TS_ERR_UNSUPPORTED, "ts_rccl_allreduce_f64: no communicator");
  TS_REQUIRE(buf && count > 0, TS_ERR_INVALID_ARGUMENT, "ts_rccl_allreduce_f64: bad arguments");
  const int rc = g_rccl.AllReduce(buf, buf, (size_t)count, TS_NCCL_FLOAT64, TS_NCCL_SUM, (ts_nccl_comm)comm,
                                  (hipStream_t)stream);
  return rc == 0 ? TS_OK : rccl_fail("ncclAllReduce", rc);
}

__global__ void bn_count_batch_kernel(int64_t *num_batches_tracked) { *num_batches_tracked += 1; }

#define TS_TRY(expr)          \
  do {                        \
    const int rc_ = (expr);   \
    if (rc_ != TS_OK) return rc_; \
  } while (0)

// act(SyncBN(x) [+ residual]), training mode.  pack [2C + 1] doubles (scratch; holds the global sums afterwards, its
// last element the global row count the backward pass needs).  half = 1: IEEE-half activations.
extern "C" int ts_bn_sync_forward(void *comm, const void *x, const void *residual, const float *weight,
                                  const float *bias, float *running_mean, float *running_var,
                                  int64_t *num_batches_tracked, int64_t n, int32_t c,
                                  float eps, float momentum, int32_t relu, int32_t half, double *pack, float *mean,
                                  float *invstd, void *out, uint8_t *mask, void *ws, size_t ws_bytes,
                                  ts_stream_t stream) {
  TS_REQUIRE(pack, TS_ERR_INVALID_ARGUMENT, "ts_bn_sync_forward: null pointer");
  // comm = TS_COMM_CALLER_PRE / _POST: the caller runs the all-reduce itself (torch.distributed) between two calls
  if (comm != TS_COMM_CALLER_POST) {
    if (num_batches_tracked) bn_count_batch_kernel<<<1, 1, 0, (hipStream_t)stream>>>(num_batches_tracked);
    if (half)
      TS_TRY(ts_bn_sync_stats_f16(x, n, c, pack, ws, ws_bytes, stream));
    else
      TS_TRY(ts_bn_sync_stats((const float *)x, n, c, pack, ws, ws_bytes, stream));
    if (comm == TS_COMM_CALLER_PRE) return TS_OK;
    TS_TRY(ts_rccl_allreduce_f64(comm, pack, 2 * (int64_t)c + 1, stream));
  }
  TS_TRY(ts_bn_finalize(pack, pack + 2 * c, (double)n, c, eps, momentum, running_mean, running_var, mean, invstd, stream));
  if (half)
    return ts_bn_act_forward_f16(x, residual, mean, invstd, weight, bias, n, c, relu, out, mask, stream);
  return ts_bn_act_forward((const float *)x, (const float *)residual, mean, invstd, weight, bias, n, c, relu,
                           (float *)out, mask, stream);
}

// Backward of ts_bn_sync_forward.  sums [2C] doubles (scratch); total_dev = the forward pack's last element.
// grad_weight / grad_bias are this rank's (the gradient all-reduce of the optimizer step averages them).
extern "C" int ts_bn_sync_backward(void *comm, const void *grad_out, const uint8_t *mask, const void *x,
                                   const float *mean, const float *invstd, const float *weight,
                                   const double *total_dev, int64_t n, int32_t c, int32_t half, double *sums,
                                   void *grad_x, void *grad_residual, float *grad_weight, float *grad_bias, void *ws,
                                   size_t ws_bytes, ts_stream_t stream) {
  TS_REQUIRE(sums && total_dev, TS_ERR_INVALID_ARGUMENT, "ts_bn_sync_backward: null pointer");
  if (comm != TS_COMM_CALLER_POST) {
    if (half)
      TS_TRY(ts_bn_sync_backward_reduce_f16(grad_out, mask, x, mean, invstd, n, c, sums, grad_weight, grad_bias, ws, ws_bytes,
                                            stream));
    else
      TS_TRY(ts_bn_sync_backward_reduce((const float *)grad_out, mask, (const float *)x, mean, invstd, n, c, sums,
                                        grad_weight, grad_bias, ws, ws_bytes, stream));
    if (comm == TS_COMM_CALLER_PRE) return TS_OK;
    TS_TRY(ts_rccl_allreduce_f64(comm, sums, 2 * (int64_t)c, stream));
  }
  if (half)
    return ts_bn_act_backward_f16(grad_out, mask, x, mean, invstd, weight, sums, total_dev, (double)n, n, c, grad_x,
                                  grad_residual, ws, ws_bytes, stream);
  return ts_bn_act_backward((const float *)grad_out, mask, (const float *)x, mean, invstd, weight, sums, total_dev,
                            (double)n, n, c, (float *)grad_x, (float *)grad_residual, stream);
}
