// Coordinate hashing, hash-table build / query, histogram count.
// Reference semantics: torchsparse backend/hash/hash_cuda.cu, backend/others/query_cuda.cu
// (interface only: 0 = miss, idx+1 = hit), backend/others/count_cuda.cu.
#include <stdarg.h>

#include "common.h"

// ---------------------------------------------------------------- error plumbing
static thread_local char g_err[512] = "";

void ts_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *ts_last_error(void) { return g_err; }
extern "C" const char *ts_version(void) { return "taseg_hip 0.1.0 gfx950"; }

// ---------------------------------------------------------------- tuning values (set by the host side's options object)
static int64_t g_opt[TS_OPT_COUNT] = {0};

extern "C" int ts_set_option(int32_t key, int64_t value) {
  TS_REQUIRE(key >= 0 && key < TS_OPT_COUNT, TS_ERR_INVALID_ARGUMENT, "ts_set_option: unknown key %d", key);
  if (key == TS_OPT_WGRAD_WGS)
    TS_REQUIRE(value == 0 || (value >= 64 && value <= 8192), TS_ERR_INVALID_ARGUMENT, "ts_set_option: workgroup target %lld outside 64 .. 8192",
               (long long)value);
  g_opt[key] = value;
  return TS_OK;
}

extern "C" int64_t ts_get_option(int32_t key) { return key >= 0 && key < TS_OPT_COUNT ? g_opt[key] : 0; }

// ---------------------------------------------------------------- multi-buffer fill
struct TsFillSegs {
  TsFillSeg s[TS_FILL_MAX];
};
__global__ __launch_bounds__(256) void fill_segments_kernel(TsFillSegs a) {
  const TsFillSeg sg = a.s[blockIdx.y];
  uint32_t *p = (uint32_t *)sg.p;
  const size_t nw = sg.bytes >> 2;
  size_t head = ((16 - ((uintptr_t)p & 15)) & 15) >> 2;      // words up to the first 16-byte boundary
  if (head > nw) head = nw;
  const size_t nv = (nw - head) >> 2;
  const uint4 v = make_uint4(sg.word, sg.word, sg.word, sg.word);
  uint4 *pv = (uint4 *)(p + head);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) pv[i] = v;
  if (blockIdx.x == 0 && threadIdx.x < 4) {
    if (threadIdx.x < head) p[threadIdx.x] = sg.word;
    const size_t t0 = head + 4 * nv + threadIdx.x;
    if (t0 < nw) p[t0] = sg.word;
  }
}

int ts_fill_segments(const TsFillSeg *segs, int n, hipStream_t stream) {
  TsFillSegs a;
  int cnt = 0;
  size_t big = 0;
  for (int i = 0; i < n; ++i) {
    if (!segs[i].p || segs[i].bytes == 0) continue;
    TS_REQUIRE(cnt < TS_FILL_MAX && (((uintptr_t)segs[i].p | segs[i].bytes) & 3) == 0, TS_ERR_INVALID_ARGUMENT,
               "ts_fill_segments: more than %d segments or a segment that is not 4-byte granular", TS_FILL_MAX);
    a.s[cnt++] = segs[i];
    big = std::max(big, segs[i].bytes);
  }
  if (cnt == 0) return TS_OK;
  dim3 grid((unsigned)std::max<size_t>(1, std::min<size_t>(ts_cdiv((int64_t)(big >> 4), 1024), 2048)), (unsigned)cnt);
  fill_segments_kernel<<<grid, 256, 0, stream>>>(a);
  TS_CHECK_LAUNCH("ts_fill_segments");
  return TS_OK;
}

// ---------------------------------------------------------------- table init
int ts_table_init(TsTable *t, int64_t n, void *ws, size_t ws_bytes, hipStream_t stream, size_t *used,
                  const TsFillSeg *extra, int n_extra) {
  size_t cap = ts_table_capacity(n);
  size_t kb = ts_align_up(cap * 8, 256), vb = ts_align_up(cap * 4, 256);
  TS_REQUIRE(ws != nullptr && ws_bytes >= kb + vb, TS_ERR_WORKSPACE_TOO_SMALL,
             "hash table: workspace %zu < %zu bytes", ws_bytes, kb + vb);
  TS_REQUIRE(((uintptr_t)ws & 7) == 0, TS_ERR_INVALID_ARGUMENT, "workspace must be 8-byte aligned");
  t->keys = (unsigned long long *)ws;
  t->vals = (int *)((char *)ws + kb);
  t->mask = (uint32_t)(cap - 1);
  TS_REQUIRE(n_extra >= 0 && n_extra <= TS_FILL_MAX - 2, TS_ERR_INVALID_ARGUMENT, "hash table: too many extra fill segments");
  TsFillSeg segs[TS_FILL_MAX] = {{t->keys, cap * 8, 0xFFFFFFFFu}, {t->vals, cap * 4, 0x7F7F7F7Fu}};
  for (int i = 0; i < n_extra; ++i) segs[2 + i] = extra[i];
  if (used) *used = kb + vb;
  return ts_fill_segments(segs, 2 + n_extra, stream);
}

// ---------------------------------------------------------------- K1: hash
__global__ __launch_bounds__(256) void hash_kernel(const int4 *__restrict__ coords, int64_t n,
                                                   int64_t *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    int4 c = coords[i];  // one 16-B load per voxel, coalesced
    out[i] = (int64_t)ts_fnv60(c.x, c.y, c.z, c.w);
  }
}

extern "C" int ts_hash(const int32_t *coords, int64_t n, int64_t *out, ts_stream_t stream) {
  TS_REQUIRE(n >= 0, TS_ERR_INVALID_ARGUMENT, "ts_hash: n < 0");
  if (n == 0) return TS_OK;
  TS_REQUIRE(coords && out, TS_ERR_INVALID_ARGUMENT, "ts_hash: null pointer");
  TS_REQUIRE(((uintptr_t)coords & 15) == 0, TS_ERR_INVALID_ARGUMENT, "ts_hash: coords must be 16-byte aligned");
  int grid = (int)std::min<int64_t>(ts_cdiv(n, 256), 2048);
  hash_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const int4 *)coords, n, out);
  TS_CHECK_LAUNCH("ts_hash");
  return TS_OK;
}

// ---------------------------------------------------------------- K2: kernel hash
// One thread per voxel, looping over the K offsets: the coordinate is loaded once
// (16 B), each out[k*n + i] store is coalesced across i.  Offsets are read
// through wave-uniform (scalar) loads.
__global__ __launch_bounds__(256) void kernel_hash_kernel(const int4 *__restrict__ coords, int64_t n,
                                                          const int *__restrict__ offsets, int K,
                                                          int64_t *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    int4 c = coords[i];
    for (int k = 0; k < K; ++k) {
      int ox = offsets[3 * k], oy = offsets[3 * k + 1], oz = offsets[3 * k + 2];
      out[(int64_t)k * n + i] = (int64_t)ts_fnv60(c.x + ox, c.y + oy, c.z + oz, c.w);
    }
  }
}

extern "C" int ts_kernel_hash(const int32_t *coords, int64_t n, const int32_t *offsets,
                              int32_t n_offsets, int64_t *out, ts_stream_t stream) {
  TS_REQUIRE(n >= 0 && n_offsets >= 0, TS_ERR_INVALID_ARGUMENT, "ts_kernel_hash: negative size");
  if (n == 0 || n_offsets == 0) return TS_OK;
  TS_REQUIRE(coords && offsets && out, TS_ERR_INVALID_ARGUMENT, "ts_kernel_hash: null pointer");
  TS_REQUIRE(((uintptr_t)coords & 15) == 0, TS_ERR_INVALID_ARGUMENT, "ts_kernel_hash: coords must be 16-byte aligned");
  int grid = (int)std::min<int64_t>(ts_cdiv(n, 256), 4096);
  kernel_hash_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const int4 *)coords, n, offsets,
                                                            n_offsets, out);
  TS_CHECK_LAUNCH("ts_kernel_hash");
  return TS_OK;
}

// ---------------------------------------------------------------- K3-K5: hash query
__global__ __launch_bounds__(256) void table_insert_keys_kernel(TsTable t, const int64_t *__restrict__ keys,
                                                                int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    uint64_t k = (uint64_t)keys[i];
    if (k != TS_EMPTY_KEY) ts_table_insert(t, k, (int)i);
  }
}

__global__ __launch_bounds__(256) void table_query_kernel(TsTable t, const int64_t *__restrict__ query,
                                                          int64_t nq, const int64_t *__restrict__ ref_idx,
                                                          int64_t *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < nq; i += step) {
    uint64_t k = (uint64_t)query[i];
    int pos = (k == TS_EMPTY_KEY) ? -1 : ts_table_find(t, k);
    int64_t r = 0;
    if (pos >= 0) r = (ref_idx ? ref_idx[pos] : (int64_t)pos) + 1;
    out[i] = r;
  }
}

extern "C" size_t ts_hash_query_workspace_bytes(int64_t n_ref) { return ts_table_bytes(n_ref < 0 ? 0 : n_ref); }

extern "C" int ts_hash_query(const int64_t *query, int64_t n_query, const int64_t *ref_hash,
                             const int64_t *ref_idx, int64_t n_ref, int64_t *out, void *ws,
                             size_t ws_bytes, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n_query >= 0 && n_ref >= 0, TS_ERR_INVALID_ARGUMENT, "ts_hash_query: negative size");
  TS_REQUIRE(n_ref < (1LL << 30), TS_ERR_UNSUPPORTED, "ts_hash_query: n_ref too large");
  if (n_query == 0) return TS_OK;
  TS_REQUIRE(query && out && (ref_hash || n_ref == 0), TS_ERR_INVALID_ARGUMENT, "ts_hash_query: null pointer");
  TsTable t;
  int rc = ts_table_init(&t, n_ref, ws, ws_bytes, stream, nullptr);
  if (rc != TS_OK) return rc;
  if (n_ref > 0) {
    int grid = (int)std::min<int64_t>(ts_cdiv(n_ref, 256), 4096);
    table_insert_keys_kernel<<<grid, 256, 0, stream>>>(t, ref_hash, n_ref);
    TS_CHECK_LAUNCH("ts_hash_query/insert");
  }
  int grid = (int)std::min<int64_t>(ts_cdiv(n_query, 256), 8192);
  table_query_kernel<<<grid, 256, 0, stream>>>(t, query, n_query, ref_idx, out);
  TS_CHECK_LAUNCH("ts_hash_query/lookup");
  return TS_OK;
}

// ---------------------------------------------------------------- K6: count
__global__ __launch_bounds__(256) void count_kernel(const int *__restrict__ idx, int64_t n, int64_t n_out,
                                                    int *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    int v = idx[i];
    if (v >= 0 && v < n_out) atomicAdd(&out[v], 1);
  }
}

extern "C" int ts_count(const int32_t *idx, int64_t n, int32_t *out, int64_t n_out, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && n_out >= 0, TS_ERR_INVALID_ARGUMENT, "ts_count: negative size");
  if (n_out == 0) return TS_OK;
  TS_REQUIRE(out && (idx || n == 0), TS_ERR_INVALID_ARGUMENT, "ts_count: null pointer");
  TS_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)n_out * 4, stream), "ts_count memset");
  if (n == 0) return TS_OK;
  int grid = (int)std::min<int64_t>(ts_cdiv(n, 256), 4096);
  count_kernel<<<grid, 256, 0, stream>>>(idx, n, n_out, out);
  TS_CHECK_LAUNCH("ts_count");
  return TS_OK;
}
