// Rulebook construction on device: strided down-sampling (sort + unique),
// int64 unique/inverse, kernel-map (neighbour table + reference-order pair list),
// trilinear point<->voxel maps.
//
// Reference semantics restated here:
//   torchsparse nn/functional/downsample.py:25-51, nn/functional/conv.py:156-176,
//   nn/functional/devoxelize.py:10-48, pcseg/model/segmentor/voxel/minkunet/utils.py:11-36,69-82.
// Sorting / scanning use rocPRIM device primitives (the AMD-native ones torch itself
// uses on ROCm); everything coordinate-specific is hand-written below.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>

#include "common.h"

// ------------------------------------------------------------------ packing
// (b, x, y, z) -> one uint64 whose unsigned order == lexicographic signed order.
#define TS_CBIAS (1 << 17)
#define TS_CMASK ((1u << 18) - 1)

__device__ __forceinline__ bool ts_pack_ok(int x, int y, int z, int b) {
  return (unsigned)(x + TS_CBIAS) <= TS_CMASK && (unsigned)(y + TS_CBIAS) <= TS_CMASK &&
         (unsigned)(z + TS_CBIAS) <= TS_CMASK && (unsigned)b < 1024u;
}
__device__ __forceinline__ uint64_t ts_pack(int x, int y, int z, int b) {
  return ((uint64_t)(unsigned)b << 54) | ((uint64_t)(unsigned)(x + TS_CBIAS) << 36) |
         ((uint64_t)(unsigned)(y + TS_CBIAS) << 18) | (uint64_t)(unsigned)(z + TS_CBIAS);
}
__device__ __forceinline__ int4 ts_unpack(uint64_t k) {
  int4 c;
  c.w = (int)(k >> 54);
  c.x = (int)((k >> 36) & TS_CMASK) - TS_CBIAS;
  c.y = (int)((k >> 18) & TS_CMASK) - TS_CBIAS;
  c.z = (int)(k & TS_CMASK) - TS_CBIAS;
  return c;
}

// ------------------------------------------------------------------ downsample
__global__ __launch_bounds__(256) void ds_pack_kernel(const int4 *__restrict__ coords, int64_t n, int sx,
                                                      int sy, int sz, uint64_t *__restrict__ keys,
                                                      int *__restrict__ err) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    int4 c = coords[i];
    // trunc(c / s) * s  (C integer division truncates toward zero, like
    // torch.div(int, int).trunc() on exactly-representable values)
    int x = (c.x / sx) * sx, y = (c.y / sy) * sy, z = (c.z / sz) * sz;
    if (!ts_pack_ok(x, y, z, c.w)) *err = 1;
    keys[i] = ts_pack(x, y, z, c.w);
  }
}

__global__ __launch_bounds__(256) void ds_unpack_kernel(const uint64_t *__restrict__ uniq,
                                                        const unsigned *__restrict__ count,
                                                        const int *__restrict__ err,
                                                        int4 *__restrict__ out, int *__restrict__ out_count) {
  int64_t m = *count;
  if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = (*err) ? -1 : (int)m;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < m; i += step) out[i] = ts_unpack(uniq[i]);
}

static size_t sort_unique_ws_bytes(int64_t n) {
  // 3 key arrays + counters + generous rocPRIM scratch (verified at run time)
  size_t nn = (size_t)(n < 1 ? 1 : n);
  return 3 * ts_align_up(nn * 8, 256) + 256 + ts_align_up(nn * 16 + (4u << 20), 256);
}

struct SortUniqueWs {
  uint64_t *a, *b, *u;
  unsigned *count;
  int *err;
  void *tmp;
  size_t tmp_bytes;
};

static int carve_sort_unique(SortUniqueWs *w, int64_t n, void *ws, size_t ws_bytes) {
  TS_REQUIRE(ws && ws_bytes >= sort_unique_ws_bytes(n), TS_ERR_WORKSPACE_TOO_SMALL,
             "sort/unique: workspace %zu < %zu bytes", ws_bytes, sort_unique_ws_bytes(n));
  TS_REQUIRE(((uintptr_t)ws & 255) == 0, TS_ERR_INVALID_ARGUMENT, "workspace must be 256-byte aligned");
  size_t nn = (size_t)(n < 1 ? 1 : n);
  size_t kb = ts_align_up(nn * 8, 256);
  char *p = (char *)ws;
  w->a = (uint64_t *)p;
  p += kb;
  w->b = (uint64_t *)p;
  p += kb;
  w->u = (uint64_t *)p;
  p += kb;
  w->count = (unsigned *)p;
  w->err = (int *)(p + 64);
  p += 256;
  w->tmp = p;
  w->tmp_bytes = ws_bytes - (size_t)(p - (char *)ws);
  return TS_OK;
}

// keys in w->a  ->  sorted unique keys in w->u, *w->count
static int sort_unique(SortUniqueWs *w, int64_t n, unsigned end_bit, hipStream_t stream) {
  size_t need = 0;
  TS_CHECK_HIP(rocprim::radix_sort_keys(nullptr, need, w->a, w->b, (size_t)n, 0u, end_bit, stream),
               "radix_sort size query");
  TS_REQUIRE(need <= w->tmp_bytes, TS_ERR_WORKSPACE_TOO_SMALL, "radix sort scratch %zu > %zu", need, w->tmp_bytes);
  size_t tb = w->tmp_bytes;
  TS_CHECK_HIP(rocprim::radix_sort_keys(w->tmp, tb, w->a, w->b, (size_t)n, 0u, end_bit, stream), "radix_sort");
  need = 0;
  TS_CHECK_HIP(rocprim::unique(nullptr, need, w->b, w->u, w->count, (size_t)n,
                               rocprim::equal_to<uint64_t>(), stream),
               "unique size query");
  TS_REQUIRE(need <= w->tmp_bytes, TS_ERR_WORKSPACE_TOO_SMALL, "unique scratch %zu > %zu", need, w->tmp_bytes);
  tb = w->tmp_bytes;
  TS_CHECK_HIP(rocprim::unique(w->tmp, tb, w->b, w->u, w->count, (size_t)n, rocprim::equal_to<uint64_t>(), stream),
               "unique");
  return TS_OK;
}

extern "C" size_t ts_downsample_workspace_bytes(int64_t n) { return sort_unique_ws_bytes(n); }

extern "C" int ts_downsample(const int32_t *coords, int64_t n, int32_t sx, int32_t sy, int32_t sz,
                             int32_t *out_coords, int32_t *out_count, void *ws, size_t ws_bytes,
                             ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && n < (1LL << 31), TS_ERR_INVALID_ARGUMENT, "ts_downsample: bad n");
  TS_REQUIRE(sx > 0 && sy > 0 && sz > 0, TS_ERR_INVALID_ARGUMENT, "ts_downsample: strides must be positive");
  TS_REQUIRE(out_count, TS_ERR_INVALID_ARGUMENT, "ts_downsample: null out_count");
  if (n == 0) {
    TS_CHECK_HIP(hipMemsetAsync(out_count, 0, 4, stream), "ts_downsample memset");
    return TS_OK;
  }
  TS_REQUIRE(coords && out_coords, TS_ERR_INVALID_ARGUMENT, "ts_downsample: null pointer");
  TS_REQUIRE(((uintptr_t)coords & 15) == 0 && ((uintptr_t)out_coords & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_downsample: coords must be 16-byte aligned");
  SortUniqueWs w;
  int rc = carve_sort_unique(&w, n, ws, ws_bytes);
  if (rc != TS_OK) return rc;
  TS_CHECK_HIP(hipMemsetAsync(w.count, 0, 256, stream), "ts_downsample memset");
  int grid = (int)std::min<int64_t>(ts_cdiv(n, 256), 2048);
  ds_pack_kernel<<<grid, 256, 0, stream>>>((const int4 *)coords, n, sx, sy, sz, w.a, w.err);
  TS_CHECK_LAUNCH("ts_downsample/pack");
  rc = sort_unique(&w, n, 64, stream);
  if (rc != TS_OK) return rc;
  ds_unpack_kernel<<<grid, 256, 0, stream>>>(w.u, w.count, w.err, (int4 *)out_coords, out_count);
  TS_CHECK_LAUNCH("ts_downsample/unpack");
  return TS_OK;
}

// ------------------------------------------------------------------ unique int64 (+ inverse)
__global__ __launch_bounds__(256) void uq_copy_kernel(const int64_t *__restrict__ keys, int64_t n,
                                                      uint64_t *__restrict__ out, int *__restrict__ err) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    int64_t k = keys[i];
    if (k < 0 || k >= (1LL << 62)) *err = 1;
    out[i] = (uint64_t)k;
  }
}

__global__ __launch_bounds__(256) void uq_insert_kernel(TsTable t, const uint64_t *__restrict__ uniq,
                                                        const unsigned *__restrict__ count,
                                                        int64_t *__restrict__ uniq_out) {
  int64_t m = *count;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < m; i += step) {
    uint64_t k = uniq[i];
    if (t.keys) ts_table_insert(t, k, (int)i);
    if (uniq_out) uniq_out[i] = (int64_t)k;
  }
}

__global__ __launch_bounds__(256) void uq_inverse_kernel(TsTable t, const int64_t *__restrict__ keys, int64_t n,
                                                         const unsigned *__restrict__ count,
                                                         const int *__restrict__ err, int *__restrict__ inverse,
                                                         int *__restrict__ out_count) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = (*err) ? -1 : (int)(*count);
  if (!inverse) return;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) inverse[i] = ts_table_find(t, (uint64_t)keys[i]);
}

extern "C" size_t ts_unique_workspace_bytes(int64_t n) {
  return sort_unique_ws_bytes(n) + ts_table_bytes(n < 0 ? 0 : n);
}

extern "C" int ts_unique_i64(const int64_t *keys, int64_t n, int64_t *uniq, int32_t *inverse,
                             int32_t *out_count, void *ws, size_t ws_bytes, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && n < (1LL << 30), TS_ERR_INVALID_ARGUMENT, "ts_unique_i64: bad n");
  TS_REQUIRE(out_count, TS_ERR_INVALID_ARGUMENT, "ts_unique_i64: null out_count");
  if (n == 0) {
    TS_CHECK_HIP(hipMemsetAsync(out_count, 0, 4, stream), "ts_unique memset");
    return TS_OK;
  }
  TS_REQUIRE(keys, TS_ERR_INVALID_ARGUMENT, "ts_unique_i64: null keys");
  TS_REQUIRE(ws_bytes >= ts_unique_workspace_bytes(n), TS_ERR_WORKSPACE_TOO_SMALL, "ts_unique_i64: workspace too small");
  size_t su = sort_unique_ws_bytes(n);
  SortUniqueWs w;
  int rc = carve_sort_unique(&w, n, ws, su);
  if (rc != TS_OK) return rc;
  TS_CHECK_HIP(hipMemsetAsync(w.count, 0, 256, stream), "ts_unique memset");
  int grid = (int)std::min<int64_t>(ts_cdiv(n, 256), 2048);
  uq_copy_kernel<<<grid, 256, 0, stream>>>(keys, n, w.a, w.err);
  TS_CHECK_LAUNCH("ts_unique/copy");
  rc = sort_unique(&w, n, 62, stream);
  if (rc != TS_OK) return rc;
  TsTable t{nullptr, nullptr, 0};
  if (inverse) {
    rc = ts_table_init(&t, n, (char *)ws + su, ws_bytes - su, stream, nullptr);
    if (rc != TS_OK) return rc;
  }
  uq_insert_kernel<<<grid, 256, 0, stream>>>(t, w.u, w.count, uniq);
  TS_CHECK_LAUNCH("ts_unique/insert");
  uq_inverse_kernel<<<grid, 256, 0, stream>>>(t, keys, n, w.count, w.err, inverse, out_count);
  TS_CHECK_LAUNCH("ts_unique/inverse");
  return TS_OK;
}

// ------------------------------------------------------------------ kernel map
__global__ __launch_bounds__(256) void table_insert_coords_kernel(TsTable t, const int4 *__restrict__ coords,
                                                                  int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    int4 c = coords[i];
    ts_table_insert(t, ts_fnv60(c.x, c.y, c.z, c.w), (int)i);
  }
}

#define KM_BLOCK 256
// One block = 256 consecutive output voxels; every thread keeps its coordinate in
// registers and probes all K offsets.  nbr stores are coalesced across j for each
// k; hit counts per (k, block) go through LDS counters.
__global__ __launch_bounds__(KM_BLOCK) void kmap_probe_kernel(TsTable t, const int4 *__restrict__ out_coords,
                                                             int64_t n_out, const int *__restrict__ offsets, int K,
                                                             int *__restrict__ nbr, unsigned *__restrict__ blk_counts,
                                                             int nblk) {
  // Probes are issued in batches of KM_MLP offsets (ts_table_find_n: independent first-slot loads in flight together).  The
  // round-2 form probed offset after offset - 27 x 2 dependent round trips per lane, 57 us per call (2.0 TB/s of its
  // 115 MB); same results, same table.
  constexpr int KM_MLP = 9;
  extern __shared__ unsigned lds_cnt[];
  for (int k = threadIdx.x; k < K; k += KM_BLOCK) lds_cnt[k] = 0;
  __syncthreads();
  int64_t j = (int64_t)blockIdx.x * KM_BLOCK + threadIdx.x;
  bool valid = j < n_out;
  int4 c = valid ? out_coords[j] : make_int4(0, 0, 0, 0);
  for (int k0 = 0; k0 < K; k0 += KM_MLP) {
    unsigned long long want[KM_MLP];
    int r[KM_MLP];
#pragma unroll
    for (int i = 0; i < KM_MLP; ++i) {
      const int k = min(k0 + i, K - 1);
      want[i] = ts_fnv60(c.x + offsets[3 * k], c.y + offsets[3 * k + 1], c.z + offsets[3 * k + 2], c.w);
    }
    ts_table_find_n<KM_MLP>(t, want, valid, r);
#pragma unroll
    for (int i = 0; i < KM_MLP; ++i) {
      const int k = k0 + i;
      if (k < K) {                                               // (uniform)
        if (valid) nbr[(int64_t)k * n_out + j] = r[i];
        unsigned long long m = __ballot(valid && r[i] >= 0);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(&lds_cnt[k], (unsigned)__popcll(m));
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += KM_BLOCK) blk_counts[(int64_t)k * nblk + blockIdx.x] = lds_cnt[k];
}

// Submanifold maps (in == out, unique coordinates, offsets[K - 1 - k] == -offsets[k]: nn/utils/kernel.py:11-32 enumerates an odd
// volume that way): offset k of voxel j finding r IS offset K - 1 - k of voxel r finding j.  Every thread probes the first half of
// the offsets and the centre (14 look-ups instead of 27) and stores the mirrored entry nbr[K - 1 - k][r] = j beside its own - one
// writer per entry, since r's neighbour at that offset is unique; the mirrored half was filled with -1 before.  The centre probe
// must return j itself: anything else is a duplicated coordinate, for which the symmetry does not hold - *dup is set and the caller
// builds the map with the full probe.
// KM_MLP look-ups in flight per round: 14 = the 13 + 1 probes of a 3x3x3 map in ONE round (two dependent round trips in all; rounds
// of 9 made it 9 + 9 with four of the second round's probes clamped duplicates of the centre - 18 look-ups for 14).
template <int KM_MLP>
__global__ __launch_bounds__(KM_BLOCK) void kmap_probe_sym_kernel(TsTable t, const int4 *__restrict__ coords, int64_t n,
                                                                 const int *__restrict__ offsets, int K, int *__restrict__ nbr,
                                                                 unsigned *__restrict__ blk_counts, int nblk, int *__restrict__ dup) {
  extern __shared__ unsigned lds_cnt[];
  const int half = K / 2;
  for (int k = threadIdx.x; k <= half; k += KM_BLOCK) lds_cnt[k] = 0;
  __syncthreads();
  const int64_t j = (int64_t)blockIdx.x * KM_BLOCK + threadIdx.x;
  const bool valid = j < n;
  const int4 c = valid ? coords[j] : make_int4(0, 0, 0, 0);
  for (int k0 = 0; k0 <= half; k0 += KM_MLP) {
    unsigned long long want[KM_MLP];
    int r[KM_MLP];
#pragma unroll
    for (int i = 0; i < KM_MLP; ++i) {
      const int k = min(k0 + i, half);
      want[i] = ts_fnv60(c.x + offsets[3 * k], c.y + offsets[3 * k + 1], c.z + offsets[3 * k + 2], c.w);
    }
    ts_table_find_n<KM_MLP>(t, want, valid, r);
#pragma unroll
    for (int i = 0; i < KM_MLP; ++i) {
      const int k = k0 + i;
      if (k <= half) {                                            // (uniform)
        if (valid) {
          nbr[(int64_t)k * n + j] = r[i];
          if (k < half && r[i] >= 0) nbr[(int64_t)(K - 1 - k) * n + r[i]] = (int)j;
          if (k == half && r[i] != (int)j) *dup = 1;
        }
        unsigned long long m = __ballot(valid && r[i] >= 0);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(&lds_cnt[k], (unsigned)__popcll(m));
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k <= half; k += KM_BLOCK) blk_counts[(int64_t)k * nblk + blockIdx.x] = lds_cnt[k];
}

// hits per (offset, block of 256 rows) of the mirrored half, counted from the table (grid (nblk, K / 2))
__global__ __launch_bounds__(KM_BLOCK) void kmap_count_kernel(const int *__restrict__ nbr, int64_t n, int K,
                                                             unsigned *__restrict__ blk_counts, int nblk) {
  __shared__ unsigned wave_cnt[KM_BLOCK / 64];
  const int k = K / 2 + 1 + (int)blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * KM_BLOCK + threadIdx.x;
  const bool hit = j < n && nbr[(int64_t)k * n + j] >= 0;
  const unsigned long long m = __ballot(hit);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = (unsigned)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) blk_counts[(int64_t)k * nblk + blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ void kmap_sizes_kernel(const unsigned *__restrict__ blk_offs, int K, int nblk, int *__restrict__ nbsizes,
                                  int *__restrict__ nboffs, const int *__restrict__ dup) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k <= K) {
    unsigned o = blk_offs[(int64_t)k * nblk];
    // (symmetric builder: a duplicated coordinate voids the map - the pair total reads -1 and the caller probes in full)
    if (nboffs) nboffs[k] = (k == K && dup && *dup) ? -1 : (int)o;
    if (k < K && nbsizes) nbsizes[k] = (int)(blk_offs[(int64_t)(k + 1) * nblk] - o);
  }
}

// grid (nblk, K): stable compaction of the hits of offset k inside one block of
// 256 outputs -> pairs (in, out) at blk_offs[k][blk] + rank; also fills the
// inverse table nbr_t[k][in] = out.
__global__ __launch_bounds__(KM_BLOCK) void kmap_compact_kernel(const int *__restrict__ nbr, int64_t n_out,
                                                               int64_t n_in, const unsigned *__restrict__ blk_offs,
                                                               int nblk, int2 *__restrict__ nbmaps,
                                                               int *__restrict__ nbr_t, int *__restrict__ pos_out,
                                                               int *__restrict__ pos_in) {
  __shared__ unsigned wave_cnt[KM_BLOCK / 64];
  int k = blockIdx.y;
  int64_t j = (int64_t)blockIdx.x * KM_BLOCK + threadIdx.x;
  int r = (j < n_out) ? nbr[(int64_t)k * n_out + j] : -1;
  bool hit = r >= 0;
  unsigned long long m = __ballot(hit);
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned rank = (unsigned)__popcll(m & ((1ULL << lane) - 1ULL));
  if (lane == 0) wave_cnt[wave] = (unsigned)__popcll(m);
  __syncthreads();
  unsigned base = blk_offs[(int64_t)k * nblk + blockIdx.x];
  for (int w = 0; w < wave; ++w) base += wave_cnt[w];
  if (hit) {
    if (nbmaps) nbmaps[base + rank] = make_int2(r, (int)j);
    if (nbr_t) nbr_t[(int64_t)k * n_in + r] = (int)j;
    if (pos_in) pos_in[(int64_t)k * n_in + r] = (int)(base + rank);
  }
  // position of pair (k, j) in the rulebook, per output row (coalesced; -1 = no pair)
  if (pos_out && j < n_out) pos_out[(int64_t)k * n_out + j] = hit ? (int)(base + rank) : -1;
}

extern "C" size_t ts_build_kmap_workspace_bytes(int64_t n_in, int64_t n_out, int32_t K) {
  int64_t nblk = ts_cdiv(n_out < 1 ? 1 : n_out, KM_BLOCK);
  size_t cnt = ts_align_up(((size_t)K * nblk + 1) * 4, 256);
  return ts_table_bytes(n_in < 0 ? 0 : n_in) + 2 * cnt + ts_align_up(cnt + (1u << 20), 256) + 256;      // (+ the symmetric builder's flag)
}

static int build_kmap_impl(const int32_t *in_coords, int64_t n_in, const int32_t *out_coords, int64_t n_out,
                           const int32_t *offsets, int32_t K, int32_t *nbr, int32_t *nbr_t, int32_t *nbmaps,
                           int32_t *nbsizes, int32_t *nboffs, int32_t *pos_out, int32_t *pos_in, void *ws,
                           size_t ws_bytes, ts_stream_t stream_, bool sym) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n_in >= 0 && n_out >= 0 && K > 0 && K <= 4096, TS_ERR_INVALID_ARGUMENT, "ts_build_kmap: bad sizes");
  TS_REQUIRE(n_in < (1LL << 30) && n_out < (1LL << 30) && (int64_t)K * n_out < (1LL << 31), TS_ERR_UNSUPPORTED,
             "ts_build_kmap: problem too large for int32 indexing");
  TS_REQUIRE(offsets && (nbr || n_out == 0), TS_ERR_INVALID_ARGUMENT, "ts_build_kmap: null pointer");
  TS_REQUIRE(ws_bytes >= ts_build_kmap_workspace_bytes(n_in, n_out, K), TS_ERR_WORKSPACE_TOO_SMALL,
             "ts_build_kmap: workspace %zu < %zu", ws_bytes, ts_build_kmap_workspace_bytes(n_in, n_out, K));
  // the tables only the hits write (-1 = no pair): pos_in, nbr_t; one fill launch together with the hash table below
  const TsFillSeg inverse[2] = {{pos_in, (size_t)K * n_in * 4, 0xFFFFFFFFu}, {nbr_t, (size_t)K * n_in * 4, 0xFFFFFFFFu}};
  if (n_out == 0) {                              // no output rows: empty rulebook
    const TsFillSeg empty[4] = {inverse[0], inverse[1], {nbsizes, (size_t)K * 4, 0u}, {nboffs, (size_t)(K + 1) * 4, 0u}};
    return ts_fill_segments(empty, 4, stream);
  }
  TS_REQUIRE(out_coords && (in_coords || n_in == 0), TS_ERR_INVALID_ARGUMENT, "ts_build_kmap: null coords");
  TS_REQUIRE(((uintptr_t)in_coords & 15) == 0 && ((uintptr_t)out_coords & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_build_kmap: coords must be 16-byte aligned");
  TsTable t;
  int nblk = (int)ts_cdiv(n_out, KM_BLOCK);
  size_t n_cnt = (size_t)K * nblk + 1;
  size_t cnt_bytes = ts_align_up(n_cnt * 4, 256);
  size_t used = ts_table_bytes(n_in);
  TS_REQUIRE(used <= ws_bytes, TS_ERR_WORKSPACE_TOO_SMALL, "ts_build_kmap: workspace too small");
  // table keys / values, the two inverse tables and the closing element of the per-block counts: ONE fill launch (symmetric
  // builder: + the mirrored half of nbr, which only hits write, + the duplicate flag, kept in the last word of the scan scratch)
  const int half = K / 2;
  // (the flag sits in the last 256 bytes of what ts_build_kmap_workspace_bytes asks for, behind the scan's scratch)
  const size_t flag_at = ts_build_kmap_workspace_bytes(n_in, n_out, K) - 256;
  TS_REQUIRE(!sym || ws_bytes >= flag_at + 256, TS_ERR_WORKSPACE_TOO_SMALL, "ts_build_kmap_sym: workspace too small for the duplicate flag");
  int *dup = sym ? (int *)((char *)ws + flag_at) : nullptr;
  const TsFillSeg extra[5] = {inverse[0], inverse[1], {(unsigned *)((char *)ws + used) + (n_cnt - 1), 4, 0u},
                              {nbr + (size_t)(half + 1) * n_out, (size_t)half * n_out * 4, 0xFFFFFFFFu}, {dup, 4, 0u}};
  int rc = ts_table_init(&t, n_in, ws, ws_bytes, stream, &used, extra, sym ? 5 : 3);
  if (rc != TS_OK) return rc;
  unsigned *blk_counts = (unsigned *)((char *)ws + used);
  unsigned *blk_offs = (unsigned *)((char *)ws + used + cnt_bytes);
  void *tmp = (char *)ws + used + 2 * cnt_bytes;
  size_t tmp_bytes = (sym ? flag_at : ws_bytes) - used - 2 * cnt_bytes;

  if (n_in > 0) {
    int grid = (int)std::min<int64_t>(ts_cdiv(n_in, 256), 4096);
    table_insert_coords_kernel<<<grid, 256, 0, stream>>>(t, (const int4 *)in_coords, n_in);
    TS_CHECK_LAUNCH("ts_build_kmap/insert");
  }
  if (sym) {
    if (half + 1 <= 14 && half + 1 > 9)
      kmap_probe_sym_kernel<14><<<nblk, KM_BLOCK, (size_t)(half + 1) * 4, stream>>>(t, (const int4 *)out_coords, n_out, offsets, K, nbr,
                                                                                    blk_counts, nblk, dup);
    else
      kmap_probe_sym_kernel<9><<<nblk, KM_BLOCK, (size_t)(half + 1) * 4, stream>>>(t, (const int4 *)out_coords, n_out, offsets, K, nbr,
                                                                                   blk_counts, nblk, dup);
    TS_CHECK_LAUNCH("ts_build_kmap_sym/probe");
    kmap_count_kernel<<<dim3(nblk, half), KM_BLOCK, 0, stream>>>(nbr, n_out, K, blk_counts, nblk);
    TS_CHECK_LAUNCH("ts_build_kmap_sym/count");
  } else {
    kmap_probe_kernel<<<nblk, KM_BLOCK, (size_t)K * 4, stream>>>(t, (const int4 *)out_coords, n_out, offsets, K, nbr,
                                                                 blk_counts, nblk);
    TS_CHECK_LAUNCH("ts_build_kmap/probe");
  }
  if (!nbmaps && !nbr_t && !nbsizes && !nboffs && !pos_out && !pos_in) return TS_OK;
  size_t need = 0;
  TS_CHECK_HIP(rocprim::exclusive_scan(nullptr, need, blk_counts, blk_offs, 0u, n_cnt, rocprim::plus<unsigned>(), stream),
               "scan size query");
  TS_REQUIRE(need <= tmp_bytes, TS_ERR_WORKSPACE_TOO_SMALL, "ts_build_kmap: scan scratch %zu > %zu", need, tmp_bytes);
  TS_CHECK_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, blk_counts, blk_offs, 0u, n_cnt, rocprim::plus<unsigned>(), stream),
               "scan");
  kmap_sizes_kernel<<<(int)ts_cdiv(K + 1, 256), 256, 0, stream>>>(blk_offs, K, nblk, nbsizes, nboffs, dup);
  TS_CHECK_LAUNCH("ts_build_kmap/sizes");
  if (nbmaps || nbr_t || pos_out || pos_in) {
    dim3 grid(nblk, K);
    kmap_compact_kernel<<<grid, KM_BLOCK, 0, stream>>>(nbr, n_out, n_in, blk_offs, nblk, (int2 *)nbmaps, nbr_t,
                                                       pos_out, pos_in);
    TS_CHECK_LAUNCH("ts_build_kmap/compact");
  }
  return TS_OK;
}

extern "C" int ts_build_kmap(const int32_t *in_coords, int64_t n_in, const int32_t *out_coords, int64_t n_out,
                             const int32_t *offsets, int32_t K, int32_t *nbr, int32_t *nbr_t, int32_t *nbmaps,
                             int32_t *nbsizes, int32_t *nboffs, int32_t *pos_out, int32_t *pos_in, void *ws,
                             size_t ws_bytes, ts_stream_t stream) {
  return build_kmap_impl(in_coords, n_in, out_coords, n_out, offsets, K, nbr, nbr_t, nbmaps, nbsizes, nboffs, pos_out, pos_in, ws,
                         ws_bytes, stream, false);
}

// ts_build_kmap for a SUBMANIFOLD map (one coordinate set, odd K, offsets[K - 1 - k] == -offsets[k]: conv.py:156-176 with
// kernel_size odd and stride 1) on half the probes; same tables, bit for bit.  Precondition the call checks on the device:
// UNIQUE coordinates - otherwise nboffs[K] reads -1 (nothing else is meaningful) and the caller falls back to ts_build_kmap.
extern "C" int ts_build_kmap_sym(const int32_t *coords, int64_t n, const int32_t *offsets, int32_t K, int32_t *nbr, int32_t *nbr_t,
                                 int32_t *nbmaps, int32_t *nbsizes, int32_t *nboffs, int32_t *pos_out, int32_t *pos_in, void *ws,
                                 size_t ws_bytes, ts_stream_t stream) {
  TS_REQUIRE(K > 1 && (K & 1) && nboffs && nbr && n > 0, TS_ERR_INVALID_ARGUMENT,
             "ts_build_kmap_sym: an odd number of offsets > 1, rows and the nbr / nboffs outputs are required");
  return build_kmap_impl(coords, n, coords, n, offsets, K, nbr, nbr_t, nbmaps, nbsizes, nboffs, pos_out, pos_in, ws, ws_bytes, stream, true);
}

// ------------------------------------------------------------------ nbr from an explicit rulebook
__global__ __launch_bounds__(256) void nbr_from_nbmaps_kernel(const int2 *__restrict__ nbmaps,
                                                              const int *__restrict__ nboffs, int K, int col_in,
                                                              int64_t n_rows, int *__restrict__ nbr) {
  int k = blockIdx.y;
  int beg = nboffs[k], end = nboffs[k + 1];
  for (int p = beg + blockIdx.x * blockDim.x + threadIdx.x; p < end; p += gridDim.x * blockDim.x) {
    int2 pr = nbmaps[p];
    int i = col_in ? pr.y : pr.x, o = col_in ? pr.x : pr.y;
    if (o >= 0 && o < n_rows && i >= 0) nbr[(int64_t)k * n_rows + o] = i;
  }
}

extern "C" int ts_nbr_from_nbmaps(const int32_t *nbmaps, const int32_t *nboffs, int32_t K, int32_t col_in,
                                  int64_t n_rows, int32_t *nbr, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(K > 0 && n_rows >= 0 && (int64_t)K * n_rows < (1LL << 31), TS_ERR_INVALID_ARGUMENT,
             "ts_nbr_from_nbmaps: bad sizes");
  if (n_rows == 0) return TS_OK;
  TS_REQUIRE(nbmaps && nboffs && nbr, TS_ERR_INVALID_ARGUMENT, "ts_nbr_from_nbmaps: null pointer");
  TS_CHECK_HIP(hipMemsetAsync(nbr, 0xFF, (size_t)K * n_rows * 4, stream), "nbr memset");
  dim3 grid((unsigned)std::min<int64_t>(ts_cdiv(n_rows, 256), 1024), K);
  nbr_from_nbmaps_kernel<<<grid, 256, 0, stream>>>((const int2 *)nbmaps, nboffs, K, col_in ? 1 : 0, n_rows, nbr);
  TS_CHECK_LAUNCH("ts_nbr_from_nbmaps");
  return TS_OK;
}

// ------------------------------------------------------------------ trilinear map
__global__ __launch_bounds__(256) void trilinear_kernel(TsTable t, const float4 *__restrict__ pts, int64_t n, int s,
                                                        int *__restrict__ idx, float *__restrict__ wout) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  const float fs = (float)s;
  for (; i < n; i += step) {
    float4 p = pts[i];
    // utils.py:73-76: floor(z.C[:, :3] / s).int() * s ; batch = z.C[:, -1].int()
    float fx = floorf(p.x / fs), fy = floorf(p.y / fs), fz = floorf(p.z / fs);
    int bx = (int)fx * s, by = (int)fy * s, bz = (int)fz * s, b = (int)p.w;
    // devoxelize.py:14-32 (pf = floor(p / s) * s in float, pc = pf + s)
    float xf, yf, zf;
    if (s != 1) {
      xf = fx * fs;
      yf = fy * fs;
      zf = fz * fs;
    } else {
      xf = floorf(p.x);
      yf = floorf(p.y);
      zf = floorf(p.z);
    }
    float xc = xf + fs, yc = yf + fs, zc = zf + fs;
    float wx[2] = {xc - p.x, p.x - xf}, wy[2] = {yc - p.y, p.y - yf}, wz[2] = {zc - p.z, p.z - zf};
    int id[8];
    float w[8];
    float sum = 0.f;
    const float s3 = (float)(s * s * s);
    unsigned long long want[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)      // get_kernel_offsets(2): x outermost
      want[k] = ts_fnv60(bx + ((k >> 2) & 1) * s, by + ((k >> 1) & 1) * s, bz + (k & 1) * s, b);
    ts_table_find_n<8>(t, want, true, id);       // the 8 corner look-ups in flight together
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int ix = (k >> 2) & 1, iy = (k >> 1) & 1, iz = k & 1;
      float v = (wx[ix] * wy[iy]) * wz[iz];
      if (s != 1) v /= s3;  // w /= scale**3
      if (id[k] < 0) v = 0.f;
      w[k] = v;
      sum += v;
    }
    sum += 1e-8f;
    int4 *ip = (int4 *)(idx + i * 8);
    ip[0] = make_int4(id[0], id[1], id[2], id[3]);
    ip[1] = make_int4(id[4], id[5], id[6], id[7]);
    float4 *wp = (float4 *)(wout + i * 8);
    wp[0] = make_float4(w[0] / sum, w[1] / sum, w[2] / sum, w[3] / sum);
    wp[1] = make_float4(w[4] / sum, w[5] / sum, w[6] / sum, w[7] / sum);
  }
}

extern "C" size_t ts_trilinear_workspace_bytes(int64_t n_vox) { return ts_table_bytes(n_vox < 0 ? 0 : n_vox); }

extern "C" int ts_trilinear_map(const float *points, int64_t n_points, const int32_t *vox_coords, int64_t n_vox,
                                int32_t stride, int32_t *idx, float *weight, void *ws, size_t ws_bytes,
                                ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n_points >= 0 && n_vox >= 0 && stride > 0, TS_ERR_INVALID_ARGUMENT, "ts_trilinear_map: bad sizes");
  TS_REQUIRE(n_vox < (1LL << 30), TS_ERR_UNSUPPORTED, "ts_trilinear_map: too many voxels");
  if (n_points == 0) return TS_OK;
  TS_REQUIRE(points && idx && weight && (vox_coords || n_vox == 0), TS_ERR_INVALID_ARGUMENT,
             "ts_trilinear_map: null pointer");
  TS_REQUIRE(((uintptr_t)points & 15) == 0 && ((uintptr_t)vox_coords & 15) == 0 && ((uintptr_t)idx & 15) == 0 &&
                 ((uintptr_t)weight & 15) == 0,
             TS_ERR_INVALID_ARGUMENT, "ts_trilinear_map: pointers must be 16-byte aligned");
  TsTable t;
  int rc = ts_table_init(&t, n_vox, ws, ws_bytes, stream, nullptr);
  if (rc != TS_OK) return rc;
  if (n_vox > 0) {
    int grid = (int)std::min<int64_t>(ts_cdiv(n_vox, 256), 4096);
    table_insert_coords_kernel<<<grid, 256, 0, stream>>>(t, (const int4 *)vox_coords, n_vox);
    TS_CHECK_LAUNCH("ts_trilinear_map/insert");
  }
  int grid = (int)std::min<int64_t>(ts_cdiv(n_points, 256), 4096);
  trilinear_kernel<<<grid, 256, 0, stream>>>(t, (const float4 *)points, n_points, stride, idx, weight);
  TS_CHECK_LAUNCH("ts_trilinear_map");
  return TS_OK;
}

// ------------------------------------------------------------------ devoxelize run order
// Permutation of the points that brings points of the same interpolation cell (identical 8-corner index tuple)
// next to each other: key = 8 * (first present corner voxel) + (its corner number), which identifies the cell.
// ts_devoxelize_backward_runs walks the points in this order and adds a whole run of equal tuples to the voxel
// gradient once.  Any permutation is valid input there; this one makes the runs long (n / #cells points).
__global__ __launch_bounds__(256) void devox_key_kernel(const int *__restrict__ idx, int64_t n,
                                                        unsigned *__restrict__ keys, int *__restrict__ vals) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 a = *(const int4 *)(idx + i * 8), b = *(const int4 *)(idx + i * 8 + 4);
  const int id[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned key = 0xFFFFFFFFu;
#pragma unroll
  for (int k = 7; k >= 0; --k)
    if (id[k] >= 0) key = ((unsigned)id[k] << 3) | (unsigned)k;
  keys[i] = key;
  vals[i] = (int)i;
}

extern "C" size_t ts_devox_order_workspace_bytes(int64_t n) {
  size_t nn = (size_t)(n < 0 ? 0 : n);
  return 3 * ts_align_up(nn * 4, 256) + ts_align_up(nn * 16 + (4u << 20), 256);
}

extern "C" int ts_devox_order(const int32_t *idx, int64_t n, int64_t n_vox, int32_t *order, void *ws, size_t ws_bytes,
                              ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && n_vox >= 0 && n < (1LL << 31), TS_ERR_INVALID_ARGUMENT, "ts_devox_order: bad sizes");
  TS_REQUIRE(n_vox < (1LL << 28), TS_ERR_UNSUPPORTED, "ts_devox_order: too many voxels for a 32-bit run key");
  if (n == 0) return TS_OK;
  TS_REQUIRE(idx && order && ws, TS_ERR_INVALID_ARGUMENT, "ts_devox_order: null pointer");
  TS_REQUIRE((((uintptr_t)idx) & 15) == 0, TS_ERR_INVALID_ARGUMENT, "ts_devox_order: idx must be 16-byte aligned");
  TS_REQUIRE(ws_bytes >= ts_devox_order_workspace_bytes(n), TS_ERR_INVALID_ARGUMENT, "ts_devox_order: workspace too small");
  char *p = (char *)ws;
  const size_t kb = ts_align_up((size_t)n * 4, 256);
  unsigned *keys = (unsigned *)p;
  unsigned *keys_out = (unsigned *)(p + kb);
  int *vals = (int *)(p + 2 * kb);
  void *tmp = p + 3 * kb;
  size_t tmp_bytes = ws_bytes - 3 * kb, need = 0;
  devox_key_kernel<<<(unsigned)ts_cdiv(n, 256), 256, 0, stream>>>(idx, n, keys, vals);
  TS_CHECK_LAUNCH("ts_devox_order/keys");
  unsigned end_bit = 32;
  TS_CHECK_HIP(rocprim::radix_sort_pairs(nullptr, need, keys, keys_out, vals, order, (size_t)n, 0u, end_bit, stream),
               "radix_sort_pairs size query");
  TS_REQUIRE(need <= tmp_bytes, TS_ERR_INVALID_ARGUMENT, "ts_devox_order: workspace too small for the sort");
  TS_CHECK_HIP(rocprim::radix_sort_pairs(tmp, need, keys, keys_out, vals, order, (size_t)n, 0u, end_bit, stream),
               "radix_sort_pairs");
  return TS_OK;
}


// ------------------------------------------------------------------ devoxelize inverse map (voxel -> contributions)
// CSR transpose of a trilinear map: for every voxel the list of (point, corner) slots that carry a non-zero weight
// onto it, slots in ascending order (stable radix sort on the voxel id).  ts_devoxelize_backward_csr gathers along it:
// every voxel row of the gradient is written once, no atomics, and the summation order is fixed.  Pays off where a
// point touches few voxels on average (stride 1: exactly one; stride 4: ~3); at stride 16 (~4 live corners per point,
// 60 points per voxel) the run-wise atomic kernel reads every gradient row once and stays ahead.
__global__ __launch_bounds__(256) void devox_csr_key_kernel(const int *__restrict__ idx, const float *__restrict__ w,
                                                            int64_t n8, int64_t n_vox, unsigned *__restrict__ keys,
                                                            int *__restrict__ vals) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n8) return;
  const int v = idx[e];
  keys[e] = (v >= 0 && v < n_vox && w[e] != 0.f) ? (unsigned)v : 0xFFFFFFFFu;
  vals[e] = (int)e;
}

__global__ __launch_bounds__(256) void devox_csr_offsets_kernel(const unsigned *__restrict__ keys, int64_t n8,
                                                                int64_t n_vox, int *__restrict__ off) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v > n_vox) return;
  // first position whose key is >= v (keys sorted; dead slots carry 0xFFFFFFFF)
  int64_t lo = 0, hi = n8;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)keys[mid] < v) lo = mid + 1; else hi = mid;
  }
  off[v] = (int)lo;
}

extern "C" size_t ts_devox_csr_workspace_bytes(int64_t n) {
  const size_t n8 = (size_t)(n < 0 ? 0 : n) * 8;
  return 3 * ts_align_up(n8 * 4, 256) + ts_align_up(n8 * 16 + (4u << 20), 256);
}

// idx / weight [n, 8] (a ts_trilinear_map result); offsets [n_vox + 1] and entries [8 n] int32 out: the slots of voxel v are
// entries[offsets[v] .. offsets[v + 1]), slot = point * 8 + corner
extern "C" int ts_devox_csr(const int32_t *idx, const float *weight, int64_t n, int64_t n_vox, int32_t *offsets,
                            int32_t *entries, void *ws, size_t ws_bytes, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && n_vox >= 0 && n < (1LL << 27), TS_ERR_INVALID_ARGUMENT, "ts_devox_csr: bad sizes");
  TS_REQUIRE(n_vox < (1LL << 31) - 1, TS_ERR_UNSUPPORTED, "ts_devox_csr: too many voxels");
  TS_REQUIRE(offsets, TS_ERR_INVALID_ARGUMENT, "ts_devox_csr: null pointer");
  const int64_t n8 = n * 8;
  if (n == 0) {
    TS_CHECK_HIP(hipMemsetAsync(offsets, 0, (size_t)(n_vox + 1) * 4, stream), "ts_devox_csr memset");
    return TS_OK;
  }
  TS_REQUIRE(idx && weight && entries && ws, TS_ERR_INVALID_ARGUMENT, "ts_devox_csr: null pointer");
  TS_REQUIRE(ws_bytes >= ts_devox_csr_workspace_bytes(n), TS_ERR_INVALID_ARGUMENT, "ts_devox_csr: workspace too small");
  char *p = (char *)ws;
  const size_t kb = ts_align_up((size_t)n8 * 4, 256);
  unsigned *keys = (unsigned *)p;
  unsigned *keys_out = (unsigned *)(p + kb);
  int *vals = (int *)(p + 2 * kb);
  void *tmp = p + 3 * kb;
  size_t tmp_bytes = ws_bytes - 3 * kb, need = 0;
  devox_csr_key_kernel<<<(unsigned)ts_cdiv(n8, 256), 256, 0, stream>>>(idx, weight, n8, n_vox, keys, vals);
  TS_CHECK_LAUNCH("ts_devox_csr/keys");
  TS_CHECK_HIP(rocprim::radix_sort_pairs(nullptr, need, keys, keys_out, vals, entries, (size_t)n8, 0u, 32u, stream),
               "radix_sort_pairs size query");
  TS_REQUIRE(need <= tmp_bytes, TS_ERR_INVALID_ARGUMENT, "ts_devox_csr: workspace too small for the sort");
  TS_CHECK_HIP(rocprim::radix_sort_pairs(tmp, need, keys, keys_out, vals, entries, (size_t)n8, 0u, 32u, stream),
               "radix_sort_pairs");
  devox_csr_offsets_kernel<<<(unsigned)ts_cdiv(n_vox + 1, 256), 256, 0, stream>>>(keys_out, n8, n_vox, offsets);
  TS_CHECK_LAUNCH("ts_devox_csr/offsets");
  return TS_OK;
}
