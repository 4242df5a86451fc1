// Dataset-side voxelisation moved on device.
// Reference (host numpy, one CPU worker per scan):
//   pcseg/data/dataset/semantickitti/semantickitti_voxel.py:119-127  (np.round(xyz / vs).astype(int32), min shift)
//   torchsparse utils/quantize.py:9-46                                (ravel_hash + np.unique(return_index, return_inverse))
// Semantics kept: voxel order = ascending (b, x, y, z); representative of a voxel = its FIRST
// point in input order (np.unique's first occurrence); inverse[i] = rank of point i's voxel.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <climits>

#include "common.h"

#define TQ_CBIAS (1 << 17)
#define TQ_CMASK ((1u << 18) - 1)

// ------------------------------------------------------------------ float points -> integer voxel coords
// The per-sample minima are reduced in registers (one sample) or LDS (<= 64 samples) first: one global atomicMin per
// workgroup and coordinate.  (A global atomicMin per point serialises millions of updates on three addresses:
// 3.9 ms for 500k points.)
#define VC_LDS_BATCH 64
__global__ __launch_bounds__(256) void vc_round_kernel(const float *__restrict__ pts, int64_t n, int pstride, float vs,
                                                       const int *__restrict__ batch_idx, int n_batch,
                                                       int *__restrict__ mins, int4 *__restrict__ out) {
  __shared__ int smin[3 * VC_LDS_BATCH];
  const bool lds = mins && n_batch > 1 && n_batch <= VC_LDS_BATCH;
  if (lds) {
    for (int t = threadIdx.x; t < 3 * n_batch; t += blockDim.x) smin[t] = INT_MAX;
    __syncthreads();
  }
  int mx = INT_MAX, my = INT_MAX, mz = INT_MAX;   // n_batch == 1: thread-local minima
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const float *p = pts + i * pstride;
    // np.round == round-half-to-even == rintf; the division is float32 / float32
    int x = (int)rintf(__fdiv_rn(p[0], vs)), y = (int)rintf(__fdiv_rn(p[1], vs)), z = (int)rintf(__fdiv_rn(p[2], vs));
    int b = batch_idx ? batch_idx[i] : 0;
    out[i] = make_int4(x, y, z, b);
    if (mins && b >= 0 && b < n_batch) {
      if (n_batch == 1) {
        mx = min(mx, x);
        my = min(my, y);
        mz = min(mz, z);
      } else if (lds) {
        atomicMin(&smin[3 * b + 0], x);
        atomicMin(&smin[3 * b + 1], y);
        atomicMin(&smin[3 * b + 2], z);
      } else {
        atomicMin(&mins[3 * b + 0], x);
        atomicMin(&mins[3 * b + 1], y);
        atomicMin(&mins[3 * b + 2], z);
      }
    }
  }
  if (!mins) return;
  if (n_batch == 1) {
    // wave minima -> workgroup minima -> ONE atomic triple per workgroup: atomics on the same three addresses serialise
    // in L2 (~4 ns each; one per wave of a 500k-point cloud took 90 us)
    __shared__ int wmin[3][4];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      mx = min(mx, __shfl_xor(mx, d, 64));
      my = min(my, __shfl_xor(my, d, 64));
      mz = min(mz, __shfl_xor(mz, d, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      wmin[0][threadIdx.x >> 6] = mx;
      wmin[1][threadIdx.x >> 6] = my;
      wmin[2][threadIdx.x >> 6] = mz;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
      const int v = min(min(wmin[threadIdx.x][0], wmin[threadIdx.x][1]), min(wmin[threadIdx.x][2], wmin[threadIdx.x][3]));
      if (v != INT_MAX) atomicMin(&mins[threadIdx.x], v);
    }
  } else if (lds) {
    __syncthreads();
    for (int t = threadIdx.x; t < 3 * n_batch; t += blockDim.x)
      if (smin[t] != INT_MAX) atomicMin(&mins[t], smin[t]);
  }
}

__global__ __launch_bounds__(256) void vc_shift_kernel(int4 *__restrict__ c, int64_t n, const int *__restrict__ shift,
                                                       int n_batch) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    int4 v = c[i];
    if (v.w >= 0 && v.w < n_batch) {
      v.x -= shift[3 * v.w];
      v.y -= shift[3 * v.w + 1];
      v.z -= shift[3 * v.w + 2];
      c[i] = v;
    }
  }
}

extern "C" int ts_voxel_coords(const float *points, int64_t n, int32_t point_stride, float voxel_size,
                               const int32_t *batch_idx, int32_t n_batch, const int32_t *shift_in,
                               int32_t *mins_out, int32_t *out_coords, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && point_stride >= 3 && voxel_size > 0.f && n_batch >= 1, TS_ERR_INVALID_ARGUMENT,
             "ts_voxel_coords: bad arguments");
  TS_REQUIRE(shift_in || mins_out, TS_ERR_INVALID_ARGUMENT, "ts_voxel_coords: need shift_in or mins_out");
  if (mins_out && !shift_in)
    TS_CHECK_HIP(hipMemsetAsync(mins_out, 0x7F, (size_t)n_batch * 3 * 4, stream), "voxel_coords memset");
  if (n == 0) return TS_OK;
  TS_REQUIRE(points && out_coords, TS_ERR_INVALID_ARGUMENT, "ts_voxel_coords: null pointer");
  TS_REQUIRE(((uintptr_t)out_coords & 15) == 0, TS_ERR_INVALID_ARGUMENT, "ts_voxel_coords: out must be 16-byte aligned");
  int grid = (int)std::min<int64_t>(ts_cdiv(n, 256), 4096);
  const int rgrid = std::min(grid, 1024);       // the round pass ends in atomics per workgroup: fewer, longer workgroups
  vc_round_kernel<<<rgrid, 256, 0, stream>>>(points, n, point_stride, voxel_size, batch_idx, n_batch,
                                            shift_in ? nullptr : mins_out, (int4 *)out_coords);
  TS_CHECK_LAUNCH("ts_voxel_coords/round");
  vc_shift_kernel<<<grid, 256, 0, stream>>>((int4 *)out_coords, n, shift_in ? shift_in : mins_out, n_batch);
  TS_CHECK_LAUNCH("ts_voxel_coords/shift");
  return TS_OK;
}

// ------------------------------------------------------------------ per-sample minimum of the float coordinates
// out[b, 0..2] = min over the points of sample b of (x, y, z) - what the multi-scan stage clamps every fused cloud to
// (semantickitti_voxel_ms.py:121-124: `points_ms[:, :3] >= points[:, :3].min(0)`), for all samples of a batch in one launch.  A
// float minimum is exact in any order.  Per workgroup the minima of the <= 64 samples are reduced in LDS on order-preserving
// integer images of the floats, then ONE compare-and-swap minimum per (workgroup, sample, coordinate) reaches global memory
// (torch's scatter_reduce_(amin) issues one float atomic per element on 3 B addresses: 21 ms for 240k points).
__device__ __forceinline__ int sm_enc(float f) {
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__global__ __launch_bounds__(256) void seg_min3_kernel(const float *__restrict__ pts, int64_t n, int pstride,
                                                       const int64_t *__restrict__ seg, int n_seg, float *__restrict__ out) {
  __shared__ int smin[3 * VC_LDS_BATCH];
  for (int t = threadIdx.x; t < 3 * n_seg; t += blockDim.x) smin[t] = 0x7F800000;      // +inf
  __syncthreads();
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const int b = (int)seg[i];
    if (b < 0 || b >= n_seg) continue;
    const float *p = pts + i * pstride;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float v = p[d];
      if (v == v) atomicMin(&smin[3 * b + d], sm_enc(v));            // (NaN never wins, as in numpy's min of finite data)
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 3 * n_seg; t += blockDim.x) {
    const int e = smin[t];
    if (e == 0x7F800000) continue;
    const float v = __int_as_float(e >= 0 ? e : e ^ 0x7FFFFFFF);
    int *addr = (int *)(out + t);
    int old = *addr;
    while (v < __int_as_float(old)) {
      const int assumed = old;
      old = atomicCAS(addr, assumed, __float_as_int(v));
      if (old == assumed) break;
    }
  }
}

extern "C" int ts_segment_min3(const float *points, int64_t n, int32_t point_stride, const int64_t *seg, int32_t n_seg,
                               float *out, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && point_stride >= 3 && n_seg >= 1 && n_seg <= VC_LDS_BATCH, TS_ERR_INVALID_ARGUMENT,
             "ts_segment_min3: need point_stride >= 3 and 1 .. 64 segments");
  TS_REQUIRE(out, TS_ERR_INVALID_ARGUMENT, "ts_segment_min3: null pointer");
  const TsFillSeg fill = {out, (size_t)n_seg * 3 * 4, 0x7F800000u};                    // +inf
  const int rc = ts_fill_segments(&fill, 1, stream);
  if (rc != TS_OK) return rc;
  if (n == 0) return TS_OK;
  TS_REQUIRE(points && seg, TS_ERR_INVALID_ARGUMENT, "ts_segment_min3: null pointer");
  const int grid = (int)std::min<int64_t>(ts_cdiv(n, 256), 512);
  seg_min3_kernel<<<grid, 256, 0, stream>>>(points, n, point_stride, seg, n_seg, out);
  TS_CHECK_LAUNCH("ts_segment_min3");
  return TS_OK;
}

// ------------------------------------------------------------------ sparse_quantize
__global__ __launch_bounds__(256) void sq_pack_kernel(const int4 *__restrict__ c, int64_t n, uint64_t *__restrict__ keys,
                                                      int *__restrict__ vals, int *__restrict__ err) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    int4 v = c[i];
    bool ok = (unsigned)(v.x + TQ_CBIAS) <= TQ_CMASK && (unsigned)(v.y + TQ_CBIAS) <= TQ_CMASK &&
              (unsigned)(v.z + TQ_CBIAS) <= TQ_CMASK && (unsigned)v.w < 1024u;
    if (!ok) *err = 1;
    keys[i] = ((uint64_t)(unsigned)v.w << 54) | ((uint64_t)(unsigned)(v.x + TQ_CBIAS) << 36) |
              ((uint64_t)(unsigned)(v.y + TQ_CBIAS) << 18) | (uint64_t)(unsigned)(v.z + TQ_CBIAS);
    vals[i] = (int)i;
  }
}

__global__ __launch_bounds__(256) void sq_flag_kernel(const uint64_t *__restrict__ keys, int64_t n,
                                                      unsigned *__restrict__ flags) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// ranks = inclusive scan of head flags; rank-1 = voxel id of sorted position i
__global__ __launch_bounds__(256) void sq_scatter_kernel(const uint64_t *__restrict__ keys, const int *__restrict__ vals,
                                                         const unsigned *__restrict__ ranks, int64_t n,
                                                         const int *__restrict__ err, int *__restrict__ out_index,
                                                         int *__restrict__ out_inverse, int *__restrict__ out_count) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  if (i == 0) *out_count = (*err) ? -1 : (int)ranks[n - 1];
  for (; i < n; i += step) {
    int v = (int)ranks[i] - 1;
    bool head = (i == 0) || keys[i] != keys[i - 1];
    int src = vals[i];
    // stable radix sort: inside a run of equal keys the original indices ascend,
    // so the head of the run is the voxel's first point
    if (head && out_index) out_index[v] = src;
    if (out_inverse) out_inverse[src] = v;
  }
}

extern "C" size_t ts_quantize_workspace_bytes(int64_t n) {
  size_t nn = (size_t)(n < 1 ? 1 : n);
  return 2 * ts_align_up(nn * 8, 256) + 4 * ts_align_up(nn * 4, 256) + 256 + ts_align_up(nn * 24 + (4u << 20), 256);
}

extern "C" int ts_sparse_quantize(const int32_t *coords, int64_t n, int32_t *out_index, int32_t *out_inverse,
                                  int32_t *out_count, void *ws, size_t ws_bytes, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && n < (1LL << 30), TS_ERR_INVALID_ARGUMENT, "ts_sparse_quantize: bad n");
  TS_REQUIRE(out_count, TS_ERR_INVALID_ARGUMENT, "ts_sparse_quantize: null out_count");
  if (n == 0) {
    TS_CHECK_HIP(hipMemsetAsync(out_count, 0, 4, stream), "quantize memset");
    return TS_OK;
  }
  TS_REQUIRE(coords, TS_ERR_INVALID_ARGUMENT, "ts_sparse_quantize: null coords");
  TS_REQUIRE(((uintptr_t)coords & 15) == 0, TS_ERR_INVALID_ARGUMENT, "ts_sparse_quantize: coords must be 16-byte aligned");
  TS_REQUIRE(ws && ws_bytes >= ts_quantize_workspace_bytes(n), TS_ERR_WORKSPACE_TOO_SMALL,
             "ts_sparse_quantize: workspace %zu < %zu", ws_bytes, ts_quantize_workspace_bytes(n));
  TS_REQUIRE(((uintptr_t)ws & 255) == 0, TS_ERR_INVALID_ARGUMENT, "workspace must be 256-byte aligned");
  size_t nn = (size_t)n;
  size_t kb = ts_align_up(nn * 8, 256), vb = ts_align_up(nn * 4, 256);
  char *p = (char *)ws;
  uint64_t *ka = (uint64_t *)p;
  p += kb;
  uint64_t *kbuf = (uint64_t *)p;
  p += kb;
  int *va = (int *)p;
  p += vb;
  int *vbuf = (int *)p;
  p += vb;
  unsigned *flags = (unsigned *)p;
  p += vb;
  unsigned *ranks = (unsigned *)p;
  p += vb;
  int *err = (int *)p;
  p += 256;
  void *tmp = p;
  size_t tmp_bytes = ws_bytes - (size_t)(p - (char *)ws);

  TS_CHECK_HIP(hipMemsetAsync(err, 0, 256, stream), "quantize memset");
  int grid = (int)std::min<int64_t>(ts_cdiv(n, 256), 4096);
  sq_pack_kernel<<<grid, 256, 0, stream>>>((const int4 *)coords, n, ka, va, err);
  TS_CHECK_LAUNCH("ts_sparse_quantize/pack");
  size_t need = 0;
  TS_CHECK_HIP(rocprim::radix_sort_pairs(nullptr, need, ka, kbuf, va, vbuf, nn, 0u, 64u, stream), "sort size query");
  TS_REQUIRE(need <= tmp_bytes, TS_ERR_WORKSPACE_TOO_SMALL, "ts_sparse_quantize: sort scratch %zu > %zu", need, tmp_bytes);
  size_t tb = tmp_bytes;
  TS_CHECK_HIP(rocprim::radix_sort_pairs(tmp, tb, ka, kbuf, va, vbuf, nn, 0u, 64u, stream), "radix_sort_pairs");
  sq_flag_kernel<<<grid, 256, 0, stream>>>(kbuf, n, flags);
  TS_CHECK_LAUNCH("ts_sparse_quantize/flag");
  need = 0;
  TS_CHECK_HIP(rocprim::inclusive_scan(nullptr, need, flags, ranks, nn, rocprim::plus<unsigned>(), stream),
               "scan size query");
  TS_REQUIRE(need <= tmp_bytes, TS_ERR_WORKSPACE_TOO_SMALL, "ts_sparse_quantize: scan scratch too small");
  tb = tmp_bytes;
  TS_CHECK_HIP(rocprim::inclusive_scan(tmp, tb, flags, ranks, nn, rocprim::plus<unsigned>(), stream), "inclusive_scan");
  sq_scatter_kernel<<<grid, 256, 0, stream>>>(kbuf, vbuf, ranks, n, err, out_index, out_inverse, out_count);
  TS_CHECK_LAUNCH("ts_sparse_quantize/scatter");
  return TS_OK;
}
