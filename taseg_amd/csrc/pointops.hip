// Point <-> voxel feature transfer and the multi-scan pose fuse.
// Reference semantics: torchsparse backend/voxelize/voxelize_cuda.cu,
// backend/devoxelize/devoxelize_cuda.cu (the CUDA files are the authority: the CPU
// devoxelize backward is buggy), and
// pcseg/data/dataset/semantickitti/semantickitti_ms.py:403-417.
//
// All four feature kernels are HBM-bound row movers: a group of lanes owns one
// row and walks its channels with 16-byte accesses where C % 4 == 0.
#include "common.h"

// four consecutive channels of a row stored as float32 or IEEE half; arithmetic is float32 either way and a half
// result is rounded once at the store (what `.half()` of the float32 result gives)
typedef _Float16 ts_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float *p) { return *(const float4 *)p; }
__device__ __forceinline__ float4 ld4(const _Float16 *p) {
  const ts_h4 v = *(const ts_h4 *)p;
  return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
}
__device__ __forceinline__ void st4(float *p, float4 v) { *(float4 *)p = v; }
__device__ __forceinline__ void st4(_Float16 *p, float4 v) {
  ts_h4 o;
  o.x = (_Float16)v.x; o.y = (_Float16)v.y; o.z = (_Float16)v.z; o.w = (_Float16)v.w;
  *(ts_h4 *)p = o;
}

// ------------------------------------------------------------------ voxelize
// forward: out[idx[i]] += feat[i] / counts[idx[i]]   (float atomics, like the reference)
__global__ __launch_bounds__(256) void voxelize_fwd_kernel(const float *__restrict__ feat,
                                                           const int *__restrict__ idx,
                                                           const int *__restrict__ counts, int64_t n, int c,
                                                           int64_t m, float *__restrict__ out) {
  int64_t total = n * c;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; e < total; e += step) {
    int64_t i = e / c;
    int j = (int)(e - i * c);
    int pos = idx[i];
    if (pos < 0 || pos >= m) continue;
    int cnt = counts[pos];
    if (cnt == 0) continue;
    atomicAdd(&out[(int64_t)pos * c + j], feat[e] / (float)cnt);
  }
}

__global__ __launch_bounds__(256) void voxelize_bwd_kernel(const float *__restrict__ gout,
                                                           const int *__restrict__ idx,
                                                           const int *__restrict__ counts, int64_t n, int c,
                                                           int64_t m, float *__restrict__ gfeat) {
  int64_t total = n * c;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; e < total; e += step) {
    int64_t i = e / c;
    int j = (int)(e - i * c);
    int pos = idx[i];
    float v = 0.f;
    if (pos >= 0 && pos < m) {
      int cnt = counts[pos];
      if (cnt != 0) v = gout[(int64_t)pos * c + j] / (float)cnt;
    }
    gfeat[e] = v;
  }
}

extern "C" int ts_voxelize_forward(const float *feat, const int32_t *idx, const int32_t *counts, int64_t n,
                                   int32_t c, int64_t m, float *out, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && m >= 0 && c > 0, TS_ERR_INVALID_ARGUMENT, "ts_voxelize_forward: bad sizes");
  if (m == 0) return TS_OK;
  TS_REQUIRE(out && counts, TS_ERR_INVALID_ARGUMENT, "ts_voxelize_forward: null pointer");
  TS_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)m * c * 4, stream), "voxelize memset");
  if (n == 0) return TS_OK;
  TS_REQUIRE(feat && idx, TS_ERR_INVALID_ARGUMENT, "ts_voxelize_forward: null pointer");
  int grid = (int)std::min<int64_t>(ts_cdiv(n * c, 256), 8192);
  voxelize_fwd_kernel<<<grid, 256, 0, stream>>>(feat, idx, counts, n, c, m, out);
  TS_CHECK_LAUNCH("ts_voxelize_forward");
  return TS_OK;
}

extern "C" int ts_voxelize_backward(const float *grad_out, const int32_t *idx, const int32_t *counts, int64_t n,
                                    int32_t c, int64_t m, float *grad_feat, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && m >= 0 && c > 0, TS_ERR_INVALID_ARGUMENT, "ts_voxelize_backward: bad sizes");
  if (n == 0) return TS_OK;
  TS_REQUIRE(grad_feat && idx && (m == 0 || (grad_out && counts)), TS_ERR_INVALID_ARGUMENT,
             "ts_voxelize_backward: null pointer");
  int grid = (int)std::min<int64_t>(ts_cdiv(n * c, 256), 8192);
  voxelize_bwd_kernel<<<grid, 256, 0, stream>>>(grad_out, idx, counts, n, c, m, grad_feat);
  TS_CHECK_LAUNCH("ts_voxelize_backward");
  return TS_OK;
}

// ------------------------------------------------------------------ devoxelize
// forward: out[i, :] = sum_k w[i,k] * feat[idx[i,k], :], accumulated in k order
// like the reference's `out += w * f` loop (devoxelize_cuda.cu:26-31).
// VEC = 4: lane group of (c/4) lanes per point, float4 loads; VEC = 1 generic.
template <int VEC, typename T = float>
__global__ __launch_bounds__(256) void devoxelize_fwd_kernel(const T *__restrict__ feat,
                                                             const int *__restrict__ idx,
                                                             const float *__restrict__ w, int64_t n, int c,
                                                             T *__restrict__ out, int64_t out_ld) {
  const int cv = c / VEC;
  int64_t total = n * cv;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; e < total; e += step) {
    int64_t i = e / cv;
    int j = (int)(e - i * cv) * VEC;
    const int *ip = idx + i * 8;
    const float *wp = w + i * 8;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int id = ip[k];
      float wk = wp[k];
      if (VEC == 4) {
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id >= 0 && wk != 0.f) f = ld4(feat + (int64_t)id * c + j);  // w == 0: nothing to fetch
        acc[0] += wk * f.x;
        acc[1 % VEC] += wk * f.y;
        acc[2 % VEC] += wk * f.z;
        acc[3 % VEC] += wk * f.w;
      } else {
        float f = 0.f;
        if (id >= 0 && wk != 0.f) f = (float)feat[(int64_t)id * c + j];
        acc[0] += wk * f;
      }
    }
    if (VEC == 4) {
      st4(out + i * out_ld + j, make_float4(acc[0], acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]));
    } else {
      out[i * out_ld + j] = (T)acc[0];
    }
  }
}

// backward: gfeat[idx[i,k], :] += w[i,k] * gout[i, :]   (float atomics)
__global__ __launch_bounds__(256) void devoxelize_bwd_kernel(const float *__restrict__ gout,
                                                             const int *__restrict__ idx,
                                                             const float *__restrict__ w, int64_t n, int c,
                                                             int64_t m, float *__restrict__ gfeat) {
  int64_t total = n * c;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; e < total; e += step) {
    int64_t i = e / c;
    int j = (int)(e - i * c);
    float g = gout[e];
    const int *ip = idx + i * 8;
    const float *wp = w + i * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int id = ip[k];
      // corners with zero weight (points sitting on a voxel centre: 7 of 8) add nothing: skip the atomic
      if (id >= 0 && id < m && wp[k] != 0.f) atomicAdd(&gfeat[(int64_t)id * c + j], wp[k] * g);
    }
  }
}

// backward over runs of points with identical corner tuples (points of one interpolation cell): the lane group of
// a run accumulates w[i,k] * gout[i,:] for the 8 corners in registers and issues the atomics once per run.
// `order` (optional) is the walk order (ts_devox_order groups equal tuples); each group of c/4 lanes takes
// `run_len` consecutive positions.  With n / #cells points per cell this divides the atomic count - and the
// contention on the few coarse voxels (stride 16: ~70 points per cell, 8 x 256 atomics per point before) - by
// the run length.
__global__ __launch_bounds__(256) void devoxelize_bwd_runs_kernel(const float *__restrict__ gout,
                                                                  const int *__restrict__ idx,
                                                                  const float *__restrict__ w,
                                                                  const int *__restrict__ order, int64_t n, int c,
                                                                  int64_t m, int run_len, float *__restrict__ gfeat,
                                                                  int64_t go_ld) {
  const int cq = c >> 2, groups = 256 / cq;
  const int grp = threadIdx.x / cq, lane = threadIdx.x - grp * cq;
  if (grp >= groups) return;
  const int64_t p_beg = ((int64_t)blockIdx.x * groups + grp) * run_len;
  const int64_t p_end = min(n, p_beg + run_len);
  int cur[8];
  float4 acc[8];
  unsigned live = 0;  // corners that received a non-zero weight in the current run
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    cur[k] = -1;
    acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  auto flush = [&]() {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if ((live >> k) & 1) {
        float *dst = gfeat + (int64_t)cur[k] * c + 4 * lane;
        atomicAdd(dst + 0, acc[k].x);
        atomicAdd(dst + 1, acc[k].y);
        atomicAdd(dst + 2, acc[k].z);
        atomicAdd(dst + 3, acc[k].w);
      }
      acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    live = 0;
  };
  for (int64_t p = p_beg; p < p_end; ++p) {
    const int64_t i = order ? order[p] : p;
    const int4 ia = *(const int4 *)(idx + i * 8), ib = *(const int4 *)(idx + i * 8 + 4);
    const float4 wa = *(const float4 *)(w + i * 8), wb = *(const float4 *)(w + i * 8 + 4);
    const int id[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
    const float wk[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
    bool same = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) same = same && (id[k] == cur[k]);
    if (!same) {
      if (live) flush();
#pragma unroll
      for (int k = 0; k < 8; ++k) cur[k] = id[k];
    }
    const float4 g = *(const float4 *)(gout + i * go_ld + 4 * lane);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (id[k] >= 0 && id[k] < m && wk[k] != 0.f) {
        acc[k].x += wk[k] * g.x;
        acc[k].y += wk[k] * g.y;
        acc[k].z += wk[k] * g.z;
        acc[k].w += wk[k] * g.w;
        live |= 1u << k;
      }
    }
  }
  if (live) flush();
}

extern "C" int ts_devoxelize_forward(const float *feat, const int32_t *idx, const float *weight, int64_t n,
                                     int32_t c, int64_t m, float *out, ts_stream_t stream_) {
  return ts_devoxelize_forward_ld(feat, idx, weight, n, c, m, out, c, stream_);
}

// out rows `out_ld` floats apart (out_ld >= c): writes a column block of a wider matrix in place of a later torch.cat
extern "C" int ts_devoxelize_forward_ld(const float *feat, const int32_t *idx, const float *weight, int64_t n, int32_t c,
                                        int64_t m, float *out, int64_t out_ld, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && m >= 0 && c > 0 && out_ld >= c, TS_ERR_INVALID_ARGUMENT, "ts_devoxelize_forward: bad sizes");
  if (n == 0) return TS_OK;
  TS_REQUIRE(out && idx && weight && (feat || m == 0), TS_ERR_INVALID_ARGUMENT,
             "ts_devoxelize_forward: null pointer");
  bool vec = (c % 4 == 0) && (out_ld % 4 == 0) && (((uintptr_t)feat & 15) == 0) && (((uintptr_t)out & 15) == 0);
  if (vec) {
    int grid = (int)std::min<int64_t>(ts_cdiv(n * (c / 4), 256), 16384);
    devoxelize_fwd_kernel<4><<<grid, 256, 0, stream>>>(feat, idx, weight, n, c, out, out_ld);
  } else {
    int grid = (int)std::min<int64_t>(ts_cdiv(n * c, 256), 16384);
    devoxelize_fwd_kernel<1><<<grid, 256, 0, stream>>>(feat, idx, weight, n, c, out, out_ld);
  }
  TS_CHECK_LAUNCH("ts_devoxelize_forward");
  return TS_OK;
}

extern "C" int ts_devoxelize_backward(const float *grad_out, const int32_t *idx, const float *weight, int64_t n,
                                      int32_t c, int64_t m, float *grad_feat, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && m >= 0 && c > 0, TS_ERR_INVALID_ARGUMENT, "ts_devoxelize_backward: bad sizes");
  if (m == 0) return TS_OK;
  TS_REQUIRE(grad_feat, TS_ERR_INVALID_ARGUMENT, "ts_devoxelize_backward: null pointer");
  TS_CHECK_HIP(hipMemsetAsync(grad_feat, 0, (size_t)m * c * 4, stream), "devoxelize memset");
  if (n == 0) return TS_OK;
  TS_REQUIRE(grad_out && idx && weight, TS_ERR_INVALID_ARGUMENT, "ts_devoxelize_backward: null pointer");
  int grid = (int)std::min<int64_t>(ts_cdiv(n * c, 256), 16384);
  devoxelize_bwd_kernel<<<grid, 256, 0, stream>>>(grad_out, idx, weight, n, c, m, grad_feat);
  TS_CHECK_LAUNCH("ts_devoxelize_backward");
  return TS_OK;
}

// ------------------------------------------------------------------ multi-scan pose fuse
// new = sum_k hp[k] * pose^T[k][:]  (k = 0..3, summed in k order like np.sum(axis=1));
// out = sum_k (new - t0)[k] * R0[k][:]  (k = 0..2).  float32 throughout.
__global__ __launch_bounds__(256) void fuse_scan_kernel(const float4 *__restrict__ pts, int64_t n,
                                                        const float *__restrict__ pose0,
                                                        const float *__restrict__ pose, float4 *__restrict__ out) {
#pragma clang fp contract(off)  // numpy multiplies, then adds: no fused multiply-add anywhere below
  // 32 wave-uniform scalars
  float P[16], Q[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    P[i] = pose[i];
    Q[i] = pose0[i];
  }
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    float4 p = pts[i];
    float h[4] = {p.x, p.y, p.z, 1.0f};
    float nw[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      // (hpoints[:, k, None] * pose.T[k, j]) summed over k;  pose.T[k][j] = pose[j][k]
      float s = __fmul_rn(h[0], P[j * 4 + 0]);
      s = __fadd_rn(s, __fmul_rn(h[1], P[j * 4 + 1]));
      s = __fadd_rn(s, __fmul_rn(h[2], P[j * 4 + 2]));
      s = __fadd_rn(s, __fmul_rn(h[3], P[j * 4 + 3]));
      nw[j] = __fsub_rn(s, Q[j * 4 + 3]);  // - pose0[:3, 3]
    }
    float o[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      // sum_k new[k] * pose0[:3, :3][k][j]
      float s = __fmul_rn(nw[0], Q[0 * 4 + j]);
      s = __fadd_rn(s, __fmul_rn(nw[1], Q[1 * 4 + j]));
      s = __fadd_rn(s, __fmul_rn(nw[2], Q[2 * 4 + j]));
      o[j] = s;
    }
    out[i] = make_float4(o[0], o[1], o[2], p.w);
  }
}

// All history scans of a sample in one launch: point i uses poses[scan_idx[i]].  Same arithmetic, same order.
// PER_SCAN: pose0 is a table too (pose0[scan_idx[i]]) - the history scans of a whole BATCH of samples in one launch, every scan
// with the current-frame pose of its own sample.
template <bool PER_SCAN>
__global__ __launch_bounds__(256) void fuse_scans_kernel(const float4 *__restrict__ pts,
                                                         const int *__restrict__ scan_idx, int64_t n,
                                                         const float *__restrict__ pose0,
                                                         const float *__restrict__ poses, int n_scans,
                                                         float4 *__restrict__ out) {
#pragma clang fp contract(off)
  float Q[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) Q[i] = PER_SCAN ? 0.f : pose0[i];
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const float4 p = pts[i];
    const int sidx = min(max(scan_idx[i], 0), n_scans - 1);
    const float *P = poses + 16 * sidx;
    if (PER_SCAN) {
#pragma unroll
      for (int q = 0; q < 16; ++q) Q[q] = pose0[16 * sidx + q];
    }
    const float h[4] = {p.x, p.y, p.z, 1.0f};
    float nw[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float s = __fmul_rn(h[0], P[j * 4 + 0]);
      s = __fadd_rn(s, __fmul_rn(h[1], P[j * 4 + 1]));
      s = __fadd_rn(s, __fmul_rn(h[2], P[j * 4 + 2]));
      s = __fadd_rn(s, __fmul_rn(h[3], P[j * 4 + 3]));
      nw[j] = __fsub_rn(s, Q[j * 4 + 3]);
    }
    float o[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float s = __fmul_rn(nw[0], Q[0 * 4 + j]);
      s = __fadd_rn(s, __fmul_rn(nw[1], Q[1 * 4 + j]));
      s = __fadd_rn(s, __fmul_rn(nw[2], Q[2 * 4 + j]));
      o[j] = s;
    }
    out[i] = make_float4(o[0], o[1], o[2], p.w);
  }
}

extern "C" int ts_fuse_scans(const float *points, const int32_t *scan_idx, int64_t n, const float *pose0,
                             const float *poses, int32_t n_scans, float *out, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && n_scans > 0, TS_ERR_INVALID_ARGUMENT, "ts_fuse_scans: bad sizes");
  if (n == 0) return TS_OK;
  TS_REQUIRE(points && scan_idx && pose0 && poses && out, TS_ERR_INVALID_ARGUMENT, "ts_fuse_scans: null pointer");
  TS_REQUIRE(((uintptr_t)points & 15) == 0 && ((uintptr_t)out & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_fuse_scans: points/out must be 16-byte aligned");
  int grid = (int)std::min<int64_t>(ts_cdiv(n, 256), 4096);
  fuse_scans_kernel<false><<<grid, 256, 0, stream>>>((const float4 *)points, scan_idx, n, pose0, poses, n_scans,
                                                     (float4 *)out);
  TS_CHECK_LAUNCH("ts_fuse_scans");
  return TS_OK;
}

// the history scans of a whole batch: pose0s [n_scans, 16] = the current-frame pose of the sample scan s belongs to
extern "C" int ts_fuse_scans_batch(const float *points, const int32_t *scan_idx, int64_t n, const float *pose0s,
                                   const float *poses, int32_t n_scans, float *out, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && n_scans > 0, TS_ERR_INVALID_ARGUMENT, "ts_fuse_scans_batch: bad sizes");
  if (n == 0) return TS_OK;
  TS_REQUIRE(points && scan_idx && pose0s && poses && out, TS_ERR_INVALID_ARGUMENT, "ts_fuse_scans_batch: null pointer");
  TS_REQUIRE(((uintptr_t)points & 15) == 0 && ((uintptr_t)out & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_fuse_scans_batch: points/out must be 16-byte aligned");
  int grid = (int)std::min<int64_t>(ts_cdiv(n, 256), 4096);
  fuse_scans_kernel<true><<<grid, 256, 0, stream>>>((const float4 *)points, scan_idx, n, pose0s, poses, n_scans,
                                                    (float4 *)out);
  TS_CHECK_LAUNCH("ts_fuse_scans_batch");
  return TS_OK;
}

// nuScenes multi-scan fuse (nuscenes_ms.py:280-318, 348-373): see include/taseg_hip.h.  One thread per point; the 28
// doubles of a sweep are wave-uniform for long runs of points (sweeps are concatenated).
__global__ __launch_bounds__(256) void fuse_sweeps_kernel(const float *__restrict__ pts,
                                                          const int *__restrict__ sweep_idx, int64_t n,
                                                          const double *__restrict__ params, int n_sweeps,
                                                          float *__restrict__ out, uint8_t *__restrict__ keep) {
#pragma clang fp contract(off)  // only the explicit fma() below fuse: numpy's dgemm accumulates x_k * R_kj with FMAs in k order
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const float *p = pts + i * 5;
    const float x = p[0], y = p[1], z = p[2];
    const int s = min(max(sweep_idx[i], 0), n_sweeps - 1);
    const double *P = params + 28 * s;
    keep[i] = !(fabsf(x) < 1.0f && fabsf(y) < 1.5f);
    float q[3] = {x, y, z};
    if (P[12] != 0.0) {        // p @ A^T : out_j = sum_k p_k A[j][k]; then += a (each stored as float32)
      const double d0 = q[0], d1 = q[1], d2 = q[2];
      float r[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double acc = __builtin_fma(d2, P[3 * j + 2], __builtin_fma(d1, P[3 * j + 1], d0 * P[3 * j + 0]));
        r[j] = (float)acc;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) q[j] = (float)((double)r[j] + P[9 + j]);
    }
    if (P[25] != 0.0) {        // p @ B + b : out_j = sum_k p_k B[k][j] + b_j
      const double d0 = q[0], d1 = q[1], d2 = q[2];
      const double *B = P + 13;
      float r[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double acc = __builtin_fma(d2, B[6 + j], __builtin_fma(d1, B[3 + j], d0 * B[j]));
        r[j] = (float)(acc + B[9 + j]);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) q[j] = r[j];
    }
    float *o = out + i * 5;
    o[0] = q[0];
    o[1] = q[1];
    o[2] = q[2];
    o[3] = p[3];
    o[4] = (float)P[26];
  }
}

extern "C" int ts_fuse_sweeps(const float *points, const int32_t *sweep_idx, int64_t n, const double *params,
                              int32_t n_sweeps, float *out, uint8_t *keep, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && n_sweeps > 0, TS_ERR_INVALID_ARGUMENT, "ts_fuse_sweeps: bad sizes");
  if (n == 0) return TS_OK;
  TS_REQUIRE(points && sweep_idx && params && out && keep, TS_ERR_INVALID_ARGUMENT, "ts_fuse_sweeps: null pointer");
  const int grid = (int)std::min<int64_t>(ts_cdiv(n, 256), 4096);
  fuse_sweeps_kernel<<<grid, 256, 0, stream>>>(points, sweep_idx, n, params, n_sweeps, out, keep);
  TS_CHECK_LAUNCH("ts_fuse_sweeps");
  return TS_OK;
}

// TIAF camera projection (semantickitti_ms_mm.py:419-457): see include/taseg_hip.h
__global__ __launch_bounds__(256) void project_fov_kernel(const float4 *__restrict__ pts, int64_t n,
                                                          const double *__restrict__ proj, int img_w, int img_h,
                                                          int crop_h, int crop_w, float row_offset,
                                                          float2 *__restrict__ pix, uint8_t *__restrict__ keep) {
#pragma clang fp contract(off)
  double P[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) P[i] = proj[i];
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const float4 p = pts[i];
    const double x = p.x, y = p.y, z = p.z;
    double uvz[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)      // sum_k P[j][k] * (x, y, z, 1)[k], accumulated with FMAs in k order
      uvz[j] = __builtin_fma(P[4 * j + 3], 1.0, __builtin_fma(P[4 * j + 2], z, __builtin_fma(P[4 * j + 1], y, P[4 * j] * x)));
    const double u = uvz[0] / uvz[2], v = uvz[1] / uvz[2];
    bool ok = p.x > 0.f && u > 0.0 && v > 0.0 && u < (double)img_w && v < (double)img_h;
    int row = 0, col = 0;
    if (ok) {
      row = (int)v;                  // astype(int): truncation
      col = (int)u;
      ok = row < crop_h && col < crop_w;
    }
    keep[i] = ok ? 1 : 0;
    pix[i] = make_float2(__fadd_rn((float)row, row_offset), (float)col);
  }
}

extern "C" int ts_project_fov(const float *points, int64_t n, const double *proj, int32_t img_w, int32_t img_h,
                              int32_t crop_h, int32_t crop_w, float row_offset, float *pix, uint8_t *keep,
                              ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && img_w > 0 && img_h > 0 && crop_h > 0 && crop_w > 0, TS_ERR_INVALID_ARGUMENT, "ts_project_fov: bad sizes");
  if (n == 0) return TS_OK;
  TS_REQUIRE(points && proj && pix && keep, TS_ERR_INVALID_ARGUMENT, "ts_project_fov: null pointer");
  TS_REQUIRE(((((uintptr_t)points) & 15) | (((uintptr_t)pix) & 7)) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_project_fov: points must be 16-byte, pix 8-byte aligned");
  const int grid = (int)std::min<int64_t>(ts_cdiv(n, 256), 4096);
  project_fov_kernel<<<grid, 256, 0, stream>>>((const float4 *)points, n, proj, img_w, img_h, crop_h, crop_w, row_offset,
                                               (float2 *)pix, keep);
  TS_CHECK_LAUNCH("ts_project_fov");
  return TS_OK;
}

// nuScenes TIAF camera projection (nuscenes_ms_mm.py:349-398): see include/taseg_hip.h.  cam = 57 doubles:
//   M1[9] t1[3] (lidar -> ego), M2[9] t2[3] (ego -> global), t3[3] M3[9] (global -> camera ego: subtract, then M3 =
//   rotation^T), t4[3] M4[9] (camera ego -> camera), K[9] (intrinsic; the devkit pads it to 4 x 4 with a zero last column).
// numpy evaluates every 3 x 3 product as an FMA chain over k in ascending order (dgemm), the additions separately.
__global__ __launch_bounds__(256) void project_cam_kernel(const float4 *__restrict__ pts, int64_t n,
                                                          const double *__restrict__ cam, int img_w, int img_h,
                                                          int crop_top, float row_offset, float2 *__restrict__ pix,
                                                          uint8_t *__restrict__ keep) {
#pragma clang fp contract(off)
  double C[57];
#pragma unroll
  for (int i = 0; i < 57; ++i) C[i] = cam[i];
  auto mat = [](const double *M, const double *v, double *o) {
#pragma unroll
    for (int j = 0; j < 3; ++j) o[j] = __builtin_fma(M[3 * j + 2], v[2], __builtin_fma(M[3 * j + 1], v[1], M[3 * j] * v[0]));
  };
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const float4 p = pts[i];
    double a[3] = {(double)p.x, (double)p.y, (double)p.z}, b[3];
    mat(C, a, b);
#pragma unroll
    for (int j = 0; j < 3; ++j) b[j] = b[j] + C[9 + j];
    mat(C + 12, b, a);
#pragma unroll
    for (int j = 0; j < 3; ++j) a[j] = (a[j] + C[21 + j]) - C[24 + j];
    mat(C + 27, a, b);
#pragma unroll
    for (int j = 0; j < 3; ++j) b[j] = b[j] - C[36 + j];
    mat(C + 39, b, a);
    const double depth = a[2];
    // viewpad @ (x, y, z, 1): the 4th column of the padded intrinsic is zero, fma(0, 1, acc) = acc + 0
    double q[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
      q[j] = __builtin_fma(0.0, 1.0, __builtin_fma(C[48 + 3 * j + 2], a[2], __builtin_fma(C[48 + 3 * j + 1], a[1], C[48 + 3 * j] * a[0])));
    const float u = (float)(q[0] / q[2]), v = (float)(q[1] / q[2]);
    bool ok = depth > 0.0 && u > 0.f && u < (float)img_w && v > 0.f && v < (float)img_h;
    int row = 0, col = 0;
    if (ok) {
      row = ((int)v) >> 1;           // astype(int), then floor(0.5 * .) written back into the integer array
      col = ((int)u) >> 1;
      ok = row >= crop_top;
      row -= crop_top;
    }
    keep[i] = ok ? 1 : 0;
    pix[i] = make_float2(__fadd_rn((float)row, row_offset), (float)col);
  }
}

extern "C" int ts_project_cam(const float *points, int64_t n, const double *cam, int32_t img_w, int32_t img_h,
                              int32_t crop_top, float row_offset, float *pix, uint8_t *keep, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && img_w > 0 && img_h > 0 && crop_top >= 0, TS_ERR_INVALID_ARGUMENT, "ts_project_cam: bad sizes");
  if (n == 0) return TS_OK;
  TS_REQUIRE(points && cam && pix && keep, TS_ERR_INVALID_ARGUMENT, "ts_project_cam: null pointer");
  TS_REQUIRE(((((uintptr_t)points) & 15) | (((uintptr_t)pix) & 7)) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_project_cam: points must be 16-byte, pix 8-byte aligned");
  const int grid = (int)std::min<int64_t>(ts_cdiv(n, 256), 4096);
  project_cam_kernel<<<grid, 256, 0, stream>>>((const float4 *)points, n, cam, img_w, img_h, crop_top, row_offset,
                                               (float2 *)pix, keep);
  TS_CHECK_LAUNCH("ts_project_cam");
  return TS_OK;
}

extern "C" int ts_fuse_scan(const float *points, int64_t n, const float *pose0, const float *pose, float *out,
                            ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0, TS_ERR_INVALID_ARGUMENT, "ts_fuse_scan: n < 0");
  if (n == 0) return TS_OK;
  TS_REQUIRE(points && pose0 && pose && out, TS_ERR_INVALID_ARGUMENT, "ts_fuse_scan: null pointer");
  TS_REQUIRE(((uintptr_t)points & 15) == 0 && ((uintptr_t)out & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_fuse_scan: points/out must be 16-byte aligned");
  int grid = (int)std::min<int64_t>(ts_cdiv(n, 256), 4096);
  fuse_scan_kernel<<<grid, 256, 0, stream>>>((const float4 *)points, n, pose0, pose, (float4 *)out);
  TS_CHECK_LAUNCH("ts_fuse_scan");
  return TS_OK;
}

extern "C" int ts_devoxelize_backward_runs(const float *grad_out, const int32_t *idx, const float *weight,
                                           const int32_t *order, int64_t n, int32_t c, int64_t m, float *grad_feat,
                                           ts_stream_t stream_) {
  return ts_devoxelize_backward_runs_ld(grad_out, c, idx, weight, order, n, c, m, grad_feat, stream_);
}

// grad_out rows `go_ld` floats apart: reads a column block of a wider gradient matrix without a contiguous copy
extern "C" int ts_devoxelize_backward_runs_ld(const float *grad_out, int64_t go_ld, const int32_t *idx,
                                              const float *weight, const int32_t *order, int64_t n, int32_t c, int64_t m,
                                              float *grad_feat, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && m >= 0 && c > 0 && go_ld >= c && (go_ld & 3) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_devoxelize_backward_runs: bad sizes");
  TS_REQUIRE((c & 3) == 0 && c <= 1024, TS_ERR_UNSUPPORTED, "ts_devoxelize_backward_runs: C must be a multiple of 4, <= 1024");
  if (m == 0) return TS_OK;
  TS_REQUIRE(grad_feat, TS_ERR_INVALID_ARGUMENT, "ts_devoxelize_backward_runs: null pointer");
  TS_CHECK_HIP(hipMemsetAsync(grad_feat, 0, (size_t)m * c * 4, stream), "devoxelize memset");
  if (n == 0) return TS_OK;
  TS_REQUIRE(grad_out && idx && weight, TS_ERR_INVALID_ARGUMENT, "ts_devoxelize_backward_runs: null pointer");
  TS_REQUIRE(((((uintptr_t)grad_out) | ((uintptr_t)idx) | ((uintptr_t)weight) | ((uintptr_t)grad_feat)) & 15) == 0,
             TS_ERR_INVALID_ARGUMENT, "ts_devoxelize_backward_runs: pointers must be 16-byte aligned");
  const int run_len = 32;
  const int groups = 256 / (c >> 2);
  const int64_t n_groups = ts_cdiv(n, run_len);
  devoxelize_bwd_runs_kernel<<<(unsigned)ts_cdiv(n_groups, groups), 256, 0, stream>>>(grad_out, idx, weight, order, n,
                                                                                      c, m, run_len, grad_feat, go_ld);
  TS_CHECK_LAUNCH("ts_devoxelize_backward_runs");
  return TS_OK;
}


// backward along the inverse map of ts_devox_csr: gfeat[v, :] = sum over the slots (point, corner) of voxel v of
// weight[slot] * gout[point, :].  One group of c / 4 lanes per voxel, slots taken four at a time (independent loads);
// every row is written exactly once (zeros for a voxel without slots): no fill, no atomics, fixed summation order.
// SHIFT = 3: slot = point * 8 + corner, row = point, weighted by w[slot].  SHIFT = 0 (second stage of the cell-reduced
// form below): slot = row of a matrix of partial sums, weight 1.
template <int SHIFT, typename TG = float, typename TO = float>
__global__ __launch_bounds__(256) void devoxelize_bwd_csr_kernel(const TG *__restrict__ gout,
                                                                 const float *__restrict__ w,
                                                                 const int *__restrict__ off,
                                                                 const int *__restrict__ ent, int64_t m, int c,
                                                                 TO *__restrict__ gfeat, int64_t go_ld) {
  const int cq = c >> 2, groups = 256 / cq;
  const int grp = threadIdx.x / cq, lane = threadIdx.x - grp * cq;
  if (grp >= groups) return;
  const int64_t v = (int64_t)blockIdx.x * groups + grp;
  if (v >= m) return;
  const int beg = off[v], end = off[v + 1];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int e = beg;
  for (; e + 4 <= end; e += 4) {
    int s[4];
    float wt[4];
    float4 g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = ent[e + u];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      wt[u] = SHIFT ? w[s[u]] : 1.f;
      g[u] = ld4(gout + (int64_t)(s[u] >> SHIFT) * go_ld + 4 * lane);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc.x += wt[u] * g[u].x; acc.y += wt[u] * g[u].y; acc.z += wt[u] * g[u].z; acc.w += wt[u] * g[u].w;
    }
  }
  for (; e < end; ++e) {
    const int s = ent[e];
    const float wt = SHIFT ? w[s] : 1.f;
    const float4 g = ld4(gout + (int64_t)(s >> SHIFT) * go_ld + 4 * lane);
    acc.x += wt * g.x; acc.y += wt * g.y; acc.z += wt * g.z; acc.w += wt * g.w;
  }
  st4(gfeat + v * c + 4 * lane, acc);
}

extern "C" int ts_devoxelize_backward_csr(const float *grad_out, const float *weight, const int32_t *offsets,
                                          const int32_t *entries, int64_t n, int32_t c, int64_t m, float *grad_feat,
                                          ts_stream_t stream_) {
  return ts_devoxelize_backward_csr_ld(grad_out, c, weight, offsets, entries, n, c, m, grad_feat, stream_);
}

extern "C" int ts_devoxelize_backward_csr_ld(const float *grad_out, int64_t go_ld, const float *weight,
                                             const int32_t *offsets, const int32_t *entries, int64_t n, int32_t c,
                                             int64_t m, float *grad_feat, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && m >= 0 && c > 0 && go_ld >= c && (go_ld & 3) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_devoxelize_backward_csr: bad sizes");
  TS_REQUIRE((c & 3) == 0 && c <= 1024, TS_ERR_UNSUPPORTED, "ts_devoxelize_backward_csr: C must be a multiple of 4, <= 1024");
  if (m == 0) return TS_OK;
  TS_REQUIRE(grad_feat && offsets && (n == 0 || (grad_out && weight && entries)), TS_ERR_INVALID_ARGUMENT,
             "ts_devoxelize_backward_csr: null pointer");
  TS_REQUIRE(((((uintptr_t)grad_out) | ((uintptr_t)grad_feat)) & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_devoxelize_backward_csr: rows must be 16-byte aligned");
  const int groups = 256 / (c >> 2);
  devoxelize_bwd_csr_kernel<3><<<(unsigned)ts_cdiv(m, groups), 256, 0, stream>>>(grad_out, weight, offsets, entries, m, c,
                                                                                grad_feat, go_ld);
  TS_CHECK_LAUNCH("ts_devoxelize_backward_csr");
  return TS_OK;
}

// ------------------------------------------------------------------ cell-reduced backward (coarse strides)
// At stride 16 a voxel collects ~700 (point, corner) contributions and a point feeds ~4 voxels: the inverse-map gather
// above reads every gradient row once per live corner (8 x 246 MB per launch for 240k points x 256 channels).  Points of
// the same interpolation cell share their 8-corner tuple, so the sum is split in two fixed-order stages:
//   1. the points, walked in cell order (ts_devox_order), are cut into segments of equal tuple and at most 64 points
//      (ts_devox_segments); a group of c / 4 lanes per segment forms the 8 weighted sums of its points' gradient rows -
//      every row is read ONCE - and stores them as rows 8 s .. 8 s + 7 of `part`;
//   2. the voxels gather the partial rows along the inverse map of the segments' corner tuples (ts_devox_csr on the
//      [n_seg, 8] tuples): ~8x fewer rows than the point-wise gather.
// No atomics, every output row written once, summation order fixed by the plan.
__global__ __launch_bounds__(256) void devox_seg_flag_kernel(const int *__restrict__ idx, const int *__restrict__ order,
                                                             int64_t n, int max_len, int *__restrict__ flags) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  bool cut = p == 0 || (p % max_len) == 0;
  if (!cut) {
    const int a = order[p], b = order[p - 1];
    const int4 a0 = *(const int4 *)(idx + (int64_t)a * 8), a1 = *(const int4 *)(idx + (int64_t)a * 8 + 4);
    const int4 b0 = *(const int4 *)(idx + (int64_t)b * 8), b1 = *(const int4 *)(idx + (int64_t)b * 8 + 4);
    cut = a0.x != b0.x || a0.y != b0.y || a0.z != b0.z || a0.w != b0.w || a1.x != b1.x || a1.y != b1.y || a1.z != b1.z ||
          a1.w != b1.w;
  }
  flags[p] = cut ? 1 : 0;
}

extern "C" int ts_devox_segments(const int32_t *idx, const int32_t *order, int64_t n, int32_t max_len, int32_t *flags,
                                 ts_stream_t stream_) {
  TS_REQUIRE(n >= 0 && max_len > 0, TS_ERR_INVALID_ARGUMENT, "ts_devox_segments: bad sizes");
  if (n == 0) return TS_OK;
  TS_REQUIRE(idx && order && flags && (((uintptr_t)idx) & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_devox_segments: null or misaligned pointer");
  devox_seg_flag_kernel<<<(unsigned)ts_cdiv(n, 256), 256, 0, (hipStream_t)stream_>>>(idx, order, n, max_len, flags);
  TS_CHECK_LAUNCH("ts_devox_segments");
  return TS_OK;
}

template <typename TG>
__global__ __launch_bounds__(256) void devoxelize_bwd_cells_kernel(const TG *__restrict__ gout,
                                                                   const float *__restrict__ w,
                                                                   const int *__restrict__ order,
                                                                   const int *__restrict__ seg_start, int64_t n_seg,
                                                                   int c, float *__restrict__ part, int64_t go_ld) {
  const int cq = c >> 2, groups = 256 / cq;
  const int grp = threadIdx.x / cq, lane = threadIdx.x - grp * cq;
  if (grp >= groups) return;
  const int64_t s = (int64_t)blockIdx.x * groups + grp;
  if (s >= n_seg) return;
  const int beg = seg_start[s], end = seg_start[s + 1];
  float4 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  int p = beg;
  for (; p + 2 <= end; p += 2) {                       // two rows in flight
    const int i0 = order[p], i1 = order[p + 1];
    const float4 g0 = ld4(gout + (int64_t)i0 * go_ld + 4 * lane);
    const float4 g1 = ld4(gout + (int64_t)i1 * go_ld + 4 * lane);
    const float4 wa0 = *(const float4 *)(w + (int64_t)i0 * 8), wb0 = *(const float4 *)(w + (int64_t)i0 * 8 + 4);
    const float4 wa1 = *(const float4 *)(w + (int64_t)i1 * 8), wb1 = *(const float4 *)(w + (int64_t)i1 * 8 + 4);
    const float w0[8] = {wa0.x, wa0.y, wa0.z, wa0.w, wb0.x, wb0.y, wb0.z, wb0.w};
    const float w1[8] = {wa1.x, wa1.y, wa1.z, wa1.w, wb1.x, wb1.y, wb1.z, wb1.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc[k].x += w0[k] * g0.x; acc[k].y += w0[k] * g0.y; acc[k].z += w0[k] * g0.z; acc[k].w += w0[k] * g0.w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc[k].x += w1[k] * g1.x; acc[k].y += w1[k] * g1.y; acc[k].z += w1[k] * g1.z; acc[k].w += w1[k] * g1.w;
    }
  }
  for (; p < end; ++p) {
    const int i0 = order[p];
    const float4 g0 = ld4(gout + (int64_t)i0 * go_ld + 4 * lane);
    const float4 wa0 = *(const float4 *)(w + (int64_t)i0 * 8), wb0 = *(const float4 *)(w + (int64_t)i0 * 8 + 4);
    const float w0[8] = {wa0.x, wa0.y, wa0.z, wa0.w, wb0.x, wb0.y, wb0.z, wb0.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc[k].x += w0[k] * g0.x; acc[k].y += w0[k] * g0.y; acc[k].z += w0[k] * g0.z; acc[k].w += w0[k] * g0.w;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) *(float4 *)(part + (s * 8 + k) * c + 4 * lane) = acc[k];
}

// grad_feat [m, c] = adjoint of the trilinear map (idx, weight) [n, 8] applied to grad_out [n, ld] through the plan
// (order [n], seg_start [n_seg + 1], offsets [m + 1] / entries of ts_devox_csr on the segments' tuples); part = scratch
// of n_seg * 8 * c floats
extern "C" int ts_devoxelize_backward_cells_ld(const float *grad_out, int64_t go_ld, const float *weight,
                                               const int32_t *order, const int32_t *seg_start, int64_t n_seg,
                                               const int32_t *offsets, const int32_t *entries, int64_t n, int32_t c,
                                               int64_t m, float *part, float *grad_feat, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && m >= 0 && n_seg >= 0 && c > 0 && go_ld >= c && (go_ld & 3) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_devoxelize_backward_cells: bad sizes");
  TS_REQUIRE((c & 3) == 0 && c <= 1024, TS_ERR_UNSUPPORTED, "ts_devoxelize_backward_cells: C must be a multiple of 4, <= 1024");
  if (m == 0) return TS_OK;
  TS_REQUIRE(grad_feat && offsets && (n_seg == 0 || (grad_out && weight && order && seg_start && entries && part)),
             TS_ERR_INVALID_ARGUMENT, "ts_devoxelize_backward_cells: null pointer");
  TS_REQUIRE(((((uintptr_t)grad_out) | ((uintptr_t)grad_feat) | ((uintptr_t)part) | ((uintptr_t)weight)) & 15) == 0,
             TS_ERR_INVALID_ARGUMENT, "ts_devoxelize_backward_cells: rows must be 16-byte aligned");
  const int groups = 256 / (c >> 2);
  if (n_seg > 0) {
    devoxelize_bwd_cells_kernel<float><<<(unsigned)ts_cdiv(n_seg, groups), 256, 0, stream>>>(grad_out, weight, order,
                                                                                            seg_start, n_seg, c, part, go_ld);
    TS_CHECK_LAUNCH("ts_devoxelize_backward_cells/partials");
  }
  devoxelize_bwd_csr_kernel<0><<<(unsigned)ts_cdiv(m, groups), 256, 0, stream>>>(part, nullptr, offsets, entries, m, c,
                                                                                grad_feat, c);
  TS_CHECK_LAUNCH("ts_devoxelize_backward_cells/gather");
  return TS_OK;
}

// ------------------------------------------------------------------ half-storage forms (the AMP path)
// Point / voxel feature rows and their gradients stored as IEEE half, every sum in float32, one rounding at the store:
// bit for bit what the float32 entry points give between a `.float()` of the inputs and a `.half()` of the result -
// without those two passes over the matrices and with half the bytes gathered.  `void *` = half arrays, leading
// dimensions in elements.  C % 4 == 0, rows 8-byte aligned.
extern "C" int ts_devoxelize_forward_f16_ld(const void *feat, const int32_t *idx, const float *weight, int64_t n, int32_t c,
                                            int64_t m, void *out, int64_t out_ld, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && m >= 0 && c > 0 && out_ld >= c, TS_ERR_INVALID_ARGUMENT, "ts_devoxelize_forward_f16: bad sizes");
  TS_REQUIRE((c & 3) == 0 && (out_ld & 3) == 0, TS_ERR_UNSUPPORTED, "ts_devoxelize_forward_f16: C and ld must be multiples of 4");
  if (n == 0) return TS_OK;
  TS_REQUIRE(out && idx && weight && (feat || m == 0), TS_ERR_INVALID_ARGUMENT, "ts_devoxelize_forward_f16: null pointer");
  TS_REQUIRE(((((uintptr_t)feat) | ((uintptr_t)out)) & 7) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_devoxelize_forward_f16: rows must be 8-byte aligned");
  int grid = (int)std::min<int64_t>(ts_cdiv(n * (c / 4), 256), 16384);
  devoxelize_fwd_kernel<4, _Float16><<<grid, 256, 0, stream>>>((const _Float16 *)feat, idx, weight, n, c, (_Float16 *)out,
                                                               out_ld);
  TS_CHECK_LAUNCH("ts_devoxelize_forward_f16");
  return TS_OK;
}

extern "C" int ts_devoxelize_backward_csr_f16_ld(const void *grad_out, int64_t go_ld, const float *weight,
                                                 const int32_t *offsets, const int32_t *entries, int64_t n, int32_t c,
                                                 int64_t m, void *grad_feat, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && m >= 0 && c > 0 && go_ld >= c && (go_ld & 3) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_devoxelize_backward_csr_f16: bad sizes");
  TS_REQUIRE((c & 3) == 0 && c <= 1024, TS_ERR_UNSUPPORTED, "ts_devoxelize_backward_csr_f16: C must be a multiple of 4, <= 1024");
  if (m == 0) return TS_OK;
  TS_REQUIRE(grad_feat && offsets && (n == 0 || (grad_out && weight && entries)), TS_ERR_INVALID_ARGUMENT,
             "ts_devoxelize_backward_csr_f16: null pointer");
  TS_REQUIRE(((((uintptr_t)grad_out) | ((uintptr_t)grad_feat)) & 7) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_devoxelize_backward_csr_f16: rows must be 8-byte aligned");
  const int groups = 256 / (c >> 2);
  devoxelize_bwd_csr_kernel<3, _Float16, _Float16><<<(unsigned)ts_cdiv(m, groups), 256, 0, stream>>>(
      (const _Float16 *)grad_out, weight, offsets, entries, m, c, (_Float16 *)grad_feat, go_ld);
  TS_CHECK_LAUNCH("ts_devoxelize_backward_csr_f16");
  return TS_OK;
}

// `part` stays float32 (n_seg * 8 * c floats): only the two ends of the two-stage sum are half
extern "C" int ts_devoxelize_backward_cells_f16_ld(const void *grad_out, int64_t go_ld, const float *weight,
                                                   const int32_t *order, const int32_t *seg_start, int64_t n_seg,
                                                   const int32_t *offsets, const int32_t *entries, int64_t n, int32_t c,
                                                   int64_t m, float *part, void *grad_feat, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n >= 0 && m >= 0 && n_seg >= 0 && c > 0 && go_ld >= c && (go_ld & 3) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_devoxelize_backward_cells_f16: bad sizes");
  TS_REQUIRE((c & 3) == 0 && c <= 1024, TS_ERR_UNSUPPORTED, "ts_devoxelize_backward_cells_f16: C must be a multiple of 4, <= 1024");
  if (m == 0) return TS_OK;
  TS_REQUIRE(grad_feat && offsets && (n_seg == 0 || (grad_out && weight && order && seg_start && entries && part)),
             TS_ERR_INVALID_ARGUMENT, "ts_devoxelize_backward_cells_f16: null pointer");
  TS_REQUIRE(((((uintptr_t)grad_out) | ((uintptr_t)grad_feat)) & 7) == 0 && ((((uintptr_t)part) | ((uintptr_t)weight)) & 15) == 0,
             TS_ERR_INVALID_ARGUMENT, "ts_devoxelize_backward_cells_f16: misaligned rows");
  const int groups = 256 / (c >> 2);
  if (n_seg > 0) {
    devoxelize_bwd_cells_kernel<_Float16><<<(unsigned)ts_cdiv(n_seg, groups), 256, 0, stream>>>(
        (const _Float16 *)grad_out, weight, order, seg_start, n_seg, c, part, go_ld);
    TS_CHECK_LAUNCH("ts_devoxelize_backward_cells_f16/partials");
  }
  devoxelize_bwd_csr_kernel<0, float, _Float16><<<(unsigned)ts_cdiv(m, groups), 256, 0, stream>>>(
      part, nullptr, offsets, entries, m, c, (_Float16 *)grad_feat, c);
  TS_CHECK_LAUNCH("ts_devoxelize_backward_cells_f16/gather");
  return TS_OK;
}
