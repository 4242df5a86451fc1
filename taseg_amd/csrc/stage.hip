// Batched data stage (round 4): the per-point decisions and the layout of the fused clouds of a WHOLE batch in a few launches.
// Reference (numpy in DataLoader workers, one sample at a time): R/pcseg/data/dataset/semantickitti/semantickitti_ms.py:303-308
// (class-step rule), semantickitti_voxel_ms.py:121-124 (clamp of the fused cloud to the current scan's minimum), :127-212 (two
// voxelisations + collate); nuscenes/nuscenes_ms.py:320-328, nuscenes_voxel_ms.py:77-212.  taseg_amd/data/stage.py::voxelize_batch_ms
// chains these kernels with ts_fuse_scans_batch / ts_fuse_sweeps, ts_segment_min3, ts_voxel_coords and ts_sparse_quantize; every
// tensor of the resulting batch_dict equals the per-sample form bit for bit (tests/test_gpu_ops.py).
#include "common.h"

// keep[i] = pre[i] (optional) AND table[scan[i]][class(i)] AND (x, y, z)(i) >= lo[sample(i)]  ;  sample[i] = sample_of_scan[scan[i]]
//   class(i) = cls[i], or the column neg_col where cls[i] < 0 (KITTI: a pseudo label that is no class's canonical raw id: never kept)
__global__ __launch_bounds__(256) void stage_keep_kernel(const float *__restrict__ pts, int64_t n, int pstride,
                                                         const unsigned char *__restrict__ pre, const int *__restrict__ scan,
                                                         const int64_t *__restrict__ cls, const unsigned char *__restrict__ table,
                                                         int n_scans, int cols, int neg_col,
                                                         const int64_t *__restrict__ sample_of_scan, const float *__restrict__ lo,
                                                         int n_samples, unsigned char *__restrict__ keep,
                                                         int64_t *__restrict__ sample) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const int s = min(max(scan[i], 0), n_scans - 1);
    int64_t c = cls[i];
    if (c < 0) c = neg_col;
    const int64_t b = sample_of_scan[s];
    bool k = c >= 0 && c < cols && table[(int64_t)s * cols + c] != 0;
    if (pre) k = k && pre[i] != 0;
    if (k && b >= 0 && b < n_samples) {
      const float *p = pts + i * pstride;
      const float *l = lo + 3 * b;
      k = p[0] >= l[0] && p[1] >= l[1] && p[2] >= l[2];
    } else {
      k = false;
    }
    keep[i] = k ? 1 : 0;
    sample[i] = b;
  }
}

extern "C" int ts_stage_keep_flags(const float *points, int64_t n, int32_t point_stride, const uint8_t *pre_keep,
                                   const int32_t *scan_idx, const int64_t *cls, const uint8_t *table, int32_t n_scans,
                                   int32_t table_cols, int32_t neg_col, const int64_t *sample_of_scan, const float *lo,
                                   int32_t n_samples, uint8_t *keep, int64_t *sample, ts_stream_t stream) {
  TS_REQUIRE(n >= 0 && point_stride >= 3 && n_scans > 0 && table_cols > 0 && n_samples > 0, TS_ERR_INVALID_ARGUMENT,
             "ts_stage_keep_flags: bad sizes");
  if (n == 0) return TS_OK;
  TS_REQUIRE(points && scan_idx && cls && table && sample_of_scan && lo && keep && sample, TS_ERR_INVALID_ARGUMENT,
             "ts_stage_keep_flags: null pointer");
  const int grid = (int)std::min<int64_t>(ts_cdiv(n, 256), 4096);
  stage_keep_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(points, n, point_stride, pre_keep, scan_idx, cls, table, n_scans,
                                                         table_cols, neg_col, sample_of_scan, lo, n_samples, keep, sample);
  TS_CHECK_LAUNCH("ts_stage_keep_flags");
  return TS_OK;
}

// The fused clouds of a batch, sample-major, current scan first, kept history behind it in its own order, written straight to
// their rows: current point i of sample b goes to row i + kept_start[b] (kept_start[b] = kept history points of the samples
// before b), kept history point j (= history row idx[j], sample hb) to row cur_start[hb + 1] + j.
__global__ __launch_bounds__(256) void stage_layout_kernel(const float *__restrict__ cur, const int64_t *__restrict__ cur_lab,
                                                           const int64_t *__restrict__ cur_b, int64_t n_cur,
                                                           const float *__restrict__ hist, const int64_t *__restrict__ hist_lab,
                                                           const int64_t *__restrict__ hist_b, const int64_t *__restrict__ idx,
                                                           int64_t n_kept, int f, const int64_t *__restrict__ cur_start,
                                                           const int64_t *__restrict__ kept_start, float *__restrict__ pts,
                                                           int64_t *__restrict__ lab, int64_t *__restrict__ sample,
                                                           int *__restrict__ sample32, unsigned char *__restrict__ is_cur) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; t < n_cur + n_kept; t += step) {
    const float *src;
    int64_t dst, b, l;
    if (t < n_cur) {
      b = cur_b[t];
      dst = t + kept_start[b];
      src = cur + t * f;
      l = cur_lab[t];
    } else {
      const int64_t j = t - n_cur, h = idx[j];
      b = hist_b[h];
      dst = cur_start[b + 1] + j;
      src = hist + h * f;
      l = hist_lab[h];
    }
    float *d = pts + dst * f;
    for (int c = 0; c < f; ++c) d[c] = src[c];
    lab[dst] = l;
    sample[dst] = b;
    sample32[dst] = (int)b;
    is_cur[dst] = t < n_cur ? 1 : 0;
  }
}

extern "C" int ts_stage_layout(const float *cur, const int64_t *cur_lab, const int64_t *cur_b, int64_t n_cur, const float *hist,
                               const int64_t *hist_lab, const int64_t *hist_b, const int64_t *idx, int64_t n_kept, int32_t f,
                               const int64_t *cur_start, const int64_t *kept_start, float *pts, int64_t *lab, int64_t *sample,
                               int32_t *sample32, uint8_t *is_cur, ts_stream_t stream) {
  TS_REQUIRE(n_cur >= 0 && n_kept >= 0 && f > 0, TS_ERR_INVALID_ARGUMENT, "ts_stage_layout: bad sizes");
  if (n_cur + n_kept == 0) return TS_OK;
  TS_REQUIRE(cur_lab && cur_b && cur_start && kept_start && pts && lab && sample && sample32 && is_cur && (n_cur == 0 || cur) &&
                 (n_kept == 0 || (hist && hist_lab && hist_b && idx)),
             TS_ERR_INVALID_ARGUMENT, "ts_stage_layout: null pointer");
  const int grid = (int)std::min<int64_t>(ts_cdiv(n_cur + n_kept, 256), 8192);
  stage_layout_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(cur, cur_lab, cur_b, n_cur, hist, hist_lab, hist_b, idx, n_kept, f,
                                                           cur_start, kept_start, pts, lab, sample, sample32, is_cur);
  TS_CHECK_LAUNCH("ts_stage_layout");
  return TS_OK;
}

// After ts_sparse_quantize on the whole batch (voxels ordered by (sample, x, y, z)): vox[v] = coords4[index[v]], the cumulative voxel
// counts per sample (offset[b] = voxels of samples 0 .. b: the batch_dict's `offset` tensors) and the LOCAL voxel index of every
// point (inverse[i] minus the voxels of the samples before the point's) - what sparse_quantize returns sample by sample.
__global__ __launch_bounds__(256) void stage_vox_kernel(const int4 *__restrict__ coords, const int *__restrict__ index, int64_t m,
                                                        int4 *__restrict__ vox) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; v < m; v += step) vox[v] = coords[index[v]];
}
__global__ void stage_starts_kernel(const int4 *__restrict__ vox, int64_t m, int n_samples, int64_t *__restrict__ start,
                                    int *__restrict__ offset) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;                // start[b] = first voxel with sample >= b
  if (b > n_samples) return;
  int64_t lo = 0, hi = m;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (vox[mid].w < b) lo = mid + 1; else hi = mid;
  }
  start[b] = lo;
  if (b >= 1) offset[b - 1] = (int)lo;
}
__global__ __launch_bounds__(256) void stage_local_kernel(const int *__restrict__ inverse, const int64_t *__restrict__ row_b,
                                                          const int64_t *__restrict__ start, int64_t n, int64_t *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) out[i] = (int64_t)inverse[i] - start[row_b[i]];
}

extern "C" int ts_stage_split_voxels(const int32_t *coords4, const int32_t *index, int64_t m, const int32_t *inverse,
                                     const int64_t *row_sample, int64_t n, int32_t n_samples, int32_t *vox, int64_t *start,
                                     int32_t *offset, int64_t *inverse_local, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(m >= 0 && n >= 0 && n_samples > 0 && n_samples <= 1024, TS_ERR_INVALID_ARGUMENT, "ts_stage_split_voxels: bad sizes");
  TS_REQUIRE(start && offset && (m == 0 || (coords4 && index && vox)) && (n == 0 || (inverse && row_sample && inverse_local)),
             TS_ERR_INVALID_ARGUMENT, "ts_stage_split_voxels: null pointer");
  TS_REQUIRE(((((uintptr_t)coords4) | ((uintptr_t)vox)) & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_stage_split_voxels: coordinate arrays must be 16-byte aligned");
  if (m > 0) {
    stage_vox_kernel<<<(int)std::min<int64_t>(ts_cdiv(m, 256), 4096), 256, 0, stream>>>((const int4 *)coords4, index, m, (int4 *)vox);
    TS_CHECK_LAUNCH("ts_stage_split_voxels/vox");
  }
  stage_starts_kernel<<<(unsigned)ts_cdiv(n_samples + 1, 64), 64, 0, stream>>>((const int4 *)vox, m, n_samples, start, offset);
  TS_CHECK_LAUNCH("ts_stage_split_voxels/starts");
  if (n > 0) {
    stage_local_kernel<<<(int)std::min<int64_t>(ts_cdiv(n, 256), 4096), 256, 0, stream>>>(inverse, row_sample, start, n, inverse_local);
    TS_CHECK_LAUNCH("ts_stage_split_voxels/local");
  }
  return TS_OK;
}
