// The evaluation tail of the segmentors for a whole batch (R/pcseg/model/segmentor/voxel/minkunet/minkunet.py:435-455,
// minkunet_ms.py:433-458: per scene `out[scene mask][inverse_map of the scene]`, arg-max, trimmed to the scan's own points).
// The reference loops over the scenes with boolean masks (~12 launches and 6 host reads per scene); the tensor form of round 3 did
// the batch at once with three stable sorts by scene and ~45 small tensor ops - 1 ms of host time per pass, which made the
// evaluation loop under autocast host-bound.  Here:
//   ts_scene_counts  rows per scene of the voxel / point / label index arrays (+ are they grouped by scene in ascending order, as
//                    sparse_collate builds them?  + any index outside the batch?) - one launch, counters through LDS
//   ts_unvoxelise    mapped[p, :] = logits[start_v[scene_p] + inverse_map[p], :] and arg-max (first maximum), a wave per 64 / C
//                    points: rows of C <= 64 values move as whole rows - one launch
// For arrays grouped by scene (every collated batch) the scene-major order IS the array order: no sort.  Anything else is
// flagged, and the caller falls back to the sorted form (taseg_amd/pcseg/model/.../minkunet.py::unvoxelise_predictions).
#include "common.h"

#define ET_MAX_SCENES 64

__global__ __launch_bounds__(256) void scene_counts_kernel(const int *__restrict__ b0, int64_t s0, int64_t n0,
                                                           const int *__restrict__ b1, int64_t s1, int64_t n1,
                                                           const int *__restrict__ b2, int64_t s2, int64_t n2, int n_scenes,
                                                           unsigned long long *__restrict__ counts, int *__restrict__ flags) {
  __shared__ unsigned cnt[3][ET_MAX_SCENES];
  for (int i = threadIdx.x; i < 3 * ET_MAX_SCENES; i += 256) (&cnt[0][0])[i] = 0;
  __syncthreads();
  int bad = 0;
  const int64_t step = (int64_t)gridDim.x * 256;
  auto walk = [&](const int *b, int64_t stride, int64_t n, int which) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step) {
      const int v = b[i * stride];
      if (v < 0 || v >= n_scenes) {
        bad |= 1;
      } else {
        atomicAdd(&cnt[which][v], 1u);
      }
      if (i > 0 && b[(i - 1) * stride] > v) bad |= 2;
    }
  };
  walk(b0, s0, n0, 0);
  walk(b1, s1, n1, 1);
  walk(b2, s2, n2, 2);
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * n_scenes; i += 256) {
    const unsigned c = cnt[i / n_scenes][i % n_scenes];
    if (c) atomicAdd(&counts[i], (unsigned long long)c);
  }
  if (bad) atomicOr(flags, bad);
}

extern "C" int ts_scene_counts(const int32_t *b_vox, int64_t stride_vox, int64_t n_vox, const int32_t *b_pts, int64_t stride_pts,
                               int64_t n_pts, const int32_t *b_lab, int64_t stride_lab, int64_t n_lab, int32_t n_scenes, int64_t *counts,
                               int32_t *flags, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n_scenes > 0 && n_scenes <= ET_MAX_SCENES && n_vox >= 0 && n_pts >= 0 && n_lab >= 0 && counts && flags &&
                 stride_vox > 0 && stride_pts > 0 && stride_lab > 0,
             TS_ERR_INVALID_ARGUMENT, "ts_scene_counts: bad arguments (at most %d scenes)", ET_MAX_SCENES);
  TS_REQUIRE((n_vox == 0 || b_vox) && (n_pts == 0 || b_pts) && (n_lab == 0 || b_lab), TS_ERR_INVALID_ARGUMENT, "ts_scene_counts: null pointer");
  TS_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)3 * n_scenes * 8, stream), "ts_scene_counts: memset");
  TS_CHECK_HIP(hipMemsetAsync(flags, 0, 4, stream), "ts_scene_counts: memset");
  const int64_t n = std::max(n_vox, std::max(n_pts, n_lab));
  if (n == 0) return TS_OK;
  const unsigned grid = (unsigned)std::min<int64_t>(ts_cdiv(n, 256 * 8), 1024);
  scene_counts_kernel<<<grid, 256, 0, stream>>>(b_vox, stride_vox, n_vox, b_pts, stride_pts, n_pts, b_lab, stride_lab, n_lab, n_scenes,
                                                (unsigned long long *)counts, flags);
  TS_CHECK_LAUNCH("ts_scene_counts");
  return TS_OK;
}

// one thread per (point, 4-column piece): a row of C values leaves as C / 4 16-byte stores (C % 4 == 0) or element-wise
template <typename T>
__global__ __launch_bounds__(256) void unvoxelise_kernel(const T *__restrict__ logits, int C, const long long *__restrict__ counts,
                                                         int n_scenes, const int *__restrict__ b_pts, int64_t stride_pts,
                                                         const long long *__restrict__ inv, int64_t n_pts, T *__restrict__ mapped,
                                                         long long *__restrict__ pred, int *__restrict__ flags) {
  __shared__ long long start[ET_MAX_SCENES + 1];
  if (threadIdx.x == 0) {
    long long acc = 0;
    for (int s = 0; s < n_scenes; ++s) {
      start[s] = acc;
      acc += counts[s];
    }
    start[n_scenes] = acc;
  }
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n_pts) return;
  const int scene = b_pts[p * stride_pts];
  const long long local = inv[p];
  long long row = -1;
  if (scene >= 0 && scene < n_scenes) {
    if (local >= 0 && local < start[scene + 1] - start[scene]) row = start[scene] + local;
  }
  if (row < 0) {
    atomicOr(flags, 4);                 // the reference's indexing would raise
    if (pred) pred[p] = 0;
    if (mapped)
      for (int c = 0; c < C; ++c) mapped[p * C + c] = (T)0.f;
    return;
  }
  const T *src = logits + row * C;
  float best = -INFINITY;
  int arg = 0;
  for (int c = 0; c < C; ++c) {
    const T v = src[c];
    if (mapped) mapped[p * C + c] = v;
    const float f = (float)v;
    if (f > best || (f != f && best == best)) {      // first maximum; a NaN wins like torch.argmax
      best = f;
      arg = c;
    }
  }
  if (pred) pred[p] = arg;
}

extern "C" int ts_unvoxelise(const void *logits, int32_t half, int32_t C, const int64_t *counts, int32_t n_scenes, const int32_t *b_pts,
                             int64_t stride_pts, const int64_t *inv, int64_t n_pts, void *mapped, int64_t *pred, int32_t *flags,
                             ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(C > 0 && C <= 4096 && n_scenes > 0 && n_scenes <= ET_MAX_SCENES && n_pts >= 0 && stride_pts > 0 && counts && flags,
             TS_ERR_INVALID_ARGUMENT, "ts_unvoxelise: bad arguments");
  if (n_pts == 0) return TS_OK;
  TS_REQUIRE(logits && b_pts && inv && (mapped || pred), TS_ERR_INVALID_ARGUMENT, "ts_unvoxelise: null pointer");
  const unsigned grid = (unsigned)ts_cdiv(n_pts, 256);
  if (half)
    unvoxelise_kernel<_Float16><<<grid, 256, 0, stream>>>((const _Float16 *)logits, C, (const long long *)counts, n_scenes, b_pts, stride_pts,
                                                          (const long long *)inv, n_pts, (_Float16 *)mapped, (long long *)pred, flags);
  else
    unvoxelise_kernel<float><<<grid, 256, 0, stream>>>((const float *)logits, C, (const long long *)counts, n_scenes, b_pts, stride_pts,
                                                       (const long long *)inv, n_pts, (float *)mapped, (long long *)pred, flags);
  TS_CHECK_LAUNCH("ts_unvoxelise");
  return TS_OK;
}
