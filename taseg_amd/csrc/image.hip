// TIAF image -> point gather (and its adjoint).
//
// Reference: R/pcseg/model/segmentor/voxel/minkunet/unet2d.py:180-214 makes NHWC copies
// (`permute(0, 2, 3, 1)`) of five full feature stacks, reshapes every sample's T frames into one tall
// (T*H, W, C) image and fancy-indexes it with the (row, col) pairs stored in the last two feature columns of the
// FOV cloud; the 1/4-scale map is indexed with (row // 4, col // 4).  Here the gather reads the NCHW stack in
// place: out[n, c] = feat[first_frame(batch n) + row_n / H, c, (row_n % H) >> shift, col_n >> shift], one thread
// per (point, channel), channel fastest (coalesced output rows; the scattered reads of one point stay inside
// C cache lines).  The adjoint accumulates with float atomics (several points can share a pixel).
#include "common.h"

__device__ __forceinline__ bool image_pixel(const float *__restrict__ pix, const int *__restrict__ pbatch,
                                            const int *__restrict__ frame_end, int64_t n, int n_batch, int T, int H,
                                            int W, int shift, int &frame, int &r, int &c) {
  const int row = (int)pix[2 * n], col = (int)pix[2 * n + 1];   // `.long()` of a non-negative float: truncation
  const int b = pbatch[n];
  if (b < 0 || b >= n_batch || row < 0 || col < 0) return false;
  const int start = b > 0 ? frame_end[b - 1] : 0;
  frame = start + row / H;
  r = (row % H) >> shift;
  c = col >> shift;
  return frame < T && frame < frame_end[b] && c < (W >> shift);
}

__global__ __launch_bounds__(256) void image_gather_fwd_kernel(const float *__restrict__ feat,
                                                               const float *__restrict__ pix,
                                                               const int *__restrict__ pbatch,
                                                               const int *__restrict__ frame_end, int64_t n_pts,
                                                               int n_batch, int T, int C, int H, int W, int shift,
                                                               float *__restrict__ out, int *__restrict__ err) {
  const int hs = H >> shift, ws = W >> shift;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = n_pts * C, step = (int64_t)gridDim.x * blockDim.x;
  for (; e < total; e += step) {
    const int64_t n = e / C;
    const int ch = (int)(e - n * C);
    int f, r, c;
    float v = 0.f;
    if (image_pixel(pix, pbatch, frame_end, n, n_batch, T, H, W, shift, f, r, c))
      v = feat[(((int64_t)f * C + ch) * hs + r) * ws + c];
    else if (ch == 0)
      *err = 1;   // the reference would raise an IndexError
    out[e] = v;
  }
}

__global__ __launch_bounds__(256) void image_gather_bwd_kernel(const float *__restrict__ gout,
                                                               const float *__restrict__ pix,
                                                               const int *__restrict__ pbatch,
                                                               const int *__restrict__ frame_end, int64_t n_pts,
                                                               int n_batch, int T, int C, int H, int W, int shift,
                                                               float *__restrict__ gfeat) {
  const int hs = H >> shift, ws = W >> shift;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = n_pts * C, step = (int64_t)gridDim.x * blockDim.x;
  for (; e < total; e += step) {
    const int64_t n = e / C;
    const int ch = (int)(e - n * C);
    int f, r, c;
    if (image_pixel(pix, pbatch, frame_end, n, n_batch, T, H, W, shift, f, r, c))
      atomicAdd(&gfeat[(((int64_t)f * C + ch) * hs + r) * ws + c], gout[e]);
  }
}

static int image_check(const char *what, const void *a, const void *pix, const void *pbatch, const void *frame_end,
                       int64_t n_pts, int n_batch, int T, int C, int H, int W, int shift) {
  TS_REQUIRE(n_pts >= 0 && n_batch > 0 && T > 0 && C > 0 && H > 0 && W > 0 && shift >= 0 && shift < 8,
             TS_ERR_INVALID_ARGUMENT, "%s: bad sizes", what);
  TS_REQUIRE((H % (1 << shift)) == 0 && (W % (1 << shift)) == 0, TS_ERR_INVALID_ARGUMENT,
             "%s: H and W must be multiples of the scale", what);
  TS_REQUIRE(a && frame_end && (n_pts == 0 || (pix && pbatch)), TS_ERR_INVALID_ARGUMENT, "%s: null pointer", what);
  return TS_OK;
}

extern "C" int ts_image_gather_forward(const float *feat, const float *pix, const int32_t *pbatch,
                                       const int32_t *frame_end, int64_t n_pts, int32_t n_batch, int32_t T, int32_t C,
                                       int32_t H, int32_t W, int32_t shift, float *out, int32_t *err,
                                       ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = image_check("ts_image_gather_forward", feat, pix, pbatch, frame_end, n_pts, n_batch, T, C, H, W, shift);
  if (rc != TS_OK) return rc;
  if (n_pts == 0) return TS_OK;
  TS_REQUIRE(out && err, TS_ERR_INVALID_ARGUMENT, "ts_image_gather_forward: null pointer");
  const unsigned grid = (unsigned)std::min<int64_t>(ts_cdiv(n_pts * C, 256), 1 << 16);
  image_gather_fwd_kernel<<<grid, 256, 0, stream>>>(feat, pix, pbatch, frame_end, n_pts, n_batch, T, C, H, W, shift,
                                                    out, err);
  TS_CHECK_LAUNCH("ts_image_gather_forward");
  return TS_OK;
}

extern "C" int ts_image_gather_backward(const float *grad_out, const float *pix, const int32_t *pbatch,
                                        const int32_t *frame_end, int64_t n_pts, int32_t n_batch, int32_t T,
                                        int32_t C, int32_t H, int32_t W, int32_t shift, float *grad_feat,
                                        ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = image_check("ts_image_gather_backward", grad_feat, pix, pbatch, frame_end, n_pts, n_batch, T, C, H, W, shift);
  if (rc != TS_OK) return rc;
  TS_CHECK_HIP(hipMemsetAsync(grad_feat, 0, (size_t)T * C * (H >> shift) * (W >> shift) * 4, stream), "image gather memset");
  if (n_pts == 0) return TS_OK;
  TS_REQUIRE(grad_out, TS_ERR_INVALID_ARGUMENT, "ts_image_gather_backward: null pointer");
  const unsigned grid = (unsigned)std::min<int64_t>(ts_cdiv(n_pts * C, 256), 1 << 16);
  image_gather_bwd_kernel<<<grid, 256, 0, stream>>>(grad_out, pix, pbatch, frame_end, n_pts, n_batch, T, C, H, W,
                                                    shift, grad_feat);
  TS_CHECK_LAUNCH("ts_image_gather_backward");
  return TS_OK;
}
