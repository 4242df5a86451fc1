// TIAF image -> point gather and its adjoint.
//
// Reference: R/pcseg/model/segmentor/voxel/minkunet/unet2d.py:180-214 makes NHWC copies (`permute(0, 2, 3, 1)`) of five full
// feature stacks, reshapes every sample's T frames into one tall (T*H, W, C) image and fancy-indexes it with the (row, col) pairs
// stored in the last two feature columns of the FOV cloud; the 1/4-scale map is indexed with (row // 4, col // 4); the adjoint is
// index_put(accumulate) into a zero-filled copy of the stack.
//
// Here the stacks stay NCHW and nothing is copied.  Once per batch and scale the points are put in RASTER order
// (`ts_image_plan`: pixel address = (frame * hs + r) * ws + c as sort key, stable radix sort, run lengths of equal addresses).
// Then, per feature map:
//   * gather  (`image_gather_rows_kernel`): a workgroup takes 64 consecutive points of that order.  Lanes run along the POINTS for
//     one channel at a time, so a wave reads neighbouring pixels of ONE plane (LiDAR returns of a scan line are 2-3 pixels apart:
//     a handful of cache lines per wave instead of 64 planes x 4 useful bytes with lanes along the channels); the 64 x 32 tile is
//     transposed through LDS and leaves as contiguous 128-byte pieces of the [n, C] rows (original point order).
//   * adjoint (`image_scatter_rows_kernel`): the same tiles the other way round - gradient rows in, transposed through LDS, and
//     the points of one pixel (adjacent in raster order) summed in index order by the lane of the run's first point, which then
//     adds the sum to the pixel with a plain read-modify-write: every pixel has exactly one such lane in the whole grid.  No
//     atomics, run-to-run identical, and only the pixels that have points are touched: the caller hands in the gradient the map
//     already has from its dense consumer (unet2d.py: classifier / next decoder stage) and gets the sum back in place - no
//     zero-filled T x C x H x W tensor, no second dense add.
#include <cstring>

#include <hip/hip_fp16.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

// pixel address (frame * hs + r) * ws + c of point n in the [T, hs, ws] stack, or `limit` when the point falls outside its
// sample's frames.  Written without early exits and with ONE select at the end: with the key initialised to `limit` and
// overwritten in a branch, hipcc (ROCm 7.2) kept the key in the register its integer division used as scratch, and an invalid
// point left the kernel with row / H as its key (found in round 6: `err` was right, the address was not).
__device__ __forceinline__ unsigned image_pixel_key(const float *__restrict__ pix, const int *__restrict__ pbatch,
                                                    const int *__restrict__ frame_end, int64_t n, int n_batch, int T, int H, int W,
                                                    int shift, unsigned limit) {
  const int hs = H >> shift, ws = W >> shift;
  const int row = (int)pix[2 * n], col = (int)pix[2 * n + 1];   // `.long()` of a non-negative float: truncation
  const int b = pbatch[n];
  const bool in_batch = b >= 0 && b < n_batch;
  const int bc = in_batch ? b : 0;
  const int start = bc > 0 ? frame_end[bc - 1] : 0;
  const int end = frame_end[bc];
  const int rowc = row < 0 ? 0 : row;
  const int frame = start + rowc / H;
  const int r = (rowc % H) >> shift;
  const int c = (col < 0 ? 0 : col) >> shift;
  const bool ok = in_batch && row >= 0 && col >= 0 && frame < T && frame < end && c < ws;
  const unsigned key = (unsigned)((frame * hs + r) * ws + c);
  return ok ? key : limit;
}

// key = pixel address in the [T, hs, ws] stack, `limit` (= T * hs * ws) for a point outside its sample's frames (sorts last)
__global__ __launch_bounds__(256) void image_key_kernel(const float *__restrict__ pix, const int *__restrict__ pbatch,
                                                        const int *__restrict__ frame_end, int64_t n_pts, int n_batch, int T, int H,
                                                        int W, int shift, unsigned limit, unsigned *__restrict__ keys,
                                                        int *__restrict__ idx, int *__restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pts) return;
  const unsigned key = image_pixel_key(pix, pbatch, frame_end, i, n_batch, T, H, W, shift, limit);
  if (key >= limit) *err = 1;          // the reference would raise an IndexError
  keys[i] = key;
  idx[i] = (int)i;
}

// paddr[i] = pixel address of the i-th point in raster order (-1: none); run[i] = number of consecutive points on that pixel if i is
// the first of them, else 0
__global__ __launch_bounds__(256) void image_runs_kernel(const unsigned *__restrict__ keys, int64_t n_pts, unsigned limit,
                                                         int *__restrict__ paddr, int *__restrict__ run) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pts) return;
  const unsigned k = keys[i];
  paddr[i] = k < limit ? (int)k : -1;
  int len = 0;
  if (k < limit && (i == 0 || keys[i - 1] != k)) {
    len = 1;
    while (i + len < n_pts && keys[i + len] == k) ++len;
  }
  run[i] = len;
}

static size_t image_sort_bytes(int64_t n) {
  size_t need = 0;
  (void)rocprim::radix_sort_pairs(nullptr, need, (unsigned *)nullptr, (unsigned *)nullptr, (int *)nullptr, (int *)nullptr, (size_t)std::max<int64_t>(n, 1), 0,
                                  32, (hipStream_t)0);
  return need;
}

extern "C" size_t ts_image_plan_workspace_bytes(int64_t n_pts) {
  const size_t n = (size_t)std::max<int64_t>(n_pts, 1);
  return ts_align_up(n * 4, 256) * 3 + ts_align_up(image_sort_bytes(n_pts), 256);
}

extern "C" int ts_image_plan(const float *pix, const int32_t *pbatch, const int32_t *frame_end, int64_t n_pts, int32_t n_batch, int32_t T,
                             int32_t H, int32_t W, int32_t shift, int32_t *perm, int32_t *paddr, int32_t *run, int32_t *err, void *ws,
                             size_t ws_bytes, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n_pts >= 0 && n_batch > 0 && T > 0 && H > 0 && W > 0 && shift >= 0 && shift < 8 && n_pts < (1LL << 31), TS_ERR_INVALID_ARGUMENT,
             "ts_image_plan: bad sizes");
  TS_REQUIRE((H % (1 << shift)) == 0 && (W % (1 << shift)) == 0, TS_ERR_INVALID_ARGUMENT, "ts_image_plan: H and W must be multiples of the scale");
  const int64_t limit64 = (int64_t)T * (H >> shift) * (W >> shift);
  TS_REQUIRE(limit64 < (1LL << 31), TS_ERR_INVALID_ARGUMENT, "ts_image_plan: more than 2^31 pixels in the stack");
  if (n_pts == 0) return TS_OK;
  TS_REQUIRE(pix && pbatch && frame_end && perm && paddr && run && err && ws && ws_bytes >= ts_image_plan_workspace_bytes(n_pts),
             TS_ERR_INVALID_ARGUMENT, "ts_image_plan: null pointer / workspace too small");
  char *p = (char *)ws;
  unsigned *keys = (unsigned *)p;
  p += ts_align_up((size_t)n_pts * 4, 256);
  unsigned *keys_sorted = (unsigned *)p;
  p += ts_align_up((size_t)n_pts * 4, 256);
  int *idx = (int *)p;
  p += ts_align_up((size_t)n_pts * 4, 256);
  size_t tmp_bytes = ws_bytes - (size_t)(p - (char *)ws);
  const unsigned limit = (unsigned)limit64;
  const unsigned grid = (unsigned)ts_cdiv(n_pts, 256);
  image_key_kernel<<<grid, 256, 0, stream>>>(pix, pbatch, frame_end, n_pts, n_batch, T, H, W, shift, limit, keys, idx, err);
  TS_CHECK_LAUNCH("ts_image_plan/keys");
  int bits = 1;
  while ((1ULL << bits) <= (unsigned long long)limit) ++bits;      // keys are <= limit
  TS_CHECK_HIP(rocprim::radix_sort_pairs(p, tmp_bytes, keys, keys_sorted, idx, perm, (size_t)n_pts, 0, bits, stream), "ts_image_plan: sort");
  image_runs_kernel<<<grid, 256, 0, stream>>>(keys_sorted, n_pts, limit, paddr, run);
  TS_CHECK_LAUNCH("ts_image_plan/runs");
  return TS_OK;
}

#define IG_PTS 64      // points per workgroup tile
#define IG_CH 32       // channels per LDS pass

__global__ __launch_bounds__(256) void image_gather_rows_kernel(const float *__restrict__ feat, int C, int64_t hw,
                                                                const int *__restrict__ perm, const int *__restrict__ paddr,
                                                                int64_t n_pts, float *__restrict__ out) {
  __shared__ float tile[IG_PTS][IG_CH + 1];
  __shared__ int s_perm[IG_PTS];
  __shared__ int64_t s_off[IG_PTS];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t base = (int64_t)blockIdx.x * IG_PTS;
  if (t < IG_PTS) {
    const int64_t i = base + t;
    int pm = -1;
    int64_t off = -1;
    if (i < n_pts) {
      pm = perm[i];
      const int a = paddr[i];
      if (a >= 0) {
        const int64_t f = a / hw;
        off = f * C * hw + (a - f * hw);       // element (f, channel 0, r, c) of the NCHW stack
      }
    }
    s_perm[t] = pm;
    s_off[t] = off;
  }
  __syncthreads();
  const int64_t my_off = s_off[lane];
  for (int c0 = 0; c0 < C; c0 += IG_CH) {
    // lanes along the points, one channel per wave and round: neighbouring pixels of one plane
    for (int cc = wave; cc < IG_CH; cc += 4) {
      const int c = c0 + cc;
      tile[lane][cc] = (c < C && my_off >= 0) ? feat[my_off + (int64_t)c * hw] : 0.f;
    }
    __syncthreads();
    // rows out: 32 consecutive lanes write 128 contiguous bytes of one point's row
    for (int e = t; e < IG_PTS * IG_CH; e += 256) {
      const int r = e / IG_CH, cc = e - r * IG_CH;
      const int pm = s_perm[r];
      if (pm >= 0 && c0 + cc < C) out[(int64_t)pm * C + c0 + cc] = tile[r][cc];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void image_scatter_rows_kernel(const float *__restrict__ gout, int C, int64_t hw,
                                                                 const int *__restrict__ perm, const int *__restrict__ paddr,
                                                                 const int *__restrict__ run, int64_t n_pts, float *__restrict__ gfeat) {
  __shared__ float tile[IG_PTS][IG_CH + 1];
  __shared__ int s_perm[IG_PTS], s_run[IG_PTS];
  __shared__ int64_t s_off[IG_PTS];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t base = (int64_t)blockIdx.x * IG_PTS;
  if (t < IG_PTS) {
    const int64_t i = base + t;
    int pm = -1, rn = 0;
    int64_t off = -1;
    if (i < n_pts) {
      pm = perm[i];
      rn = run[i];
      const int a = paddr[i];
      if (a >= 0) {
        const int64_t f = a / hw;
        off = f * C * hw + (a - f * hw);
      }
    }
    s_perm[t] = pm;
    s_run[t] = rn;
    s_off[t] = off;
  }
  __syncthreads();
  const int64_t my_off = s_off[lane];
  const int my_run = s_run[lane];
  for (int c0 = 0; c0 < C; c0 += IG_CH) {
    for (int e = t; e < IG_PTS * IG_CH; e += 256) {
      const int r = e / IG_CH, cc = e - r * IG_CH;
      const int pm = s_perm[r];
      tile[r][cc] = (pm >= 0 && c0 + cc < C) ? gout[(int64_t)pm * C + c0 + cc] : 0.f;
    }
    __syncthreads();
    if (my_run > 0 && my_off >= 0) {
      for (int cc = wave; cc < IG_CH; cc += 4) {
        const int c = c0 + cc;
        if (c >= C) break;
        float sum = 0.f;
        for (int q = 0; q < my_run; ++q) {
          const int r = lane + q;
          // (a run that leaves the tile: its tail comes straight from memory - a pixel rarely holds more than a few points)
          sum += r < IG_PTS ? tile[r][cc] : gout[(int64_t)perm[base + r] * C + c];
        }
        float *dst = gfeat + my_off + (int64_t)c * hw;
        *dst += sum;
      }
    }
    __syncthreads();
  }
}

static int image_rows_check(const char *what, const void *a, const void *b, const void *perm, const void *paddr, int64_t n_pts, int32_t C,
                            int64_t hw) {
  TS_REQUIRE(n_pts >= 0 && C > 0 && hw > 0 && n_pts < (1LL << 31), TS_ERR_INVALID_ARGUMENT, "%s: bad sizes", what);
  TS_REQUIRE(n_pts == 0 || (a && b && perm && paddr), TS_ERR_INVALID_ARGUMENT, "%s: null pointer", what);
  return TS_OK;
}

// out[n, C] (every row written; a point outside its sample's frames - plan's err - reads as zeros); feat [T, C, hs, ws], hw = hs * ws
extern "C" int ts_image_gather_forward(const float *feat, int32_t C, int64_t hw, const int32_t *perm, const int32_t *paddr, int64_t n_pts,
                                       float *out, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int rc = image_rows_check("ts_image_gather_forward", feat, out, perm, paddr, n_pts, C, hw);
  if (rc != TS_OK) return rc;
  if (n_pts == 0) return TS_OK;
  image_gather_rows_kernel<<<(unsigned)ts_cdiv(n_pts, IG_PTS), 256, 0, stream>>>(feat, C, hw, perm, paddr, n_pts, out);
  TS_CHECK_LAUNCH("ts_image_gather_forward");
  return TS_OK;
}

// grad_feat [T, C, hs, ws] += adjoint(grad_out [n, C]) on the pixels that have points.  accumulate == 0: grad_feat (n_feat elements)
// is zero-filled first (the stand-alone adjoint); != 0: the caller's tensor already holds the map's other gradient.
extern "C" int ts_image_gather_backward(const float *grad_out, int32_t C, int64_t hw, const int32_t *perm, const int32_t *paddr,
                                        const int32_t *run, int64_t n_pts, float *grad_feat, int64_t n_feat, int32_t accumulate,
                                        ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(grad_feat && n_feat >= 0, TS_ERR_INVALID_ARGUMENT, "ts_image_gather_backward: null pointer");
  const int rc = image_rows_check("ts_image_gather_backward", grad_out, grad_feat, perm, paddr, n_pts, C, hw);
  if (rc != TS_OK) return rc;
  if (!accumulate) TS_CHECK_HIP(hipMemsetAsync(grad_feat, 0, (size_t)n_feat * 4, stream), "image gather memset");
  if (n_pts == 0) return TS_OK;
  TS_REQUIRE(run, TS_ERR_INVALID_ARGUMENT, "ts_image_gather_backward: null pointer");
  image_scatter_rows_kernel<<<(unsigned)ts_cdiv(n_pts, IG_PTS), 256, 0, stream>>>(grad_out, C, hw, perm, paddr, run, n_pts, grad_feat);
  TS_CHECK_LAUNCH("ts_image_gather_backward");
  return TS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same gather on a CHANNELS-LAST stack ([T, hs, ws, C] in memory: the layout the reference itself indexes, unet2d.py:183-187
// `permute(0, 2, 3, 1)`, and the one MIOpen's fp16 convolutions produce).  A pixel is ONE contiguous row of C * elem bytes at
// byte offset paddr * C * elem, so the gather is a row copy: thread (point i of the raster order, piece j) moves one 16-byte piece
// (8 / 4 / 2 where the row length asks for it); a 96-channel fp32 row is 24 lanes, a wave moves 2 2/3 rows.  Raster order keeps
// neighbouring rows of the map in neighbouring lanes (LiDAR returns of a scan line are 2-3 pixels apart) and duplicates of a pixel
// together; the output rows are whole 128-byte lines wherever C * elem is a multiple of 128.  No LDS, no transpose.
// The adjoint: the thread group of a pixel's FIRST point sums the gradient rows of its run in index order (fp32 accumulate) and
// read-modify-writes the pixel's row once - every row has one owner in the grid: no atomics, run-to-run identical.
template <typename V>
__global__ __launch_bounds__(256) void image_rows_gather_kernel(const char *__restrict__ feat, int pieces, const int *__restrict__ perm,
                                                                const int *__restrict__ paddr, int64_t n_pts, char *__restrict__ out) {
  // (32-bit index arithmetic: the entry checks n_pts * pieces < 2^31)
  const unsigned e = blockIdx.x * 256u + threadIdx.x;
  const unsigned i = e / (unsigned)pieces;
  if (i >= (unsigned)n_pts) return;
  const unsigned j = e - i * (unsigned)pieces;
  const int a = paddr[i];
  V v;
  memset(&v, 0, sizeof(V));            // a point outside its sample's frames (plan's err) reads as zeros
  if (a >= 0) v = ((const V *)(feat + (size_t)a * pieces * sizeof(V)))[j];
  ((V *)(out + (size_t)perm[i] * pieces * sizeof(V)))[j] = v;
}

template <typename T>
struct RowAcc;
template <>
struct RowAcc<float> {
  static __device__ __forceinline__ float load(const float *p) { return *p; }
  static __device__ __forceinline__ void store(float *p, float v) { *p = v; }
};
template <>
struct RowAcc<__half> {
  static __device__ __forceinline__ float load(const __half *p) { return __half2float(*p); }
  static __device__ __forceinline__ void store(__half *p, float v) { *p = __float2half(v); }
};

// VE elements of type T per thread (VE * sizeof(T) = 16, 8, 4 or 2 bytes); `pieces` = C / VE threads per row
template <typename T, int VE>
__global__ __launch_bounds__(256) void image_rows_scatter_kernel(const T *__restrict__ gout, int pieces, const int *__restrict__ perm,
                                                                 const int *__restrict__ paddr, const int *__restrict__ run, int64_t n_pts,
                                                                 T *__restrict__ gfeat) {
  struct alignas(VE * sizeof(T)) Vec {
    T x[VE];
  };
  const unsigned e = blockIdx.x * 256u + threadIdx.x;
  const unsigned i = e / (unsigned)pieces;
  if (i >= (unsigned)n_pts) return;
  const int len = run[i];
  const int a = paddr[i];
  if (len <= 0 || a < 0) return;
  const unsigned j = e - i * (unsigned)pieces;
  const size_t row = (size_t)pieces * VE;
  float acc[VE];
#pragma unroll
  for (int k = 0; k < VE; ++k) acc[k] = 0.f;
  for (int q = 0; q < len; ++q) {
    const Vec g = ((const Vec *)(gout + (size_t)perm[i + q] * row))[j];
#pragma unroll
    for (int k = 0; k < VE; ++k) acc[k] += RowAcc<T>::load(&g.x[k]);
  }
  Vec *dst = (Vec *)(gfeat + (size_t)a * row) + j;
  Vec d = *dst;
#pragma unroll
  for (int k = 0; k < VE; ++k) RowAcc<T>::store(&d.x[k], RowAcc<T>::load(&d.x[k]) + acc[k]);
  *dst = d;
}

// largest piece of 16 / 8 / 4 / 2 / 1 bytes that divides the row and keeps every base pointer aligned
static int image_piece_bytes(size_t row_bytes, const void *a, const void *b) {
  for (int v = 16; v > 1; v >>= 1)
    if (row_bytes % v == 0 && ((uintptr_t)a % v) == 0 && ((uintptr_t)b % v) == 0) return v;
  return 1;
}

// out[n, C] (original point order; every row written) from feat [T, hs, ws, C] (channels-last memory), elements of `elem_bytes`
// bytes (a pure move: any dtype)
extern "C" int ts_image_gather_rows_forward(const void *feat, int32_t C, int32_t elem_bytes, const int32_t *perm, const int32_t *paddr,
                                            int64_t n_pts, void *out, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n_pts >= 0 && C > 0 && n_pts < (1LL << 31) && (elem_bytes == 1 || elem_bytes == 2 || elem_bytes == 4 || elem_bytes == 8),
             TS_ERR_INVALID_ARGUMENT, "ts_image_gather_rows_forward: bad sizes");
  if (n_pts == 0) return TS_OK;
  TS_REQUIRE(feat && out && perm && paddr, TS_ERR_INVALID_ARGUMENT, "ts_image_gather_rows_forward: null pointer");
  const size_t row = (size_t)C * elem_bytes;
  const int v = image_piece_bytes(row, feat, out);
  const int pieces = (int)(row / v);
  TS_REQUIRE(n_pts * pieces < (1LL << 31), TS_ERR_UNSUPPORTED, "ts_image_gather_rows_forward: more than 2^31 row pieces");
  const unsigned grid = (unsigned)ts_cdiv(n_pts * pieces, 256);
  const char *f = (const char *)feat;
  char *o = (char *)out;
  switch (v) {
    case 16: image_rows_gather_kernel<uint4><<<grid, 256, 0, stream>>>(f, pieces, perm, paddr, n_pts, o); break;
    case 8: image_rows_gather_kernel<uint2><<<grid, 256, 0, stream>>>(f, pieces, perm, paddr, n_pts, o); break;
    case 4: image_rows_gather_kernel<uint32_t><<<grid, 256, 0, stream>>>(f, pieces, perm, paddr, n_pts, o); break;
    case 2: image_rows_gather_kernel<uint16_t><<<grid, 256, 0, stream>>>(f, pieces, perm, paddr, n_pts, o); break;
    default: image_rows_gather_kernel<uint8_t><<<grid, 256, 0, stream>>>(f, pieces, perm, paddr, n_pts, o); break;
  }
  TS_CHECK_LAUNCH("ts_image_gather_rows_forward");
  return TS_OK;
}

template <typename T>
static int image_rows_scatter(const T *g, int C, const int32_t *perm, const int32_t *paddr, const int32_t *run, int64_t n_pts, T *gf,
                              hipStream_t stream) {
  const int v = image_piece_bytes((size_t)C * sizeof(T), g, gf);
  TS_REQUIRE(v >= (int)sizeof(T), TS_ERR_INVALID_ARGUMENT, "ts_image_gather_rows_backward: pointers must be aligned to the element size");
  const int ve = v / (int)sizeof(T);
  const int pieces = C / ve;
  TS_REQUIRE(n_pts * pieces < (1LL << 31), TS_ERR_UNSUPPORTED, "ts_image_gather_rows_backward: more than 2^31 row pieces");
  const unsigned grid = (unsigned)ts_cdiv(n_pts * pieces, 256);
  if constexpr (sizeof(T) == 2) {
    if (ve == 8) {
      image_rows_scatter_kernel<T, 8><<<grid, 256, 0, stream>>>(g, pieces, perm, paddr, run, n_pts, gf);
      return TS_OK;
    }
  }
  if (ve == 4)
    image_rows_scatter_kernel<T, 4><<<grid, 256, 0, stream>>>(g, pieces, perm, paddr, run, n_pts, gf);
  else if (ve == 2)
    image_rows_scatter_kernel<T, 2><<<grid, 256, 0, stream>>>(g, pieces, perm, paddr, run, n_pts, gf);
  else
    image_rows_scatter_kernel<T, 1><<<grid, 256, 0, stream>>>(g, pieces, perm, paddr, run, n_pts, gf);
  return TS_OK;
}

// grad_feat [T, hs, ws, C] (channels-last memory, n_feat elements) += adjoint(grad_out [n, C]); half != 0: IEEE-half rows and map
// (fp32 accumulation over a pixel's run, one rounding into the map).  accumulate == 0: grad_feat is zero-filled first.
extern "C" int ts_image_gather_rows_backward(const void *grad_out, int32_t C, int32_t half, const int32_t *perm, const int32_t *paddr,
                                             const int32_t *run, int64_t n_pts, void *grad_feat, int64_t n_feat, int32_t accumulate,
                                             ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(grad_feat && n_feat >= 0 && n_pts >= 0 && C > 0 && n_pts < (1LL << 31), TS_ERR_INVALID_ARGUMENT,
             "ts_image_gather_rows_backward: bad arguments");
  if (!accumulate) TS_CHECK_HIP(hipMemsetAsync(grad_feat, 0, (size_t)n_feat * (half ? 2 : 4), stream), "image gather memset");
  if (n_pts == 0) return TS_OK;
  TS_REQUIRE(grad_out && perm && paddr && run, TS_ERR_INVALID_ARGUMENT, "ts_image_gather_rows_backward: null pointer");
  const int rc = half ? image_rows_scatter<__half>((const __half *)grad_out, C, perm, paddr, run, n_pts, (__half *)grad_feat, stream)
                      : image_rows_scatter<float>((const float *)grad_out, C, perm, paddr, run, n_pts, (float *)grad_feat, stream);
  if (rc != TS_OK) return rc;
  TS_CHECK_LAUNCH("ts_image_gather_rows_backward");
  return TS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// AvgPool2d(kernel 3, stride 2, padding 1, count_include_pad = True) of UNet2D's encoder (R/.../unet2d.py:58-62,72-77:
// `nn.AvgPool2d(kernel_size=(3, 3), stride=2, padding=1)`) on a CHANNELS-LAST stack.  Hand-written because the library kernel this
// PyTorch-ROCm dispatches to for channels-last gradients (`avg_pool2d_backward_out_cuda_frame_nhwc`) returns wrong values (found in
// round 6 by the TIAF golden: relative error 1.0 against the NCHW form on the same input, fp32 and fp16), and because both passes
// are plain row moves here: a thread owns one 16-byte piece of one output (forward) / input (backward) pixel row, sums the <= 9
// (<= 4) contributing rows in fp32 in a fixed order and divides by 9 - no atomics, run-to-run identical.
template <typename T, int VE>
struct alignas(VE * sizeof(T)) PoolVec {
  T x[VE];
};

template <typename T, int VE>
__global__ __launch_bounds__(256) void avgpool3s2_fwd_kernel(const T *__restrict__ X, int H, int W, int Ho, int Wo, int pieces,
                                                             unsigned total, T *__restrict__ Y) {
  using V = PoolVec<T, VE>;
  const unsigned e = blockIdx.x * 256u + threadIdx.x;
  if (e >= total) return;
  const unsigned j = e % (unsigned)pieces, px = e / (unsigned)pieces;
  const unsigned wo = px % (unsigned)Wo, rest = px / (unsigned)Wo;
  const unsigned ho = rest % (unsigned)Ho, t = rest / (unsigned)Ho;
  float acc[VE];
#pragma unroll
  for (int k = 0; k < VE; ++k) acc[k] = 0.f;
  const V *base = (const V *)X + (size_t)t * H * W * pieces + j;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    const int h = 2 * (int)ho + dy;
    if (h < 0 || h >= H) continue;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int w = 2 * (int)wo + dx;
      if (w < 0 || w >= W) continue;
      const V v = base[((size_t)h * W + w) * pieces];
#pragma unroll
      for (int k = 0; k < VE; ++k) acc[k] += RowAcc<T>::load(&v.x[k]);
    }
  }
  V out;
#pragma unroll
  for (int k = 0; k < VE; ++k) RowAcc<T>::store(&out.x[k], acc[k] / 9.f);
  ((V *)Y)[(size_t)px * pieces + j] = out;
}

template <typename T, int VE>
__global__ __launch_bounds__(256) void avgpool3s2_bwd_kernel(const T *__restrict__ GY, int H, int W, int Ho, int Wo, int pieces,
                                                             unsigned total, T *__restrict__ GX) {
  using V = PoolVec<T, VE>;
  const unsigned e = blockIdx.x * 256u + threadIdx.x;
  if (e >= total) return;
  const unsigned j = e % (unsigned)pieces, px = e / (unsigned)pieces;
  const unsigned w = px % (unsigned)W, rest = px / (unsigned)W;
  const unsigned h = rest % (unsigned)H, t = rest / (unsigned)H;
  // output rows whose window (2 ho - 1 .. 2 ho + 1) holds h: h / 2 for an even h, (h - 1) / 2 and (h + 1) / 2 for an odd one
  const int ho0 = (int)h / 2, ho1 = (h & 1) ? ho0 + 1 : ho0;
  const int wo0 = (int)w / 2, wo1 = (w & 1) ? wo0 + 1 : wo0;
  float acc[VE];
#pragma unroll
  for (int k = 0; k < VE; ++k) acc[k] = 0.f;
  const V *base = (const V *)GY + (size_t)t * Ho * Wo * pieces + j;
  for (int ho = ho0; ho <= ho1; ++ho) {
    if (ho >= Ho) continue;
    for (int wo = wo0; wo <= wo1; ++wo) {
      if (wo >= Wo) continue;
      const V v = base[((size_t)ho * Wo + wo) * pieces];
#pragma unroll
      for (int k = 0; k < VE; ++k) acc[k] += RowAcc<T>::load(&v.x[k]) / 9.f;
    }
  }
  V out;
#pragma unroll
  for (int k = 0; k < VE; ++k) RowAcc<T>::store(&out.x[k], acc[k]);
  ((V *)GX)[(size_t)px * pieces + j] = out;
}

template <typename T>
static int avgpool3s2_launch(const char *what, bool backward, const T *in, int T_, int H, int W, int C, T *out, hipStream_t stream) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int v = image_piece_bytes((size_t)C * sizeof(T), in, out);
  TS_REQUIRE(v >= (int)sizeof(T), TS_ERR_INVALID_ARGUMENT, "%s: pointers must be aligned to the element size", what);
  const int ve = v / (int)sizeof(T), pieces = C / ve;
  const int64_t total = (int64_t)T_ * (backward ? (int64_t)H * W : (int64_t)Ho * Wo) * pieces;
  TS_REQUIRE(total < (1LL << 32), TS_ERR_UNSUPPORTED, "%s: more than 2^32 row pieces", what);
  if (total == 0) return TS_OK;
  const unsigned grid = (unsigned)ts_cdiv(total, 256);
#define TS_POOL(VE_)                                                                                                          \
  (backward ? avgpool3s2_bwd_kernel<T, VE_><<<grid, 256, 0, stream>>>(in, H, W, Ho, Wo, pieces, (unsigned)total, out)         \
            : avgpool3s2_fwd_kernel<T, VE_><<<grid, 256, 0, stream>>>(in, H, W, Ho, Wo, pieces, (unsigned)total, out))
  if constexpr (sizeof(T) == 2) {
    if (ve == 8) {
      TS_POOL(8);
      return TS_OK;
    }
  }
  if (ve == 4)
    TS_POOL(4);
  else if (ve == 2)
    TS_POOL(2);
  else
    TS_POOL(1);
#undef TS_POOL
  return TS_OK;
}

// y [T, Ho, Wo, C] = AvgPool2d(3, stride 2, padding 1)(x [T, H, W, C]); Ho = (H - 1) / 2 + 1, Wo likewise; half != 0: IEEE half
extern "C" int ts_avgpool3s2_rows_forward(const void *x, int32_t T, int32_t H, int32_t W, int32_t C, int32_t half, void *y,
                                          ts_stream_t stream_) {
  TS_REQUIRE(T >= 0 && H > 0 && W > 0 && C > 0, TS_ERR_INVALID_ARGUMENT, "ts_avgpool3s2_rows_forward: bad sizes");
  if (T == 0) return TS_OK;
  TS_REQUIRE(x && y, TS_ERR_INVALID_ARGUMENT, "ts_avgpool3s2_rows_forward: null pointer");
  const int rc = half ? avgpool3s2_launch<__half>("ts_avgpool3s2_rows_forward", false, (const __half *)x, T, H, W, C, (__half *)y, (hipStream_t)stream_)
                      : avgpool3s2_launch<float>("ts_avgpool3s2_rows_forward", false, (const float *)x, T, H, W, C, (float *)y, (hipStream_t)stream_);
  if (rc != TS_OK) return rc;
  TS_CHECK_LAUNCH("ts_avgpool3s2_rows_forward");
  return TS_OK;
}

// grad_x [T, H, W, C] (every element written) from grad_y [T, Ho, Wo, C]
extern "C" int ts_avgpool3s2_rows_backward(const void *grad_y, int32_t T, int32_t H, int32_t W, int32_t C, int32_t half, void *grad_x,
                                           ts_stream_t stream_) {
  TS_REQUIRE(T >= 0 && H > 0 && W > 0 && C > 0, TS_ERR_INVALID_ARGUMENT, "ts_avgpool3s2_rows_backward: bad sizes");
  if (T == 0) return TS_OK;
  TS_REQUIRE(grad_y && grad_x, TS_ERR_INVALID_ARGUMENT, "ts_avgpool3s2_rows_backward: null pointer");
  const int rc = half ? avgpool3s2_launch<__half>("ts_avgpool3s2_rows_backward", true, (const __half *)grad_y, T, H, W, C, (__half *)grad_x, (hipStream_t)stream_)
                      : avgpool3s2_launch<float>("ts_avgpool3s2_rows_backward", true, (const float *)grad_y, T, H, W, C, (float *)grad_x, (hipStream_t)stream_);
  if (rc != TS_OK) return rc;
  TS_CHECK_LAUNCH("ts_avgpool3s2_rows_backward");
  return TS_OK;
}
