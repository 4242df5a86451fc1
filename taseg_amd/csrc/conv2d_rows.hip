// Dense 3 x 3 convolutions of TIAF's camera branch on channels-last half rows - the full-resolution layers of UNet2D
// (R/pcseg/model/segmentor/voxel/minkunet/unet2d.py:10-31,41-60: `nn.Conv2d(32, 32, (3, 3), padding = 1)` and the dilated twin
// `dilation = 2, padding = 2`; stem x 6, stage1 x 1: 4.9 M pixels each at bs 2).
//
// Why hand-written: MIOpen's best solver for this shape (its own find-db: ConvAsmImplicitGemmGTCDynamicFwdXdlopsNHWC) takes 0.60 ms
// for the plain and 1.5 ms for the dilated layer; the layer moves 315 MB in and 315 MB out (0.08 ms at 8 TB/s) and needs 90 GFLOP
// (0.04 ms of fp16 MFMA).  With 32 channels the whole weight (9 x 32 x 32 halfs = 18 KB) fits a wave's registers in MFMA operand
// form, so the kernel is: weights -> 72 VGPRs once per wave, then per 32-pixel segment of an output row 18 x
// `v_mfma_f32_32x32x16_f16` (M = output channel, N = pixel, K = 16 input channels: 2 per tap) whose B operands are 16-byte reads
// of the input rows (lane = pixel, half = channel block) from a [pixel][channel] image of the tile's rows in LDS.
// The four waves of a workgroup take four vertically adjacent segments and share the 4 + 2 D staged input rows; the result leaves
// as 16-byte stores (the two lanes of a pixel trade accumulator groups first: c2_pair_up).
//   forward        Y[t, y, x, co] = bias[co] + sum_{ky, kx, ci} X[t, y + (ky - 1) D, x + (kx - 1) D, ci] W[co, ci, ky, kx]
//   data gradient  dX = the same kernel over dY with the packed weights of mode 1 (taps mirrored, channels swapped), no bias
// fp32 accumulation (the MFMA's), one rounding to half at the store - what MIOpen's fp16 solvers do.
#include <hip/hip_fp16.h>

#include "common.h"

typedef _Float16 c2_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 c2_h4 __attribute__((ext_vector_type(4)));
typedef float c2_f16 __attribute__((ext_vector_type(16)));

#define C2_C 32

// Result layout of v_mfma_f32_32x32x16: lane (pixel r, half h) holds, per group g = 0 .. 3, the channels 8 g + 4 h .. + 3 of its pixel -
// stored as they come that is four 8-byte pieces per lane, and 8-byte stores of 64 lanes to 32 rows are what such a kernel then spends
// its time issuing.  One v_permlane32_swap per dword (gfx950) trades pieces between the two lanes of a pixel: the h = 0 lane ends up
// with channels 0 .. 15, the h = 1 lane with 16 .. 31, each as two 16-byte pieces side by side (p0: channels 16 h .. + 7, p1: + 8).
// Every lane of the wave must be active.
__device__ inline void c2_pair_up(const float (&a)[16], c2_h8 &p0, c2_h8 &p1) {
  union { c2_h4 v; unsigned d[2]; } g[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) g[k].v = (c2_h4){(_Float16)a[4 * k], (_Float16)a[4 * k + 1], (_Float16)a[4 * k + 2], (_Float16)a[4 * k + 3]};
  union { c2_h8 v; unsigned d[4]; } q0, q1;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    // swap(x, y): x' = {lower lanes: x, upper lanes: y of the lower lanes}, y' = {lower lanes: x of the upper lanes, upper lanes: y}
    const auto s02 = __builtin_amdgcn_permlane32_swap(g[0].d[d], g[2].d[d], false, false);
    const auto s13 = __builtin_amdgcn_permlane32_swap(g[1].d[d], g[3].d[d], false, false);
    q0.d[d] = s02[0]; q0.d[2 + d] = s02[1];
    q1.d[d] = s13[0]; q1.d[2 + d] = s13[1];
  }
  p0 = q0.v; p1 = q1.v;
}

// weight [co][ci][ky][kx] with element strides (s_co, s_ci, s_ky, s_kx) -> packed[tap][kb][lane] (8 halfs per lane): the A operand of
// v_mfma_f32_32x32x16_f16 for output row m = lane & 31 and reduction index k = 16 kb + 8 (lane >> 5) + j
//   mode 0 (forward):        m = co, k = ci, tap = (ky, kx):          W[m][k][ky][kx]
//   mode 1 (data gradient):  m = ci, k = co, tap = (ky, kx) mirrored: W[k][m][2 - ky][2 - kx]
__global__ __launch_bounds__(64) void conv3x3c32_pack_kernel(const _Float16 *__restrict__ w, int64_t s_co, int64_t s_ci, int64_t s_ky,
                                                             int64_t s_kx, int mode, c2_h8 *__restrict__ packed) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int tap = blockIdx.x / 2, kb = blockIdx.x % 2;
  const int ky = tap / 3, kx = tap % 3;
  c2_h8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 16 * kb + 8 * h + j;
    v[j] = mode == 0 ? w[r * s_co + k * s_ci + ky * s_ky + kx * s_kx] : w[k * s_co + r * s_ci + (2 - ky) * s_ky + (2 - kx) * s_kx];
  }
  packed[(size_t)blockIdx.x * 64 + lane] = v;
}

#define C2_PITCH 40            // halfs per staged pixel row (32 channels + 8: 16-byte multiple, 4 mod 8 dwords - the 16-byte fragment
                               // reads of a 16-lane group then cover the 64 banks exactly once)

// One tile = 4 output rows x 32 pixels.  The workgroup stages the 4 + 2 D input rows of the tile with their D-pixel halo in LDS
// ([pixel][channel] rows, zeros where the image ends), wave w computes output row w: per tap two 16-byte LDS reads per lane (B
// operand: lane = pixel, half = channel block) and two MFMAs against the weight fragments it holds in registers.  The NEXT tile's
// rows are already on their way from HBM while the current one is multiplied (global loads into registers before the MFMAs, LDS
// store after the barrier): the first form, which read its operands straight from global memory, ran at the latency of those
// reads (0.32 ms per layer); the weight gradient below uses the same pipeline.
template <int DIL>
__global__ __launch_bounds__(256) void conv3x3c32_rows_kernel(const _Float16 *__restrict__ X, const c2_h8 *__restrict__ Wp,
                                                              const float *__restrict__ bias, _Float16 *__restrict__ Y, int H, int W,
                                                              int tiles_x, int tiles_y, int n_tiles) {
  constexpr int XR = 4 + 2 * DIL, XW = 32 + 2 * DIL, N_X = XR * XW * 4, X_IT = (N_X + 255) / 256;
  __shared__ __attribute__((aligned(16))) _Float16 xs[XR * XW * C2_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  c2_h8 wr[9][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    wr[tap][0] = Wp[(tap * 2 + 0) * 64 + lane];
    wr[tap][1] = Wp[(tap * 2 + 1) * 64 + lane];
  }
  // accumulator register i of this lane belongs to output channel (i & 3) + 8 (i >> 2) + 4 h (column = pixel r)
  c2_f16 init;
#pragma unroll
  for (int i = 0; i < 16; ++i) init[i] = bias ? bias[(i & 3) + 8 * (i >> 2) + 4 * h] : 0.f;
  const c2_h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  c2_h8 nx[X_IT];
  auto fetch = [&](int tile) {                                   // the tile's input rows -> registers (zeros outside the image)
    const int xsi = tile % tiles_x, rest = tile / tiles_x;
    const int y0 = (rest % tiles_y) * 4, x0 = xsi * 32;
    const _Float16 *img = X + (size_t)(rest / tiles_y) * H * W * C2_C;
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int e = tid + it * 256;
      const int c8 = (e & 3) * 8, px = (e >> 2) % XW, row = (e >> 2) / XW;
      const int yy = y0 - DIL + row, xx = x0 - DIL + px;
      nx[it] = (e < N_X && yy >= 0 && yy < H && xx >= 0 && xx < W) ? *(const c2_h8 *)(img + ((size_t)yy * W + xx) * C2_C + c8) : zero;
    }
  };
  int tile = blockIdx.x;
  if (tile < n_tiles) fetch(tile);
  for (; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();                                             // (the previous tile's fragments have been read)
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int e = tid + it * 256;
      if (e < N_X) *(c2_h8 *)&xs[((e >> 2) / XW * XW + (e >> 2) % XW) * C2_PITCH + (e & 3) * 8] = nx[it];
    }
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < n_tiles) fetch(next);                             // in flight while this tile is multiplied
    const int xsi = tile % tiles_x, rest = tile / tiles_x;
    const int yb = rest % tiles_y, t = rest / tiles_y;
    const int y = yb * 4 + wave, x = xsi * 32 + r;
    c2_f16 acc = init;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const _Float16 *xrow = xs + ((wave + ky * DIL) * XW + r) * C2_PITCH + 8 * h;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const c2_h8 b0 = *(const c2_h8 *)(xrow + kx * DIL * C2_PITCH), b1 = *(const c2_h8 *)(xrow + kx * DIL * C2_PITCH + 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[ky * 3 + kx][0], b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[ky * 3 + kx][1], b1, acc, 0, 0, 0);
      }
    }
    float af[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) af[i] = acc[i];
    c2_h8 p0, p1;
    c2_pair_up(af, p0, p1);                                      // (all lanes: before the bounds check)
    if (y < H && x < W) {
      _Float16 *o = Y + (((size_t)t * H + y) * W + x) * C2_C + 16 * h;
      *(c2_h8 *)o = p0;
      *(c2_h8 *)(o + 8) = p1;
    }
  }
}

extern "C" size_t ts_conv3x3c32_packed_bytes(void) { return (size_t)9 * 2 * 64 * 16; }

// weight: the nn.Conv2d parameter [32 co][32 ci][3][3] (IEEE half) with its element strides; mode 0 = forward operand, 1 = data
// gradient operand; packed: ts_conv3x3c32_packed_bytes() bytes
extern "C" int ts_conv3x3c32_pack(const void *weight, int64_t s_co, int64_t s_ci, int64_t s_ky, int64_t s_kx, int32_t mode, void *packed,
                                  ts_stream_t stream) {
  TS_REQUIRE(weight && packed && (mode == 0 || mode == 1), TS_ERR_INVALID_ARGUMENT, "ts_conv3x3c32_pack: bad arguments");
  TS_REQUIRE((((uintptr_t)packed) & 15) == 0, TS_ERR_INVALID_ARGUMENT, "ts_conv3x3c32_pack: packed must be 16-byte aligned");
  conv3x3c32_pack_kernel<<<18, 64, 0, (hipStream_t)stream>>>((const _Float16 *)weight, s_co, s_ci, s_ky, s_kx, mode, (c2_h8 *)packed);
  TS_CHECK_LAUNCH("ts_conv3x3c32_pack");
  return TS_OK;
}

// y [T, H, W, 32] = conv3x3(x [T, H, W, 32], packed weights) (+ bias [32] float, may be NULL); stride 1, padding = dilation (1 or 2).
// x and y channels-last IEEE half, 16-byte aligned, not overlapping.
extern "C" int ts_conv3x3c32_rows(const void *x, const void *packed, const float *bias, int32_t T, int32_t H, int32_t W, int32_t dilation,
                                  void *y, ts_stream_t stream) {
  TS_REQUIRE(T >= 0 && H > 0 && W > 0 && (dilation == 1 || dilation == 2), TS_ERR_INVALID_ARGUMENT, "ts_conv3x3c32_rows: bad sizes");
  if (T == 0) return TS_OK;
  TS_REQUIRE(x && packed && y && x != y, TS_ERR_INVALID_ARGUMENT, "ts_conv3x3c32_rows: null / aliased pointer");
  TS_REQUIRE(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)packed)) & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_conv3x3c32_rows: pointers must be 16-byte aligned");
  const int tiles_x = (int)ts_cdiv(W, 32), tiles_y = (int)ts_cdiv(H, 4);
  const int64_t n_tiles = (int64_t)T * tiles_x * tiles_y;
  TS_REQUIRE(n_tiles < (1LL << 31) && (int64_t)T * H * W * C2_C < (1LL << 40), TS_ERR_UNSUPPORTED, "ts_conv3x3c32_rows: stack too large");
  // persistent workgroups: the 18 KB of packed weights are read once per wave, not once per tile
  const unsigned grid = (unsigned)std::min<int64_t>(n_tiles, 256 * 8);
  if (dilation == 1)
    conv3x3c32_rows_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>((const _Float16 *)x, (const c2_h8 *)packed, bias, (_Float16 *)y, H, W,
                                                                     tiles_x, tiles_y, (int)n_tiles);
  else
    conv3x3c32_rows_kernel<2><<<grid, 256, 0, (hipStream_t)stream>>>((const _Float16 *)x, (const c2_h8 *)packed, bias, (_Float16 *)y, H, W,
                                                                     tiles_x, tiles_y, (int)n_tiles);
  TS_CHECK_LAUNCH("ts_conv3x3c32_rows");
  return TS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same layers:  dW[co, ci, ky, kx] = sum_{t, y, x} X[t, y + (ky - 1) D, x + (kx - 1) D, ci] dY[t, y, x, co].
// Per tap a 32 x 32 product whose REDUCTION index is the pixel: both operands arrive pixel-major (channels-last rows), so a
// workgroup stages the rows it needs as [pixel][channel] images in LDS (4 output rows x 32 pixels of dY, the 4 + 2 D input rows with
// their D-pixel halo, zero where the image ends) and the MFMA fragments - 8 consecutive pixels of one channel per lane - come out of
// gfx950's transposing LDS read (ds_read_b64_tr_b16), as in the sparse weight gradient (conv_pairs_h.hip).  Wave w of a workgroup owns
// output row w of the tile and keeps all 9 x (32 x 32) sums in registers (144 accumulators) over the tiles of a persistent
// workgroup; the four waves' sums meet in LDS once at the end, the workgroup writes ONE partial [9][32][32] and a second launch adds
// the partials in index order: no atomics, run-to-run identical.  MIOpen's best solver for this shape: 0.95 ms; bytes: X and dY
// once, 630 MB.
#define CW_PITCH 40            // halfs per staged pixel row (32 channels + 8: 16-byte multiple, 4 mod 8 dwords)
typedef __fp16 c2_hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float c2_f4 __attribute__((ext_vector_type(4)));

template <int DIL>
__global__ __launch_bounds__(256) void conv3x3c32_wgrad_kernel(const _Float16 *__restrict__ X, const _Float16 *__restrict__ DY, int H,
                                                               int W, int tiles_x, int tiles_y, int n_tiles, float *__restrict__ part,
                                                               float *__restrict__ bias_part) {
  constexpr int XR = 4 + 2 * DIL, XW = 32 + 2 * DIL;
  constexpr int XS_HALFS = XR * XW * CW_PITCH, YS_HALFS = 4 * 32 * CW_PITCH;
  __shared__ __attribute__((aligned(16))) _Float16 smem[XS_HALFS + YS_HALFS];
  _Float16 *xs = smem, *ys = smem + XS_HALFS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kg = lane >> 4, tq = r16 >> 2, tp = lane & 3;
  const c2_h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  // transposed fragment: pixels r0 .. r0 + 7 (rows of the image) of channel c0 + r16
  auto frag = [&](const _Float16 *img, int r0, int c0) -> c2_h8 {
    typedef c2_hv4 __attribute__((address_space(3))) * lds_hv4;
    const c2_hv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_hv4)(img + (r0 + tq) * CW_PITCH + c0 + 4 * tp));
    const c2_hv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_hv4)(img + (r0 + 4 + tq) * CW_PITCH + c0 + 4 * tp));
    c2_h8 v;
    v[0] = (_Float16)lo[0]; v[1] = (_Float16)lo[1]; v[2] = (_Float16)lo[2]; v[3] = (_Float16)lo[3];
    v[4] = (_Float16)hi[0]; v[5] = (_Float16)hi[1]; v[6] = (_Float16)hi[2]; v[7] = (_Float16)hi[3];
    return v;
  };
  c2_f4 acc[9][2][2];
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[t9][a][b] = (c2_f4){0.f, 0.f, 0.f, 0.f};
  // the bias gradient (column sums of dY) rides along: one more product per co block, against a fragment of ones
  const c2_h8 ones = {1, 1, 1, 1, 1, 1, 1, 1};
  c2_f4 bacc[2] = {(c2_f4){0.f, 0.f, 0.f, 0.f}, (c2_f4){0.f, 0.f, 0.f, 0.f}};

  constexpr int N_X = XR * XW * 4, X_IT = (N_X + 255) / 256;
  c2_h8 nx[X_IT], ny[2];
  auto fetch = [&](int tile) {                                   // the tile's rows -> registers (zeros outside the image)
    const int xsi = tile % tiles_x, rest = tile / tiles_x;
    const int y0 = (rest % tiles_y) * 4, x0 = xsi * 32;
    const _Float16 *ximg = X + (size_t)(rest / tiles_y) * H * W * C2_C, *yimg = DY + (size_t)(rest / tiles_y) * H * W * C2_C;
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {                          // input rows y0 - D .. y0 + 3 + D, pixels x0 - D .. x0 + 31 + D
      const int e = tid + it * 256;
      const int c8 = (e & 3) * 8, px = (e >> 2) % XW, row = (e >> 2) / XW;
      const int yy = y0 - DIL + row, xx = x0 - DIL + px;
      nx[it] = (e < N_X && yy >= 0 && yy < H && xx >= 0 && xx < W) ? *(const c2_h8 *)(ximg + ((size_t)yy * W + xx) * C2_C + c8) : zero;
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {                             // output rows y0 .. y0 + 3, pixels x0 .. x0 + 31
      const int e = tid + it * 256;
      const int c8 = (e & 3) * 8, px = (e >> 2) & 31, row = e >> 7;
      const int yy = y0 + row, xx = x0 + px;
      ny[it] = (yy < H && xx < W) ? *(const c2_h8 *)(yimg + ((size_t)yy * W + xx) * C2_C + c8) : zero;
    }
  };
  int tile = blockIdx.x;
  if (tile < n_tiles) fetch(tile);
  for (; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();                                       // (the previous tile's fragments have been read)
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int e = tid + it * 256;
      if (e < N_X) *(c2_h8 *)&xs[((e >> 2) / XW * XW + (e >> 2) % XW) * CW_PITCH + (e & 3) * 8] = nx[it];
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int e = tid + it * 256;
      *(c2_h8 *)&ys[((e >> 7) * 32 + ((e >> 2) & 31)) * CW_PITCH + (e & 3) * 8] = ny[it];
    }
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < n_tiles) fetch(next);                       // in flight while this tile is multiplied
    // (no lane-dependent control flow from here to the end of the tile: the transposing reads need every lane)
    const _Float16 *yrow = ys + wave * 32 * CW_PITCH;
    const c2_h8 b0 = frag(yrow, 8 * kg, 0), b1 = frag(yrow, 8 * kg, 16);
    bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, b0, bacc[0], 0, 0, 0);
    bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, b1, bacc[1], 0, 0, 0);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const _Float16 *xrow = xs + (wave + ky * DIL) * XW * CW_PITCH;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int t9 = ky * 3 + kx;
        const c2_h8 a0 = frag(xrow, 8 * kg + kx * DIL, 0), a1 = frag(xrow, 8 * kg + kx * DIL, 16);
        acc[t9][0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[t9][0][0], 0, 0, 0);
        acc[t9][0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, acc[t9][0][1], 0, 0, 0);
        acc[t9][1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, acc[t9][1][0], 0, 0, 0);
        acc[t9][1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc[t9][1][1], 0, 0, 0);
      }
    }
  }
  // the four waves' sums, tap by tap through LDS (the staging images are free now), wave order 0 .. 3
  __syncthreads();
  float *red = (float *)smem;                              // 4 waves x [32][32] floats = 16 KB <= the two images together
  static_assert(sizeof(_Float16) * (XS_HALFS + YS_HALFS) >= 4 * 1024 * sizeof(float) && (XS_HALFS % 8) == 0,
                "LDS images too small for the reduction / second image misaligned");
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)                          // accumulator register q: ci = 16 a + 4 kg + q, co = 16 b + r16
          red[wave * 1024 + (16 * a + 4 * kg + q) * 32 + 16 * b + r16] = acc[t9][a][b][q];
    __syncthreads();
    for (int e = tid; e < 1024; e += 256)
      part[((size_t)blockIdx.x * 9 + t9) * 1024 + e] = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
    __syncthreads();
  }
  if (kg == 0) {                                           // (every row of the ones product holds the column sums: row 0)
    red[wave * 32 + r16] = bacc[0][0];
    red[wave * 32 + 16 + r16] = bacc[1][0];
  }
  __syncthreads();
  if (tid < 32) bias_part[(size_t)blockIdx.x * 32 + tid] = ((red[tid] + red[32 + tid]) + red[64 + tid]) + red[96 + tid];
}

// dW[co][ci][ky][kx] (element strides given, IEEE half) = sum over the workgroups' partials [n_part][9][ci][co] in index order.
// A workgroup owns 64 consecutive elements; its four waves each add a quarter of the partials (8 loads in flight per lane) and the
// four sums meet in LDS in wave order - a fixed order, run-to-run identical.  (The first form, one thread per element walking all
// 512 partials, took 0.12 ms per layer: as long as half the weight gradient itself.)
__global__ __launch_bounds__(256) void conv3x3c32_wgrad_reduce_kernel(const float *__restrict__ part, const float *__restrict__ bias_part,
                                                                      int n_part, _Float16 *__restrict__ dw, int64_t s_co, int64_t s_ci,
                                                                      int64_t s_ky, int64_t s_kx, float *__restrict__ db, int center_only) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;                    // (tap, ci, co): 9 * 1024 = 144 * 64 elements, then the 32 of the bias
  const bool is_w = e < 9 * 1024, live = e < 9 * 1024 + 32 && !(center_only && is_w && (e >> 10) != 4);   // (uniform per workgroup)
  const float *src = is_w ? part + e : bias_part + (e - 9 * 1024);
  const size_t pitch = is_w ? (size_t)9 * 1024 : (size_t)32;
  const int per = (n_part + 3) / 4, p0 = q * per, p1 = min(n_part, p0 + per);
  float s = 0.f;
  if (live) {
    int p = p0;
    for (; p + 8 <= p1; p += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(p + j) * pitch];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; p < p1; ++p) s += src[(size_t)p * pitch];
  }
  red[q][lane] = s;
  __syncthreads();
  if (q != 0 || !live) return;
  s = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
  if (is_w) {
    const int t9 = e >> 10, ci = (e >> 5) & 31, co = e & 31;
    dw[co * s_co + ci * s_ci + (center_only ? 0 : (t9 / 3) * s_ky + (t9 % 3) * s_kx)] = (_Float16)s;
  } else if (db) {
    db[e - 9 * 1024] = s;
  }
}

#define C2_WGRAD_WGS 512
extern "C" size_t ts_conv3x3c32_wgrad_workspace_bytes(void) { return (size_t)C2_WGRAD_WGS * (9 * 1024 + 32) * sizeof(float); }

// grad_weight [32 co][32 ci][3][3] (IEEE half, element strides given: the layout of the weight it belongs to) and grad_bias [32]
// (float, may be NULL) from x and grad_y [T, H, W, 32] channels-last half; ws >= ts_conv3x3c32_wgrad_workspace_bytes()
extern "C" int ts_conv3x3c32_wgrad(const void *x, const void *grad_y, int32_t T, int32_t H, int32_t W, int32_t dilation, void *grad_weight,
                                   int64_t s_co, int64_t s_ci, int64_t s_ky, int64_t s_kx, float *grad_bias, void *ws, size_t ws_bytes,
                                   ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  // (dilation 0: the 1 x 1 layer of ts_conv1x1c32_wgrad below - the same pass, only the centre tap is written)
  TS_REQUIRE(T > 0 && H > 0 && W > 0 && dilation >= 0 && dilation <= 2, TS_ERR_INVALID_ARGUMENT, "ts_conv3x3c32_wgrad: bad sizes");
  TS_REQUIRE(x && grad_y && grad_weight && ws && ws_bytes >= ts_conv3x3c32_wgrad_workspace_bytes(), TS_ERR_INVALID_ARGUMENT,
             "ts_conv3x3c32_wgrad: null pointer / workspace too small");
  TS_REQUIRE(((((uintptr_t)x) | ((uintptr_t)grad_y) | ((uintptr_t)ws)) & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_conv3x3c32_wgrad: pointers must be 16-byte aligned");
  const int tiles_x = (int)ts_cdiv(W, 32), tiles_y = (int)ts_cdiv(H, 4);
  const int64_t n_tiles = (int64_t)T * tiles_x * tiles_y;
  TS_REQUIRE(n_tiles < (1LL << 31), TS_ERR_UNSUPPORTED, "ts_conv3x3c32_wgrad: stack too large");
  const int grid = (int)std::min<int64_t>(n_tiles, C2_WGRAD_WGS);
  float *part = (float *)ws, *bias_part = part + (size_t)C2_WGRAD_WGS * 9 * 1024;
  if (dilation <= 1)
    conv3x3c32_wgrad_kernel<1><<<grid, 256, 0, stream>>>((const _Float16 *)x, (const _Float16 *)grad_y, H, W, tiles_x, tiles_y, (int)n_tiles,
                                                         part, bias_part);
  else
    conv3x3c32_wgrad_kernel<2><<<grid, 256, 0, stream>>>((const _Float16 *)x, (const _Float16 *)grad_y, H, W, tiles_x, tiles_y, (int)n_tiles,
                                                         part, bias_part);
  conv3x3c32_wgrad_reduce_kernel<<<145, 256, 0, stream>>>(part, bias_part, grid, (_Float16 *)grad_weight, s_co, s_ci, s_ky, s_kx, grad_bias,
                                                          dilation == 0);
  TS_CHECK_LAUNCH("ts_conv3x3c32_wgrad");
  return TS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The 1 x 1, 32 -> 32 channel layers in front of the blocks (ResContextBlock.conv1 of stem[1], stem[2], ResBlock.conv1 of stage 1:
// unet2d.py:10-12,41-43 - `act1(conv1(x))`, a LeakyReLU straight behind the layer): a [pixels, 32] x [32, 32] product, bias and
// activation in the epilogue.  No LDS: a lane's two 16-byte loads ARE the B operand (lane = pixel, half = channel block), a wave keeps
// four 32-pixel segments in flight.  630 MB per layer at full scale; MIOpen + ATen's LeakyReLU take 0.31 + 0.07 ms.
// With a mode-1 pack (channels swapped) and no activation the same kernel is the data gradient.
template <bool LEAKY>
__global__ __launch_bounds__(256) void conv1x1c32_kernel(const _Float16 *__restrict__ X, const c2_h8 *__restrict__ Wp,
                                                         const float *__restrict__ bias, float slope, _Float16 *__restrict__ Y, int64_t n_px) {
  constexpr int UNR = 4;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const c2_h8 w0 = Wp[lane], w1 = Wp[64 + lane];
  c2_f16 init;
#pragma unroll
  for (int i = 0; i < 16; ++i) init[i] = bias ? bias[(i & 3) + 8 * (i >> 2) + 4 * h] : 0.f;
  const c2_h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t n_seg = (n_px + 31) / 32, waves = (int64_t)gridDim.x * 4;
  for (int64_t s0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * UNR; s0 < n_seg; s0 += waves * UNR) {
    c2_h8 b[UNR][2];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t px = (s0 + u) * 32 + r;
      const bool ok = px < n_px;
      const c2_h8 *p = (const c2_h8 *)(X + (ok ? px : 0) * C2_C + 8 * h);
      b[u][0] = ok ? p[0] : zero;
      b[u][1] = ok ? p[2] : zero;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      c2_f16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, b[u][0], init, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b[u][1], acc, 0, 0, 0);
      const int64_t px = (s0 + u) * 32 + r;
      float af[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const _Float16 y = (_Float16)acc[i];                     // (rounded first: LeakyReLU acts on the half value, as the module pair does)
        af[i] = LEAKY ? (float)(y > (_Float16)0 ? y : (_Float16)((float)y * slope)) : (float)y;
      }
      c2_h8 p0, p1;
      c2_pair_up(af, p0, p1);                                    // (all lanes: before the bounds check)
      if (px < n_px) {
        _Float16 *o = Y + px * C2_C + 16 * h;
        *(c2_h8 *)o = p0;
        *(c2_h8 *)(o + 8) = p1;
      }
    }
  }
}

// packed operand of a [32 co][32 ci] weight (element strides given): 2 x 64 x 16 bytes; mode 0 forward, 1 data gradient
extern "C" size_t ts_conv1x1c32_packed_bytes(void) { return (size_t)2 * 64 * 16; }

extern "C" int ts_conv1x1c32_pack(const void *weight, int64_t s_co, int64_t s_ci, int32_t mode, void *packed, ts_stream_t stream) {
  TS_REQUIRE(weight && packed && (mode == 0 || mode == 1), TS_ERR_INVALID_ARGUMENT, "ts_conv1x1c32_pack: bad arguments");
  TS_REQUIRE((((uintptr_t)packed) & 15) == 0, TS_ERR_INVALID_ARGUMENT, "ts_conv1x1c32_pack: packed must be 16-byte aligned");
  // (the 3 x 3 pack with tap strides 0: blocks 0 and 1 are tap (0, 0) = the weight itself)
  conv3x3c32_pack_kernel<<<2, 64, 0, (hipStream_t)stream>>>((const _Float16 *)weight, s_co, s_ci, 0, 0, mode, (c2_h8 *)packed);
  TS_CHECK_LAUNCH("ts_conv1x1c32_pack");
  return TS_OK;
}

// y [n_pixels, 32] = x [n_pixels, 32] W^T + bias [32] (float, may be NULL), then LeakyReLU(slope) if leaky != 0; IEEE-half rows
// (a channels-last stack is its [T H W, 32] rows), 16-byte aligned, not overlapping
extern "C" int ts_conv1x1c32_rows(const void *x, const void *packed, const float *bias, int64_t n_pixels, int32_t leaky, float slope, void *y,
                                  ts_stream_t stream) {
  TS_REQUIRE(n_pixels >= 0 && n_pixels < (1LL << 40), TS_ERR_INVALID_ARGUMENT, "ts_conv1x1c32_rows: bad sizes");
  if (n_pixels == 0) return TS_OK;
  TS_REQUIRE(x && packed && y && x != y, TS_ERR_INVALID_ARGUMENT, "ts_conv1x1c32_rows: null / aliased pointer");
  TS_REQUIRE(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)packed)) & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_conv1x1c32_rows: pointers must be 16-byte aligned");
  const int64_t n_seg = (n_pixels + 31) / 32;
  const unsigned grid = (unsigned)std::min<int64_t>(ts_cdiv(n_seg, 16), 256 * 16);
  if (leaky)
    conv1x1c32_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>((const _Float16 *)x, (const c2_h8 *)packed, bias, slope, (_Float16 *)y, n_pixels);
  else
    conv1x1c32_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>((const _Float16 *)x, (const c2_h8 *)packed, bias, slope, (_Float16 *)y, n_pixels);
  TS_CHECK_LAUNCH("ts_conv1x1c32_rows");
  return TS_OK;
}

// grad_weight [32 co][32 ci] (IEEE half, element strides given) and grad_bias [32] (float, may be NULL) of the 1 x 1 layer from x and
// g = the gradient at the layer's output (behind the activation's own backward), [T, H, W, 32] rows: the 3 x 3 weight gradient's pass
// (its centre tap), workspace as there
extern "C" int ts_conv3x3c32_wgrad(const void *, const void *, int32_t, int32_t, int32_t, int32_t, void *, int64_t, int64_t, int64_t, int64_t,
                                   float *, void *, size_t, ts_stream_t);
extern "C" int ts_conv1x1c32_wgrad(const void *x, const void *g, int32_t T, int32_t H, int32_t W, void *grad_weight, int64_t s_co, int64_t s_ci,
                                   float *grad_bias, void *ws, size_t ws_bytes, ts_stream_t stream) {
  return ts_conv3x3c32_wgrad(x, g, T, H, W, 0, grad_weight, s_co, s_ci, 0, 0, grad_bias, ws, ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The decoder's wide layers: UpBlock.conv1 of up3 (96 -> 96 channels at 1/2 scale) and up4 (56 -> 96 at full scale)
// (R/pcseg/model/segmentor/voxel/minkunet/unet2d.py:81-115; plain 3 x 3, stride 1, padding 1).  MIOpen's best solvers for the up4
// shape take 4.4 ms forward and 1.5 ms for the data gradient (tools/unet2d_layers.py; the maps are 550 + 945 MB: 0.19 ms of HBM, 0.2 ms
// of fp16 MFMA).  Same scheme as the 32-channel kernel above with the channel counts opened up:
//   - C_in any multiple of 8 up to 96 (staged in LDS zero-padded to 16 KB channels: KB = 2, 4, 6 K-blocks per tap),
//   - C_out any multiple of 8, in blocks of 32.  A WAVE owns one output block and keeps its 9 x KB weight fragments in registers
//     (36 KB VGPRs of the 512 a lone wave of a SIMD may hold); the NCOB waves of the blocks share ONE staged tile of input rows, so
//     the input is read from HBM and written to LDS once, not once per block (the first form - a workgroup per block - ran at the
//     bytes it kept in flight: 1.0 ms forward at the up4 shape), and RG such wave groups split the tile's rows,
//   - a wave walks its rows two at a time: a B fragment (one input row, tap column, K block) read from LDS once feeds both output
//     rows it belongs to - at one row per pass the LDS reads (1 KB per MFMA) are as long as the MFMAs themselves,
//   - the rows of the next tile are on their way while a tile is multiplied, and the last row pair's result leaves during the next
//     tile (stores count against the same counter as loads here: a store just before the loop's wait would be waited for).
template <int KB, int NCOB, int RG, int TR>
__global__ __launch_bounds__(64 * NCOB * RG) void conv3x3_rows_kernel(const _Float16 *__restrict__ X, int x_ch, const c2_h8 *__restrict__ Wp,
                                                                      const float *__restrict__ bias, _Float16 *__restrict__ Y, int y_ch,
                                                                      int H, int W, int tiles_x, int tiles_y, int n_tiles, int n_cob,
                                                                      int n_groups) {
  constexpr int NT = 64 * NCOB * RG, RPWV = TR / RG, NP = RPWV / 2;
  static_assert(TR % RG == 0 && RPWV % 2 == 0, "rows per wave must be even");
  constexpr int XR = TR + 2, XW = 34, NCH = 2 * KB, PITCH = 16 * KB + 8;
  constexpr int N_X = XR * XW * NCH, X_IT = (N_X + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) _Float16 xs_dyn[];
  _Float16 *xs = xs_dyn;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int group = blockIdx.x % n_groups, first = blockIdx.x / n_groups, step = gridDim.x / n_groups;
  const int cob = group * NCOB + wave % NCOB, row0 = (wave / NCOB) * RPWV;    // this wave's output block and first row in the tile
  const bool live = cob < n_cob;                                              // (wave-uniform; idle waves still stage and wait)
  const int x_chunks = x_ch >> 3;
  c2_h8 wr[9][KB];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) wr[tap][kb] = Wp[((size_t)((live ? cob : 0) * 9 + tap) * KB + kb) * 64 + lane];
  // accumulator register i of this lane belongs to output channel 32 cob + (i & 3) + 8 (i >> 2) + 4 h (column = pixel r)
  c2_f16 init;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int co = 32 * cob + (i & 3) + 8 * (i >> 2) + 4 * h;
    init[i] = (bias && co < y_ch) ? bias[co] : 0.f;
  }
  const c2_h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  c2_h8 nx[X_IT];
  // (the offsets of a thread's pieces are recomputed per tile on purpose: kept in registers across tiles they push the prefetched
  // rows out to AGPRs behind a wait per load - KB = 6 ran twice as long that way)
  auto fetch = [&](int tile) {                                   // the tile's input rows -> registers (zeros outside the image / beyond C_in)
    const int xsi = tile % tiles_x, rest = tile / tiles_x;
    const int y0 = (rest % tiles_y) * TR, x0 = xsi * 32;
    const _Float16 *img = X + (size_t)(rest / tiles_y) * H * W * x_ch;
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int e = tid + it * NT;
      const int ch = e % NCH, px = (e / NCH) % XW, row = (e / NCH) / XW;
      const int yy = y0 - 1 + row, xx = x0 - 1 + px;
      nx[it] = (e < N_X && ch < x_chunks && yy >= 0 && yy < H && xx >= 0 && xx < W)
                   ? *(const c2_h8 *)(img + ((size_t)yy * W + xx) * x_ch + 8 * ch) : zero;
    }
  };
  c2_h8 pend[2][2];
  int pend_tile = -1;
  auto store = [&](int tile, int rows_at, const c2_h8 (&v)[2][2]) {      // two output rows of this wave's block (c2_pair_up's pieces)
    const int xsi = tile % tiles_x, rest = tile / tiles_x;
    const int yb = rest % tiles_y, t = rest / tiles_y;
    const int x = xsi * 32 + r;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int y = yb * TR + rows_at + o;
      if (live && y < H && x < W) {
        _Float16 *op = Y + (((size_t)t * H + y) * W + x) * y_ch + 32 * cob + 16 * h;
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (32 * cob + 16 * h + 8 * k < y_ch) *(c2_h8 *)(op + 8 * k) = v[o][k];
      }
    }
  };
  int tile = first;
  if (tile < n_tiles) fetch(tile);
  for (; tile < n_tiles; tile += step) {
    __syncthreads();                                             // (the previous tile's fragments have been read)
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int e = tid + it * NT;
      if (e < N_X) *(c2_h8 *)&xs[(e / NCH) * PITCH + (e % NCH) * 8] = nx[it];
    }
    __syncthreads();
    const int next = tile + step;
    if (next < n_tiles) fetch(next);                             // in flight while this tile is multiplied
    if (pend_tile >= 0) store(pend_tile, row0 + 2 * (NP - 1), pend);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      c2_f16 acc[2] = {init, init};
      // The 12 (input row, tap column) groups of a row pair, software-pipelined by hand: group g + 1's KB fragments are requested
      // before group g's MFMAs (left to itself the compiler keeps ONE LDS read ahead of each MFMA, and with one wave on the SIMD the
      // rest of the read's latency is a stall per MFMA: forward at the up4 shape 0.84 -> 0.71 ms with whole rows requested at once,
      // -> this form)
      const _Float16 *xbase = xs + ((row0 + 2 * p) * XW + r) * PITCH + 8 * h;
      c2_h8 bf[2][KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) bf[0][kb] = *(const c2_h8 *)(xbase + 16 * kb);
#pragma unroll
      for (int g = 0; g < 12; ++g) {
        const int ir = g / 3, kx = g % 3;                        // input row row0 + 2 p + ir of the staged image, tap column kx
        if (g + 1 < 12) {
          const int ir1 = (g + 1) / 3, kx1 = (g + 1) % 3;
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) bf[(g + 1) & 1][kb] = *(const c2_h8 *)(xbase + (ir1 * XW + kx1) * PITCH + 16 * kb);
        }
        __builtin_amdgcn_sched_barrier(0);                       // (the scheduler would sink each read back to its MFMA)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
          for (int o = 0; o < 2; ++o) {
            const int ky = ir - o;                               // output row o takes this input row at tap row ky
            if (ky >= 0 && ky < 3) acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[ky * 3 + kx][kb], bf[g & 1][kb], acc[o], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      c2_h8 out[2][2];
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        float af[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) af[i] = acc[o][i];
        c2_pair_up(af, out[o][0], out[o][1]);                    // (all lanes: the bounds check is in store)
      }
      if (p + 1 < NP) {
        store(tile, row0 + 2 * p, out);
      } else {
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int k = 0; k < 2; ++k) pend[o][k] = out[o][k];
      }
    }
    pend_tile = tile;
  }
  if (pend_tile >= 0) store(pend_tile, row0 + 2 * (NP - 1), pend);
}

// packed[cob][tap][kb][lane] for the kernel above: rows m = 32 cob + (lane & 31) < M, reduction k = 16 kb + 8 (lane >> 5) + j < K,
// zero beyond;  mode 0: (M, K) = (C_out, C_in), W[m][k][ky][kx];  mode 1: (M, K) = (C_in, C_out), W[k][m][2 - ky][2 - kx]
__global__ __launch_bounds__(64) void conv3x3_pack_kernel(const _Float16 *__restrict__ w, int64_t s_co, int64_t s_ci, int64_t s_ky,
                                                          int64_t s_kx, int mode, int M, int K, int KB, c2_h8 *__restrict__ packed) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int kb = blockIdx.x % KB, tap = (blockIdx.x / KB) % 9, cob = blockIdx.x / (9 * KB);
  const int ky = tap / 3, kx = tap % 3, m = 32 * cob + r;
  c2_h8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 16 * kb + 8 * h + j;
    v[j] = (m < M && k < K) ? (mode == 0 ? w[m * s_co + k * s_ci + ky * s_ky + kx * s_kx] : w[k * s_co + m * s_ci + (2 - ky) * s_ky + (2 - kx) * s_kx])
                            : (_Float16)0;
  }
  packed[(size_t)blockIdx.x * 64 + lane] = v;
}

static int c2_kb_of(int k_channels) { return k_channels <= 32 ? 2 : k_channels <= 64 ? 4 : 6; }

// bytes of the packed operand of a layer with k_channels reduction channels (the input of the call: C_in forward, C_out for the data
// gradient) and m_channels result channels; 0 if the kernel does not take the layer
extern "C" size_t ts_conv3x3_rows_packed_bytes(int32_t k_channels, int32_t m_channels) {
  if (k_channels < 8 || k_channels > 96 || (k_channels & 7) || m_channels < 8 || (m_channels & 7)) return 0;
  return (size_t)ts_cdiv(m_channels, 32) * 9 * c2_kb_of(k_channels) * 64 * 16;
}

// weight: the nn.Conv2d parameter [c_out][c_in][3][3] (IEEE half) with its element strides; mode 0 = forward operand, 1 = data
// gradient operand; packed: ts_conv3x3_rows_packed_bytes(mode 0: c_in, c_out; mode 1: c_out, c_in) bytes
extern "C" int ts_conv3x3_rows_pack(const void *weight, int32_t c_out, int32_t c_in, int64_t s_co, int64_t s_ci, int64_t s_ky, int64_t s_kx,
                                    int32_t mode, void *packed, ts_stream_t stream) {
  TS_REQUIRE(weight && packed && (mode == 0 || mode == 1), TS_ERR_INVALID_ARGUMENT, "ts_conv3x3_rows_pack: bad arguments");
  const int M = mode == 0 ? c_out : c_in, K = mode == 0 ? c_in : c_out;
  TS_REQUIRE(ts_conv3x3_rows_packed_bytes(K, M) != 0, TS_ERR_UNSUPPORTED,
             "ts_conv3x3_rows_pack: reduction channels a multiple of 8 up to 96, result channels a multiple of 8");
  TS_REQUIRE((((uintptr_t)packed) & 15) == 0, TS_ERR_INVALID_ARGUMENT, "ts_conv3x3_rows_pack: packed must be 16-byte aligned");
  const int KB = c2_kb_of(K), n_cob = (int)ts_cdiv(M, 32);
  conv3x3_pack_kernel<<<n_cob * 9 * KB, 64, 0, (hipStream_t)stream>>>((const _Float16 *)weight, s_co, s_ci, s_ky, s_kx, mode, M, K, KB,
                                                                      (c2_h8 *)packed);
  TS_CHECK_LAUNCH("ts_conv3x3_rows_pack");
  return TS_OK;
}

template <int KB, int NCOB, int RG, int TR>
static int conv3x3_rows_launch(const _Float16 *x, int x_ch, const c2_h8 *packed, const float *bias, int T, int H, int W, _Float16 *y, int y_ch,
                               hipStream_t stream) {
  constexpr size_t lds = (size_t)(TR + 2) * 34 * (16 * KB + 8) * sizeof(_Float16);
  static_assert(lds <= 160 * 1024, "staged rows exceed the LDS of a CU");
  static bool attr_set = false;
  auto kern = conv3x3_rows_kernel<KB, NCOB, RG, TR>;
  if (lds > 64 * 1024 && !attr_set) {
    TS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
    attr_set = true;
  }
  const int tiles_x = (int)ts_cdiv(W, 32), tiles_y = (int)ts_cdiv(H, TR), n_cob = (int)ts_cdiv(y_ch, 32);
  const int n_groups = (int)ts_cdiv(n_cob, NCOB);                // (more than NCOB output blocks: workgroups side by side per tile)
  const int64_t n_tiles = (int64_t)T * tiles_x * tiles_y;
  TS_REQUIRE(n_tiles < (1LL << 31) && (int64_t)T * H * W * std::max(x_ch, y_ch) < (1LL << 40), TS_ERR_UNSUPPORTED,
             "ts_conv3x3_rows: stack too large");
  // persistent workgroups: the packed weights are read once per wave, not once per tile
  const unsigned per_group = (unsigned)std::min<int64_t>(n_tiles, 1024);
  kern<<<per_group * n_groups, 64 * NCOB * RG, lds, stream>>>(x, x_ch, packed, bias, y, y_ch, H, W, tiles_x, tiles_y, (int)n_tiles, n_cob,
                                                              n_groups);
  TS_CHECK_LAUNCH("ts_conv3x3_rows");
  return TS_OK;
}

template <int KB>
static int conv3x3_rows_by_blocks(const _Float16 *x, int x_ch, const c2_h8 *packed, const float *bias, int T, int H, int W, _Float16 *y, int y_ch,
                                  hipStream_t stream) {
  // waves = output blocks x row groups: 1 x 4, 2 x 2, 3 x 1, 4 x 1 (more blocks: several workgroups per tile)
  constexpr int TR = KB == 6 ? 4 : 8;
  switch ((int)ts_cdiv(y_ch, 32)) {
    case 1: return conv3x3_rows_launch<KB, 1, 4, 8>(x, x_ch, packed, bias, T, H, W, y, y_ch, stream);
    case 2: return conv3x3_rows_launch<KB, 2, 2, TR>(x, x_ch, packed, bias, T, H, W, y, y_ch, stream);
    case 3: return conv3x3_rows_launch<KB, 3, 1, TR>(x, x_ch, packed, bias, T, H, W, y, y_ch, stream);
    default: return conv3x3_rows_launch<KB, 4, 1, TR>(x, x_ch, packed, bias, T, H, W, y, y_ch, stream);
  }
}

// y [T, H, W, y_channels] = conv3x3(x [T, H, W, x_channels], packed weights) (+ bias [y_channels] float, may be NULL); stride 1,
// padding 1.  x and y channels-last IEEE half, 16-byte aligned, not overlapping; with a mode-1 pack and x = grad_y: grad_x.
extern "C" int ts_conv3x3_rows(const void *x, int32_t x_channels, const void *packed, const float *bias, int32_t T, int32_t H, int32_t W,
                               void *y, int32_t y_channels, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(T >= 0 && H > 0 && W > 0, TS_ERR_INVALID_ARGUMENT, "ts_conv3x3_rows: bad sizes");
  TS_REQUIRE(ts_conv3x3_rows_packed_bytes(x_channels, y_channels) != 0, TS_ERR_UNSUPPORTED,
             "ts_conv3x3_rows: input channels a multiple of 8 up to 96, output channels a multiple of 8");
  if (T == 0) return TS_OK;
  TS_REQUIRE(x && packed && y && x != y, TS_ERR_INVALID_ARGUMENT, "ts_conv3x3_rows: null / aliased pointer");
  TS_REQUIRE(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)packed)) & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_conv3x3_rows: pointers must be 16-byte aligned");
  const _Float16 *xp = (const _Float16 *)x;
  _Float16 *yp = (_Float16 *)y;
  const c2_h8 *pp = (const c2_h8 *)packed;
  switch (c2_kb_of(x_channels)) {
    case 2: return conv3x3_rows_by_blocks<2>(xp, x_channels, pp, bias, T, H, W, yp, y_channels, stream);
    case 4: return conv3x3_rows_by_blocks<4>(xp, x_channels, pp, bias, T, H, W, yp, y_channels, stream);
    default: return conv3x3_rows_by_blocks<6>(xp, x_channels, pp, bias, T, H, W, yp, y_channels, stream);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weight (and bias) gradient of the general layers:  dW[co, ci, ky, kx] = sum_{t, y, x} X[t, y + ky - 1, x + kx - 1, ci] dY[t, y, x, co],
// db[co] = sum dY[t, y, x, co].  MIOpen's best solver: 2.4 ms at the up4 shape (56 -> 96 channels, full scale), 1.2 ms at up3's; ATen's
// bias sum over the 96-channel map another 0.2 - 0.4 ms.  The 9 x C_in x C_out fp32 sums (83 k for 96 x 96) stay in REGISTERS for the
// whole run of a persistent workgroup: its four waves split the (ci, co) plane 2 x 2, wave (a, b) owning NI x NO blocks of 16 x 16
// (v_mfma_f32_16x16x32_f16, K = 32 pixels of a row: 36 NI NO accumulator registers, 324 for 96 x 96), and every wave walks all TR
// rows of the staged tile.  Operands as in the 32-channel kernel: [pixel][channel] images of the tile's rows in LDS, fragments through
// the transposing read.  The bias sums ride along as one more product against a fragment of ones.  Per workgroup one partial
// [9][CIP][COP] (+ [COP]); a second launch adds the partials in a fixed order.
template <int NI, int NO, int TR>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const _Float16 *__restrict__ X, int x_ch, int x_off, int x_chunks,
                                                            const _Float16 *__restrict__ DY, int y_ch, int H, int W, int tiles_x, int tiles_y,
                                                            int n_tiles, float *__restrict__ part, float *__restrict__ bias_part) {
  constexpr int CIP = 32 * NI, COP = 32 * NO, PI = CIP + 8, PO = COP + 8, XR = TR + 2, XW = 34;
  constexpr int XS_HALFS = XR * XW * PI;
  constexpr int NCI = CIP / 8, NCO = COP / 8, N_X = XR * XW * NCI, N_Y = TR * 32 * NCO, X_IT = (N_X + 255) / 256, Y_IT = (N_Y + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) _Float16 wg_dyn[];
  _Float16 *xs = wg_dyn, *ys = wg_dyn + XS_HALFS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wa = wave >> 1, wb = wave & 1;
  const int r16 = lane & 15, kg = lane >> 4, tq = r16 >> 2, tp = lane & 3;
  const int y_chunks = y_ch >> 3;                              // (x: channels x_off .. x_off + 8 x_chunks - 1 of the x_ch in a row)
  const c2_h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  const c2_h8 ones = {1, 1, 1, 1, 1, 1, 1, 1};
  // transposed fragment: pixels r0 .. r0 + 7 (rows of the image, `pitch` halfs apart) of channel c0 + r16
  auto frag = [&](const _Float16 *img, int pitch, int r0, int c0) -> c2_h8 {
    typedef c2_hv4 __attribute__((address_space(3))) * lds_hv4;
    const c2_hv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_hv4)(img + (r0 + tq) * pitch + c0 + 4 * tp));
    const c2_hv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_hv4)(img + (r0 + 4 + tq) * pitch + c0 + 4 * tp));
    c2_h8 v;
    v[0] = (_Float16)lo[0]; v[1] = (_Float16)lo[1]; v[2] = (_Float16)lo[2]; v[3] = (_Float16)lo[3];
    v[4] = (_Float16)hi[0]; v[5] = (_Float16)hi[1]; v[6] = (_Float16)hi[2]; v[7] = (_Float16)hi[3];
    return v;
  };
  c2_f4 acc[9][NI][NO], bacc[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    bacc[o] = (c2_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[t9][i][o] = (c2_f4){0.f, 0.f, 0.f, 0.f};
  }
  c2_h8 nx[X_IT], ny[Y_IT];
  auto fetch = [&](int tile) {                                   // the tile's rows -> registers (zeros outside the image / beyond the channels)
    const int xsi = tile % tiles_x, rest = tile / tiles_x;
    const int y0 = (rest % tiles_y) * TR, x0 = xsi * 32;
    const _Float16 *ximg = X + (size_t)(rest / tiles_y) * H * W * x_ch + x_off, *yimg = DY + (size_t)(rest / tiles_y) * H * W * y_ch;
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {                          // input rows y0 - 1 .. y0 + TR, pixels x0 - 1 .. x0 + 32
      const int e = tid + it * 256;
      const int ch = e % NCI, px = (e / NCI) % XW, row = (e / NCI) / XW;
      const int yy = y0 - 1 + row, xx = x0 - 1 + px;
      nx[it] = (e < N_X && ch < x_chunks && yy >= 0 && yy < H && xx >= 0 && xx < W)
                   ? *(const c2_h8 *)(ximg + ((size_t)yy * W + xx) * x_ch + 8 * ch) : zero;
    }
#pragma unroll
    for (int it = 0; it < Y_IT; ++it) {                          // output rows y0 .. y0 + TR - 1, pixels x0 .. x0 + 31
      const int e = tid + it * 256;
      const int ch = e % NCO, px = (e / NCO) & 31, row = (e / NCO) >> 5;
      const int yy = y0 + row, xx = x0 + px;
      ny[it] = (e < N_Y && ch < y_chunks && yy < H && xx < W) ? *(const c2_h8 *)(yimg + ((size_t)yy * W + xx) * y_ch + 8 * ch) : zero;
    }
  };
  int tile = blockIdx.x;
  if (tile < n_tiles) fetch(tile);
  for (; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();                                             // (the previous tile's fragments have been read)
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int e = tid + it * 256;
      if (e < N_X) *(c2_h8 *)&xs[(e / NCI) * PI + (e % NCI) * 8] = nx[it];
    }
#pragma unroll
    for (int it = 0; it < Y_IT; ++it) {
      const int e = tid + it * 256;
      if (e < N_Y) *(c2_h8 *)&ys[(e / NCO) * PO + (e % NCO) * 8] = ny[it];
    }
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < n_tiles) fetch(next);                             // in flight while this tile is multiplied
    // (no lane-dependent control flow from here to the end of the tile: the transposing reads need every lane)
#pragma unroll
    for (int row = 0; row < TR; ++row) {
      const _Float16 *yrow = ys + row * 32 * PO;
      c2_h8 b[NO];
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        b[o] = frag(yrow, PO, 8 * kg, (wb * NO + o) * 16);
        bacc[o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, b[o], bacc[o], 0, 0, 0);
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const _Float16 *xrow = xs + (row + ky) * XW * PI;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const c2_h8 a = frag(xrow, PI, 8 * kg + kx, (wa * NI + i) * 16);     // (requesting a tap row's fragments ahead of their
#pragma unroll                                                                   // MFMAs, as the forward kernel does: no gain here)
            for (int o = 0; o < NO; ++o)
              acc[ky * 3 + kx][i][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[o], acc[ky * 3 + kx][i][o], 0, 0, 0);
          }
        }
      }
    }
  }
  // accumulator register q of block (i, o): ci = 16 (wa NI + i) + 4 kg + q, co = 16 (wb NO + o) + r16
  float *mine = part + (size_t)blockIdx.x * 9 * CIP * COP;
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          mine[((size_t)t9 * CIP + 16 * (wa * NI + i) + 4 * kg + q) * COP + 16 * (wb * NO + o) + r16] = acc[t9][i][o][q];
  if (wa == 0 && kg == 0)                                        // (every row of the ones product holds the column sums: row 0)
#pragma unroll
    for (int o = 0; o < NO; ++o) bias_part[(size_t)blockIdx.x * COP + 16 * (wb * NO + o) + r16] = bacc[o][0];
}

// dW[co][ci][ky][kx] (element strides given, IEEE half) and db[co] (float) = sums over the workgroups' partials in index order;
// elements [0, 9 CIP COP): (tap, ci, co) of the weight, [9 CIP COP, 9 CIP COP + COP): the bias
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float *__restrict__ part, const float *__restrict__ bias_part, int n_part,
                                                                   int CIP, int COP, int ci_off, int c_in, int c_out, _Float16 *__restrict__ dw, int64_t s_co,
                                                                   int64_t s_ci, int64_t s_ky, int64_t s_kx, float *__restrict__ db) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int n_w = 9 * CIP * COP, e = blockIdx.x * 64 + lane;
  const bool is_w = e < n_w, live = e < n_w + COP;
  const float *src = is_w ? part + e : bias_part + (e - n_w);
  const size_t pitch = is_w ? (size_t)n_w : (size_t)COP;
  const int per = (n_part + 3) / 4, p0 = q * per, p1 = min(n_part, p0 + per);
  float s = 0.f;
  if (live) {
    int p = p0;
    for (; p + 8 <= p1; p += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(p + j) * pitch];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; p < p1; ++p) s += src[(size_t)p * pitch];
  }
  red[q][lane] = s;
  __syncthreads();
  if (q != 0 || !live) return;
  s = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
  if (is_w) {
    const int t9 = e / (CIP * COP), ci = (e / COP) % CIP, co = e % COP;
    if (ci_off + ci < c_in && co < c_out) dw[co * s_co + (ci_off + ci) * s_ci + (t9 / 3) * s_ky + (t9 % 3) * s_kx] = (_Float16)s;
  } else if (db && e - n_w < c_out) {
    db[e - n_w] = s;
  }
}

#define C2G_WGRAD_WGS 512
static int c2_pad32(int c) { return (c + 31) / 32 * 32; }
// input channels per pass: with 96 output channels 64 (216 accumulator registers per wave; 96 x 96 in one pass does not fit the
// register file - the compiler spills 487 of them), else all
static int c2g_pass_ci(int c_in, int c_out) { return (c2_pad32(c_in) / 32) * (c2_pad32(c_out) / 32) > 6 ? 64 : c2_pad32(c_in); }

// 0 if the kernel does not take the layer (channel counts multiples of 8 up to 96)
extern "C" size_t ts_conv3x3_wgrad_workspace_bytes(int32_t c_in, int32_t c_out) {
  if (c_in < 8 || c_in > 96 || (c_in & 7) || c_out < 8 || c_out > 96 || (c_out & 7)) return 0;
  return (size_t)C2G_WGRAD_WGS * ((size_t)9 * c2g_pass_ci(c_in, c_out) * c2_pad32(c_out) + c2_pad32(c_out)) * sizeof(float);
}

template <int NI, int NO, int TR>
static int conv3x3_wgrad_launch(const _Float16 *x, int x_ch, int x_off, int x_chunks, const _Float16 *gy, int y_ch, int T, int H, int W,
                                float *part, float *bias_part, int *n_part, hipStream_t stream) {
  constexpr size_t lds = ((size_t)(TR + 2) * 34 * (32 * NI + 8) + (size_t)TR * 32 * (32 * NO + 8)) * sizeof(_Float16);
  static_assert(lds <= 160 * 1024, "staged rows exceed the LDS of a CU");
  static bool attr_set = false;
  auto kern = conv3x3_wgrad_kernel<NI, NO, TR>;
  if (lds > 64 * 1024 && !attr_set) {
    TS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
    attr_set = true;
  }
  const int tiles_x = (int)ts_cdiv(W, 32), tiles_y = (int)ts_cdiv(H, TR);
  const int64_t n_tiles = (int64_t)T * tiles_x * tiles_y;
  TS_REQUIRE(n_tiles < (1LL << 31), TS_ERR_UNSUPPORTED, "ts_conv3x3_wgrad: stack too large");
  const int grid = (int)std::min<int64_t>(n_tiles, C2G_WGRAD_WGS);
  kern<<<grid, 256, lds, stream>>>(x, x_ch, x_off, x_chunks, gy, y_ch, H, W, tiles_x, tiles_y, (int)n_tiles, part, bias_part);
  *n_part = grid;
  return TS_OK;
}

// grad_weight [c_out][c_in][3][3] (IEEE half, element strides given) and grad_bias [c_out] (float, may be NULL) of a 3 x 3, stride 1,
// padding 1 layer from x [T, H, W, c_in] and grad_y [T, H, W, c_out], channels-last half; ws >= ts_conv3x3_wgrad_workspace_bytes()
extern "C" int ts_conv3x3_wgrad(const void *x, int32_t c_in, const void *grad_y, int32_t c_out, int32_t T, int32_t H, int32_t W,
                                void *grad_weight, int64_t s_co, int64_t s_ci, int64_t s_ky, int64_t s_kx, float *grad_bias, void *ws,
                                size_t ws_bytes, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(T > 0 && H > 0 && W > 0, TS_ERR_INVALID_ARGUMENT, "ts_conv3x3_wgrad: bad sizes");
  const size_t need = ts_conv3x3_wgrad_workspace_bytes(c_in, c_out);
  TS_REQUIRE(need != 0, TS_ERR_UNSUPPORTED, "ts_conv3x3_wgrad: channel counts multiples of 8 up to 96");
  TS_REQUIRE(x && grad_y && grad_weight && ws && ws_bytes >= need, TS_ERR_INVALID_ARGUMENT, "ts_conv3x3_wgrad: null pointer / workspace too small");
  TS_REQUIRE(((((uintptr_t)x) | ((uintptr_t)grad_y) | ((uintptr_t)ws)) & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_conv3x3_wgrad: pointers must be 16-byte aligned");
  const int pass_ci = c2g_pass_ci(c_in, c_out), COP = c2_pad32(c_out), NO = COP / 32;
  const _Float16 *xp = (const _Float16 *)x, *gp = (const _Float16 *)grad_y;
  for (int ci_off = 0; ci_off < c_in; ci_off += pass_ci) {       // (passes on one stream: the workspace is free again when the next starts)
    const int ci_n = std::min(pass_ci, c_in - ci_off), CIP = c2_pad32(ci_n), NI = CIP / 32;
    float *part = (float *)ws, *bias_part = part + (size_t)C2G_WGRAD_WGS * 9 * CIP * COP;
    int n_part = 0, rc = TS_ERR_UNSUPPORTED;
#define C2G_CASE(NI_, NO_, TR_) \
    if (NI == NI_ && NO == NO_)   \
      rc = conv3x3_wgrad_launch<NI_, NO_, TR_>(xp, c_in, ci_off, ci_n / 8, gp, c_out, T, H, W, part, bias_part, &n_part, stream);
    C2G_CASE(1, 1, 8) C2G_CASE(1, 2, 8) C2G_CASE(1, 3, 8) C2G_CASE(2, 1, 8) C2G_CASE(2, 2, 8) C2G_CASE(2, 3, 4) C2G_CASE(3, 1, 8) C2G_CASE(3, 2, 4)
#undef C2G_CASE
    if (rc != TS_OK) return rc;
    TS_CHECK_LAUNCH("ts_conv3x3_wgrad");
    const int n_elems = 9 * CIP * COP + COP;
    conv3x3_wgrad_reduce_kernel<<<(n_elems + 63) / 64, 256, 0, stream>>>(part, bias_part, n_part, CIP, COP, ci_off, c_in, c_out,
                                                                         (_Float16 *)grad_weight, s_co, s_ci, s_ky, s_kx,
                                                                         ci_off == 0 ? grad_bias : nullptr);
    TS_CHECK_LAUNCH("ts_conv3x3_wgrad (reduce)");
  }
  return TS_OK;
}
