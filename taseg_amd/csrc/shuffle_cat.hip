// UpBlock's entry (R/pcseg/model/segmentor/voxel/minkunet/unet2d.py:98-108): `upA = nn.PixelShuffle(2)(x)`, `Dropout2d`,
// `torch.cat((upA, skip), dim = 1)`, `Dropout2d` - on channels-last rows, in ONE pass over the result.
//   cat[t, Y, X, c]       = s[t, c]      * x[t, Y / 2, X / 2, 4 c + 2 (Y % 2) + (X % 2)]     c <  Cq = C / 4      (PixelShuffle)
//   cat[t, Y, X, Cq + c]  = s[t, Cq + c] * skip[t, Y, X, c]                                  c <  Cs
// s = the two Dropout2d masks folded into one per-(frame, channel) factor (NULL: none).  ATen runs this as a strided copy
// (PixelShuffle), two multiplications and a concatenation along the innermost dimension (3.1 ms per step for the four UpBlocks at the
// TIAF shape, plus the strided gradient slices the concatenation's backward leaves behind); it is 1.1 GB of traffic at full scale.
// A thread owns 16 bytes of an INPUT pixel row of x (VE channels = VE / 4 output channels of each of the four pixels it spreads
// to: four 4-byte stores, neighbouring threads neighbouring bytes), or 16 bytes of a skip row (one 16-byte store).  The gradient is
// the same walk backwards: grad_x gathers its four 4-byte pieces, grad_skip is a 16-byte copy of its slice.
#include <hip/hip_fp16.h>

#include "common.h"

template <typename T> struct ScVec;
template <> struct ScVec<__half> { typedef _Float16 e; typedef _Float16 v __attribute__((ext_vector_type(8))); typedef _Float16 q __attribute__((ext_vector_type(2))); static constexpr int VE = 8; };
template <> struct ScVec<float> { typedef float e; typedef float v __attribute__((ext_vector_type(4))); typedef float q __attribute__((ext_vector_type(1))); static constexpr int VE = 4; };

// BWD = false: x, skip -> cat;  BWD = true: cat (= grad_cat) -> x (= grad_x), skip (= grad_skip)
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void shuffle_cat_kernel(typename ScVec<T>::e *__restrict__ x, typename ScVec<T>::e *__restrict__ skip,
                                                          typename ScVec<T>::e *__restrict__ cat, const float *__restrict__ scale, int h,
                                                          int w, int C, int Cs, unsigned n_up, unsigned n_total) {
  typedef typename ScVec<T>::v vec;
  typedef typename ScVec<T>::q quad;
  typedef typename ScVec<T>::e elem;
  constexpr int VE = ScVec<T>::VE, QE = VE / 4;
  const unsigned item = blockIdx.x * 256u + threadIdx.x;
  if (item >= n_total) return;
  const int Cq = C >> 2, Cc = Cq + Cs;
  if (item < n_up) {
    const unsigned pieces = (unsigned)C / VE, j = item % pieces, pix = item / pieces;
    const unsigned X = pix % (unsigned)w, rest = pix / (unsigned)w, Y = rest % (unsigned)h, t = rest / (unsigned)h;
    const int co = (int)j * QE;                                        // first output channel of this piece
    elem *xp = x + (size_t)pix * C + (size_t)j * VE;
    elem *cp = cat + (((size_t)t * 2 * h + 2 * Y) * 2 * w + 2 * X) * Cc + co;
    float s[QE];
#pragma unroll
    for (int q = 0; q < QE; ++q) s[q] = scale ? scale[(size_t)t * Cc + co + q] : 1.f;
    if (!BWD) {
      const vec v = *(const vec *)xp;
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        quad o;
#pragma unroll
        for (int q = 0; q < QE; ++q) o[q] = scale ? (elem)((float)v[4 * q + sub] * s[q]) : v[4 * q + sub];
        *(quad *)(cp + ((size_t)(sub >> 1) * 2 * w + (sub & 1)) * Cc) = o;
      }
    } else {
      vec v;
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const quad o = *(const quad *)(cp + ((size_t)(sub >> 1) * 2 * w + (sub & 1)) * Cc);
#pragma unroll
        for (int q = 0; q < QE; ++q) v[4 * q + sub] = scale ? (elem)((float)o[q] * s[q]) : o[q];
      }
      *(vec *)xp = v;
    }
  } else {
    const unsigned it2 = item - n_up, pieces = (unsigned)Cs / VE, j = it2 % pieces, pix = it2 / pieces;   // pix = (t, Y, X) of the result
    elem *sp = skip + (size_t)pix * Cs + (size_t)j * VE;
    elem *cp = cat + (size_t)pix * Cc + Cq + (size_t)j * VE;
    vec v = BWD ? *(const vec *)cp : *(const vec *)sp;
    if (scale) {
      const unsigned t = pix / ((unsigned)(2 * h) * (unsigned)(2 * w));
      const float *sc = scale + (size_t)t * Cc + Cq + j * VE;
#pragma unroll
      for (int k = 0; k < VE; ++k) v[k] = (elem)((float)v[k] * sc[k]);
    }
    if (BWD) *(vec *)sp = v; else *(vec *)cp = v;
  }
}

template <typename T>
static int shuffle_cat_launch(const char *what, bool backward, void *x, void *skip, void *cat, const float *scale, int T_, int h, int w, int C,
                              int Cs, hipStream_t stream) {
  constexpr int VE = ScVec<T>::VE;
  typedef typename ScVec<T>::e elem;
  TS_REQUIRE(C % (4 * VE) == 0 && Cs % VE == 0 && Cs > 0, TS_ERR_UNSUPPORTED,
             "ts_shuffle_cat_rows: C / 4 and the skip channels must be multiples of the 16-byte piece");
  const int64_t n_up = (int64_t)T_ * h * w * (C / VE), n_skip = (int64_t)T_ * 4 * h * w * (Cs / VE);
  TS_REQUIRE(n_up + n_skip < (1LL << 32) - 256, TS_ERR_UNSUPPORTED, "ts_shuffle_cat_rows: stack too large for 32-bit item indices");
  const unsigned grid = (unsigned)ts_cdiv(n_up + n_skip, 256);
  if (backward)
    shuffle_cat_kernel<T, true><<<grid, 256, 0, stream>>>((elem *)x, (elem *)skip, (elem *)cat, scale, h, w, C, Cs, (unsigned)n_up,
                                                          (unsigned)(n_up + n_skip));
  else
    shuffle_cat_kernel<T, false><<<grid, 256, 0, stream>>>((elem *)x, (elem *)skip, (elem *)cat, scale, h, w, C, Cs, (unsigned)n_up,
                                                           (unsigned)(n_up + n_skip));
  TS_CHECK_LAUNCH(what);
  return TS_OK;
}

// cat [T, 2h, 2w, C / 4 + Cs] = concat(PixelShuffle(2)(x [T, h, w, C]), skip [T, 2h, 2w, Cs]) * scale [T, C / 4 + Cs] (float, may be
// NULL); channels-last rows, 16-byte aligned, half != 0: IEEE half, else float.
extern "C" int ts_shuffle_cat_rows_forward(const void *x, const void *skip, const float *scale, int32_t T, int32_t h, int32_t w, int32_t C,
                                           int32_t Cs, int32_t half, void *cat, ts_stream_t stream_) {
  TS_REQUIRE(T >= 0 && h > 0 && w > 0 && C > 0 && Cs > 0, TS_ERR_INVALID_ARGUMENT, "ts_shuffle_cat_rows_forward: bad sizes");
  if (T == 0) return TS_OK;
  TS_REQUIRE(x && skip && cat, TS_ERR_INVALID_ARGUMENT, "ts_shuffle_cat_rows_forward: null pointer");
  TS_REQUIRE(((((uintptr_t)x) | ((uintptr_t)skip) | ((uintptr_t)cat)) & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_shuffle_cat_rows_forward: pointers must be 16-byte aligned");
  return half ? shuffle_cat_launch<__half>("ts_shuffle_cat_rows_forward", false, (void *)x, (void *)skip, cat, scale, T, h, w, C, Cs, (hipStream_t)stream_)
              : shuffle_cat_launch<float>("ts_shuffle_cat_rows_forward", false, (void *)x, (void *)skip, cat, scale, T, h, w, C, Cs, (hipStream_t)stream_);
}

// the adjoint: grad_x [T, h, w, C] and grad_skip [T, 2h, 2w, Cs] from grad_cat (same scale)
extern "C" int ts_shuffle_cat_rows_backward(const void *grad_cat, const float *scale, int32_t T, int32_t h, int32_t w, int32_t C, int32_t Cs,
                                            int32_t half, void *grad_x, void *grad_skip, ts_stream_t stream_) {
  TS_REQUIRE(T >= 0 && h > 0 && w > 0 && C > 0 && Cs > 0, TS_ERR_INVALID_ARGUMENT, "ts_shuffle_cat_rows_backward: bad sizes");
  if (T == 0) return TS_OK;
  TS_REQUIRE(grad_cat && grad_x && grad_skip, TS_ERR_INVALID_ARGUMENT, "ts_shuffle_cat_rows_backward: null pointer");
  TS_REQUIRE(((((uintptr_t)grad_cat) | ((uintptr_t)grad_x) | ((uintptr_t)grad_skip)) & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_shuffle_cat_rows_backward: pointers must be 16-byte aligned");
  return half ? shuffle_cat_launch<__half>("ts_shuffle_cat_rows_backward", true, grad_x, grad_skip, (void *)grad_cat, scale, T, h, w, C, Cs, (hipStream_t)stream_)
              : shuffle_cat_launch<float>("ts_shuffle_cat_rows_backward", true, grad_x, grad_skip, (void *)grad_cat, scale, T, h, w, C, Cs, (hipStream_t)stream_);
}
