// BatchNorm over [N, C] voxel features, training mode.
//
// The reference runs nn.BatchNorm1d / nn.SyncBatchNorm on the feature matrix of every sparse conv
// (R/pcseg/model/segmentor/voxel/minkunet/minkunet.py:23-29, TS/torchsparse/nn/utils/apply.py:10-16), followed by
// the residual add and the ReLU as separate passes.  This file holds
//   * the elementwise halves: BN apply (+ residual) (+ ReLU) forward / backward (ts_bn_act_forward / _backward),
//     statistics finalisation (ts_bn_finalize);
//   * the sliced reductions + finish kernels and the one-call-per-direction training entry points
//     (ts_bn_act_train_forward / _backward, fp32 and half storage);
//   * the SyncBatchNorm reduction halves whose double sums the host all-reduces (ts_bn_sync_*).
// Accumulation: float per lane over one slice (<= ~350 rows), double across slices.
#include <stdlib.h>

#include "common.h"

// mean / invstd from the double sums + running-statistics update (nn.BatchNorm1d semantics: biased variance
// for normalisation, unbiased for running_var, momentum update), one tiny launch instead of ~8 tensor ops.
__global__ void bn_finalize_kernel(const double *__restrict__ sums, const double *__restrict__ total_dev,
                                   double total_host, int c, float eps, float momentum,
                                   float *__restrict__ running_mean, float *__restrict__ running_var,
                                   float *__restrict__ mean, float *__restrict__ invstd) {
  int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const double total = total_dev ? *total_dev : total_host;
  const double m = sums[ch] / total;
  double var = sums[c + ch] / total - m * m;
  if (var < 0.0) var = 0.0;
  mean[ch] = (float)m;
  invstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)m;
  if (running_var) {
    const double unbiased = total > 1.0 ? var * total / (total - 1.0) : var;
    running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unbiased;
  }
}

extern "C" int ts_bn_finalize(const double *sums, const double *total_dev, double total_host, int32_t c, float eps,
                              float momentum, float *running_mean, float *running_var, float *mean, float *invstd,
                              ts_stream_t stream) {
  TS_REQUIRE(sums && mean && invstd && c > 0, TS_ERR_INVALID_ARGUMENT, "ts_bn_finalize: bad arguments");
  TS_REQUIRE(total_dev || total_host > 0.0, TS_ERR_INVALID_ARGUMENT, "ts_bn_finalize: empty batch");
  bn_finalize_kernel<<<(unsigned)ts_cdiv(c, 256), 256, 0, (hipStream_t)stream>>>(
      sums, total_dev, total_host, c, eps, momentum, running_mean, running_var, mean, invstd);
  TS_CHECK_LAUNCH("ts_bn_finalize");
  return TS_OK;
}

// ------------------------------------------------------------------------------------------------------
// Fused BatchNorm-apply (+ residual add) (+ ReLU) and its backward: one pass each instead of
// batch_norm_elemt -> add -> relu (forward) and relu-backward -> reduce -> batch_norm_backward_elemt (backward).
//   forward   out = act((x - mean) * invstd * w + b [+ res])
//   backward  g = gout * (out > 0)          (act = relu; the forward stores a 4-bit sign mask per float4)
//             sums[0] = sum g, sums[1] = sum g (x - mean)                            (ts_bn_act_backward_reduce)
//             gx = (g - sums[0]/N - (x - mean) invstd^2 sums[1]/N) invstd w ,  gres = g   (ts_bn_act_backward)
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float4 *__restrict__ X, const float4 *__restrict__ RES,
                                                         const float *__restrict__ mean,
                                                         const float *__restrict__ invstd,
                                                         const float *__restrict__ w, const float *__restrict__ b,
                                                         int64_t total4, int cq, int relu, float4 *__restrict__ OUT,
                                                         unsigned char *__restrict__ MASK) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; e < total4; e += step) {
    const int q = (int)(e % cq) * 4;
    const float4 x = X[e];
    const float4 m = *(const float4 *)(mean + q), s = *(const float4 *)(invstd + q);
    const float4 ww = *(const float4 *)(w + q), bb = *(const float4 *)(b + q);
    float4 y;
    y.x = (x.x - m.x) * s.x * ww.x + bb.x;
    y.y = (x.y - m.y) * s.y * ww.y + bb.y;
    y.z = (x.z - m.z) * s.z * ww.z + bb.z;
    y.w = (x.w - m.w) * s.w * ww.w + bb.w;
    if (RES) {
      const float4 r = RES[e];
      y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
    }
    if (relu) {
      // 4-bit sign mask per float4 (1 byte per 16 bytes of activations): all the backward needs of `out`
      if (MASK) MASK[e] = (unsigned char)((y.x > 0.f) | ((y.y > 0.f) << 1) | ((y.z > 0.f) << 2) | ((y.w > 0.f) << 3));
      y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
    }
    OUT[e] = y;
  }
}

__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const float4 *__restrict__ GOUT,
                                                         const unsigned char *__restrict__ MASK,
                                                         const float4 *__restrict__ X, const float *__restrict__ mean,
                                                         const float *__restrict__ invstd,
                                                         const float *__restrict__ w, const double *__restrict__ sums,
                                                         const double *__restrict__ total_dev, double total_host,
                                                         int64_t total4, int c, float4 *__restrict__ GX,
                                                         float4 *__restrict__ GRES) {
  const int cq = c >> 2;
  const double total = total_dev ? *total_dev : total_host;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; e < total4; e += step) {
    const int q = (int)(e % cq) * 4;
    float4 g = GOUT[e];
    if (MASK) {
      const unsigned mk = MASK[e];
      g.x = (mk & 1) ? g.x : 0.f; g.y = (mk & 2) ? g.y : 0.f;
      g.z = (mk & 4) ? g.z : 0.f; g.w = (mk & 8) ? g.w : 0.f;
    }
    if (GRES) GRES[e] = g;
    const float4 x = X[e];
    const float4 m = *(const float4 *)(mean + q), s = *(const float4 *)(invstd + q), ww = *(const float4 *)(w + q);
    float gx[4];
    const float gv[4] = {g.x, g.y, g.z, g.w}, xv[4] = {x.x, x.y, x.z, x.w}, mv[4] = {m.x, m.y, m.z, m.w},
                sv[4] = {s.x, s.y, s.z, s.w}, wv[4] = {ww.x, ww.y, ww.z, ww.w};
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const float mdy = (float)(sums[q + l] / total);
      const float k2 = (float)(sums[c + q + l] / total) * sv[l] * sv[l];
      gx[l] = (gv[l] - mdy - (xv[l] - mv[l]) * k2) * sv[l] * wv[l];
    }
    GX[e] = make_float4(gx[0], gx[1], gx[2], gx[3]);
  }
}

static bool bn_aligned(const void *p) { return (((uintptr_t)p) & 15) == 0; }

extern "C" int ts_bn_act_forward(const float *x, const float *residual, const float *mean, const float *invstd,
                                 const float *weight, const float *bias, int64_t n, int32_t c, int32_t relu, float *out,
                                 uint8_t *mask, ts_stream_t stream) {
  TS_REQUIRE(n >= 0 && c > 0 && (c & 3) == 0, TS_ERR_UNSUPPORTED, "ts_bn_act_forward: C must be a multiple of 4");
  if (n == 0) return TS_OK;
  TS_REQUIRE(x && mean && invstd && weight && bias && out, TS_ERR_INVALID_ARGUMENT, "ts_bn_act_forward: null pointer");
  TS_REQUIRE(bn_aligned(x) && bn_aligned(out) && bn_aligned(mean) && bn_aligned(invstd) && bn_aligned(weight) &&
                 bn_aligned(bias) && (!residual || bn_aligned(residual)),
             TS_ERR_INVALID_ARGUMENT, "ts_bn_act_forward: pointers must be 16-byte aligned");
  const int64_t total4 = n * (c / 4);
  const unsigned grid = (unsigned)std::min<int64_t>(ts_cdiv(total4, 256), 1 << 16);
  bn_act_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const float4 *)x, (const float4 *)residual, mean, invstd,
                                                           weight, bias, total4, c / 4, relu, (float4 *)out, mask);
  TS_CHECK_LAUNCH("ts_bn_act_forward");
  return TS_OK;
}

extern "C" int ts_bn_act_backward(const float *grad_out, const uint8_t *mask, const float *x, const float *mean,
                                  const float *invstd, const float *weight, const double *sums, const double *total_dev,
                                  double total_host, int64_t n, int32_t c, float *grad_x, float *grad_residual,
                                  ts_stream_t stream) {
  TS_REQUIRE(n >= 0 && c > 0 && (c & 3) == 0, TS_ERR_UNSUPPORTED, "ts_bn_act_backward: C must be a multiple of 4");
  if (n == 0) return TS_OK;
  TS_REQUIRE(grad_out && x && mean && invstd && weight && sums && grad_x, TS_ERR_INVALID_ARGUMENT,
             "ts_bn_act_backward: null pointer");
  TS_REQUIRE(total_dev || total_host > 0.0, TS_ERR_INVALID_ARGUMENT, "ts_bn_act_backward: empty batch");
  TS_REQUIRE(bn_aligned(grad_out) && bn_aligned(x) && bn_aligned(grad_x) &&
                 (!grad_residual || bn_aligned(grad_residual)) && bn_aligned(mean) && bn_aligned(invstd) &&
                 bn_aligned(weight),
             TS_ERR_INVALID_ARGUMENT, "ts_bn_act_backward: pointers must be 16-byte aligned");
  const int64_t total4 = n * (c / 4);
  const unsigned grid = (unsigned)std::min<int64_t>(ts_cdiv(total4, 256), 1 << 16);
  bn_act_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const float4 *)grad_out, mask,
                                                           (const float4 *)x, mean, invstd, weight, sums, total_dev,
                                                           total_host, total4, c, (float4 *)grad_x,
                                                           (float4 *)grad_residual);
  TS_CHECK_LAUNCH("ts_bn_act_backward");
  return TS_OK;
}

typedef _Float16 bn_h8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------------
// Single-process training BatchNorm (+ residual) (+ ReLU), one host call per direction.
//
// The reductions above end in 2 C double atomics per workgroup into a zeroed buffer and are sized at 512 rows per
// workgroup - 59 workgroups for a 30k-row matrix on a 256-CU device.  Here every launch is cut into <= 512
// slices of at least 32 rows, each slice writes its float partial sums to scratch (no memset, no atomics, and the
// result does not depend on the order workgroups finish in), and a small second kernel adds the partials in
// double and finishes the per-channel math: mean / invstd / running statistics in the forward pass; grad_weight,
// grad_bias and the two coefficients of the input gradient in the backward pass (which the elementwise kernel
// then reads as floats instead of dividing doubles per element).
//   forward   ts_bn_act_train_forward  = bn_partial<0> -> bn_fwd_finish -> bn_act_fwd
//   backward  ts_bn_act_train_backward = bn_partial<1|2> -> bn_bwd_finish -> bn_act_bwd_coef
#define BN_MAX_SLICES 512

// MODE 0: (x, x^2)   1: (dy, dy (x - mean))   2: same with the ReLU mask applied to dy
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float *__restrict__ X, const float *__restrict__ DY,
                                                         const unsigned char *__restrict__ MASK,
                                                         const float *__restrict__ mean, int64_t n, int c,
                                                         int rows_per_wg, float *__restrict__ part) {
  __shared__ float red[2][256 * 4];
  const int cq = c >> 2, rpp = 256 / cq;
  const int tid = threadIdx.x, ty = tid / cq, tx = tid - ty * cq;
  const bool active = ty < rpp;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, mu = s0;
  if (MODE != 0 && active) mu = *(const float4 *)(mean + 4 * tx);
  const int64_t r_beg = (int64_t)blockIdx.x * rows_per_wg, r_end = min(n, r_beg + rows_per_wg);
  if (active) {
#pragma unroll 4
    for (int64_t r = r_beg + ty; r < r_end; r += rpp) {
      const float4 x = *(const float4 *)(X + r * c + 4 * tx);
      if (MODE == 0) {
        s0.x += x.x; s0.y += x.y; s0.z += x.z; s0.w += x.w;
        s1.x += x.x * x.x; s1.y += x.y * x.y; s1.z += x.z * x.z; s1.w += x.w * x.w;
      } else {
        float4 d = *(const float4 *)(DY + r * c + 4 * tx);
        if (MODE == 2) {
          const unsigned mk = MASK[r * cq + tx];
          d.x = (mk & 1) ? d.x : 0.f; d.y = (mk & 2) ? d.y : 0.f;
          d.z = (mk & 4) ? d.z : 0.f; d.w = (mk & 8) ? d.w : 0.f;
        }
        s0.x += d.x; s0.y += d.y; s0.z += d.z; s0.w += d.w;
        s1.x += d.x * (x.x - mu.x); s1.y += d.y * (x.y - mu.y);
        s1.z += d.z * (x.z - mu.z); s1.w += d.w * (x.w - mu.w);
      }
    }
  }
  *(float4 *)&red[0][tid * 4] = s0;
  *(float4 *)&red[1][tid * 4] = s1;
  __syncthreads();
  float *out = part + (int64_t)blockIdx.x * 2 * c;
  for (int ch = tid; ch < c; ch += 256) {
    const int q = ch >> 2, l = ch & 3;
    float a = 0.f, b = 0.f;
    for (int y = 0; y < rpp; ++y) {
      a += red[0][(y * cq + q) * 4 + l];
      b += red[1][(y * cq + q) * 4 + l];
    }
    out[ch] = a;
    out[c + ch] = b;
  }
}

// 8 channels per workgroup, 32 lanes per channel over the slices (<= 16 partials per lane, all loads issued
// before the first add); double accumulation in a fixed order.
#define BN_FIN_CH 8
#define BN_FIN_LANES 32
__device__ __forceinline__ void bn_sum_slices(const float *__restrict__ part, int slices, int c, int ch, int sl,
                                              double (&red)[2][BN_FIN_CH][BN_FIN_LANES + 1], double &s0, double &s1) {
  constexpr int PER = BN_MAX_SLICES / BN_FIN_LANES;
  // unconditional loads from clamped addresses, masked afterwards: a predicated load compiles to a branch plus
  // a full wait per element (32 serialised round trips, 10 us for this tiny kernel)
  float va[PER], vb[PER];
  const int chc = min(ch, c - 1);
#pragma unroll
  for (int t = 0; t < PER; ++t) {
    const int g = min(sl + t * BN_FIN_LANES, slices - 1);
    va[t] = part[(int64_t)g * 2 * c + chc];
    vb[t] = part[(int64_t)g * 2 * c + c + chc];
  }
  double a = 0.0, b = 0.0;
#pragma unroll
  for (int t = 0; t < PER; ++t) {
    const bool ok = sl + t * BN_FIN_LANES < slices;
    a += ok ? (double)va[t] : 0.0;
    b += ok ? (double)vb[t] : 0.0;
  }
  const int cl = threadIdx.x % BN_FIN_CH;
  red[0][cl][sl] = a;
  red[1][cl][sl] = b;
  __syncthreads();
  s0 = s1 = 0.0;
  if (sl == 0) {
    for (int i = 0; i < BN_FIN_LANES; ++i) {
      s0 += red[0][cl][i];
      s1 += red[1][cl][i];
    }
  }
}

__global__ __launch_bounds__(256) void bn_fwd_finish_kernel(const float *__restrict__ part, int slices, double total,
                                                            int c, float eps, float momentum,
                                                            float *__restrict__ running_mean,
                                                            float *__restrict__ running_var, float *__restrict__ mean,
                                                            float *__restrict__ invstd,
                                                            int64_t *__restrict__ num_batches_tracked) {
  __shared__ double red[2][BN_FIN_CH][BN_FIN_LANES + 1];
  const int ch = blockIdx.x * BN_FIN_CH + (threadIdx.x % BN_FIN_CH), sl = threadIdx.x / BN_FIN_CH;
  if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
  double s0, s1;
  bn_sum_slices(part, slices, c, ch, sl, red, s0, s1);
  if (sl != 0 || ch >= c) return;
  const double m = s0 / total;
  double var = s1 / total - m * m;
  if (var < 0.0) var = 0.0;
  mean[ch] = (float)m;
  invstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)m;
  if (running_var) {
    const double unbiased = total > 1.0 ? var * total / (total - 1.0) : var;
    running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unbiased;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finish_kernel(const float *__restrict__ part, int slices, double total,
                                                            int c, const float *__restrict__ invstd,
                                                            float *__restrict__ coef, float *__restrict__ grad_weight,
                                                            float *__restrict__ grad_bias) {
  __shared__ double red[2][BN_FIN_CH][BN_FIN_LANES + 1];
  const int ch = blockIdx.x * BN_FIN_CH + (threadIdx.x % BN_FIN_CH), sl = threadIdx.x / BN_FIN_CH;
  double s0, s1;
  bn_sum_slices(part, slices, c, ch, sl, red, s0, s1);
  if (sl != 0 || ch >= c) return;
  const float is = invstd[ch];
  coef[ch] = (float)(s0 / total);                  // mean of dy
  coef[c + ch] = (float)(s1 / total) * is * is;    // mean of dy (x - mean), times invstd^2
  if (grad_bias) grad_bias[ch] = (float)s0;
  if (grad_weight) grad_weight[ch] = (float)s1 * is;
}

__global__ __launch_bounds__(256) void bn_act_bwd_coef_kernel(const float4 *__restrict__ GOUT,
                                                              const unsigned char *__restrict__ MASK,
                                                              const float4 *__restrict__ X,
                                                              const float *__restrict__ mean,
                                                              const float *__restrict__ invstd,
                                                              const float *__restrict__ w,
                                                              const float *__restrict__ coef, int64_t total4, int c,
                                                              float4 *__restrict__ GX, float4 *__restrict__ GRES) {
  const int cq = c >> 2;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; e < total4; e += step) {
    const int q = (int)(e % cq) * 4;
    float4 g = GOUT[e];
    if (MASK) {
      const unsigned mk = MASK[e];
      g.x = (mk & 1) ? g.x : 0.f; g.y = (mk & 2) ? g.y : 0.f;
      g.z = (mk & 4) ? g.z : 0.f; g.w = (mk & 8) ? g.w : 0.f;
    }
    if (GRES) GRES[e] = g;
    const float4 x = X[e];
    const float4 m = *(const float4 *)(mean + q), s = *(const float4 *)(invstd + q), ww = *(const float4 *)(w + q);
    const float4 c0 = *(const float4 *)(coef + q), c1 = *(const float4 *)(coef + c + q);
    float4 gx;
    gx.x = (g.x - c0.x - (x.x - m.x) * c1.x) * s.x * ww.x;
    gx.y = (g.y - c0.y - (x.y - m.y) * c1.y) * s.y * ww.y;
    gx.z = (g.z - c0.z - (x.z - m.z) * c1.z) * s.z * ww.z;
    gx.w = (g.w - c0.w - (x.w - m.w) * c1.w) * s.w * ww.w;
    GX[e] = gx;
  }
}

static inline int bn_rows_per_slice(int64_t n, int c) {
  const int rpp = 256 / (c >> 2);                       // rows one pass of a workgroup covers
  int64_t rows = std::max<int64_t>(ts_cdiv(n, BN_MAX_SLICES), 32);
  rows = ts_cdiv(rows, rpp) * rpp;                      // whole passes
  return (int)rows;
}

// TS_OPT_DEBUG_BN_ABLATE (diagnostic, fp32 path).  Bits 0 / 1 (WRONG results, timing only): skip the forward statistics pass
// (bn_partial<0>) / the forward elementwise pass of the blocks without a residual.  Garbage activations also change what the
// chip draws, so the CLEAN measurement is bits 2 / 3: run the same pass TWICE (same results) - the step's slow-down is what one such
// pass costs in the step, launch gap included: the most "statistics in the epilogue of the pass that produces y" / "normalise where
// the consumer gathers the row" could save.
static int bn_debug_ablate() { return (int)ts_get_option(TS_OPT_DEBUG_BN_ABLATE); }

extern "C" size_t ts_bn_train_workspace_bytes(int32_t c) {
  // float partials [BN_MAX_SLICES][2][C] + float coefficients [2][C]
  return ((size_t)BN_MAX_SLICES * 2 * c + 2 * (size_t)c) * sizeof(float);
}

extern "C" int ts_bn_act_train_forward(const float *x, const float *residual, const float *weight, const float *bias,
                                       float *running_mean, float *running_var, int64_t *num_batches_tracked,
                                       int64_t n, int32_t c, float eps, float momentum, int32_t relu, float *mean,
                                       float *invstd, float *out, uint8_t *mask, void *ws, size_t ws_bytes,
                                       ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n > 0 && c > 0 && (c & 3) == 0 && c <= 1024, TS_ERR_UNSUPPORTED,
             "ts_bn_act_train_forward: need N > 0 and C a multiple of 4, <= 1024");
  TS_REQUIRE(x && weight && bias && mean && invstd && out && ws, TS_ERR_INVALID_ARGUMENT,
             "ts_bn_act_train_forward: null pointer");
  TS_REQUIRE(ws_bytes >= ts_bn_train_workspace_bytes(c), TS_ERR_INVALID_ARGUMENT,
             "ts_bn_act_train_forward: workspace too small");
  TS_REQUIRE(bn_aligned(x) && bn_aligned(out) && bn_aligned(mean) && bn_aligned(invstd) && bn_aligned(weight) &&
                 bn_aligned(bias) && (!residual || bn_aligned(residual)) && bn_aligned(ws),
             TS_ERR_INVALID_ARGUMENT, "ts_bn_act_train_forward: pointers must be 16-byte aligned");
  float *part = (float *)ws;
  const int rows = bn_rows_per_slice(n, c);
  const int slices = (int)ts_cdiv(n, rows);
  const int ablate = bn_debug_ablate();       // diagnostic (wrong results, timing only): upper bounds of two fusions, see below
  if (!(ablate & 1)) bn_partial_kernel<0><<<slices, 256, 0, stream>>>(x, nullptr, nullptr, nullptr, n, c, rows, part);
  if (ablate & 4) bn_partial_kernel<0><<<slices, 256, 0, stream>>>(x, nullptr, nullptr, nullptr, n, c, rows, part);   // the pass once more
  bn_fwd_finish_kernel<<<(unsigned)ts_cdiv(c, BN_FIN_CH), 256, 0, stream>>>(part, slices, (double)n, c, eps, momentum,
                                                                     running_mean, running_var, mean, invstd,
                                                                     num_batches_tracked);
  const int64_t total4 = n * (c / 4);
  const unsigned grid = (unsigned)std::min<int64_t>(ts_cdiv(total4, 256), 1 << 16);
  if (!((ablate & 2) && !residual && relu))
    bn_act_fwd_kernel<<<grid, 256, 0, stream>>>((const float4 *)x, (const float4 *)residual, mean, invstd, weight, bias,
                                                total4, c / 4, relu, (float4 *)out, mask);
  if ((ablate & 8) && !residual && relu)          // the same pass once more (same results)
    bn_act_fwd_kernel<<<grid, 256, 0, stream>>>((const float4 *)x, (const float4 *)residual, mean, invstd, weight, bias,
                                                total4, c / 4, relu, (float4 *)out, mask);
  TS_CHECK_LAUNCH("ts_bn_act_train_forward");
  return TS_OK;
}

extern "C" int ts_bn_act_train_backward(const float *grad_out, const uint8_t *mask, const float *x, const float *mean,
                                        const float *invstd, const float *weight, int64_t n, int32_t c, float *grad_x,
                                        float *grad_residual, float *grad_weight, float *grad_bias, void *ws,
                                        size_t ws_bytes, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n > 0 && c > 0 && (c & 3) == 0 && c <= 1024, TS_ERR_UNSUPPORTED,
             "ts_bn_act_train_backward: need N > 0 and C a multiple of 4, <= 1024");
  TS_REQUIRE(grad_out && x && mean && invstd && weight && grad_x && ws, TS_ERR_INVALID_ARGUMENT,
             "ts_bn_act_train_backward: null pointer");
  TS_REQUIRE(ws_bytes >= ts_bn_train_workspace_bytes(c), TS_ERR_INVALID_ARGUMENT,
             "ts_bn_act_train_backward: workspace too small");
  TS_REQUIRE(bn_aligned(grad_out) && bn_aligned(x) && bn_aligned(grad_x) && bn_aligned(mean) && bn_aligned(invstd) &&
                 bn_aligned(weight) && (!grad_residual || bn_aligned(grad_residual)) && bn_aligned(ws),
             TS_ERR_INVALID_ARGUMENT, "ts_bn_act_train_backward: pointers must be 16-byte aligned");
  float *part = (float *)ws;
  float *coef = part + (size_t)BN_MAX_SLICES * 2 * c;
  const int rows = bn_rows_per_slice(n, c);
  const int slices = (int)ts_cdiv(n, rows);
  if (mask)
    bn_partial_kernel<2><<<slices, 256, 0, stream>>>(x, grad_out, mask, mean, n, c, rows, part);
  else
    bn_partial_kernel<1><<<slices, 256, 0, stream>>>(x, grad_out, nullptr, mean, n, c, rows, part);
  bn_bwd_finish_kernel<<<(unsigned)ts_cdiv(c, BN_FIN_CH), 256, 0, stream>>>(part, slices, (double)n, c, invstd, coef,
                                                                     grad_weight, grad_bias);
  const int64_t total4 = n * (c / 4);
  const unsigned grid = (unsigned)std::min<int64_t>(ts_cdiv(total4, 256), 1 << 16);
  bn_act_bwd_coef_kernel<<<grid, 256, 0, stream>>>((const float4 *)grad_out, mask, (const float4 *)x, mean, invstd,
                                                   weight, coef, total4, c, (float4 *)grad_x, (float4 *)grad_residual);
  TS_CHECK_LAUNCH("ts_bn_act_train_backward");
  return TS_OK;
}

// ------------------------------------------------------------------------------------------------------
// SyncBatchNorm halves: the same sliced reductions, finished into DOUBLE sums that the host all-reduces
// (one collective of [2C + 1] doubles in the forward pass, [2C] in the backward pass - torch's SyncBatchNorm
// all-gathers per-rank mean / invstd / count instead).  The elementwise halves are ts_bn_finalize +
// ts_bn_act_forward and ts_bn_act_backward above.
__global__ __launch_bounds__(256) void bn_sums_finish_kernel(const float *__restrict__ part, int slices, int c,
                                                             double count, const float *__restrict__ invstd,
                                                             double *__restrict__ sums,
                                                             float *__restrict__ grad_weight,
                                                             float *__restrict__ grad_bias) {
  __shared__ double red[2][BN_FIN_CH][BN_FIN_LANES + 1];
  const int ch = blockIdx.x * BN_FIN_CH + (threadIdx.x % BN_FIN_CH), sl = threadIdx.x / BN_FIN_CH;
  double s0, s1;
  bn_sum_slices(part, slices, c, ch, sl, red, s0, s1);
  if (count >= 0.0 && blockIdx.x == 0 && threadIdx.x == 0) sums[2 * c] = count;   // forward pack: [sum, sumsq, n]
  if (sl != 0 || ch >= c) return;
  sums[ch] = s0;
  sums[c + ch] = s1;
  if (grad_bias) grad_bias[ch] = (float)s0;                       // local (per-rank) parameter gradients
  if (grad_weight) grad_weight[ch] = (float)s1 * invstd[ch];
}

extern "C" int ts_bn_sync_stats(const float *x, int64_t n, int32_t c, double *pack, void *ws, size_t ws_bytes,
                                ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n > 0 && c > 0 && (c & 3) == 0 && c <= 1024, TS_ERR_UNSUPPORTED,
             "ts_bn_sync_stats: need N > 0 and C a multiple of 4, <= 1024");
  TS_REQUIRE(x && pack && ws && bn_aligned(x) && bn_aligned(ws), TS_ERR_INVALID_ARGUMENT, "ts_bn_sync_stats: bad pointer");
  TS_REQUIRE(ws_bytes >= ts_bn_train_workspace_bytes(c), TS_ERR_INVALID_ARGUMENT, "ts_bn_sync_stats: workspace too small");
  float *part = (float *)ws;
  const int rows = bn_rows_per_slice(n, c);
  const int slices = (int)ts_cdiv(n, rows);
  bn_partial_kernel<0><<<slices, 256, 0, stream>>>(x, nullptr, nullptr, nullptr, n, c, rows, part);
  bn_sums_finish_kernel<<<(unsigned)ts_cdiv(c, BN_FIN_CH), 256, 0, stream>>>(part, slices, c, (double)n, nullptr, pack,
                                                                           nullptr, nullptr);
  TS_CHECK_LAUNCH("ts_bn_sync_stats");
  return TS_OK;
}

extern "C" int ts_bn_sync_backward_reduce(const float *grad_out, const uint8_t *mask, const float *x, const float *mean,
                                          const float *invstd, int64_t n, int32_t c, double *sums, float *grad_weight,
                                          float *grad_bias, void *ws, size_t ws_bytes, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n > 0 && c > 0 && (c & 3) == 0 && c <= 1024, TS_ERR_UNSUPPORTED,
             "ts_bn_sync_backward_reduce: need N > 0 and C a multiple of 4, <= 1024");
  TS_REQUIRE(grad_out && x && mean && invstd && sums && ws, TS_ERR_INVALID_ARGUMENT,
             "ts_bn_sync_backward_reduce: null pointer");
  TS_REQUIRE(bn_aligned(grad_out) && bn_aligned(x) && bn_aligned(mean) && bn_aligned(ws), TS_ERR_INVALID_ARGUMENT,
             "ts_bn_sync_backward_reduce: pointers must be 16-byte aligned");
  TS_REQUIRE(ws_bytes >= ts_bn_train_workspace_bytes(c), TS_ERR_INVALID_ARGUMENT,
             "ts_bn_sync_backward_reduce: workspace too small");
  float *part = (float *)ws;
  const int rows = bn_rows_per_slice(n, c);
  const int slices = (int)ts_cdiv(n, rows);
  if (mask)
    bn_partial_kernel<2><<<slices, 256, 0, stream>>>(x, grad_out, mask, mean, n, c, rows, part);
  else
    bn_partial_kernel<1><<<slices, 256, 0, stream>>>(x, grad_out, nullptr, mean, n, c, rows, part);
  bn_sums_finish_kernel<<<(unsigned)ts_cdiv(c, BN_FIN_CH), 256, 0, stream>>>(part, slices, c, -1.0, invstd, sums,
                                                                           grad_weight, grad_bias);
  TS_CHECK_LAUNCH("ts_bn_sync_backward_reduce");
  return TS_OK;
}

// ------------------------------------------------------------------------------------------------------
// Half-storage variants of the single-process training BatchNorm (+ residual) (+ ReLU): activations, residual and
// gradients are IEEE half in HBM, every reduction / normalisation runs in fp32 (statistics, affine parameters and
// parameter gradients stay fp32, as under torch.autocast).  One thread handles 8 channels (one 16-byte chunk);
// the ReLU mask keeps 1 bit per element (one byte per chunk).

template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_h_kernel(const _Float16 *__restrict__ X,
                                                           const _Float16 *__restrict__ DY,
                                                           const unsigned char *__restrict__ MASK,
                                                           const float *__restrict__ mean, int64_t n, int c,
                                                           int rows_per_wg, float *__restrict__ part) {
  __shared__ float red[2][256 * 8];
  const int cq = c >> 3, rpp = 256 / cq;
  const int tid = threadIdx.x, ty = tid / cq, tx = tid - ty * cq;
  const bool active = ty < rpp;
  float s0[8], s1[8], mu[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s0[i] = s1[i] = mu[i] = 0.f;
  if (MODE != 0 && active) {
#pragma unroll
    for (int i = 0; i < 8; ++i) mu[i] = mean[8 * tx + i];
  }
  const int64_t r_beg = (int64_t)blockIdx.x * rows_per_wg, r_end = min(n, r_beg + rows_per_wg);
  if (active) {
#pragma unroll 2
    for (int64_t r = r_beg + ty; r < r_end; r += rpp) {
      const bn_h8 x = *(const bn_h8 *)(X + r * c + 8 * tx);
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float v = (float)x[i];
          s0[i] += v;
          s1[i] += v * v;
        }
      } else {
        const bn_h8 d = *(const bn_h8 *)(DY + r * c + 8 * tx);
        const unsigned mk = MODE == 2 ? MASK[r * cq + tx] : 0xFFu;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float dv = ((mk >> i) & 1) ? (float)d[i] : 0.f;
          s0[i] += dv;
          s1[i] += dv * ((float)x[i] - mu[i]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    red[0][tid * 8 + i] = s0[i];
    red[1][tid * 8 + i] = s1[i];
  }
  __syncthreads();
  float *out = part + (int64_t)blockIdx.x * 2 * c;
  for (int ch = tid; ch < c; ch += 256) {
    const int q = ch >> 3, l = ch & 7;
    float a = 0.f, b = 0.f;
    for (int y = 0; y < rpp; ++y) {
      a += red[0][(y * cq + q) * 8 + l];
      b += red[1][(y * cq + q) * 8 + l];
    }
    out[ch] = a;
    out[c + ch] = b;
  }
}

__global__ __launch_bounds__(256) void bn_act_fwd_h_kernel(const bn_h8 *__restrict__ X, const bn_h8 *__restrict__ RES,
                                                           const float *__restrict__ mean,
                                                           const float *__restrict__ invstd,
                                                           const float *__restrict__ w, const float *__restrict__ b,
                                                           int64_t total8, int cq, int relu, bn_h8 *__restrict__ OUT,
                                                           unsigned char *__restrict__ MASK) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; e < total8; e += step) {
    const int q = (int)(e % cq) * 8;
    const bn_h8 x = X[e];
    bn_h8 r;
    if (RES) r = RES[e];
    bn_h8 y;
    unsigned mk = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = ((float)x[i] - mean[q + i]) * invstd[q + i] * w[q + i] + b[q + i];
      if (RES) v += (float)r[i];
      if (relu) {
        mk |= (v > 0.f ? 1u : 0u) << i;
        v = fmaxf(v, 0.f);
      }
      y[i] = (_Float16)v;
    }
    if (relu && MASK) MASK[e] = (unsigned char)mk;
    OUT[e] = y;
  }
}

__global__ __launch_bounds__(256) void bn_act_bwd_coef_h_kernel(const bn_h8 *__restrict__ GOUT,
                                                                const unsigned char *__restrict__ MASK,
                                                                const bn_h8 *__restrict__ X,
                                                                const float *__restrict__ mean,
                                                                const float *__restrict__ invstd,
                                                                const float *__restrict__ w,
                                                                const float *__restrict__ coef, int64_t total8, int c,
                                                                bn_h8 *__restrict__ GX, bn_h8 *__restrict__ GRES) {
  const int cq = c >> 3;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; e < total8; e += step) {
    const int q = (int)(e % cq) * 8;
    const bn_h8 gin = GOUT[e], x = X[e];
    const unsigned mk = MASK ? MASK[e] : 0xFFu;
    bn_h8 g, gx;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float gv = ((mk >> i) & 1) ? (float)gin[i] : 0.f;
      g[i] = (_Float16)gv;
      const float is = invstd[q + i];
      gx[i] = (_Float16)((gv - coef[q + i] - ((float)x[i] - mean[q + i]) * coef[c + q + i]) * is * w[q + i]);
    }
    if (GRES) GRES[e] = g;
    GX[e] = gx;
  }
}

static inline int bn_rows_per_slice_h(int64_t n, int c) {
  const int rpp = 256 / (c >> 3);
  int64_t rows = std::max<int64_t>(ts_cdiv(n, BN_MAX_SLICES), 32);
  rows = ts_cdiv(rows, rpp) * rpp;
  return (int)rows;
}

extern "C" int ts_bn_act_train_forward_f16(const void *x, const void *residual, const float *weight, const float *bias,
                                           float *running_mean, float *running_var, int64_t *num_batches_tracked,
                                           int64_t n, int32_t c, float eps, float momentum, int32_t relu, float *mean,
                                           float *invstd, void *out, uint8_t *mask, void *ws, size_t ws_bytes,
                                           ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n > 0 && c > 0 && (c & 7) == 0 && c <= 2048, TS_ERR_UNSUPPORTED,
             "ts_bn_act_train_forward_f16: need N > 0 and C a multiple of 8, <= 2048");
  TS_REQUIRE(x && weight && bias && mean && invstd && out && ws, TS_ERR_INVALID_ARGUMENT,
             "ts_bn_act_train_forward_f16: null pointer");
  TS_REQUIRE(ws_bytes >= ts_bn_train_workspace_bytes(c), TS_ERR_INVALID_ARGUMENT,
             "ts_bn_act_train_forward_f16: workspace too small");
  TS_REQUIRE(bn_aligned(x) && bn_aligned(out) && (!residual || bn_aligned(residual)) && bn_aligned(ws),
             TS_ERR_INVALID_ARGUMENT, "ts_bn_act_train_forward_f16: pointers must be 16-byte aligned");
  float *part = (float *)ws;
  const int rows = bn_rows_per_slice_h(n, c);
  const int slices = (int)ts_cdiv(n, rows);
  bn_partial_h_kernel<0><<<slices, 256, 0, stream>>>((const _Float16 *)x, nullptr, nullptr, nullptr, n, c, rows, part);
  bn_fwd_finish_kernel<<<(unsigned)ts_cdiv(c, BN_FIN_CH), 256, 0, stream>>>(part, slices, (double)n, c, eps, momentum,
                                                                           running_mean, running_var, mean, invstd,
                                                                           num_batches_tracked);
  const int64_t total8 = n * (c / 8);
  const unsigned grid = (unsigned)std::min<int64_t>(ts_cdiv(total8, 256), 1 << 16);
  bn_act_fwd_h_kernel<<<grid, 256, 0, stream>>>((const bn_h8 *)x, (const bn_h8 *)residual, mean, invstd, weight, bias,
                                                total8, c / 8, relu, (bn_h8 *)out, mask);
  TS_CHECK_LAUNCH("ts_bn_act_train_forward_f16");
  return TS_OK;
}

extern "C" int ts_bn_act_train_backward_f16(const void *grad_out, const uint8_t *mask, const void *x, const float *mean,
                                            const float *invstd, const float *weight, int64_t n, int32_t c,
                                            void *grad_x, void *grad_residual, float *grad_weight, float *grad_bias,
                                            void *ws, size_t ws_bytes, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n > 0 && c > 0 && (c & 7) == 0 && c <= 2048, TS_ERR_UNSUPPORTED,
             "ts_bn_act_train_backward_f16: need N > 0 and C a multiple of 8, <= 2048");
  TS_REQUIRE(grad_out && x && mean && invstd && weight && grad_x && ws, TS_ERR_INVALID_ARGUMENT,
             "ts_bn_act_train_backward_f16: null pointer");
  TS_REQUIRE(ws_bytes >= ts_bn_train_workspace_bytes(c), TS_ERR_INVALID_ARGUMENT,
             "ts_bn_act_train_backward_f16: workspace too small");
  TS_REQUIRE(bn_aligned(grad_out) && bn_aligned(x) && bn_aligned(grad_x) && (!grad_residual || bn_aligned(grad_residual)) &&
                 bn_aligned(ws), TS_ERR_INVALID_ARGUMENT, "ts_bn_act_train_backward_f16: pointers must be 16-byte aligned");
  float *part = (float *)ws;
  float *coef = part + (size_t)BN_MAX_SLICES * 2 * c;
  const int rows = bn_rows_per_slice_h(n, c);
  const int slices = (int)ts_cdiv(n, rows);
  if (mask)
    bn_partial_h_kernel<2><<<slices, 256, 0, stream>>>((const _Float16 *)x, (const _Float16 *)grad_out, mask, mean, n, c,
                                                       rows, part);
  else
    bn_partial_h_kernel<1><<<slices, 256, 0, stream>>>((const _Float16 *)x, (const _Float16 *)grad_out, nullptr, mean, n,
                                                       c, rows, part);
  bn_bwd_finish_kernel<<<(unsigned)ts_cdiv(c, BN_FIN_CH), 256, 0, stream>>>(part, slices, (double)n, c, invstd, coef,
                                                                           grad_weight, grad_bias);
  const int64_t total8 = n * (c / 8);
  const unsigned grid = (unsigned)std::min<int64_t>(ts_cdiv(total8, 256), 1 << 16);
  bn_act_bwd_coef_h_kernel<<<grid, 256, 0, stream>>>((const bn_h8 *)grad_out, mask, (const bn_h8 *)x, mean, invstd, weight,
                                                     coef, total8, c, (bn_h8 *)grad_x, (bn_h8 *)grad_residual);
  TS_CHECK_LAUNCH("ts_bn_act_train_backward_f16");
  return TS_OK;
}

// ------------------------------------------------------------------------------------------------------
// SyncBatchNorm halves for half-storage activations (the reference's actual recipe is AMP + DDP + SyncBatchNorm,
// R/dist_train.sh:17-19): the same double sums cross the ranks, activations and gradients stay half in HBM.
__global__ void bn_coef_from_sums_kernel(const double *__restrict__ sums, const double *__restrict__ total_dev,
                                         double total_host, const float *__restrict__ invstd, int c,
                                         float *__restrict__ coef) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const double total = total_dev ? *total_dev : total_host;
  const float is = invstd[ch];
  coef[ch] = (float)(sums[ch] / total);
  coef[c + ch] = (float)(sums[c + ch] / total) * is * is;
}

static int bn_h_check(const char *what, int64_t n, int32_t c) {
  TS_REQUIRE(n > 0 && c > 0 && (c & 7) == 0 && c <= 2048, TS_ERR_UNSUPPORTED, "%s: need N > 0 and C a multiple of 8, <= 2048",
             what);
  return TS_OK;
}

extern "C" int ts_bn_sync_stats_f16(const void *x, int64_t n, int32_t c, double *pack, void *ws, size_t ws_bytes,
                                    ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = bn_h_check("ts_bn_sync_stats_f16", n, c);
  if (rc != TS_OK) return rc;
  TS_REQUIRE(x && pack && ws && bn_aligned(x) && bn_aligned(ws), TS_ERR_INVALID_ARGUMENT, "ts_bn_sync_stats_f16: bad pointer");
  TS_REQUIRE(ws_bytes >= ts_bn_train_workspace_bytes(c), TS_ERR_INVALID_ARGUMENT, "ts_bn_sync_stats_f16: workspace too small");
  float *part = (float *)ws;
  const int rows = bn_rows_per_slice_h(n, c);
  const int slices = (int)ts_cdiv(n, rows);
  bn_partial_h_kernel<0><<<slices, 256, 0, stream>>>((const _Float16 *)x, nullptr, nullptr, nullptr, n, c, rows, part);
  bn_sums_finish_kernel<<<(unsigned)ts_cdiv(c, BN_FIN_CH), 256, 0, stream>>>(part, slices, c, (double)n, nullptr, pack,
                                                                           nullptr, nullptr);
  TS_CHECK_LAUNCH("ts_bn_sync_stats_f16");
  return TS_OK;
}

extern "C" int ts_bn_act_forward_f16(const void *x, const void *residual, const float *mean, const float *invstd,
                                     const float *weight, const float *bias, int64_t n, int32_t c, int32_t relu,
                                     void *out, uint8_t *mask, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = bn_h_check("ts_bn_act_forward_f16", n, c);
  if (rc != TS_OK) return rc;
  TS_REQUIRE(x && mean && invstd && weight && bias && out, TS_ERR_INVALID_ARGUMENT, "ts_bn_act_forward_f16: null pointer");
  TS_REQUIRE(bn_aligned(x) && bn_aligned(out) && (!residual || bn_aligned(residual)), TS_ERR_INVALID_ARGUMENT,
             "ts_bn_act_forward_f16: pointers must be 16-byte aligned");
  const int64_t total8 = n * (c / 8);
  const unsigned grid = (unsigned)std::min<int64_t>(ts_cdiv(total8, 256), 1 << 16);
  bn_act_fwd_h_kernel<<<grid, 256, 0, stream>>>((const bn_h8 *)x, (const bn_h8 *)residual, mean, invstd, weight, bias,
                                                total8, c / 8, relu, (bn_h8 *)out, mask);
  TS_CHECK_LAUNCH("ts_bn_act_forward_f16");
  return TS_OK;
}

extern "C" int ts_bn_sync_backward_reduce_f16(const void *grad_out, const uint8_t *mask, const void *x, const float *mean,
                                              const float *invstd, int64_t n, int32_t c, double *sums,
                                              float *grad_weight, float *grad_bias, void *ws, size_t ws_bytes,
                                              ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = bn_h_check("ts_bn_sync_backward_reduce_f16", n, c);
  if (rc != TS_OK) return rc;
  TS_REQUIRE(grad_out && x && mean && invstd && sums && ws, TS_ERR_INVALID_ARGUMENT,
             "ts_bn_sync_backward_reduce_f16: null pointer");
  TS_REQUIRE(bn_aligned(grad_out) && bn_aligned(x) && bn_aligned(ws), TS_ERR_INVALID_ARGUMENT,
             "ts_bn_sync_backward_reduce_f16: pointers must be 16-byte aligned");
  TS_REQUIRE(ws_bytes >= ts_bn_train_workspace_bytes(c), TS_ERR_INVALID_ARGUMENT,
             "ts_bn_sync_backward_reduce_f16: workspace too small");
  float *part = (float *)ws;
  const int rows = bn_rows_per_slice_h(n, c);
  const int slices = (int)ts_cdiv(n, rows);
  if (mask)
    bn_partial_h_kernel<2><<<slices, 256, 0, stream>>>((const _Float16 *)x, (const _Float16 *)grad_out, mask, mean, n, c,
                                                       rows, part);
  else
    bn_partial_h_kernel<1><<<slices, 256, 0, stream>>>((const _Float16 *)x, (const _Float16 *)grad_out, nullptr, mean, n,
                                                       c, rows, part);
  bn_sums_finish_kernel<<<(unsigned)ts_cdiv(c, BN_FIN_CH), 256, 0, stream>>>(part, slices, c, -1.0, invstd, sums,
                                                                           grad_weight, grad_bias);
  TS_CHECK_LAUNCH("ts_bn_sync_backward_reduce_f16");
  return TS_OK;
}

extern "C" int ts_bn_act_backward_f16(const void *grad_out, const uint8_t *mask, const void *x, const float *mean,
                                      const float *invstd, const float *weight, const double *sums,
                                      const double *total_dev, double total_host, int64_t n, int32_t c, void *grad_x,
                                      void *grad_residual, void *ws, size_t ws_bytes, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = bn_h_check("ts_bn_act_backward_f16", n, c);
  if (rc != TS_OK) return rc;
  TS_REQUIRE(grad_out && x && mean && invstd && weight && sums && grad_x && ws, TS_ERR_INVALID_ARGUMENT,
             "ts_bn_act_backward_f16: null pointer");
  TS_REQUIRE(total_dev || total_host > 0.0, TS_ERR_INVALID_ARGUMENT, "ts_bn_act_backward_f16: empty batch");
  TS_REQUIRE(ws_bytes >= 2 * (size_t)c * sizeof(float), TS_ERR_INVALID_ARGUMENT, "ts_bn_act_backward_f16: workspace too small");
  TS_REQUIRE(bn_aligned(grad_out) && bn_aligned(x) && bn_aligned(grad_x) && (!grad_residual || bn_aligned(grad_residual)) &&
                 bn_aligned(ws), TS_ERR_INVALID_ARGUMENT, "ts_bn_act_backward_f16: pointers must be 16-byte aligned");
  float *coef = (float *)ws;
  bn_coef_from_sums_kernel<<<(unsigned)ts_cdiv(c, 256), 256, 0, stream>>>(sums, total_dev, total_host, invstd, c, coef);
  const int64_t total8 = n * (c / 8);
  const unsigned grid = (unsigned)std::min<int64_t>(ts_cdiv(total8, 256), 1 << 16);
  bn_act_bwd_coef_h_kernel<<<grid, 256, 0, stream>>>((const bn_h8 *)grad_out, mask, (const bn_h8 *)x, mean, invstd, weight,
                                                     coef, total8, c, (bn_h8 *)grad_x, (bn_h8 *)grad_residual);
  TS_CHECK_LAUNCH("ts_bn_act_backward_f16");
  return TS_OK;
}

// ------------------------------------------------------------------------------------------------------
// LeakyReLU -> BatchNorm (training) in one chain of passes, for the dense 2-D branch of TIAF (R/pcseg/model/segmentor/voxel/
// minkunet/unet2d.py:24-30,71,108: every block computes `bn(act(conv(x)))` with `nn.LeakyReLU()`, slope 0.01).  A channels-last
// feature stack [T, H, W, C] IS a row matrix [T*H*W, C], so the sliced statistics / double finish of the sparse BatchNorm above
// serve it; what differs is the activation IN FRONT of the normalisation:
//   forward   a = leaky(x);  out = (a - mean(a)) * invstd(a) * w + b             (x = the convolution's output, kept for backward)
//   backward  gw = sum dy (a - mean) invstd,  gb = sum dy,  ga = (dy - mean(dy) - (a - mean) invstd^2 mean(dy (a - mean))) invstd w,
//             gx = ga * (x > 0 ? 1 : slope)
// Three launches per direction (partial sums -> finish -> elementwise), 2 reads + 1 write of N*C*s forward, 3 reads + 1 write
// backward, where the library path of this PyTorch-ROCm takes a LeakyReLU pass, a layout copy and three MIOpen BatchNorm kernels
// forward, and as many backward.  T = float (4 channels per thread) or _Float16 (8 per thread); statistics fp32 / double.
template <typename T, int VE>
struct alignas(VE * sizeof(T)) LbnVec {
  T x[VE];
};

__device__ __forceinline__ float lbn_leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// MODE 0: (a, a^2)   1: (dy, dy (a - mean))
template <typename T, int VE, int MODE>
__global__ __launch_bounds__(256) void lbn_partial_kernel(const T *__restrict__ X, const T *__restrict__ DY, const float *__restrict__ mean,
                                                          float slope, int64_t n, int c, int rows_per_wg, float *__restrict__ part) {
  using V = LbnVec<T, VE>;
  __shared__ float red[2][256 * VE];
  const int cq = c / VE, rpp = 256 / cq;
  const int tid = threadIdx.x, ty = tid / cq, tx = tid - ty * cq;
  const bool active = ty < rpp;
  float s0[VE], s1[VE], mu[VE];
#pragma unroll
  for (int i = 0; i < VE; ++i) s0[i] = s1[i] = mu[i] = 0.f;
  if (MODE != 0 && active) {
#pragma unroll
    for (int i = 0; i < VE; ++i) mu[i] = mean[VE * tx + i];
  }
  const int64_t r_beg = (int64_t)blockIdx.x * rows_per_wg, r_end = min(n, r_beg + rows_per_wg);
  if (active) {
    auto add = [&](const V &x, const V &d) {
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < VE; ++i) {
          const float a = lbn_leaky((float)x.x[i], slope);
          s0[i] += a;
          s1[i] += a * a;
        }
      } else {
#pragma unroll
        for (int i = 0; i < VE; ++i) {
          const float dv = (float)d.x[i];
          s0[i] += dv;
          s1[i] += dv * (lbn_leaky((float)x.x[i], slope) - mu[i]);
        }
      }
    };
    // 8 rows per trip, their loads issued together (the sums still in row order): a workgroup walks ~10 k rows of a full-resolution
    // map with one 16-byte load per thread and row - at one or two in flight per thread the 512 workgroups ran at 2.9 TB/s, the
    // latency of their loads
    constexpr int U = 8;
    int64_t r = r_beg + ty;
    for (; r + (int64_t)(U - 1) * rpp < r_end; r += (int64_t)U * rpp) {
      V x[U], d[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        x[u] = *(const V *)(X + (r + (int64_t)u * rpp) * c + VE * tx);
        if (MODE != 0) d[u] = *(const V *)(DY + (r + (int64_t)u * rpp) * c + VE * tx);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) add(x[u], d[u]);
    }
    for (; r < r_end; r += rpp) {
      const V x = *(const V *)(X + r * c + VE * tx);
      V d = x;
      if (MODE != 0) d = *(const V *)(DY + r * c + VE * tx);
      add(x, d);
    }
  }
#pragma unroll
  for (int i = 0; i < VE; ++i) {
    red[0][tid * VE + i] = s0[i];
    red[1][tid * VE + i] = s1[i];
  }
  __syncthreads();
  float *out = part + (int64_t)blockIdx.x * 2 * c;
  for (int ch = tid; ch < c; ch += 256) {
    const int q = ch / VE, l = ch % VE;
    float a = 0.f, b = 0.f;
    for (int y = 0; y < rpp; ++y) {
      a += red[0][(y * cq + q) * VE + l];
      b += red[1][(y * cq + q) * VE + l];
    }
    out[ch] = a;
    out[c + ch] = b;
  }
}

template <typename T, int VE>
__global__ __launch_bounds__(256) void lbn_fwd_kernel(const LbnVec<T, VE> *__restrict__ X, const float *__restrict__ mean,
                                                      const float *__restrict__ invstd, const float *__restrict__ w,
                                                      const float *__restrict__ b, float slope, int64_t total, int cq,
                                                      const LbnVec<T, VE> *__restrict__ RES, LbnVec<T, VE> *__restrict__ OUT) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; e < total; e += step) {
    const int q = (int)(e % cq) * VE;
    const LbnVec<T, VE> x = X[e];
    LbnVec<T, VE> y;
#pragma unroll
    for (int i = 0; i < VE; ++i)
      y.x[i] = (T)((lbn_leaky((float)x.x[i], slope) - mean[q + i]) * invstd[q + i] * w[q + i] + b[q + i]);
    if (RES) {                                             // the block's residual sum: added to the ROUNDED result, as the module pair does
      const LbnVec<T, VE> r = RES[e];
#pragma unroll
      for (int i = 0; i < VE; ++i) y.x[i] = (T)((float)y.x[i] + (float)r.x[i]);
    }
    OUT[e] = y;
  }
}

template <typename T, int VE>
__global__ __launch_bounds__(256) void lbn_bwd_kernel(const LbnVec<T, VE> *__restrict__ GOUT, const LbnVec<T, VE> *__restrict__ X,
                                                      const float *__restrict__ mean, const float *__restrict__ invstd,
                                                      const float *__restrict__ w, const float *__restrict__ coef, float slope,
                                                      int64_t total, int c, LbnVec<T, VE> *__restrict__ GX) {
  const int cq = c / VE;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; e < total; e += step) {
    const int q = (int)(e % cq) * VE;
    const LbnVec<T, VE> g = GOUT[e], x = X[e];
    LbnVec<T, VE> gx;
#pragma unroll
    for (int i = 0; i < VE; ++i) {
      const float xv = (float)x.x[i];
      const float ga = ((float)g.x[i] - coef[q + i] - (lbn_leaky(xv, slope) - mean[q + i]) * coef[c + q + i]) * invstd[q + i] * w[q + i];
      gx.x[i] = (T)(xv > 0.f ? ga : ga * slope);
    }
    GX[e] = gx;
  }
}

static int lbn_check(const char *what, int64_t n, int32_t c, int32_t half, size_t ws_bytes) {
  const int ve = half ? 8 : 4;
  TS_REQUIRE(n > 0 && c > 0 && c % ve == 0 && c / ve <= 256 && c <= 1024, TS_ERR_UNSUPPORTED,
             "%s: need N > 0 and C a multiple of %d, <= 1024", what, ve);
  TS_REQUIRE(ws_bytes >= ts_bn_train_workspace_bytes(c), TS_ERR_WORKSPACE_TOO_SMALL, "%s: workspace too small", what);
  return TS_OK;
}

template <typename T, int VE>
static inline int lbn_rows_per_slice(int64_t n, int c) {
  const int rpp = 256 / (c / VE);
  int64_t rows = std::max<int64_t>(ts_cdiv(n, BN_MAX_SLICES), 32);
  return (int)(ts_cdiv(rows, rpp) * rpp);
}

template <typename T, int VE>
static void lbn_forward_launch(const void *x, const float *weight, const float *bias, float *running_mean, float *running_var,
                               int64_t *nbt, int64_t n, int c, float eps, float momentum, float slope, float *mean, float *invstd,
                               const void *residual, void *out, float *part, hipStream_t stream) {
  const int rows = lbn_rows_per_slice<T, VE>(n, c);
  const int slices = (int)ts_cdiv(n, rows);
  lbn_partial_kernel<T, VE, 0><<<slices, 256, 0, stream>>>((const T *)x, nullptr, nullptr, slope, n, c, rows, part);
  bn_fwd_finish_kernel<<<(unsigned)ts_cdiv(c, BN_FIN_CH), 256, 0, stream>>>(part, slices, (double)n, c, eps, momentum, running_mean,
                                                                           running_var, mean, invstd, nbt);
  const int64_t total = n * (c / VE);
  const unsigned grid = (unsigned)std::min<int64_t>(ts_cdiv(total, 256), 1 << 16);
  lbn_fwd_kernel<T, VE><<<grid, 256, 0, stream>>>((const LbnVec<T, VE> *)x, mean, invstd, weight, bias, slope, total, c / VE,
                                                  (const LbnVec<T, VE> *)residual, (LbnVec<T, VE> *)out);
}

template <typename T, int VE>
static void lbn_backward_launch(const void *grad_out, const void *x, const float *mean, const float *invstd, const float *weight,
                                int64_t n, int c, float slope, void *grad_x, float *grad_weight, float *grad_bias, float *part,
                                hipStream_t stream) {
  float *coef = part + (size_t)BN_MAX_SLICES * 2 * c;
  const int rows = lbn_rows_per_slice<T, VE>(n, c);
  const int slices = (int)ts_cdiv(n, rows);
  lbn_partial_kernel<T, VE, 1><<<slices, 256, 0, stream>>>((const T *)x, (const T *)grad_out, mean, slope, n, c, rows, part);
  bn_bwd_finish_kernel<<<(unsigned)ts_cdiv(c, BN_FIN_CH), 256, 0, stream>>>(part, slices, (double)n, c, invstd, coef, grad_weight,
                                                                           grad_bias);
  const int64_t total = n * (c / VE);
  const unsigned grid = (unsigned)std::min<int64_t>(ts_cdiv(total, 256), 1 << 16);
  lbn_bwd_kernel<T, VE><<<grid, 256, 0, stream>>>((const LbnVec<T, VE> *)grad_out, (const LbnVec<T, VE> *)x, mean, invstd, weight,
                                                  coef, slope, total, c, (LbnVec<T, VE> *)grad_x);
}

// out [N, C] = BatchNorm_train(LeakyReLU_slope(x [N, C])) (+ residual [N, C], may be NULL: the block's `skip + y`, added to the rounded
// result); mean / invstd [C] (of the activated values) are kept for the backward; running statistics (may be NULL) and
// num_batches_tracked (may be NULL) updated as nn.BatchNorm2d does.  half != 0: IEEE half rows.
// ws >= ts_bn_train_workspace_bytes(c), every pointer 16-byte aligned.
extern "C" int ts_leaky_bn_train_forward(const void *x, const float *weight, const float *bias, float *running_mean, float *running_var,
                                         int64_t *num_batches_tracked, int64_t n, int32_t c, float eps, float momentum, float slope,
                                         int32_t half, float *mean, float *invstd, const void *residual, void *out, void *ws,
                                         size_t ws_bytes, ts_stream_t stream_) {
  const int rc = lbn_check("ts_leaky_bn_train_forward", n, c, half, ws_bytes);
  if (rc != TS_OK) return rc;
  TS_REQUIRE(x && weight && bias && mean && invstd && out && ws, TS_ERR_INVALID_ARGUMENT, "ts_leaky_bn_train_forward: null pointer");
  TS_REQUIRE(bn_aligned(x) && bn_aligned(out) && bn_aligned(ws) && bn_aligned(residual), TS_ERR_INVALID_ARGUMENT,
             "ts_leaky_bn_train_forward: pointers must be 16-byte aligned");
  if (half)
    lbn_forward_launch<_Float16, 8>(x, weight, bias, running_mean, running_var, num_batches_tracked, n, c, eps, momentum, slope, mean,
                                    invstd, residual, out, (float *)ws, (hipStream_t)stream_);
  else
    lbn_forward_launch<float, 4>(x, weight, bias, running_mean, running_var, num_batches_tracked, n, c, eps, momentum, slope, mean,
                                 invstd, residual, out, (float *)ws, (hipStream_t)stream_);
  TS_CHECK_LAUNCH("ts_leaky_bn_train_forward");
  return TS_OK;
}

// grad_x [N, C] (with respect to the LeakyReLU's input), grad_weight / grad_bias [C] (may be NULL) from grad_out [N, C]
extern "C" int ts_leaky_bn_train_backward(const void *grad_out, const void *x, const float *mean, const float *invstd,
                                          const float *weight, int64_t n, int32_t c, float slope, int32_t half, void *grad_x,
                                          float *grad_weight, float *grad_bias, void *ws, size_t ws_bytes, ts_stream_t stream_) {
  const int rc = lbn_check("ts_leaky_bn_train_backward", n, c, half, ws_bytes);
  if (rc != TS_OK) return rc;
  TS_REQUIRE(grad_out && x && mean && invstd && weight && grad_x && ws, TS_ERR_INVALID_ARGUMENT, "ts_leaky_bn_train_backward: null pointer");
  TS_REQUIRE(bn_aligned(grad_out) && bn_aligned(x) && bn_aligned(grad_x) && bn_aligned(ws), TS_ERR_INVALID_ARGUMENT,
             "ts_leaky_bn_train_backward: pointers must be 16-byte aligned");
  if (half)
    lbn_backward_launch<_Float16, 8>(grad_out, x, mean, invstd, weight, n, c, slope, grad_x, grad_weight, grad_bias, (float *)ws,
                                     (hipStream_t)stream_);
  else
    lbn_backward_launch<float, 4>(grad_out, x, mean, invstd, weight, n, c, slope, grad_x, grad_weight, grad_bias, (float *)ws,
                                  (hipStream_t)stream_);
  TS_CHECK_LAUNCH("ts_leaky_bn_train_backward");
  return TS_OK;
}
