// Flat-buffer SGD step with gradient-norm clipping and AMP loss-scale handling, all decided on the device.
//
// The reference's step (R/train.py:399-417) is scaler.unscale_ -> clip_grad_norm_(10) -> scaler.step(SGD) ->
// scaler.update(): torch runs it as ~10 multi-tensor launches over 380 parameters plus one device->host read
// (GradScaler.step asks found_inf.item()), which drains the launch queue once per step.  Here parameters, gradients
// and momentum live in a few flat buckets (taseg_amd.optim.FlatSGD) and the step is
//   ts_sgd_grad_stats   per bucket: sum of squares + non-finite flag of the (scaled) gradients -> stats buffer
//   ts_sgd_decide       one thread: total norm of the UNSCALED gradients, clip coefficient, skip flag, new loss scale
//   ts_sgd_apply        per bucket: p, m updated in place unless the step is skipped
// with no host read anywhere.  Arithmetic = torch.optim.SGD (momentum, weight decay, dampening 0, no nesterov - the
// reference ignores its NESTEROV key, optim/__init__.py:15-21) + torch.nn.utils.clip_grad_norm_ + GradScaler.
#include "common.h"

// state (float[8], device): 0 loss scale, 1 growth tracker, 2 clip coefficient x inv_scale (what multiplies the stored
// gradient), 3 skip flag, 4 total norm (unscaled), 5 first-step flag (momentum buffers empty)
#define SGD_STATE_FLOATS 8

__global__ __launch_bounds__(256) void sgd_grad_stats_kernel(const float *__restrict__ g, int64_t n,
                                                             double *__restrict__ sumsq, int *__restrict__ nonfinite) {
  __shared__ double red[256];
  double acc = 0.0;
  int bad = 0;
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int64_t step = (int64_t)gridDim.x * blockDim.x * 4;
  for (; i + 3 * step + 3 < n; i += 4 * step) {       // four independent 16-byte loads in flight per thread
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const float4 *)(g + i + u * step);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc += (double)v[u].x * v[u].x + (double)v[u].y * v[u].y + (double)v[u].z * v[u].z + (double)v[u].w * v[u].w;
      bad |= !(isfinite(v[u].x) && isfinite(v[u].y) && isfinite(v[u].z) && isfinite(v[u].w));
    }
  }
  for (; i + 3 < n; i += step) {
    const float4 v = *(const float4 *)(g + i);
    acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    bad |= !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w));
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t t = n & ~(int64_t)3; t < n; ++t) {   // tail (n % 4 elements)
      acc += (double)g[t] * g[t];
      bad |= !isfinite(g[t]);
    }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(sumsq, red[0]);
  if (bad) atomicOr(nonfinite, 1);
}

__global__ void sgd_decide_kernel(double *__restrict__ sumsq, int *__restrict__ nonfinite, float *__restrict__ state,
                                  float max_norm, float growth, float backoff, int growth_interval, int amp) {
  const float scale = amp ? state[0] : 1.f;
  const float inv = 1.f / scale;
  const bool bad = *nonfinite != 0 || !isfinite(*sumsq);
  const float norm = (float)sqrt(*sumsq) * inv;                        // norm of the unscaled gradients
  float coef = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.f;       // clip_grad_norm_: min(1, max / (norm + eps))
  coef = fminf(coef, 1.f);
  state[2] = coef * inv;
  state[3] = bad ? 1.f : 0.f;
  state[4] = norm;
  if (amp) {                                                           // GradScaler.update()
    if (bad) {
      state[0] = scale * backoff;
      state[1] = 0.f;
    } else {
      const float tracker = state[1] + 1.f;
      if ((int)tracker >= growth_interval) {
        state[0] = scale * growth;
        state[1] = 0.f;
      } else {
        state[1] = tracker;
      }
    }
  }
  *sumsq = 0.0;          // re-arm for the next step
  *nonfinite = 0;
}

__global__ __launch_bounds__(256) void sgd_apply_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                        float *__restrict__ m, int64_t n,
                                                        const float *__restrict__ state, float lr, float momentum,
                                                        float weight_decay, int first) {
  if (state[3] != 0.f) return;                 // non-finite gradients: skip the step (GradScaler semantics)
  const float gs = state[2];
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const float pv = p[i];
    const float d = g[i] * gs + weight_decay * pv;
    const float buf = first ? d : momentum * m[i] + d;
    m[i] = buf;
    p[i] = pv - lr * buf;
  }
}

extern "C" int ts_sgd_grad_stats(const float *grad, int64_t n, double *sumsq, int32_t *nonfinite, ts_stream_t stream) {
  TS_REQUIRE(n >= 0 && sumsq && nonfinite, TS_ERR_INVALID_ARGUMENT, "ts_sgd_grad_stats: bad arguments");
  if (n == 0) return TS_OK;
  TS_REQUIRE(grad && (((uintptr_t)grad) & 15) == 0, TS_ERR_INVALID_ARGUMENT, "ts_sgd_grad_stats: grad must be 16-byte aligned");
  // few workgroups with long loops: every workgroup ends in ONE double atomic on the same address, and ~2000 of
  // those took longer (18 us) than reading the bucket
  const unsigned grid = (unsigned)std::min<int64_t>(ts_cdiv(n, 16384), 512);
  sgd_grad_stats_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(grad, n, sumsq, nonfinite);
  TS_CHECK_LAUNCH("ts_sgd_grad_stats");
  return TS_OK;
}

extern "C" int ts_sgd_decide(double *sumsq, int32_t *nonfinite, float *state, float max_norm, float growth,
                             float backoff, int32_t growth_interval, int32_t amp, ts_stream_t stream) {
  TS_REQUIRE(sumsq && nonfinite && state, TS_ERR_INVALID_ARGUMENT, "ts_sgd_decide: null pointer");
  sgd_decide_kernel<<<1, 1, 0, (hipStream_t)stream>>>(sumsq, nonfinite, state, max_norm, growth, backoff,
                                                      growth_interval, amp);
  TS_CHECK_LAUNCH("ts_sgd_decide");
  return TS_OK;
}

extern "C" int ts_sgd_apply(float *param, const float *grad, float *momentum_buf, int64_t n, const float *state,
                            float lr, float momentum, float weight_decay, int32_t first_step, ts_stream_t stream) {
  TS_REQUIRE(n >= 0 && state, TS_ERR_INVALID_ARGUMENT, "ts_sgd_apply: bad arguments");
  if (n == 0) return TS_OK;
  TS_REQUIRE(param && grad && momentum_buf, TS_ERR_INVALID_ARGUMENT, "ts_sgd_apply: null pointer");
  const unsigned grid = (unsigned)std::min<int64_t>(ts_cdiv(n, 256), 8192);
  sgd_apply_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(param, grad, momentum_buf, n, state, lr, momentum,
                                                          weight_decay, first_step);
  TS_CHECK_LAUNCH("ts_sgd_apply");
  return TS_OK;
}
