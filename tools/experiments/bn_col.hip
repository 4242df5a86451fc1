// EXPERIMENT (round 4, measured and NOT adopted - kept out of libtaseg_hip.so): one launch per direction for the BatchNorm of
// the small matrices, every workgroup OWNING the statistics of one 16-byte channel slice of all rows.
//
// Idea: 38 of the 63 BatchNorms of MinkUNet mk34 sit at strides 8 / 16 (<= 10k rows at bs 2); their partial-sum, finish and
// elementwise kernels each run at the ~5 us floor of a dependent launch (6 launches, ~35 us per layer and direction pair for a few
// MB of data).  A column-owning workgroup needs no grid barrier and no partials: 1024 threads keep their <= 12 row chunks of the
// slice in registers (or stream the slice twice through L2), reduce, finish the per-channel math and normalise.
// Results (gpurun_out/r4a, bench.py default, one box, 30 steps after 8 warm-up):
//     three launches (product)                         15.57 ms / step     AMP 10.92 ms
//     one launch, rows <= 24 576                       16.71 ms            AMP 11.63 ms
//     one launch, rows <= 12 288 / <= 4 096            16.97 / 16.04 ms (with the direct 2x2x2 plans: 15.83 without it)
//     streaming form only                              17.35 ms
// i.e. +15 us per layer and direction instead of -10: a wave's 64 lanes read 64 different rows, so every load instruction touches
// 64 cache lines of which 16 bytes each are used, and C / 4 = 32-64 workgroups occupy 32-64 of the 256 CUs - the texture path
// of those few CUs serialises ~4 x N line requests per direction (9.2k rows: ~18 us) where the three launches stream the same
// bytes coalesced over 200+ CUs.  Together with steps 9c / 10d / 10e / 11c of HISTORY.md section 3.1 this closes the list of
// launch mergers for the BatchNorm chain that do not need a grid-wide synchronisation.  It also moved one mk34 gradient
// (up1.0.net.0.kernel) from 4.7e-4 to 1.6e-3 of the float64 evaluation - another summation order in a chaotic train-mode chain.
//
// To rebuild for measurements: paste the block below into csrc/bn.hip before "Single-process training BatchNorm" and call
// bn_col_forward<BnF32 | BnF16> / bn_col_backward<...> at the top of the four ts_bn_act_train_* entry points when bn_one_launch().
#if 0
typedef _Float16 bn_h8 __attribute__((ext_vector_type(8)));
// ------------------------------------------------------------------------------------------------------
// ONE launch per direction for the small matrices (round 4).  38 of the 63 BatchNorms of MinkUNet mk34 sit at strides 8 / 16
// (<= 10k rows at bs 2): their partial-sum, finish and elementwise kernels each run at the ~5 us floor of a dependent launch -
// 6 launches and ~35 us per layer for a few MB of data.  A reduction over all rows has to finish before the first element can
// be normalised, so one launch needs either a grid barrier (tried in round 2, step 10e: slower) or workgroups that OWN their
// statistics: here a 1024-thread workgroup owns one 16-byte channel slice (4 floats / 8 halves) of ALL rows - every thread
// keeps its <= 16 row chunks in registers, the workgroup reduces them (wave shuffles, then the 16 wave sums in double, fixed
// order: deterministic), finishes the per-channel math itself and normalises its registers: no partials, no scratch, no
// second pass over the input.  The variance is the mean of (x - mean)^2 taken from the registers (two reductions), not
// E[x^2] - mean^2.  Workgroups whose slices share a 128-byte line are placed on the same XCD (blockIdx round-robins over the
// 8 XCDs), so each XCD's L2 fetches an eighth of the matrix.  Same outputs as the three-launch path (mean, invstd, running
// statistics, mask, out / grad_x, grad_residual, grad_weight, grad_bias) to the last bits of the summation order.
// TASEG_BN_ONE_LAUNCH=0 keeps every layer on the three launches.
#define BNC_T 1024
#define BNC_WAVES (BNC_T / 64)

struct BnF32 {
  typedef float4 V;
  static constexpr int NV = 4;
  static __device__ __forceinline__ void unpack(const V &v, float (&f)[4]) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
  static __device__ __forceinline__ V pack(const float (&f)[4]) { return make_float4(f[0], f[1], f[2], f[3]); }
  static __device__ __forceinline__ unsigned mask_bits(unsigned byte) { return byte & 0xFu; }
};
struct BnF16 {
  typedef bn_h8 V;
  static constexpr int NV = 8;
  static __device__ __forceinline__ void unpack(const V &v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
  }
  static __device__ __forceinline__ V pack(const float (&f)[8]) {
    V v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (_Float16)f[i];
    return v;
  }
  static __device__ __forceinline__ unsigned mask_bits(unsigned byte) { return byte; }
};

__device__ __forceinline__ int bnc_slice(int b, int nsl) { return (nsl & 7) == 0 ? (b & 7) * (nsl >> 3) + (b >> 3) : b; }

// sum over the workgroup of NV per-thread floats: wave shuffles in float, the 16 wave sums added in double in wave order
template <int NV>
__device__ __forceinline__ void bnc_reduce(float (&v)[NV], double (*red)[NV], double (&out)[NV]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v[i] += __shfl_xor(v[i], d, 64);
  }
  __syncthreads();                 // the previous reduction's sums have been read
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wave][i] = (double)v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) out[i] = 0.0;
#pragma unroll 1                   // (unrolled, the 16 x NV doubles of all waves are live at once: > 128 VGPRs)
  for (int w = 0; w < BNC_WAVES; ++w) {
#pragma unroll
    for (int i = 0; i < NV; ++i) out[i] += red[w][i];
  }
}

template <class T, int R>
__global__ __launch_bounds__(BNC_T) void bn_col_fwd_kernel(const typename T::V *__restrict__ X,
                                                           const typename T::V *__restrict__ RES,
                                                           const float *__restrict__ w, const float *__restrict__ b,
                                                           float *__restrict__ running_mean, float *__restrict__ running_var,
                                                           int64_t *__restrict__ num_batches_tracked, int n, float eps,
                                                           float momentum, int relu, float *__restrict__ mean,
                                                           float *__restrict__ invstd, typename T::V *__restrict__ OUT,
                                                           unsigned char *__restrict__ MASK) {
  constexpr int NV = T::NV;
  typedef typename T::V V;
  __shared__ double red[BNC_WAVES][NV];
  const int nsl = gridDim.x, s = bnc_slice(blockIdx.x, nsl), tid = threadIdx.x;
  V v[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int r = min(tid + i * BNC_T, n - 1);           // clamped address, masked below (no divergent loads)
    v[i] = X[(int64_t)r * nsl + s];
  }
  float acc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = 0.f;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    float f[NV];
    T::unpack(v[i], f);
    const bool ok = tid + i * BNC_T < n;
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] += ok ? f[j] : 0.f;
  }
  double tot[NV];
  bnc_reduce<NV>(acc, red, tot);
  float m[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    m[j] = (float)(tot[j] / (double)n);
    acc[j] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < R; ++i) {
    float f[NV];
    T::unpack(v[i], f);
    const bool ok = tid + i * BNC_T < n;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const float d = f[j] - m[j];
      acc[j] += ok ? d * d : 0.f;
    }
  }
  bnc_reduce<NV>(acc, red, tot);
  float is[NV], ww[NV], bb[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const double var = tot[j] / (double)n;
    is[j] = (float)(1.0 / sqrt(var + (double)eps));
    ww[j] = w[NV * s + j];
    bb[j] = b[NV * s + j];
    if (tid == 0) {
      const int ch = NV * s + j;
      mean[ch] = m[j];
      invstd[ch] = is[j];
      if (running_mean) running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * m[j];
      if (running_var) {
        const double unbiased = n > 1 ? var * (double)n / ((double)n - 1.0) : var;
        running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unbiased;
      }
    }
  }
  if (num_batches_tracked && blockIdx.x == 0 && tid == 0) *num_batches_tracked += 1;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int r = tid + i * BNC_T;
    if (r < n) {
      const int64_t e = (int64_t)r * nsl + s;
      float f[NV], rr[NV];
      T::unpack(v[i], f);
      if (RES) T::unpack(RES[e], rr);
      unsigned mk = 0;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float y = (f[j] - m[j]) * is[j] * ww[j] + bb[j];
        if (RES) y += rr[j];
        if (relu) {
          mk |= (y > 0.f ? 1u : 0u) << j;
          y = fmaxf(y, 0.f);
        }
        f[j] = y;
      }
      if (relu && MASK) MASK[e] = (unsigned char)mk;
      OUT[e] = T::pack(f);
    }
  }
}

template <class T, int R>
__global__ __launch_bounds__(BNC_T) void bn_col_bwd_kernel(const typename T::V *__restrict__ GOUT,
                                                           const unsigned char *__restrict__ MASK,
                                                           const typename T::V *__restrict__ X,
                                                           const float *__restrict__ mean, const float *__restrict__ invstd,
                                                           const float *__restrict__ w, int n, typename T::V *__restrict__ GX,
                                                           typename T::V *__restrict__ GRES, float *__restrict__ grad_weight,
                                                           float *__restrict__ grad_bias) {
  constexpr int NV = T::NV;
  typedef typename T::V V;
  __shared__ double red[BNC_WAVES][2 * NV];
  const int nsl = gridDim.x, s = bnc_slice(blockIdx.x, nsl), tid = threadIdx.x;
  float m[NV], is[NV], ww[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    m[j] = mean[NV * s + j];
    is[j] = invstd[NV * s + j];
    ww[j] = w[NV * s + j];
  }
  V g[R];            // the masked output gradient stays in registers; x is read again in the second phase (L2-hot)
  float acc[2 * NV];
#pragma unroll
  for (int j = 0; j < 2 * NV; ++j) acc[j] = 0.f;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int r = min(tid + i * BNC_T, n - 1);
    const int64_t e = (int64_t)r * nsl + s;
    const bool ok = tid + i * BNC_T < n;
    float gf[NV], xf[NV];
    T::unpack(GOUT[e], gf);
    T::unpack(X[e], xf);
    const unsigned mk = MASK ? T::mask_bits(MASK[e]) : 0xFFu;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      gf[j] = (ok && ((mk >> j) & 1)) ? gf[j] : 0.f;
      acc[j] += gf[j];
      acc[NV + j] += gf[j] * (xf[j] - m[j]);
    }
    g[i] = T::pack(gf);
  }
  double tot[2 * NV];
  bnc_reduce<2 * NV>(acc, red, tot);
  float c0[NV], c1[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    c0[j] = (float)(tot[j] / (double)n);
    c1[j] = (float)(tot[NV + j] / (double)n) * is[j] * is[j];
    if (tid == 0) {
      if (grad_bias) grad_bias[NV * s + j] = (float)tot[j];
      if (grad_weight) grad_weight[NV * s + j] = (float)tot[NV + j] * is[j];
    }
  }
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int r = tid + i * BNC_T;
    if (r < n) {
      const int64_t e = (int64_t)r * nsl + s;
      float gf[NV], xf[NV];
      T::unpack(g[i], gf);
      T::unpack(X[e], xf);
      if (GRES) GRES[e] = g[i];
#pragma unroll
      for (int j = 0; j < NV; ++j) gf[j] = (gf[j] - c0[j] - (xf[j] - m[j]) * c1[j]) * is[j] * ww[j];
      GX[e] = T::pack(gf);
    }
  }
}

// The same for matrices whose column slice does not fit the registers (<= BNC_STREAM_ROWS rows): the workgroup streams its
// slice twice - sums (x, x^2 in float per lane over <= 24 rows, double across lanes: the three-launch path's accuracy), then the
// elementwise pass on the L2-hot slice.
#define BNC_STREAM_ROWS (24 * BNC_T)
template <class T>
__global__ __launch_bounds__(BNC_T) void bn_col_fwd_stream_kernel(const typename T::V *__restrict__ X,
                                                                  const typename T::V *__restrict__ RES,
                                                                  const float *__restrict__ w, const float *__restrict__ b,
                                                                  float *__restrict__ running_mean,
                                                                  float *__restrict__ running_var,
                                                                  int64_t *__restrict__ num_batches_tracked, int n, float eps,
                                                                  float momentum, int relu, float *__restrict__ mean,
                                                                  float *__restrict__ invstd, typename T::V *__restrict__ OUT,
                                                                  unsigned char *__restrict__ MASK) {
  constexpr int NV = T::NV;
  __shared__ double red[BNC_WAVES][2 * NV];
  const int nsl = gridDim.x, s = bnc_slice(blockIdx.x, nsl), tid = threadIdx.x;
  float acc[2 * NV];
#pragma unroll
  for (int j = 0; j < 2 * NV; ++j) acc[j] = 0.f;
#pragma unroll 4
  for (int r = tid; r < n; r += BNC_T) {
    float f[NV];
    T::unpack(X[(int64_t)r * nsl + s], f);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      acc[j] += f[j];
      acc[NV + j] += f[j] * f[j];
    }
  }
  double tot[2 * NV];
  bnc_reduce<2 * NV>(acc, red, tot);
  float m[NV], is[NV], ww[NV], bb[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const double mu = tot[j] / (double)n;
    double var = tot[NV + j] / (double)n - mu * mu;
    if (var < 0.0) var = 0.0;
    m[j] = (float)mu;
    is[j] = (float)(1.0 / sqrt(var + (double)eps));
    ww[j] = w[NV * s + j];
    bb[j] = b[NV * s + j];
    if (tid == 0) {
      const int ch = NV * s + j;
      mean[ch] = m[j];
      invstd[ch] = is[j];
      if (running_mean) running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * m[j];
      if (running_var) {
        const double unbiased = n > 1 ? var * (double)n / ((double)n - 1.0) : var;
        running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unbiased;
      }
    }
  }
  if (num_batches_tracked && blockIdx.x == 0 && tid == 0) *num_batches_tracked += 1;
#pragma unroll 4
  for (int r = tid; r < n; r += BNC_T) {
    const int64_t e = (int64_t)r * nsl + s;
    float f[NV], rr[NV];
    T::unpack(X[e], f);
    if (RES) T::unpack(RES[e], rr);
    unsigned mk = 0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      float y = (f[j] - m[j]) * is[j] * ww[j] + bb[j];
      if (RES) y += rr[j];
      if (relu) {
        mk |= (y > 0.f ? 1u : 0u) << j;
        y = fmaxf(y, 0.f);
      }
      f[j] = y;
    }
    if (relu && MASK) MASK[e] = (unsigned char)mk;
    OUT[e] = T::pack(f);
  }
}

template <class T>
__global__ __launch_bounds__(BNC_T) void bn_col_bwd_stream_kernel(const typename T::V *__restrict__ GOUT,
                                                                  const unsigned char *__restrict__ MASK,
                                                                  const typename T::V *__restrict__ X,
                                                                  const float *__restrict__ mean,
                                                                  const float *__restrict__ invstd, const float *__restrict__ w,
                                                                  int n, typename T::V *__restrict__ GX,
                                                                  typename T::V *__restrict__ GRES,
                                                                  float *__restrict__ grad_weight,
                                                                  float *__restrict__ grad_bias) {
  constexpr int NV = T::NV;
  __shared__ double red[BNC_WAVES][2 * NV];
  const int nsl = gridDim.x, s = bnc_slice(blockIdx.x, nsl), tid = threadIdx.x;
  float m[NV], is[NV], ww[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    m[j] = mean[NV * s + j];
    is[j] = invstd[NV * s + j];
    ww[j] = w[NV * s + j];
  }
  float acc[2 * NV];
#pragma unroll
  for (int j = 0; j < 2 * NV; ++j) acc[j] = 0.f;
#pragma unroll 4
  for (int r = tid; r < n; r += BNC_T) {
    const int64_t e = (int64_t)r * nsl + s;
    float gf[NV], xf[NV];
    T::unpack(GOUT[e], gf);
    T::unpack(X[e], xf);
    const unsigned mk = MASK ? T::mask_bits(MASK[e]) : 0xFFu;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const float gv = ((mk >> j) & 1) ? gf[j] : 0.f;
      acc[j] += gv;
      acc[NV + j] += gv * (xf[j] - m[j]);
    }
  }
  double tot[2 * NV];
  bnc_reduce<2 * NV>(acc, red, tot);
  float c0[NV], c1[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    c0[j] = (float)(tot[j] / (double)n);
    c1[j] = (float)(tot[NV + j] / (double)n) * is[j] * is[j];
    if (tid == 0) {
      if (grad_bias) grad_bias[NV * s + j] = (float)tot[j];
      if (grad_weight) grad_weight[NV * s + j] = (float)tot[NV + j] * is[j];
    }
  }
#pragma unroll 4
  for (int r = tid; r < n; r += BNC_T) {
    const int64_t e = (int64_t)r * nsl + s;
    float gf[NV], xf[NV];
    T::unpack(GOUT[e], gf);
    T::unpack(X[e], xf);
    const unsigned mk = MASK ? T::mask_bits(MASK[e]) : 0xFFu;
#pragma unroll
    for (int j = 0; j < NV; ++j) gf[j] = ((mk >> j) & 1) ? gf[j] : 0.f;
    if (GRES) GRES[e] = T::pack(gf);
#pragma unroll
    for (int j = 0; j < NV; ++j) gf[j] = (gf[j] - c0[j] - (xf[j] - m[j]) * c1[j]) * is[j] * ww[j];
    GX[e] = T::pack(gf);
  }
}

// TASEG_BN_ONE_LAUNCH=0: off; TASEG_BN_ONE_LAUNCH_ROWS: row limit (default / maximum BNC_STREAM_ROWS); TASEG_BN_COL_REGS: most
// 1024-row chunks the register form takes before the streaming form (default 12)
static int g_bn_one_launch = -1, g_bn_col_regs = 12;
static int64_t g_bn_col_rows = BNC_STREAM_ROWS;
static bool bn_one_launch(int64_t n, int c, int nv) {
  if (g_bn_one_launch < 0) {
    const char *e = getenv("TASEG_BN_ONE_LAUNCH");
    g_bn_one_launch = (e && e[0] == '0') ? 0 : 1;
    if ((e = getenv("TASEG_BN_ONE_LAUNCH_ROWS"))) g_bn_col_rows = std::min<int64_t>(std::max<int64_t>(atoll(e), 0), 64 * BNC_T);
    if ((e = getenv("TASEG_BN_COL_REGS"))) g_bn_col_regs = std::max(atoi(e), 0);
  }
  return g_bn_one_launch && n <= g_bn_col_rows && c % nv == 0;
}

template <class T>
static int bn_col_forward(const void *x, const void *residual, const float *weight, const float *bias, float *running_mean,
                          float *running_var, int64_t *nbt, int64_t n, int32_t c, float eps, float momentum, int32_t relu,
                          float *mean, float *invstd, void *out, uint8_t *mask, hipStream_t stream) {
  typedef typename T::V V;
  const unsigned grid = (unsigned)(c / T::NV);
  const int r = (int)ts_cdiv(n, BNC_T);
#define BNC_FWD(R)                                                                                                          \
  bn_col_fwd_kernel<T, R><<<grid, BNC_T, 0, stream>>>((const V *)x, (const V *)residual, weight, bias, running_mean,          \
                                                      running_var, nbt, (int)n, eps, momentum, relu, mean, invstd, (V *)out, \
                                                      mask)
  constexpr int RMAX = T::NV == 4 ? 12 : 8;            // (8 halves per chunk: 12 chunks spill)
  if (r > std::min(RMAX, g_bn_col_regs))
    bn_col_fwd_stream_kernel<T><<<grid, BNC_T, 0, stream>>>((const V *)x, (const V *)residual, weight, bias, running_mean,
                                                            running_var, nbt, (int)n, eps, momentum, relu, mean, invstd,
                                                            (V *)out, mask);
  else if (r <= 2) BNC_FWD(2);
  else if (r <= 4) BNC_FWD(4);
  else if (r <= 8) BNC_FWD(8);
  else if constexpr (T::NV == 4) BNC_FWD(12);
#undef BNC_FWD
  TS_CHECK_LAUNCH("bn_col_forward");
  return TS_OK;
}

template <class T>
static int bn_col_backward(const void *grad_out, const uint8_t *mask, const void *x, const float *mean, const float *invstd,
                           const float *weight, int64_t n, int32_t c, void *grad_x, void *grad_residual, float *grad_weight,
                           float *grad_bias, hipStream_t stream) {
  typedef typename T::V V;
  const unsigned grid = (unsigned)(c / T::NV);
  const int r = (int)ts_cdiv(n, BNC_T);
#define BNC_BWD(R)                                                                                                        \
  bn_col_bwd_kernel<T, R><<<grid, BNC_T, 0, stream>>>((const V *)grad_out, mask, (const V *)x, mean, invstd, weight, (int)n, \
                                                      (V *)grad_x, (V *)grad_residual, grad_weight, grad_bias)
  constexpr int RMAX = T::NV == 4 ? 12 : 8;
  if (r > std::min(RMAX, g_bn_col_regs))
    bn_col_bwd_stream_kernel<T><<<grid, BNC_T, 0, stream>>>((const V *)grad_out, mask, (const V *)x, mean, invstd, weight,
                                                            (int)n, (V *)grad_x, (V *)grad_residual, grad_weight, grad_bias);
  else if (r <= 2) BNC_BWD(2);
  else if (r <= 4) BNC_BWD(4);
  else if (r <= 8) BNC_BWD(8);
  else if constexpr (T::NV == 4) BNC_BWD(12);
#undef BNC_BWD
  TS_CHECK_LAUNCH("bn_col_backward");
  return TS_OK;
}

#endif
