// EXPERIMENT (round 3, not on the product path): the pair GEMM of conv_pairs_s.hip with a three-product IEEE-half split instead
// of the six-product bf16 split.  An operand element, scaled by a power of two s so that the tensor's largest magnitude lies in
// [2^14, 2^15), is written as  x s = h + 2^-11 l'  with h = rn_f16(x s), l' = rn_f16((x s - h) 2^11): 22 significant bits.
// A product is  h_a h_b + (2^-11 h_a) l'_b + l'_a (2^-11 h_b)  - three v_mfma_f32_16x16x32_f16, half products are exact in fp32 -
// and the dropped l' l' term is 2^-22 of the product.  LDS holds TWO planes of the gathered operand (h, l'; 2^-11 h is one
// v_pk_mul_f16 per fragment) and three of the weight (h, l', 2^-11 h).  Entry point ts_debug_pair_gemm_x; tools/x_probe.py.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define PX_BM 128
#define PX_BK 32
#define PX_AP (PX_BK + 8)

__device__ __forceinline__ unsigned pk_f16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, h2));       // v_cvt_pk_f16_f32 (RNE)
}

// 8 floats (already scaled by s) -> planes h, l'
__device__ __forceinline__ void split8x(const f32x4 &v0, const f32x4 &v1, float s, u32x4 &h, u32x4 &l) {
  const float a[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = a[2 * i] * s, x1 = a[2 * i + 1] * s;
    const unsigned hh = pk_f16(x0, x1);
    const h2 hv = __builtin_bit_cast(h2, hh);
    h[i] = hh;
    l[i] = pk_f16((x0 - (float)hv[0]) * 2048.f, (x1 - (float)hv[1]) * 2048.f);
  }
}

__device__ __forceinline__ h8 scale_down(const h8 &v) {                                   // 2^-11 v: four v_pk_mul_f16
  const _Float16 c = (_Float16)0.00048828125f;
  return v * (h8){c, c, c, c, c, c, c, c};
}

__device__ __forceinline__ h8 frag_tr_x(const unsigned short *img, int pitch, int r0, int c0, int tq, int tp) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + (r0 + tq) * pitch + c0 + 4 * tp));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + (r0 + 4 + tq) * pitch + c0 + 4 * tp));
  return __builtin_bit_cast(h8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int BN, int WR, bool WT>
__global__ __launch_bounds__(256, 3) void pair_gemm_x_kernel(const float *__restrict__ X, int R, const float *__restrict__ W,
                                                            int O_total, const int2 *__restrict__ nbmaps,
                                                            const int *__restrict__ nboffs, int K, int gcol, float sx,
                                                            float sw, float *__restrict__ Z) {
  constexpr int BM = PX_BM;
  constexpr int WC = 4 / WR;
  constexpr int MI = (BM / 16) / WR;
  constexpr int NI = (BN / 16) / WC;
  constexpr int BP = BN + 8;
  constexpr int A_PLANE = BM * PX_AP;
  constexpr int B_PLANE = WT ? BN * PX_AP : PX_BK * BP;
  constexpr int A_IT = BM * (PX_BK / 8) / 256;
  constexpr int B_CHUNKS = BN * (PX_BK / 8);
  constexpr int B_IT = (B_CHUNKS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem_x[];
  unsigned short *Ap = smem_x;                                 // 2 planes [128][PX_AP]
  unsigned short *Bp = Ap + 2 * A_PLANE;                       // 3 planes

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int tq = r16 >> 2, tp = lane & 3;
  const int wr = wave / WC, wc = wave % WC;
  const int o0 = blockIdx.y * BN;

  const int offv = nboffs[min(lane, K)];
  const int offn = nboffs[min(lane + 1, K)];
  int incl = lane < K ? (offn - offv + BM - 1) / BM : 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  const int tile = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  if (tile >= __builtin_amdgcn_readlane(incl, 63)) return;
  const int k = __builtin_popcountll(__builtin_amdgcn_ballot_w64(incl <= tile));
  const int t_in_k = tile - (k ? __builtin_amdgcn_readlane(incl, max(k - 1, 0)) : 0);
  const int p0 = __builtin_amdgcn_readlane(offv, k) + t_in_k * BM;
  const int np = min(BM, __builtin_amdgcn_readlane(offv, k + 1) - p0);

  const int arow0 = tid >> 2, acol = (tid & 3) << 3;
  const float *aptr[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int2 pr = nbmaps[p0 + min(arow0 + 64 * it, np - 1)];
    aptr[it] = X + (int64_t)(gcol ? pr.y : pr.x) * R + acol;
  }
  int boff[B_IT], bdst[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int e = min(tid + it * 256, B_CHUNKS - 1);
    if (WT) {
      const int col = e >> 2, c8 = (e & 3) << 3;
      boff[it] = col * R + c8;
      bdst[it] = col * PX_AP + c8;
    } else {
      constexpr int q8 = BN >> 3;
      const int kk = e / q8, c8 = (e - kk * q8) << 3;
      boff[it] = kk * O_total + c8;
      bdst[it] = kk * BP + c8;
    }
  }
  const float *wk = WT ? W + ((int64_t)k * O_total + o0) * R : W + (int64_t)k * R * O_total + o0;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 ra[A_IT][2], rb[B_IT][2];
  auto load_regs = [&](int c0) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      ra[it][0] = *(const f32x4 *)(aptr[it] + c0);
      ra[it][1] = *(const f32x4 *)(aptr[it] + c0 + 4);
    }
    const float *wb = WT ? wk + c0 : wk + (int64_t)c0 * O_total;
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      rb[it][0] = *(const f32x4 *)(wb + boff[it]);
      rb[it][1] = *(const f32x4 *)(wb + boff[it] + 4);
    }
  };
  auto store_lds = [&]() {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int rr = arow0 + 64 * it;
      const bool live = rr < np;
      u32x4 h, l;
      split8x(live ? ra[it][0] : zero, live ? ra[it][1] : zero, sx, h, l);
      unsigned short *dst = Ap + rr * PX_AP + acol;
      *(u32x4 *)dst = h;
      *(u32x4 *)(dst + A_PLANE) = l;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      if (B_IT * 256 == B_CHUNKS || tid + it * 256 < B_CHUNKS) {
        u32x4 h, l;
        split8x(rb[it][0], rb[it][1], sw, h, l);
        unsigned short *dst = Bp + bdst[it];
        *(u32x4 *)dst = h;
        *(u32x4 *)(dst + B_PLANE) = l;
        *(u32x4 *)(dst + 2 * B_PLANE) = __builtin_bit_cast(u32x4, scale_down(__builtin_bit_cast(h8, h)));
      }
    }
  };
  auto mma = [&]() {
    h8 a[MI][3];                                  // h, l', 2^-11 h
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int p = 0; p < 2; ++p)
        a[mi][p] = *(const h8 *)&Ap[p * A_PLANE + ((wr * MI + mi) * 16 + r16) * PX_AP + 8 * g];
      a[mi][2] = scale_down(a[mi][0]);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      h8 b[3];                                    // h, l', 2^-11 h
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        if (WT)
          b[p] = *(const h8 *)&Bp[p * B_PLANE + ((wc * NI + ni) * 16 + r16) * PX_AP + 8 * g];
        else
          b[p] = frag_tr_x(Bp + p * B_PLANE, BP, 8 * g, (wc * NI + ni) * 16, tq, tp);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi][1], b[2], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi][2], b[1], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi][0], b[0], acc[mi][ni], 0, 0, 0);
      }
    }
  };

  load_regs(0);
  for (int c0 = 0; c0 < R; c0 += PX_BK) {
    if (c0) __syncthreads();
    store_lds();
    __syncthreads();
    if (c0 + PX_BK < R) load_regs(c0 + PX_BK);
    mma();
  }
  const float inv = 1.0f / (sx * sw);
  float *zt = Z + (int64_t)p0 * O_total + o0;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = (wr * MI + mi) * 16 + 4 * g + q;
        if (row < np) zt[(int64_t)row * O_total + (wc * NI + ni) * 16 + r16] = acc[mi][ni][q] * inv;
      }
}

template <int BN, bool WT>
static int launch_x(const float *X, int R, const float *W, int O_total, const int2 *nbmaps, const int *nboffs, int K, int64_t P,
                    int gcol, float sx, float sw, float *Z, hipStream_t stream) {
  const size_t lds = (size_t)(2 * PX_BM * PX_AP + 3 * (WT ? BN * PX_AP : PX_BK * (BN + 8))) * 2;
  dim3 grid((unsigned)((ts_cdiv(P, PX_BM) + K + 7) / 8 * 8), (unsigned)(O_total / BN));
  pair_gemm_x_kernel<BN, 2, WT><<<grid, 256, lds, stream>>>(X, R, W, O_total, nbmaps, nboffs, K, gcol, sx, sw, Z);
  TS_CHECK_LAUNCH("ts_debug_pair_gemm_x");
  return TS_OK;
}

// sx, sw: powers of two that bring the largest |x| / |w| into [2^14, 2^15)
extern "C" int ts_debug_pair_gemm_x(const float *X, int32_t R, const float *W, int32_t O_total, const int32_t *nbmaps,
                                    const int32_t *nboffs, int32_t K, int64_t P, int32_t gcol, int32_t wt, float sx, float sw,
                                    float *Z, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(X && W && nbmaps && nboffs && Z && P > 0 && K > 0 && K <= 63 && R % 32 == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_debug_pair_gemm_x: bad arguments");
  TS_REQUIRE(O_total % 96 == 0 || O_total % 128 == 0, TS_ERR_UNSUPPORTED, "ts_debug_pair_gemm_x: 96- or 128-column tiles only");
  const int2 *nm = (const int2 *)nbmaps;
  if (O_total % 128 == 0)
    return wt ? launch_x<128, true>(X, R, W, O_total, nm, nboffs, K, P, gcol, sx, sw, Z, stream)
              : launch_x<128, false>(X, R, W, O_total, nm, nboffs, K, P, gcol, sx, sw, Z, stream);
  return wt ? launch_x<96, true>(X, R, W, O_total, nm, nboffs, K, P, gcol, sx, sw, Z, stream)
            : launch_x<96, false>(X, R, W, O_total, nm, nboffs, K, P, gcol, sx, sw, Z, stream);
}
