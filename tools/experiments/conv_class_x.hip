// EXPERIMENT (round 4, not on the product path): the fp32 class-sorted implicit GEMM of csrc/conv_class.hip with THREE IEEE-half MFMAs
// per product instead of six bf16 ones, and the scale problem of the round-3 prototype (conv_pairs_x.hip: one power of two per
// TENSOR, which every producer would have to emit) solved inside the kernel: the walk already flushes a step's accumulator into a
// running total, so a scale may differ per (row, 32-column slice) of the gathered operand - it factors out of that slice's MFMAs:
//   gathered row i, slice s:  e = 15 - exponent(max |x|) over the 32 columns (4 lanes: one max3 chain + two DPP steps),
//                             x 2^e = h + 2^-11 l'  (h = rn_f16, l' = rn_f16((x 2^e - h) 2^11): 22 significant bits of the slice's max)
//   weight slice of offset k: pre-scaled by 2^eb[k] on the host side of the probe (a per-offset exponent: what a planes pass at
//                             optimizer time would store), h | l' | 2^-11 h split in the kernel here (the product path would read
//                             planes)
//   acc (zeroed per slice) = l'_a (2^-11 h_b) + h_a (2^-11 l'_b) + h_a h_b ;  tot += acc 2^-(e + eb[k])
// Entry point ts_debug_class_gemm_x; tools/experiments/class_x_probe.py.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CX_BM 128
#define CX_BK 32
#define CX_AP (CX_BK + 8)

__device__ __forceinline__ unsigned cx_pk(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, h2));
}
// 8 floats times the power of two s -> planes h, l'
__device__ __forceinline__ void cx_split8(const f32x4 &v0, const f32x4 &v1, float s, u32x4 &h, u32x4 &l) {
  const float a[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = a[2 * i] * s, x1 = a[2 * i + 1] * s;
    const unsigned hh = cx_pk(x0, x1);
    const h2 hv = __builtin_bit_cast(h2, hh);
    h[i] = hh;
    l[i] = cx_pk((x0 - (float)hv[0]) * 2048.f, (x1 - (float)hv[1]) * 2048.f);
  }
}
__device__ __forceinline__ h8 cx_down(const h8 &v) {
  const _Float16 c = (_Float16)0.00048828125f;
  return v * (h8){c, c, c, c, c, c, c, c};
}
__device__ __forceinline__ h8 cx_frag_tr(const unsigned short *img, int pitch, int r0, int c0, int tq, int tp) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + (r0 + tq) * pitch + c0 + 4 * tp));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + (r0 + 4 + tq) * pitch + c0 + 4 * tp));
  return __builtin_bit_cast(h8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ float cx_pow2(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }

template <int BN, int WR, bool WT>
__global__ __launch_bounds__(256, 2) void class_gemm_x_kernel(const float *__restrict__ X, int R, const float *__restrict__ W,
                                                             int O_total, const int *__restrict__ src, int64_t m_pad,
                                                             const int2 *__restrict__ tile_info, const int *__restrict__ n_tiles,
                                                             int K, int gk, int mirror, const int *__restrict__ eb,
                                                             float *__restrict__ Zp) {
  constexpr int BM = CX_BM;
  constexpr int WC = 4 / WR;
  constexpr int MI = (BM / 16) / WR;
  constexpr int NI = (BN / 16) / WC;
  constexpr int BP = BN + 8;
  constexpr int A_PLANE = BM * CX_AP;
  constexpr int B_PLANE = WT ? BN * CX_AP : CX_BK * BP;
  constexpr int A_IT = BM * (CX_BK / 8) / 256;
  constexpr int B_CHUNKS = BN * (CX_BK / 8);
  constexpr int B_IT = (B_CHUNKS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem_cx[];
  unsigned short *Ap = smem_cx;                                // 2 planes [128][CX_AP]: h, l'
  unsigned short *Bp = Ap + 2 * A_PLANE;                       // 3 planes: h, 2^-11 l', 2^-11 h
  int *sc = (int *)(Bp + 3 * B_PLANE);                         // [128] exponent of the row's slice

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int tq = r16 >> 2, tp = lane & 3;
  const int wr = wave / WC, wc = wave % WC;
  const int o0 = blockIdx.y * BN;
  const int tile = (int)blockIdx.x;
  if (tile >= *n_tiles) return;
  const int2 info = tile_info[tile];
  const int grp = __builtin_amdgcn_readfirstlane(info.x) & 3;
  int mask = __builtin_amdgcn_readfirstlane(info.y);
  const int64_t row0 = (int64_t)(__builtin_amdgcn_readfirstlane(info.x) >> 2) * BM;
  if (mask == 0) return;

  const int arow0 = tid >> 2, acol = (tid & 3) << 3;
  int boff[B_IT], bdst[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int e = min(tid + it * 256, B_CHUNKS - 1);
    if (WT) {
      const int col = e >> 2, c8 = (e & 3) << 3;
      boff[it] = col * R + c8;
      bdst[it] = col * CX_AP + c8;
    } else {
      constexpr int q8 = BN >> 3;
      const int kk = e / q8, c8 = (e - kk * q8) << 3;
      boff[it] = kk * O_total + c8;
      bdst[it] = kk * BP + c8;
    }
  }
  f32x4 tot[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) tot[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float *aptr[A_IT];
  bool alive[A_IT];
  const float *wk = W;
  int nsrc[A_IT];
  int ebk = 0, ebk_next = 0;
  auto fetch = [&](int kl) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) nsrc[it] = src[(int64_t)kl * m_pad + row0 + arow0 + 64 * it];
  };
  auto bind = [&](int kl) {
    const int k = gk * grp + kl;
    const int kw = (WT && mirror) ? (K - 1 - k) : k;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      alive[it] = nsrc[it] >= 0;
      aptr[it] = X + (int64_t)max(nsrc[it], 0) * R + acol;
    }
    wk = WT ? W + ((int64_t)kw * O_total + o0) * R : W + (int64_t)kw * R * O_total + o0;
    ebk_next = eb[kw];
  };
  f32x4 ra[A_IT][2], rb[B_IT][2];
  bool rlive[A_IT];
  float sb_cur = 1.f;
  auto load_regs = [&](int c0) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      if (alive[it]) {
        ra[it][0] = *(const f32x4 *)(aptr[it] + c0);
        ra[it][1] = *(const f32x4 *)(aptr[it] + c0 + 4);
      }
      rlive[it] = alive[it];
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
      const f32x4 v0 = rlive[it] ? ra[it][0] : zero, v1 = rlive[it] ? ra[it][1] : zero;
      // largest magnitude of the row's 32 columns: 8 here, the other 24 in the three neighbouring lanes
      float m = fmaxf(fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3]))),
                      fmaxf(fmaxf(fabsf(v1[0]), fabsf(v1[1])), fmaxf(fabsf(v1[2]), fabsf(v1[3]))));
      m = fmaxf(m, __shfl_xor(m, 1, 64));
      m = fmaxf(m, __shfl_xor(m, 2, 64));
      // m < 2^ex: the slice's values times 2^(15 - ex) lie below 2^15 (and the largest at or above 2^14)
      const int ex = (int)((__float_as_uint(m) >> 23) & 255u) - 126;
      const int e = m > 0.f ? min(max(15 - ex, -100), 100) : 0;
      u32x4 h, l;
      cx_split8(v0, v1, cx_pow2(e), h, l);
      unsigned short *dst = Ap + rr * CX_AP + acol;
      *(u32x4 *)dst = h;
      *(u32x4 *)(dst + A_PLANE) = l;
      if ((tid & 3) == 0) sc[rr] = e;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      if (B_IT * 256 == B_CHUNKS || tid + it * 256 < B_CHUNKS) {
        u32x4 h, l;
        cx_split8(rb[it][0], rb[it][1], sb_cur, h, l);
        unsigned short *dst = Bp + bdst[it];
        *(u32x4 *)dst = h;
        *(u32x4 *)(dst + B_PLANE) = __builtin_bit_cast(u32x4, cx_down(__builtin_bit_cast(h8, l)));
        *(u32x4 *)(dst + 2 * B_PLANE) = __builtin_bit_cast(u32x4, cx_down(__builtin_bit_cast(h8, h)));
      }
    }
  };
  auto mma_flush = [&]() {
    h8 a[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int p = 0; p < 2; ++p) a[mi][p] = *(const h8 *)&Ap[p * A_PLANE + ((wr * MI + mi) * 16 + r16) * CX_AP + 8 * g];
    // per-row factors of this slice: 2^-(e_row + eb[k]); rows 4g .. 4g+3 of every 16-row block
    f32x4 rs[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int *sp = sc + (wr * MI + mi) * 16 + 4 * g;
#pragma unroll
      for (int q = 0; q < 4; ++q) rs[mi][q] = cx_pow2(-(sp[q] + ebk));
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      h8 b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        if (WT)
          b[p] = *(const h8 *)&Bp[p * B_PLANE + ((wc * NI + ni) * 16 + r16) * CX_AP + 8 * g];
        else
          b[p] = cx_frag_tr(Bp + p * B_PLANE, BP, 8 * g, (wc * NI + ni) * 16, tq, tp);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi][1], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi][0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi][0], b[0], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) tot[mi][ni][q] = fmaf(acc[q], rs[mi][q], tot[mi][ni][q]);
      }
    }
  };

  fetch(__builtin_ctz(mask));
  bind(__builtin_ctz(mask));
  mask &= mask - 1;
  load_regs(0);
  int eb_loaded = ebk_next;               // exponent of the weight slice sitting in rb
  bool first = true;
  while (true) {
    if (mask) fetch(__builtin_ctz(mask));
    for (int c0 = 0; c0 < R; c0 += CX_BK) {
      if (!first) __syncthreads();
      first = false;
      ebk = eb_loaded;
      sb_cur = cx_pow2(ebk);
      store_lds();
      __syncthreads();
      if (c0 + CX_BK < R) {
        load_regs(c0 + CX_BK);
      } else if (mask) {
        bind(__builtin_ctz(mask));
        load_regs(0);
        eb_loaded = ebk_next;
      }
      mma_flush();
    }
    if (!mask) break;
    mask &= mask - 1;
  }
  float *zt = Zp + row0 * O_total + o0;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        zt[(int64_t)((wr * MI + mi) * 16 + 4 * g + q) * O_total + (wc * NI + ni) * 16 + r16] = tot[mi][ni][q];
}

template <int BN, bool WT>
static int launch_cx(const float *X, int R, const float *W, int O_total, const int *src, int64_t m_pad, const int2 *tile_info,
                     const int *n_tiles, int K, int gk, int mirror, const int *eb, float *Zp, hipStream_t stream) {
  const size_t lds = (size_t)(2 * CX_BM * CX_AP + 3 * (WT ? BN * CX_AP : CX_BK * (BN + 8))) * 2 + CX_BM * 4;
  dim3 grid((unsigned)(m_pad / CX_BM), (unsigned)(O_total / BN));
  class_gemm_x_kernel<BN, 2, WT><<<grid, 256, lds, stream>>>(X, R, W, O_total, src, m_pad, tile_info, n_tiles, K, gk, mirror, eb, Zp);
  TS_CHECK_LAUNCH("ts_debug_class_gemm_x");
  return TS_OK;
}

// eb [K] int32 (device): exponent per offset with max |W_k| 2^eb[k] in [2^14, 2^15)
extern "C" int ts_debug_class_gemm_x(const float *feat, int32_t c_red, const float *kernel, int32_t K, int32_t groups, int32_t c_out,
                                     const int32_t *src, int64_t m_pad, const int32_t *tile_info, const int32_t *n_tiles, int32_t wt,
                                     int32_t mirror, const int32_t *eb, float *zp, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(feat && kernel && src && tile_info && n_tiles && eb && zp && c_red % 32 == 0 && K % groups == 0,
             TS_ERR_INVALID_ARGUMENT, "ts_debug_class_gemm_x: bad arguments");
  TS_REQUIRE(c_out % 96 == 0 || c_out % 128 == 0, TS_ERR_UNSUPPORTED, "ts_debug_class_gemm_x: 96- or 128-column tiles only");
  const int2 *ti = (const int2 *)tile_info;
  const int gk = K / groups;
  if (c_out % 128 == 0)
    return wt ? launch_cx<128, true>(feat, c_red, kernel, c_out, src, m_pad, ti, n_tiles, K, gk, mirror, eb, zp, stream)
              : launch_cx<128, false>(feat, c_red, kernel, c_out, src, m_pad, ti, n_tiles, K, gk, mirror, eb, zp, stream);
  return wt ? launch_cx<96, true>(feat, c_red, kernel, c_out, src, m_pad, ti, n_tiles, K, gk, mirror, eb, zp, stream)
            : launch_cx<96, false>(feat, c_red, kernel, c_out, src, m_pad, ti, n_tiles, K, gk, mirror, eb, zp, stream);
}
