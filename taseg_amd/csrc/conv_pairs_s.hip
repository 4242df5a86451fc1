// fp32 sparse-convolution GEMMs on the bf16 matrix pipe ("split" kernels): every fp32 operand element is written as
// the exact sum of three bf16 numbers, x = h + m + l (h = rne_bf16(x), m = rne_bf16(x - h), l = x - h - m: the two
// remainders are exact in fp32 and l needs at most 8 significant bits), and a product a * b is evaluated as
//      ah*bh + (ah*bm + am*bh) + (am*bm + ah*bl + al*bh)
// with v_mfma_f32_16x16x32_bf16: bf16 x bf16 products are exact in fp32 and the accumulators are fp32.  The three
// dropped terms (am*bl, al*bm, al*bl) are bounded by 2^-23 |a b| - below the rounding error the fp32 FMA chain of
// v_mfma_f32_16x16x4_f32 commits per product.  Six bf16 MFMAs of depth 32 (16 cycles each) replace eight fp32 MFMAs
// of depth 4 (32 cycles each): 2.7x less matrix-pipe time for the same fp32 operands and fp32 results, which turns
// the 96/128-wide layers from MFMA-bound into HBM-bound kernels (gfx950 has no xf32 / tf32 mode to do this natively).
//
//   pair_gemm_s_kernel   pass 1 of the two-pass convolution, Z[p, :] = X[g_p, :] @ W_k (forward: WT = false, the
//                        [K, Ci, Co] weight read through transposing LDS loads; dgrad: WT = true)
//   wgrad_s_kernel       dW_k = sum_pairs A[pa]^T B[pb]
// Tiling, tile -> offset map, chunking and flush order are those of pair_gemm_fast_kernel / wgrad_gemm_fast_kernel
// (conv_pairs.hip); the split happens once per element when a gathered slice is written to LDS (4.5 VALU ops per
// element, v_cvt_pk_bf16_f32), the LDS stage holds three bf16 planes per operand and is single buffered so that two
// workgroups share a CU.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// plain stores: non-temporal Z stores were measured 0.35 ms per step SLOWER (gather_sum re-reads Z from the caches)
#define TS_ZSTORE(v, p) (*(p) = (v))
#define PS_BM 128
#define PS_BK 32
#define PS_AP (PS_BK + 8)   // row pitch of a [row][k] plane in bf16: 80 bytes = 20 dwords (4 mod 8)

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf2));   // v_cvt_pk_bf16_f32 (RNE)
}

// 8 floats -> three planes of 8 bf16 (16 bytes each)
__device__ __forceinline__ void split8(const f32x4 &v0, const f32x4 &v1, u32x4 &h, u32x4 &m, u32x4 &l) {
  const float a[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = a[2 * i], x1 = a[2 * i + 1];
    const unsigned hh = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(hh << 16), r1 = x1 - __uint_as_float(hh & 0xffff0000u);
    const unsigned mm = pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(mm << 16), s1 = r1 - __uint_as_float(mm & 0xffff0000u);
    h[i] = hh;
    m[i] = mm;
    l[i] = pk_bf16(s0, s1);
  }
}

// fragment of a [k][col] plane: 8 consecutive k rows (r0 .. r0+7) of column c0 + (lane & 15), read with gfx950's
// transposing LDS load (lane 4q + p of a 16-lane group addresses row q, columns 4p .. 4p+3; see wgrad_h_kernel)
__device__ __forceinline__ bf8 frag_tr(const unsigned short *img, int pitch, int r0, int c0, int tq, int tp) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + (r0 + tq) * pitch + c0 + 4 * tp));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + (r0 + 4 + tq) * pitch + c0 + 4 * tp));
  return __builtin_bit_cast(bf8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// six of the nine partial products, smallest first
#define TS_SPLIT_MMA(ACC, A, B)                                                         \
  do {                                                                                  \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[2], (B)[0], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[0], (B)[2], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[1], (B)[1], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[1], (B)[0], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[0], (B)[1], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[0], (B)[0], ACC, 0, 0, 0);        \
  } while (0)

// X [*, R] fp32 rows; W = [K, R, O_total] (WT = false) or [K, O_total, R] (WT = true); Z [P, O_total] fp32
template <int BM, int BN, int WR, bool WT, bool XCD>
__global__ __launch_bounds__(256, BM == 128 ? 2 : 4) void pair_gemm_s_kernel(const float *__restrict__ X, int R,
                                                          const float *__restrict__ W, int O_total,
                                                          const int2 *__restrict__ nbmaps,
                                                          const int *__restrict__ nboffs, int K, int gcol,
                                                          float *__restrict__ Z) {
  constexpr int WC = 4 / WR;
  constexpr int MI = (BM / 16) / WR;
  constexpr int NI = (BN / 16) / WC;
  constexpr int BP = BN + 8;                                   // pitch of a [k][col] weight plane
  constexpr int A_PLANE = BM * PS_AP;
  constexpr int B_PLANE = WT ? BN * PS_AP : PS_BK * BP;
  constexpr int A_IT = BM * (PS_BK / 8) / 256;              // 8-float chunks per thread per A slice (2)
  constexpr int B_CHUNKS = BN * (PS_BK / 8);                   // 8-float chunks of a weight slice (either layout)
  constexpr int B_IT = (B_CHUNKS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem_s[];
  unsigned short *Ap = smem_s;                                 // 3 planes [128][PS_AP]
  unsigned short *Bp = Ap + 3 * A_PLANE;                       // 3 planes

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int tq = r16 >> 2, tp = lane & 3;
  const int wr = wave / WC, wc = wave % WC;
  const int o0 = blockIdx.y * BN;

  // tile -> (offset k, first pair, rows), as in pair_gemm_fast_kernel
  const int offv = nboffs[min(lane, K)];
  const int offn = nboffs[min(lane + 1, K)];
  int incl = lane < K ? (offn - offv + BM - 1) / BM : 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  // workgroups go round-robin over the 8 XCDs (gridDim.x is a multiple of 8): XCD x takes the x-th eighth of the tile
  // list, i.e. 3 - 4 consecutive offsets, so that its L2 holds the weight slices it multiplies with (27 W_k of a
  // 256 x 256 layer are 7 MB, an XCD's L2 4 MB)
  const int tile = XCD ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (tile >= __builtin_amdgcn_readlane(incl, 63)) return;
  const int k = __builtin_popcountll(__builtin_amdgcn_ballot_w64(incl <= tile));
  const int t_in_k = tile - (k ? __builtin_amdgcn_readlane(incl, max(k - 1, 0)) : 0);
  const int p0 = __builtin_amdgcn_readlane(offv, k) + t_in_k * BM;
  const int np = min(BM, __builtin_amdgcn_readlane(offv, k + 1) - p0);

  // A slots: 8-float chunk (tid & 3) of tile row (tid >> 2) + 64 it
  const int arow0 = tid >> 2, acol = (tid & 3) << 3;
  const float *aptr[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int2 pr = nbmaps[p0 + min(arow0 + 64 * it, np - 1)];
    aptr[it] = X + (int64_t)(gcol ? pr.y : pr.x) * R + acol;
  }
  // B slots
  int boff[B_IT], bdst[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int e = min(tid + it * 256, B_CHUNKS - 1);
    if (WT) {
      const int col = e >> 2, c8 = (e & 3) << 3;
      boff[it] = col * R + c8;
      bdst[it] = col * PS_AP + c8;
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
      u32x4 h, m, l;
      split8(live ? ra[it][0] : zero, live ? ra[it][1] : zero, h, m, l);
      unsigned short *dst = Ap + rr * PS_AP + acol;
      *(u32x4 *)dst = h;
      *(u32x4 *)(dst + A_PLANE) = m;
      *(u32x4 *)(dst + 2 * A_PLANE) = l;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      if (B_IT * 256 == B_CHUNKS || tid + it * 256 < B_CHUNKS) {
        u32x4 h, m, l;
        split8(rb[it][0], rb[it][1], h, m, l);
        unsigned short *dst = Bp + bdst[it];
        *(u32x4 *)dst = h;
        *(u32x4 *)(dst + B_PLANE) = m;
        *(u32x4 *)(dst + 2 * B_PLANE) = l;
      }
    }
  };
  auto mma = [&]() {
    bf8 a[MI][3];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        a[mi][p] = *(const bf8 *)&Ap[p * A_PLANE + ((wr * MI + mi) * 16 + r16) * PS_AP + 8 * g];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      bf8 b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        if (WT)
          b[p] = *(const bf8 *)&Bp[p * B_PLANE + ((wc * NI + ni) * 16 + r16) * PS_AP + 8 * g];
        else
          b[p] = frag_tr(Bp + p * B_PLANE, BP, 8 * g, (wc * NI + ni) * 16, tq, tp);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) TS_SPLIT_MMA(acc[mi][ni], a[mi], b);
    }
  };

  load_regs(0);
  for (int c0 = 0; c0 < R; c0 += PS_BK) {
    if (c0) __syncthreads();          // the previous slice's fragments have been read
    store_lds();
    __syncthreads();
    if (c0 + PS_BK < R) load_regs(c0 + PS_BK);
    mma();
  }
  float *zt = Z + (int64_t)p0 * O_total + o0;
  if (np == BM) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          TS_ZSTORE(acc[mi][ni][q], &zt[(int64_t)((wr * MI + mi) * 16 + 4 * g + q) * O_total + (wc * NI + ni) * 16 + r16]);
  } else {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = (wr * MI + mi) * 16 + 4 * g + q;
          if (row < np) TS_ZSTORE(acc[mi][ni][q], &zt[(int64_t)row * O_total + (wc * NI + ni) * 16 + r16]);
        }
  }
}

template <int BM, int BN, int WR, bool WT>
static int launch_pair_gemm_s(const float *X, int R, const float *W, int O_total, const int2 *nbmaps, const int *nboffs,
                              int K, int64_t P, int gcol, float *Z, hipStream_t stream) {
  const size_t lds = (size_t)3 * (BM * PS_AP + (WT ? BN * PS_AP : PS_BK * (BN + 8))) * 2;
  dim3 grid((unsigned)((ts_cdiv(P, BM) + K + 7) / 8 * 8), (unsigned)(O_total / BN));
  if (g_ts_conv_impl == 8)    // tiles in launch order (A/B of the XCD remap)
    pair_gemm_s_kernel<BM, BN, WR, WT, false><<<grid, 256, lds, stream>>>(X, R, W, O_total, nbmaps, nboffs, K, gcol, Z);
  else
    pair_gemm_s_kernel<BM, BN, WR, WT, true><<<grid, 256, lds, stream>>>(X, R, W, O_total, nbmaps, nboffs, K, gcol, Z);
  TS_CHECK_LAUNCH("conv_pair_gemm (split)");
  return TS_OK;
}

// called by ts_conv_pair_gemm for the full-tile shapes (R % 32 == 0, O_total % bn == 0, 16-byte aligned, K <= 63).
// Measured and NOT adopted (profiles/r01_v14_ab_tiles.txt): 64 x 64 tiles for the layers with few pairs (strides 8 / 16:
// 4x the workgroups, 4 per CU) are 0 - 10 % slower than the 128-row tiles - those layers are bound by the bytes each
// tile pulls through L2, not by latency or occupancy - and so is launch order vs the XCD-contiguous tile order
// (+-3 %).  ts_set_conv_impl(6) selects the small tiles, (8) launch-order tiles.
// Round 2 (profiles/r02_experiments_pair_gemm.txt): an ablation shows the phases of a workgroup (gather + split + LDS 70 us,
// MFMAs 85 us, Z stores 60 us of the 216 us stride-1 96 -> 96 launch) running back to back; tried against it and NOT
// adopted: persistent workgroups with cross-tile prefetch (hipcc spills at 256 VGPRs: 1.3x slower), tiles launched in
// spatial order for L2 reuse of the gathered rows (+8-15 % on the 32 / 64-wide layers, -0-12 % on the wide ones), 64- /
// 32-column tiles for more workgroups per CU (+-3 %), an LDS-free kernel whose lanes load their MFMA fragments straight
// from global memory at 4 waves per SIMD (bound by the L1 rate of the weight fragments: 1.4-1.9x slower), a producer /
// consumer kernel (4 loader waves fill a double-buffered LDS stage a slice ahead of 4 MFMA waves, one barrier per slice,
// one workgroup per CU: 1.1-1.4x slower).  With every gather served from cache the launch still takes 172 us, without Z
// stores 156 us: compute pipeline (~115 us), Z stores (~75 us) and gather (~70 us) each overlap the others only partly.
int ts_pair_gemm_split(const float *X, int R, const float *W, int O_total, const int2 *nbmaps, const int *nboffs, int K,
                       int64_t P, int gcol, float *Z, int bn, int wt, hipStream_t stream) {
#define TS_PS(BM, BN, WR)                                                                                   \
  (wt ? launch_pair_gemm_s<BM, BN, WR, true>(X, R, W, O_total, nbmaps, nboffs, K, P, gcol, Z, stream)       \
      : launch_pair_gemm_s<BM, BN, WR, false>(X, R, W, O_total, nbmaps, nboffs, K, P, gcol, Z, stream))
  if (g_ts_conv_impl == 6 && O_total % 64 == 0) return TS_PS(64, 64, 2);
  switch (bn) {
    case 32: return TS_PS(128, 32, 4);
    case 64: return TS_PS(128, 64, 2);
    case 96: return TS_PS(128, 96, 2);
    default: return TS_PS(128, 128, 2);
  }
#undef TS_PS
}

// ------------------------------------------------------------------------------------- weight gradient
//   dW_k[ci, co] = sum_{pairs p of k} A[pa_p, ci] * B[pb_p, co]     (A, B fp32 rows, dW fp32)
// Chunking / flushing as wgrad_gemm_fast_kernel; operands are staged pair-major as they arrive ([pair][channel]
// planes) and the MFMA fragments - 8 consecutive pairs of one channel per lane - come from ds_read_b64_tr_b16.
#define WS_PS 32            // pairs per step (one 32-deep MFMA k-block)
#define WS_MAXCHUNK 1024

template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void wgrad_s_kernel(const float *__restrict__ A, int CA,
                                                      const float *__restrict__ B, int CB,
                                                      const int2 *__restrict__ nbmaps, const int *__restrict__ nboffs,
                                                      int K, int P, int col_a, int chunk, float *__restrict__ dW,
                                                      float *__restrict__ part) {
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int XP = TM + 8, YP = TN + 8;              // plane pitches in bf16 (16-byte multiples, 4 mod 8 dwords)
  constexpr int X_PLANE = WS_PS * XP, Y_PLANE = WS_PS * YP;
  constexpr int A_CH = WS_PS * (TM / 8), B_CH = WS_PS * (TN / 8);
  constexpr int A_IT = (A_CH + 255) / 256, B_IT = (B_CH + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem_ws[];
  unsigned short *Xl = smem_ws;                        // 3 x [32][XP]
  unsigned short *Yl = Xl + 3 * X_PLANE;               // 3 x [32][YP]
  int *idxA = (int *)(Yl + 3 * Y_PLANE), *idxB = idxA + WS_MAXCHUNK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int tq = r16 >> 2, tp = lane & 3;
  const int wr = wave >> 1, wc = wave & 1;
  const int p_beg = blockIdx.x * chunk;
  const int p_end = min(P, p_beg + chunk);
  if (p_beg >= p_end) return;
  const int tiles_n = CB / TN;
  const int ci0 = (blockIdx.y / tiles_n) * TM, co0 = (blockIdx.y % tiles_n) * TN;

  for (int t = tid; t < p_end - p_beg; t += 256) {
    const int2 pr = nbmaps[p_beg + t];
    idxA[t] = col_a ? pr.y : pr.x;
    idxB[t] = col_a ? pr.x : pr.y;
  }
  const int offv = nboffs[min(lane, K)];
  const int k0 = __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < K && offv <= p_beg)) - 1;
  auto off_at = [&](int kk) { return __builtin_amdgcn_readlane(offv, kk); };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  struct WStep {
    int k, p0, np;
  };
  auto advance = [&](WStep st) -> WStep {
    int kend = min(off_at(st.k + 1), p_end);
    int pn = st.p0 + WS_PS;
    if (pn < kend) {
      st.p0 = pn;
      st.np = min(WS_PS, kend - pn);
      return st;
    }
    pn = kend;
    for (++st.k; st.k < K && pn < p_end; ++st.k) {
      kend = min(off_at(st.k + 1), p_end);
      if (kend > pn) {
        st.p0 = pn;
        st.np = min(WS_PS, kend - pn);
        return st;
      }
    }
    st.k = K;
    return st;
  };
  const float *abase = A + ci0, *bbase = B + co0;
  f32x4 ra[A_IT][2], rb[B_IT][2];
  auto load_regs = [&](const WStep &st) {
    const int l0 = st.p0 - p_beg, last = st.np - 1;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int e = min(tid + it * 256, A_CH - 1);
      const int pp = e / (TM / 8), c8 = (e - pp * (TM / 8)) << 3;
      const float *src = abase + (int64_t)idxA[l0 + min(pp, last)] * CA + c8;
      ra[it][0] = *(const f32x4 *)src;
      ra[it][1] = *(const f32x4 *)(src + 4);
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int e = min(tid + it * 256, B_CH - 1);
      const int pp = e / (TN / 8), c8 = (e - pp * (TN / 8)) << 3;
      const float *src = bbase + (int64_t)idxB[l0 + min(pp, last)] * CB + c8;
      rb[it][0] = *(const f32x4 *)src;
      rb[it][1] = *(const f32x4 *)(src + 4);
    }
  };
  auto store_lds = [&](const WStep &st) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int e = tid + it * 256;
      const int pp = e / (TM / 8), c8 = (e - pp * (TM / 8)) << 3;
      if (A_IT * 256 == A_CH || e < A_CH) {
        const bool live = pp < st.np;
        u32x4 h, m, l;
        split8(live ? ra[it][0] : zero, live ? ra[it][1] : zero, h, m, l);
        unsigned short *dst = Xl + pp * XP + c8;
        *(u32x4 *)dst = h;
        *(u32x4 *)(dst + X_PLANE) = m;
        *(u32x4 *)(dst + 2 * X_PLANE) = l;
      }
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int e = tid + it * 256;
      const int pp = e / (TN / 8), c8 = (e - pp * (TN / 8)) << 3;
      if (B_IT * 256 == B_CH || e < B_CH) {
        const bool live = pp < st.np;
        u32x4 h, m, l;
        split8(live ? rb[it][0] : zero, live ? rb[it][1] : zero, h, m, l);
        unsigned short *dst = Yl + pp * YP + c8;
        *(u32x4 *)dst = h;
        *(u32x4 *)(dst + Y_PLANE) = m;
        *(u32x4 *)(dst + 2 * Y_PLANE) = l;
      }
    }
  };

  WStep cur;
  cur.k = k0;
  cur.p0 = p_beg;
  cur.np = min(WS_PS, min(off_at(k0 + 1), p_end) - p_beg);
  __syncthreads();  // pair indices visible
  load_regs(cur);
  bool first = true;
  while (cur.k < K) {
    if (!first) __syncthreads();      // the previous step's fragments have been read
    first = false;
    store_lds(cur);
    __syncthreads();
    const WStep nxt = advance(cur);
    if (nxt.k < K) load_regs(nxt);
    {
      bf8 a[MI][3];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int p = 0; p < 3; ++p) a[mi][p] = frag_tr(Xl + p * X_PLANE, XP, 8 * g, (wr * MI + mi) * 16, tq, tp);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        bf8 b[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) b[p] = frag_tr(Yl + p * Y_PLANE, YP, 8 * g, (wc * NI + ni) * 16, tq, tp);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) TS_SPLIT_MMA(acc[mi][ni], a[mi], b);
      }
    }
    if (nxt.k != cur.k) {
      // deterministic form: the tile goes to slot (chunk + offset) of the partial buffer (common.h, TsWgradPlan)
      float *dwk = (part ? part + ((int64_t)blockIdx.x + cur.k) * CA * CB : dW + (int64_t)cur.k * CA * CB) +
                   (int64_t)ci0 * CB + co0;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float *dst = &dwk[(int64_t)((wr * MI + mi) * 16 + 4 * g + q) * CB + (wc * NI + ni) * 16 + r16];
            if (part)
              *dst = acc[mi][ni][q];
            else
              atomicAdd(dst, acc[mi][ni][q]);
          }
          acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    cur = nxt;
  }
}

template <int TM, int TN>
static int launch_wgrad_s(const float *A, int CA, const float *B, int CB, const int2 *nbmaps, const int *nboffs, int K,
                          int col_a, int64_t n_pairs, float *dW, hipStream_t stream) {
  const int tiles = (CA / TM) * (CB / TN);
  const TsWgradPlan plan = ts_wgrad_plan(n_pairs, tiles, K, WS_PS, WS_MAXCHUNK);
  g_ts_wgrad_plan = plan;
  dim3 grid((unsigned)plan.n_chunks, tiles);
  const size_t lds = (size_t)3 * WS_PS * ((TM + 8) + (TN + 8)) * 2 + 2 * WS_MAXCHUNK * 4;
  wgrad_s_kernel<TM, TN><<<grid, 256, lds, stream>>>(A, CA, B, CB, nbmaps, nboffs, K, (int)n_pairs, col_a, plan.chunk, dW,
                                                     g_ts_wgrad_part);
  TS_CHECK_LAUNCH("conv_wgrad (split)");
  return TS_OK;
}

// called by ts_conv_wgrad for full tiles (CA % tm == 0, CB % tn == 0, 16-byte aligned rows, K <= 63); dW pre-zeroed
int ts_wgrad_split(const float *A, int CA, const float *B, int CB, const int2 *nbmaps, const int *nboffs, int K,
                   int col_a, int64_t n_pairs, float *dW, int tm, int tn, hipStream_t stream) {
#define TS_WS(TM, TN) launch_wgrad_s<TM, TN>(A, CA, B, CB, nbmaps, nboffs, K, col_a, n_pairs, dW, stream)
#define TS_WS_ROW(TM)                    \
  switch (tn) {                          \
    case 32: return TS_WS(TM, 32);       \
    case 64: return TS_WS(TM, 64);       \
    case 96: return TS_WS(TM, 96);       \
    default: return TS_WS(TM, 128);      \
  }
  switch (tm) {
    case 32: TS_WS_ROW(32)
    case 64: TS_WS_ROW(64)
    case 96: TS_WS_ROW(96)
    default: TS_WS_ROW(128)
  }
#undef TS_WS_ROW
#undef TS_WS
}
