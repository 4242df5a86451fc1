// Class-sorted implicit GEMM: pass 1 of a submanifold 3x3x3 convolution with the sums of several offsets kept in the MFMA
// accumulators, so that Z' has one row per (output row, offset group) instead of one per rulebook pair.
//
// The two-pass convolution (conv_pairs_s.hip) writes one Z row per pair - 6.5 per voxel on a LiDAR scan - and reads them all
// again in pass 2: 2 * P * C_out * 4 bytes through the fabric per launch pair, about as long as the products themselves take.
// An output-stationary kernel over SPATIAL tiles (conv_os.hip) loses: at 24 % fill every offset brings a different subset of a
// tile's rows, so it compacts lists per offset and restages W_k for a handful of rows.  Here the tiles are not spatial:
//   * the 27 offsets are cut into three groups of nine (k / 9: one z-plane of the kernel each);
//   * per group, the output rows that have at least one neighbour in it are sorted by their 9-bit neighbour mask (a radix
//     sort of 11-bit keys, once per batch and stride, on the staging stream) and cut into 128-row tiles - rows of one tile
//     have (nearly) the same neighbours: on the bench rulebook the (tile, offset) steps are 1.12x the pair GEMM's tiles;
//   * a workgroup owns one tile and walks the offsets of the tile's union mask; every step is a pair-GEMM tile (gather the
//     128 neighbour rows, zero where a row lacks this neighbour; split; stage; multiply with W_k) but the 128 x BN sums stay
//     in the accumulators and are stored ONCE: Z' has 1.98 N rows instead of 6.5 N, pass 2 (`ts_conv_gather_sum` with the
//     3 x N position table of the plan) adds at most three rows per output instead of 6.5.
// Tiles are launched longest first, in list order (no XCD remap: every XCD gets tiles of every length).
// Measured on the bench rulebook (profiles/r03_class_gemm_probe.txt): stride-1 96 -> 96 forward 222 us against 298 us for
// pair GEMM + gather-sum (1.34x), input gradient 203 / 285 (1.40x), 128 -> 96 1.31x / 1.46x, stride-2 96 -> 96 1.14x / 1.24x,
// stride-4 128 -> 128 0.9-1.0x (stays on the two passes).  Arithmetic: the six-product bf16 split of conv_pairs_s.hip, fp32
// accumulation; within a group the offsets are added in the accumulator (ascending k), the three group sums in pass 2
// (ascending group): deterministic, 1e-6-close to the two-pass result, not bit-identical to it (another summation order).
// Reference semantics: convolution_forward_cuda / convolution_backward_cuda (backend/convolution/convolution_cuda.cu:101-278),
// the input-gradient product uses W_{26-k}^T on the same plan (the submanifold map is its own transpose with the offsets
// reversed).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CG_BM 128
#define CG_BK 32
#define CG_AP (CG_BK + 8)
#define CG_GROUPS 3
#define CG_GK 9              // offsets per group

// ---------------------------------------------------------------------------------------------------------- plan
static inline int64_t cg_npad(int64_t n) { return (n + CG_BM - 1) / CG_BM * CG_BM; }

extern "C" int64_t ts_conv_class_rows(int64_t n) { return n < 0 ? 0 : CG_GROUPS * cg_npad(n); }

// key = group * 512 + (511 - mask): inside a group, rows with more / higher neighbour bits first, rows without a neighbour in
// the group (and the padding up to a multiple of 128) last; value = row (-1: padding)
__global__ __launch_bounds__(256) void class_keys_kernel(const int *__restrict__ nbr, int64_t n, int64_t npad,
                                                        unsigned short *__restrict__ keys, int *__restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= CG_GROUPS * npad) return;
  const int g = (int)(i / npad);
  const int64_t j = i - (int64_t)g * npad;
  unsigned bits = 0;
  if (j < n) {
#pragma unroll
    for (int kl = 0; kl < CG_GK; ++kl) bits |= (unsigned)(nbr[(int64_t)(CG_GK * g + kl) * n + j] >= 0) << kl;
  }
  keys[i] = (unsigned short)(g * 512 + (511 - bits));
  vals[i] = j < n ? (int)j : -1;
}

// one 128-thread workgroup per tile of the sorted list: the neighbour table in sorted order (src), the position of every row in
// the list (pos), the union mask of the tile
__global__ __launch_bounds__(CG_BM) void class_fill_kernel(const int *__restrict__ nbr, int64_t n, int64_t npad,
                                                          const unsigned short *__restrict__ keys,
                                                          const int *__restrict__ vals, int *__restrict__ src,
                                                          int *__restrict__ pos, int *__restrict__ tile_mask) {
  __shared__ unsigned wmask[CG_BM / 64];
  const int64_t m_pad = CG_GROUPS * npad;
  const int64_t i = (int64_t)blockIdx.x * CG_BM + threadIdx.x;
  const int key = keys[i];
  const int g = key >> 9;
  const unsigned bits = 511u - (unsigned)(key & 511);
  const int j = vals[i];
  const bool live = j >= 0 && bits != 0;
#pragma unroll
  for (int kl = 0; kl < CG_GK; ++kl)
    src[(int64_t)kl * m_pad + i] = (live && ((bits >> kl) & 1)) ? nbr[(int64_t)(CG_GK * g + kl) * n + j] : -1;
  if (j >= 0) pos[(int64_t)g * n + j] = live ? (int)i : -1;
  unsigned m = live ? bits : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m |= __shfl_xor(m, d, 64);
  if ((threadIdx.x & 63) == 0) wmask[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) tile_mask[blockIdx.x] = (int)(wmask[0] | wmask[1]);
}

// live tiles, longest (most offsets) first: tile_info[t] = (group + 4 * tile, union mask); one workgroup (<= 70k tiles at the
// 3e6-voxel cap).  The order inside a length class follows the arrival of the LDS atomics - every tile owns its Z' rows, so
// the order of the list changes the schedule, never a result.
__global__ __launch_bounds__(1024) void class_tiles_kernel(const int *__restrict__ tile_mask, int n_all, int64_t npad,
                                                           int2 *__restrict__ tile_info, int *__restrict__ n_tiles) {
  __shared__ int cnt[CG_GK + 1], base[CG_GK + 1];
  if (threadIdx.x <= CG_GK) cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < n_all; t += 1024) {
    const int m = tile_mask[t];
    if (m) atomicAdd(&cnt[__builtin_popcount(m)], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0, steps = 0;
    for (int c = CG_GK; c >= 1; --c) {
      base[c] = run;
      run += cnt[c];
      steps += c * cnt[c];
    }
    n_tiles[0] = run;
    n_tiles[1] = steps;        // (tile, offset) steps of the plan: 128 * steps row-products against the rulebook's pairs
  }
  __syncthreads();
  const int tiles_per_group = (int)(npad / CG_BM);
  for (int t = threadIdx.x; t < n_all; t += 1024) {
    const int m = tile_mask[t];
    if (m) {
      const int at = atomicAdd(&base[__builtin_popcount(m)], 1);
      tile_info[at] = make_int2(t / tiles_per_group + 4 * t, m);
    }
  }
}

extern "C" size_t ts_conv_class_plan_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  const int64_t m = CG_GROUPS * cg_npad(n);
  size_t sort_bytes = 0;
  rocprim::radix_sort_pairs(nullptr, sort_bytes, (unsigned short *)nullptr, (unsigned short *)nullptr, (int *)nullptr,
                            (int *)nullptr, (size_t)m, 0, 11, (hipStream_t) nullptr);
  return ts_align_up((size_t)m * 2, 256) * 2 + ts_align_up((size_t)m * 4, 256) * 2 + ts_align_up((size_t)(m / CG_BM) * 4, 256) +
         ts_align_up(sort_bytes, 256) + 256;
}

// nbr [27][n] (ts_build_kmap of a submanifold 3x3x3 map: in == out) -> src [9][m_pad], tile_info [m_pad / 128] (x, y) pairs,
// n_tiles [2] = (live tiles, their (tile, offset) steps), pos [3][n];  m_pad = ts_conv_class_rows(n)
extern "C" int ts_conv_class_plan(const int32_t *nbr, int64_t n, int32_t K, int32_t *src, int32_t *tile_info, int32_t *n_tiles,
                                  int32_t *pos, void *ws, size_t ws_bytes, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(K == CG_GROUPS * CG_GK, TS_ERR_UNSUPPORTED, "ts_conv_class_plan: 27 offsets (3x3x3) only");
  TS_REQUIRE(n > 0 && n < (1LL << 28), TS_ERR_INVALID_ARGUMENT, "ts_conv_class_plan: bad n");
  TS_REQUIRE(nbr && src && tile_info && n_tiles && pos && ws, TS_ERR_INVALID_ARGUMENT, "ts_conv_class_plan: null pointer");
  TS_REQUIRE(ws_bytes >= ts_conv_class_plan_workspace_bytes(n), TS_ERR_WORKSPACE_TOO_SMALL, "ts_conv_class_plan: workspace too small");
  const int64_t npad = cg_npad(n), m = CG_GROUPS * npad;
  char *p = (char *)ws;
  unsigned short *k0 = (unsigned short *)p;
  p += ts_align_up((size_t)m * 2, 256);
  unsigned short *k1 = (unsigned short *)p;
  p += ts_align_up((size_t)m * 2, 256);
  int *v0 = (int *)p;
  p += ts_align_up((size_t)m * 4, 256);
  int *v1 = (int *)p;
  p += ts_align_up((size_t)m * 4, 256);
  int *tmask = (int *)p;
  p += ts_align_up((size_t)(m / CG_BM) * 4, 256);
  size_t sort_bytes = ws_bytes - (size_t)(p - (char *)ws);
  class_keys_kernel<<<(unsigned)ts_cdiv(m, 256), 256, 0, stream>>>(nbr, n, npad, k0, v0);
  TS_CHECK_LAUNCH("ts_conv_class_plan/keys");
  TS_CHECK_HIP(rocprim::radix_sort_pairs(p, sort_bytes, k0, k1, v0, v1, (size_t)m, 0, 11, stream), "ts_conv_class_plan/sort");
  class_fill_kernel<<<(unsigned)(m / CG_BM), CG_BM, 0, stream>>>(nbr, n, npad, k1, v1, src, pos, tmask);
  TS_CHECK_LAUNCH("ts_conv_class_plan/fill");
  class_tiles_kernel<<<1, 1024, 0, stream>>>(tmask, (int)(m / CG_BM), npad, (int2 *)tile_info, n_tiles);
  TS_CHECK_LAUNCH("ts_conv_class_plan/tiles");
  return TS_OK;
}

// ---------------------------------------------------------------------------------------------------------- kernel
__device__ __forceinline__ unsigned cg_pk_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf2));
}
__device__ __forceinline__ void cg_split8(const f32x4 &v0, const f32x4 &v1, u32x4 &h, u32x4 &m, u32x4 &l) {
  const float a[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = a[2 * i], x1 = a[2 * i + 1];
    const unsigned hh = cg_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(hh << 16), r1 = x1 - __uint_as_float(hh & 0xffff0000u);
    const unsigned mm = cg_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(mm << 16), s1 = r1 - __uint_as_float(mm & 0xffff0000u);
    h[i] = hh;
    m[i] = mm;
    l[i] = cg_pk_bf16(s0, s1);
  }
}
__device__ __forceinline__ bf8 cg_frag_tr(const unsigned short *img, int pitch, int r0, int c0, int tq, int tp) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + (r0 + tq) * pitch + c0 + 4 * tp));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + (r0 + 4 + tq) * pitch + c0 + 4 * tp));
  return __builtin_bit_cast(bf8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// six of the nine partial products, smallest first (conv_pairs_s.hip)
#define CG_MMA(ACC, A, B)                                                               \
  do {                                                                                  \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[2], (B)[0], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[0], (B)[2], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[1], (B)[1], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[1], (B)[0], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[0], (B)[1], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[0], (B)[0], ACC, 0, 0, 0);        \
  } while (0)

// X [n, R] fp32 rows; W [K, R, O_total] (WT = false: forward) or [K, O_total, R] (WT = true: the input gradient multiplies with
// the transposed slice of the MIRRORED offset); Zp [m_pad, O_total].  grid (upper bound of the tile count, O_total / BN).
template <int BN, int WR, bool WT>
__global__ __launch_bounds__(256, 2) void class_gemm_kernel(const float *__restrict__ X, int R, const float *__restrict__ W,
                                                           int O_total, const int *__restrict__ src, int64_t m_pad,
                                                           const int2 *__restrict__ tile_info,
                                                           const int *__restrict__ n_tiles, int K, float *__restrict__ Zp) {
  constexpr int BM = CG_BM;
  constexpr int WC = 4 / WR;
  constexpr int MI = (BM / 16) / WR;
  constexpr int NI = (BN / 16) / WC;
  constexpr int BP = BN + 8;
  constexpr int A_PLANE = BM * CG_AP;
  constexpr int B_PLANE = WT ? BN * CG_AP : CG_BK * BP;
  constexpr int A_IT = BM * (CG_BK / 8) / 256;
  constexpr int B_CHUNKS = BN * (CG_BK / 8);
  constexpr int B_IT = (B_CHUNKS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem_cg[];
  unsigned short *Ap = smem_cg;                                // 3 planes [128][CG_AP]
  unsigned short *Bp = Ap + 3 * A_PLANE;                       // 3 planes

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int tq = r16 >> 2, tp = lane & 3;
  const int wr = wave / WC, wc = wave % WC;
  const int o0 = blockIdx.y * BN;
  const int tile = (int)blockIdx.x;             // launch order = list order (longest first): every XCD gets tiles of every length
  if (tile >= *n_tiles) return;
  const int2 info = tile_info[tile];
  const int grp = __builtin_amdgcn_readfirstlane(info.x) & 3;
  int mask = __builtin_amdgcn_readfirstlane(info.y);
  const int64_t row0 = (int64_t)(__builtin_amdgcn_readfirstlane(info.x) >> 2) * BM;

  const int arow0 = tid >> 2, acol = (tid & 3) << 3;
  int boff[B_IT], bdst[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int e = min(tid + it * 256, B_CHUNKS - 1);
    if (WT) {
      const int col = e >> 2, c8 = (e & 3) << 3;
      boff[it] = col * R + c8;
      bdst[it] = col * CG_AP + c8;
    } else {
      constexpr int q8 = BN >> 3;
      const int kk = e / q8, c8 = (e - kk * q8) << 3;
      boff[it] = kk * O_total + c8;
      bdst[it] = kk * BP + c8;
    }
  }

  // acc: the product of the offset being walked (what a Z row of the two-pass form holds); tot: the offsets finished so far,
  // added with ordinary fp32 adds.  Keeping ONE accumulator across the offsets would chain 3.7x more terms through the matrix
  // pipe's internal adder and doubles the rounding noise of the result (measured against float64).
  f32x4 acc[MI][NI], tot[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = tot[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float *aptr[A_IT];
  bool alive[A_IT];
  const float *wk = W;
  int nsrc[A_IT];                       // input rows of the NEXT offset of the mask, fetched an offset ahead
  auto fetch = [&](int kl) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) nsrc[it] = src[(int64_t)kl * m_pad + row0 + arow0 + 64 * it];
  };
  auto bind = [&](int kl) {             // operand pointers of group offset kl: k = 9 grp + kl
    const int k = CG_GK * grp + kl;
    const int kw = WT ? (K - 1 - k) : k;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      alive[it] = nsrc[it] >= 0;
      aptr[it] = X + (int64_t)max(nsrc[it], 0) * R + acol;
    }
    wk = WT ? W + ((int64_t)kw * O_total + o0) * R : W + (int64_t)kw * R * O_total + o0;
  };
  f32x4 ra[A_IT][2], rb[B_IT][2];
  bool rlive[A_IT];
  auto load_regs = [&](int c0) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      if (alive[it]) {                   // a row without this neighbour loads nothing (11 % of the (row, offset) slots of a tile)
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
      u32x4 h, m, l;
      cg_split8(rlive[it] ? ra[it][0] : zero, rlive[it] ? ra[it][1] : zero, h, m, l);
      unsigned short *dst = Ap + rr * CG_AP + acol;
      *(u32x4 *)dst = h;
      *(u32x4 *)(dst + A_PLANE) = m;
      *(u32x4 *)(dst + 2 * A_PLANE) = l;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      if (B_IT * 256 == B_CHUNKS || tid + it * 256 < B_CHUNKS) {
        u32x4 h, m, l;
        cg_split8(rb[it][0], rb[it][1], h, m, l);
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
        a[mi][p] = *(const bf8 *)&Ap[p * A_PLANE + ((wr * MI + mi) * 16 + r16) * CG_AP + 8 * g];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      bf8 b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        if (WT)
          b[p] = *(const bf8 *)&Bp[p * B_PLANE + ((wc * NI + ni) * 16 + r16) * CG_AP + 8 * g];
        else
          b[p] = cg_frag_tr(Bp + p * B_PLANE, BP, 8 * g, (wc * NI + ni) * 16, tq, tp);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) CG_MMA(acc[mi][ni], a[mi], b);
    }
  };

  // (a listed tile has a non-empty mask)
  fetch(__builtin_ctz(mask));
  bind(__builtin_ctz(mask));
  mask &= mask - 1;
  load_regs(0);
  bool first = true;
  while (true) {
    if (mask) fetch(__builtin_ctz(mask));  // rows of the next offset: needed only when this offset's last slice is staged
    for (int c0 = 0; c0 < R; c0 += CG_BK) {
      if (!first) __syncthreads();         // the previous slice's fragments have been read
      first = false;
      store_lds();
      __syncthreads();
      if (c0 + CG_BK < R) {
        load_regs(c0 + CG_BK);
      } else if (mask) {
        bind(__builtin_ctz(mask));         // first slice of the next offset, in flight behind this slice's MFMAs
        load_regs(0);
      }
      mma();
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        tot[mi][ni] += acc[mi][ni];
        acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
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

template <int BN, int WR, bool WT>
static int launch_class(const float *X, int R, const float *W, int O_total, const int *src, int64_t m_pad, const int2 *tile_info,
                        const int *n_tiles, int K, float *Zp, hipStream_t stream) {
  const size_t lds = (size_t)3 * (CG_BM * CG_AP + (WT ? BN * CG_AP : CG_BK * (BN + 8))) * 2;
  dim3 grid((unsigned)(m_pad / CG_BM), (unsigned)(O_total / BN));
  class_gemm_kernel<BN, WR, WT><<<grid, 256, lds, stream>>>(X, R, W, O_total, src, m_pad, tile_info, n_tiles, K, Zp);
  TS_CHECK_LAUNCH("ts_conv_class_gemm");
  return TS_OK;
}

static int cg_tile_columns(int c_out) { return c_out % 128 == 0 ? 128 : c_out % 96 == 0 ? 96 : c_out % 64 == 0 ? 64 : c_out % 32 == 0 ? 32 : 0; }

extern "C" int32_t ts_conv_class_supported(int32_t c_red, int32_t c_out) {
  return c_red > 0 && c_red % CG_BK == 0 && cg_tile_columns(c_out) != 0;
}

// zp [m_pad, c_out] = pass 1 of the convolution on the plan (wt = 0: feat = input rows, kernel [27, c_red, c_out]; wt = 1: input
// gradient, feat = output-gradient rows [n, c_red], kernel [27, c_out, c_red] as stored, result columns = c_out = C_in)
extern "C" int ts_conv_class_gemm(const float *feat, int32_t c_red, const float *kernel, int32_t K, int32_t c_out,
                                  const int32_t *src, int64_t m_pad, const int32_t *tile_info, const int32_t *n_tiles, int32_t wt,
                                  float *zp, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(K == CG_GROUPS * CG_GK, TS_ERR_UNSUPPORTED, "ts_conv_class_gemm: 27 offsets (3x3x3) only");
  TS_REQUIRE(ts_conv_class_supported(c_red, c_out), TS_ERR_UNSUPPORTED,
             "ts_conv_class_gemm: C_red must be a multiple of 32 and C_out of 32 (got %d, %d)", c_red, c_out);
  TS_REQUIRE(feat && kernel && src && tile_info && n_tiles && zp, TS_ERR_INVALID_ARGUMENT, "ts_conv_class_gemm: null pointer");
  TS_REQUIRE(m_pad > 0 && m_pad % (CG_GROUPS * CG_BM) == 0 && m_pad < (1LL << 31), TS_ERR_INVALID_ARGUMENT,
             "ts_conv_class_gemm: m_pad must be ts_conv_class_rows(n)");
  TS_REQUIRE(((((uintptr_t)feat) | ((uintptr_t)kernel) | ((uintptr_t)zp)) & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_conv_class_gemm: pointers must be 16-byte aligned");
  const int2 *ti = (const int2 *)tile_info;
#define CG_GO(BN, WR)                                                                                             \
  (wt ? launch_class<BN, WR, true>(feat, c_red, kernel, c_out, src, m_pad, ti, n_tiles, K, zp, stream)            \
      : launch_class<BN, WR, false>(feat, c_red, kernel, c_out, src, m_pad, ti, n_tiles, K, zp, stream))
  switch (cg_tile_columns(c_out)) {
    case 128: return CG_GO(128, 2);
    case 96: return CG_GO(96, 2);
    case 64: return CG_GO(64, 2);
    default: return CG_GO(32, 4);
  }
#undef CG_GO
}

// ------------------------------------------------------------------------------------------- half storage
// The same walk for IEEE-half rows (torch.autocast; conv_pairs_h.hip): one v_mfma_f32_16x16x32_f16 per product, fp32 sums, the
// gathered operand and the weight slice double-buffered in LDS (one barrier per slice), Z' rows rounded to half ONCE per
// (row, group) - the two-pass form rounds every pair's Z row.
typedef _Float16 ch8 __attribute__((ext_vector_type(8)));
typedef __fp16 chv4t __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ ch8 cgh_frag_tr(const _Float16 *img, int pitch, int r0, int c0, int tq, int tp) {
  typedef chv4t __attribute__((address_space(3))) * lds_hv4;
  const chv4t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_hv4)(img + (r0 + tq) * pitch + c0 + 4 * tp));
  const chv4t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_hv4)(img + (r0 + 4 + tq) * pitch + c0 + 4 * tp));
  ch8 v;
  v[0] = (_Float16)lo[0]; v[1] = (_Float16)lo[1]; v[2] = (_Float16)lo[2]; v[3] = (_Float16)lo[3];
  v[4] = (_Float16)hi[0]; v[5] = (_Float16)hi[1]; v[6] = (_Float16)hi[2]; v[7] = (_Float16)hi[3];
  return v;
}

template <int BN, int WR, bool WT>
__global__ __launch_bounds__(256, 2) void class_gemm_h_kernel(const _Float16 *__restrict__ X, int R,
                                                             const _Float16 *__restrict__ W, int O_total,
                                                             const int *__restrict__ src, int64_t m_pad,
                                                             const int2 *__restrict__ tile_info,
                                                             const int *__restrict__ n_tiles, int K,
                                                             _Float16 *__restrict__ Zp) {
  constexpr int BM = CG_BM;
  constexpr int WC = 4 / WR;
  constexpr int MI = (BM / 16) / WR;
  constexpr int NI = (BN / 16) / WC;
  constexpr int A_HALVES = BM * CG_AP;
  constexpr int BP = BN + 8;
  constexpr int B_HALVES = WT ? BN * CG_AP : CG_BK * BP;
  constexpr int A_IT = BM * (CG_BK / 8) / 256;
  constexpr int B_IT = (BN * (CG_BK / 8) + 255) / 256;
  constexpr int ZP = BN + 8;
  extern __shared__ __attribute__((aligned(16))) _Float16 smem_cgh[];
  _Float16 *Abuf = smem_cgh;
  _Float16 *Bbuf = Abuf + 2 * A_HALVES;

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

  const int arow0 = tid >> 2, acol = (tid & 3) << 3;
  int boff[B_IT], bdst[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int e = min(tid + it * 256, BN * 4 - 1);
    if (WT) {
      const int col = e >> 2, c8 = (e & 3) << 3;
      boff[it] = col * R + c8;
      bdst[it] = col * CG_AP + c8;
    } else {
      constexpr int q8 = BN >> 3;
      const int kk = e / q8, c8 = (e - kk * q8) << 3;
      boff[it] = kk * O_total + c8;
      bdst[it] = kk * BP + c8;
    }
  }
  f32x4 acc[MI][NI], tot[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = tot[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const _Float16 *aptr[A_IT];
  bool alive[A_IT];
  const _Float16 *wk = W;
  int nsrc[A_IT];
  auto fetch = [&](int kl) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) nsrc[it] = src[(int64_t)kl * m_pad + row0 + arow0 + 64 * it];
  };
  auto bind = [&](int kl) {
    const int k = CG_GK * grp + kl;
    const int kw = WT ? (K - 1 - k) : k;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      alive[it] = nsrc[it] >= 0;
      aptr[it] = X + (int64_t)max(nsrc[it], 0) * R + acol;
    }
    wk = WT ? W + ((int64_t)kw * O_total + o0) * R : W + (int64_t)kw * R * O_total + o0;
  };
  ch8 ra[A_IT], rb[B_IT];
  auto load_regs = [&](int c0) {
    const ch8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      ra[it] = zero;
      if (alive[it]) ra[it] = *(const ch8 *)(aptr[it] + c0);
    }
    const _Float16 *wb = WT ? wk + c0 : wk + (int64_t)c0 * O_total;
#pragma unroll
    for (int it = 0; it < B_IT; ++it) rb[it] = *(const ch8 *)(wb + boff[it]);
  };
  auto store_lds = [&](_Float16 *At, _Float16 *Bt) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) *(ch8 *)&At[(arow0 + 64 * it) * CG_AP + acol] = ra[it];
#pragma unroll
    for (int it = 0; it < B_IT; ++it)
      if (B_IT * 256 == BN * 4 || tid + it * 256 < BN * 4) *(ch8 *)&Bt[bdst[it]] = rb[it];
  };
  auto mma = [&](const _Float16 *At, const _Float16 *Bt) {
    ch8 a[MI], b[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[mi] = *(const ch8 *)&At[((wr * MI + mi) * 16 + r16) * CG_AP + 8 * g];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      b[ni] = WT ? *(const ch8 *)&Bt[((wc * NI + ni) * 16 + r16) * CG_AP + 8 * g]
                 : cgh_frag_tr(Bt, BP, 8 * g, (wc * NI + ni) * 16, tq, tp);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
  };

  fetch(__builtin_ctz(mask));
  bind(__builtin_ctz(mask));
  mask &= mask - 1;
  load_regs(0);
  int t = 0;
  while (true) {
    if (mask) fetch(__builtin_ctz(mask));
    for (int c0 = 0; c0 < R; c0 += CG_BK, ++t) {
      _Float16 *At = Abuf + (t & 1) * A_HALVES, *Bt = Bbuf + (t & 1) * B_HALVES;
      store_lds(At, Bt);
      __syncthreads();
      if (c0 + CG_BK < R) {
        load_regs(c0 + CG_BK);
      } else if (mask) {
        bind(__builtin_ctz(mask));
        load_regs(0);
      }
      mma(At, Bt);
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        tot[mi][ni] += acc[mi][ni];
        acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    if (!mask) break;
    mask &= mask - 1;
  }
  // Z' tile: sums -> half -> LDS image [row][col] -> 16-byte chunks of whole rows
  __syncthreads();
  _Float16 *Zt = smem_cgh;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        Zt[((wr * MI + mi) * 16 + 4 * g + q) * ZP + (wc * NI + ni) * 16 + r16] = (_Float16)tot[mi][ni][q];
  __syncthreads();
  constexpr int CH = BN / 8;
  for (int e = tid; e < BM * CH; e += 256) {
    const int row = e / CH, ch = e - row * CH;
    *(ch8 *)(Zp + (row0 + row) * O_total + o0 + ch * 8) = *(const ch8 *)&Zt[row * ZP + ch * 8];
  }
}

template <int BN, int WR, bool WT>
static int launch_class_h(const _Float16 *X, int R, const _Float16 *W, int O_total, const int *src, int64_t m_pad,
                          const int2 *tile_info, const int *n_tiles, int K, _Float16 *Zp, hipStream_t stream) {
  const size_t stage = (size_t)2 * (CG_BM * CG_AP + (WT ? BN * CG_AP : CG_BK * (BN + 8))) * 2;
  const size_t ztile = (size_t)CG_BM * (BN + 8) * 2;
  dim3 grid((unsigned)(m_pad / CG_BM), (unsigned)(O_total / BN));
  class_gemm_h_kernel<BN, WR, WT><<<grid, 256, std::max(stage, ztile), stream>>>(X, R, W, O_total, src, m_pad, tile_info, n_tiles, K, Zp);
  TS_CHECK_LAUNCH("ts_conv_class_gemm_f16");
  return TS_OK;
}

// half rows: feat / zp IEEE half, w = the half weight [27, C_in, C_out] as stored (wt = 0: forward, read in place through the
// transposing LDS load; wt = 1: input gradient, rows of the mirrored offset's slice = result columns)
extern "C" int ts_conv_class_gemm_f16(const void *feat, int32_t c_red, const void *w, int32_t K, int32_t c_out,
                                      const int32_t *src, int64_t m_pad, const int32_t *tile_info, const int32_t *n_tiles,
                                      int32_t wt, void *zp, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(K == CG_GROUPS * CG_GK, TS_ERR_UNSUPPORTED, "ts_conv_class_gemm_f16: 27 offsets (3x3x3) only");
  TS_REQUIRE(ts_conv_class_supported(c_red, c_out), TS_ERR_UNSUPPORTED,
             "ts_conv_class_gemm_f16: channel counts must be multiples of 32 (got %d, %d)", c_red, c_out);
  TS_REQUIRE(feat && w && src && tile_info && n_tiles && zp, TS_ERR_INVALID_ARGUMENT, "ts_conv_class_gemm_f16: null pointer");
  TS_REQUIRE(m_pad > 0 && m_pad % (CG_GROUPS * CG_BM) == 0 && m_pad < (1LL << 31), TS_ERR_INVALID_ARGUMENT,
             "ts_conv_class_gemm_f16: m_pad must be ts_conv_class_rows(n)");
  TS_REQUIRE(((((uintptr_t)feat) | ((uintptr_t)w) | ((uintptr_t)zp)) & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_conv_class_gemm_f16: pointers must be 16-byte aligned");
  const int2 *ti = (const int2 *)tile_info;
  const _Float16 *x = (const _Float16 *)feat, *wh = (const _Float16 *)w;
#define CGH_GO(BN, WR)                                                                                                  \
  (wt ? launch_class_h<BN, WR, true>(x, c_red, wh, c_out, src, m_pad, ti, n_tiles, K, (_Float16 *)zp, stream)          \
      : launch_class_h<BN, WR, false>(x, c_red, wh, c_out, src, m_pad, ti, n_tiles, K, (_Float16 *)zp, stream))
  switch (cg_tile_columns(c_out)) {
    case 128: return CGH_GO(128, 2);
    case 96: return CGH_GO(96, 2);
    case 64: return CGH_GO(64, 2);
    default: return CGH_GO(32, 4);
  }
#undef CGH_GO
}

// One-shot and per thread (like ts_conv_planes_hint): the NEXT ts_conv_block_forward / ts_conv_block_backward of this thread
// may run its forward product / input gradient on this plan when the block is a submanifold 3x3x3 convolution over `n` rows.
thread_local TsClassHint g_ts_class_hint = {nullptr, nullptr, nullptr, nullptr, 0, 0};
extern "C" void ts_conv_class_hint(const int32_t *src, const int32_t *tile_info, const int32_t *n_tiles, const int32_t *pos,
                                   int64_t n, int64_t z_rows) {
  g_ts_class_hint = TsClassHint{src, tile_info, n_tiles, pos, n, z_rows};
}
