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
// PROBE: lane 0 of every workgroup leaves shader-clock stamps of its phases in `stamps` (ts_debug_phase_stamps;
// tools/phase_probe.py) - a diagnostic instantiation, the product launches PROBE = false.
#define TS_STAMP(i)                                                                     \
  do {                                                                                  \
    if (PROBE && tid == 0) stamps[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
template <int BM, int BN, int WR, bool WT, bool XCD, bool PROBE = false>
__global__ __launch_bounds__(256, BM == 128 ? 2 : 4) void pair_gemm_s_kernel(const float *__restrict__ X, int R,
                                                          const float *__restrict__ W, int O_total,
                                                          const int2 *__restrict__ nbmaps,
                                                          const int *__restrict__ nboffs, int K, int gcol,
                                                          float *__restrict__ Z,
                                                          unsigned long long *__restrict__ stamps = nullptr) {
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
  if (PROBE && tid == 0) {
#pragma unroll
    for (int i = 1; i < 16; ++i) stamps[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + i] = 0;
    stamps[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 13] = __builtin_amdgcn_s_memrealtime();
  }
  TS_STAMP(0);

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
  TS_STAMP(1);                                  // tile mapped (offset table arrived)

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

  if (PROBE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TS_STAMP(2);                                // pair indices arrived
  }
  load_regs(0);
  if (PROBE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TS_STAMP(3);                                // first slice arrived
  }
  for (int c0 = 0; c0 < R; c0 += PS_BK) {
    if (c0) __syncthreads();          // the previous slice's fragments have been read
    store_lds();
    __syncthreads();
    if (c0 == 0) TS_STAMP(4);                   // first slice split and staged
    if (c0 + PS_BK < R) load_regs(c0 + PS_BK);
    mma();
    if (c0 == 0) TS_STAMP(5);                   // first MFMA block issued
  }
  TS_STAMP(6);                                  // all slices issued
  if (PROBE) {
    asm volatile("s_nop 0" ::: "memory");
    float sink = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) sink += acc[mi][ni][0];
    if (sink == 12345.678f) stamps[1] = 0;      // forces the accumulators: stamp 7 = MFMAs retired
    TS_STAMP(7);
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
  TS_STAMP(8);                                  // Z stores issued
  if (PROBE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TS_STAMP(9);                                // Z stores acknowledged
    if (tid == 0) {
      stamps[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 10] = __builtin_amdgcn_s_memrealtime();
      stamps[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 11] =
          ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |      // XCC_ID
          (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);                          // HW_ID (wave, simd, cu, sh, se)
      stamps[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 12] = (unsigned long long)np;
    }
  }
}

// ---------------------------------------------------------------------------------- pre-split weights
// Planes of a convolution weight W [K, Ci, Co] (n = K * Ci * Co): three bf16 arrays of n elements, h | m | l, each in
// the layout of W itself.  The forward product reads slices along Ci through transposing LDS loads, the input gradient
// reads the same planes as [K, O = Ci, R = Co] rows (pair_gemm_d_kernel<.., WT>).  Written once per optimizer step
// (ts_conv_split_planes[_batch]: a streaming pass, one workgroup row per weight, 16 weights per launch) and handed to
// the next convolution call of the calling thread with ts_conv_planes_hint.  (A second set of planes of W^T, so that
// the input gradient could run as a forward-layout product with two LDS buffers, made that kernel 4 % faster and the
// split twice as expensive - more than it gained.)
struct TsPlaneJobs {
  TsPlaneJob job[16];
};
__global__ __launch_bounds__(256) void split_planes_kernel(TsPlaneJobs jobs) {
  const TsPlaneJob jb = jobs.job[blockIdx.y];
  const float *__restrict__ w = jb.w;
  unsigned short *__restrict__ planes = (unsigned short *)jb.planes;
  const int64_t n = (int64_t)jb.K * jb.c_in * jb.c_out;
  const int64_t n8 = n >> 3;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
    u32x4 h, m, l;
    split8(*(const f32x4 *)(w + 8 * i), *(const f32x4 *)(w + 8 * i + 4), h, m, l);
    *(u32x4 *)(planes + 8 * i) = h;
    *(u32x4 *)(planes + n + 8 * i) = m;
    *(u32x4 *)(planes + 2 * n + 8 * i) = l;
  }
}

extern "C" int ts_conv_split_planes_batch(const TsPlaneJob *jobs, int32_t n_jobs, ts_stream_t stream) {
  TS_REQUIRE(n_jobs >= 0 && (jobs || n_jobs == 0), TS_ERR_INVALID_ARGUMENT, "ts_conv_split_planes_batch: bad arguments");
  for (int32_t j0 = 0; j0 < n_jobs; j0 += 16) {
    TsPlaneJobs chunk;
    const int cnt = std::min(16, n_jobs - j0);
    int64_t big = 0;
    for (int j = 0; j < cnt; ++j) {
      const TsPlaneJob &jb = jobs[j0 + j];
      TS_REQUIRE(jb.w && jb.planes && jb.K > 0 && jb.c_in > 0 && jb.c_out > 0 && ((int64_t)jb.c_in * jb.c_out) % 8 == 0 &&
                     ((((uintptr_t)jb.w) | ((uintptr_t)jb.planes)) & 15) == 0,
                 TS_ERR_INVALID_ARGUMENT, "ts_conv_split_planes_batch: job %d: null / misaligned pointer or C_in * C_out not a multiple of 8",
                 j0 + j);
      chunk.job[j] = jb;
      big = std::max<int64_t>(big, (int64_t)jb.K * jb.c_in * jb.c_out);
    }
    dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(ts_cdiv(big / 8, 256), 128)), (unsigned)cnt);
    split_planes_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(chunk);
    TS_CHECK_LAUNCH("ts_conv_split_planes_batch");
  }
  return TS_OK;
}

extern "C" int ts_conv_split_planes(const float *w, int32_t K, int32_t c_in, int32_t c_out, void *planes, ts_stream_t stream) {
  TsPlaneJob jb = {w, planes, K, c_in, c_out};
  return ts_conv_split_planes_batch(&jb, 1, stream);
}

// one-shot hint of the calling thread: the next ts_conv_pair_gemm / ts_conv_block_forward / ts_conv_block_backward whose
// weight is `w` may read `planes` (layout above) instead of splitting the weight slice in every workgroup.  Any of these
// calls clears it, whether it used it or not; a hint never outlives the call it was made for.
thread_local TsPlanesHint g_ts_planes_hint = {nullptr, nullptr, 0, 0, 0};
extern "C" void ts_conv_planes_hint(const float *w, const void *planes, int32_t K, int32_t c_in, int32_t c_out) {
  g_ts_planes_hint = {w, (const unsigned short *)planes, K, c_in, c_out};
}

static unsigned long long *g_ts_phase_stamps = nullptr;
static int64_t g_ts_phase_cap = 0;
// ---------------------------------------------------------------------------------- pair GEMM, gathered rows direct
// The gathered operand never touches LDS: lane (r16, g) of a wave loads the 8 consecutive floats of ITS MFMA fragment
// (row r16 of a 16-row block, k = 8g .. 8g+7 of the slice) straight from the gathered row, splits them in registers and
// feeds the matrix pipe; a wave owns 32 of the tile's 128 rows and all BN columns.  The weight slice arrives pre-split
// (three bf16 planes in the fp32 weight's [K, R, O_total] layout, split_planes_kernel) and is the only thing staged in
// LDS, double buffered: ONE barrier per slice, no split work for the weights, 40 - 52 KB of LDS and <= 168 VGPRs, i.e.
// three workgroups per CU whose waves run their gather -> split -> MFMA chains independently between the barriers.
// WT = false: planes of W [K, R, O_total], slices along its rows (forward product), transposing LDS reads, two LDS
// buffers.  WT = true: the same planes read as [K, O_total, R] (input gradient: one image row per output column, the
// slice contiguous along it), plain 16-byte fragment reads; that image takes 3 x BN x 80 bytes, so it is single buffered
// (a second barrier per slice) to keep three workgroups on a CU.
template <int BN, bool WT, bool PROBE>
__global__ __launch_bounds__(256, 3) void pair_gemm_d_kernel(const float *__restrict__ X, int R,
                                                             const unsigned short *__restrict__ Wp, int64_t wplane,
                                                             int O_total, const int2 *__restrict__ nbmaps,
                                                             const int *__restrict__ nboffs, int K, int gcol,
                                                             float *__restrict__ Z,
                                                             unsigned long long *__restrict__ stamps) {
  constexpr int BM = 128, MI = 2, NI = BN / 16;
  constexpr int BP = BN + 8;
  constexpr int B_PLANE = WT ? BN * PS_AP : PS_BK * BP;
  constexpr int B_BUF = 3 * B_PLANE;
  constexpr int B_CHUNKS = BN * (PS_BK / 8);
  constexpr int B_IT = (B_CHUNKS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem_d[];
  unsigned short *Bp = smem_d;                                 // (WT ? 1 : 2) buffers x 3 planes

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int tq = r16 >> 2, tp = lane & 3;
  const int o0 = blockIdx.y * BN;
  if (PROBE && tid == 0) {
#pragma unroll
    for (int i = 1; i < 16; ++i) stamps[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + i] = 0;
    stamps[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 13] = __builtin_amdgcn_s_memrealtime();
  }
  TS_STAMP(0);

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
  TS_STAMP(1);

  const float *aptr[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int2 pr = nbmaps[p0 + min(wave * 32 + mi * 16 + r16, np - 1)];   // rows past the tile repeat its last row; never stored
    aptr[mi] = X + (int64_t)(gcol ? pr.y : pr.x) * R + 8 * g;
  }
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
  const unsigned short *wkp = WT ? Wp + ((int64_t)k * O_total + o0) * R : Wp + (int64_t)k * R * O_total + o0;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 ra[MI][2];
  u32x4 rp[B_IT][3];
  bf8 a[MI][3];
  auto load_a = [&](int c0) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      ra[mi][0] = *(const f32x4 *)(aptr[mi] + c0);
      ra[mi][1] = *(const f32x4 *)(aptr[mi] + c0 + 4);
    }
  };
  auto load_b = [&](int c0) {
    const unsigned short *wb = WT ? wkp + c0 : wkp + (int64_t)c0 * O_total;
#pragma unroll
    for (int it = 0; it < B_IT; ++it)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) rp[it][pl] = *(const u32x4 *)(wb + pl * wplane + boff[it]);
  };
  auto write_b = [&](int buf) {
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      if (B_IT * 256 == B_CHUNKS || tid + it * 256 < B_CHUNKS) {
        unsigned short *dst = Bp + buf * B_BUF + bdst[it];
        *(u32x4 *)dst = rp[it][0];
        *(u32x4 *)(dst + B_PLANE) = rp[it][1];
        *(u32x4 *)(dst + 2 * B_PLANE) = rp[it][2];
      }
    }
  };
  auto split_a = [&]() {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      u32x4 h, m, l;
      split8(ra[mi][0], ra[mi][1], h, m, l);
      a[mi][0] = __builtin_bit_cast(bf8, h);
      a[mi][1] = __builtin_bit_cast(bf8, m);
      a[mi][2] = __builtin_bit_cast(bf8, l);
    }
  };
  // (tried: weight fragment as the first matrix operand, so that a lane's four accumulator values are four consecutive
  // channels of one pair row and leave as one 16-byte store - 7 - 10 % slower than the four 4-byte stores, and the
  // matrix pipe does not round the two operand orders alike)
  auto mma = [&](int buf) {
    const unsigned short *bb = Bp + buf * B_BUF;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      bf8 b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        if (WT)
          b[p] = *(const bf8 *)&bb[p * B_PLANE + (ni * 16 + r16) * PS_AP + 8 * g];
        else
          b[p] = frag_tr(bb + p * B_PLANE, BP, 8 * g, ni * 16, tq, tp);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) TS_SPLIT_MMA(acc[mi][ni], a[mi], b);
    }
  };

  if (PROBE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TS_STAMP(2);
  }
  const int S = R / PS_BK;
  load_a(0);
  load_b(0);
  if (PROBE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TS_STAMP(3);
  }
  write_b(0);
  split_a();
  __syncthreads();
  TS_STAMP(4);
  // step s: the gathered fragments and the weights of slice s + 1 are issued, the MFMA block of slice s runs on the
  // split fragments in registers and on LDS buffer s & 1, then the weights of s + 1 go to the other buffer, the raw
  // fragments of s + 1 are split, and one barrier closes the step
  for (int s = 0; s < S; ++s) {
    const bool more = s + 1 < S;
    if (more) {
      load_a((s + 1) * PS_BK);
      load_b((s + 1) * PS_BK);
    }
    mma(WT ? 0 : (s & 1));
    if (s == 0) TS_STAMP(5);
    if (more) {
      if (WT) __syncthreads();        // single buffer: every wave has read the slice before it is overwritten
      write_b(WT ? 0 : ((s + 1) & 1));
      split_a();
      __syncthreads();
    }
  }
  TS_STAMP(6);
  if (PROBE) {
    float sink = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) sink += acc[mi][ni][0];
    if (sink == 12345.678f) stamps[1] = 0;
    TS_STAMP(7);
  }
  float *zt = Z + (int64_t)p0 * O_total + o0;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = wave * 32 + mi * 16 + 4 * g + q;
      if (np == BM || row < np) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) TS_ZSTORE(acc[mi][ni][q], &zt[(int64_t)row * O_total + ni * 16 + r16]);
      }
    }
  TS_STAMP(8);
  if (PROBE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TS_STAMP(9);
    if (tid == 0) {
      stamps[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 10] = __builtin_amdgcn_s_memrealtime();
      stamps[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 11] =
          ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
          (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
      stamps[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 12] = (unsigned long long)np;
    }
  }
}
template <int BN, bool WT>
static int launch_pair_gemm_d(const float *X, int R, const unsigned short *planes, int64_t wplane, int O_total,
                              const int2 *nbmaps, const int *nboffs, int K, int64_t P, int gcol, float *Z,
                              unsigned long long *stamps, hipStream_t stream) {
  const size_t lds = WT ? (size_t)3 * BN * PS_AP * 2 : (size_t)2 * 3 * PS_BK * (BN + 8) * 2;
  dim3 grid((unsigned)((ts_cdiv(P, 128) + K + 7) / 8 * 8), (unsigned)(O_total / BN));
  if (stamps)
    pair_gemm_d_kernel<BN, WT, true><<<grid, 256, lds, stream>>>(X, R, planes, wplane, O_total, nbmaps, nboffs, K, gcol, Z, stamps);
  else
    pair_gemm_d_kernel<BN, WT, false><<<grid, 256, lds, stream>>>(X, R, planes, wplane, O_total, nbmaps, nboffs, K, gcol, Z, nullptr);
  TS_CHECK_LAUNCH("conv_pair_gemm (direct rows)");
  return TS_OK;
}


// diagnostic: the next full-tile 128-row split pair GEMMs write 16 stamps per workgroup into `stamps` (device memory,
// `capacity` workgroups); nullptr switches back to the product kernels.
extern "C" void ts_debug_phase_stamps(unsigned long long *stamps, int64_t capacity) {
  g_ts_phase_stamps = stamps;
  g_ts_phase_cap = capacity;
}

// The direct-rows kernel on the planes of the weight (`plane_n` elements apart), wt = 0: forward product, 1: input
// gradient.  Taken by ts_conv_pair_gemm when the caller left a hint (ts_conv_planes_hint) and ts_pair_gemm_direct_ok says
// the kernel wins: 128-column tiles (3 instead of 2 workgroups per CU: 12 - 18 % on the stride-4 / 8 / 16 layers); on
// 96-column tiles it ties with the LDS-staged kernel (ts_set_conv_impl(11) takes it there too, (13) never takes it).
bool ts_pair_gemm_direct_ok(int bn) {
  return g_ts_conv_impl == 11 ? (bn == 96 || bn == 128) : (g_ts_conv_impl == 0 && bn == 128);
}
int ts_pair_gemm_direct(const float *X, int R, const unsigned short *planes, int64_t plane_n, int O_total,
                        const int2 *nbmaps, const int *nboffs, int K, int64_t P, int gcol, float *Z, int bn, int wt,
                        hipStream_t stream) {
  if (g_ts_phase_stamps)
    TS_CHECK_HIP(hipMemsetAsync(g_ts_phase_stamps, 0, (size_t)g_ts_phase_cap * 16 * 8, stream), "phase stamps memset");
#define TS_PD(BN)                                                                                                           \
  (wt ? launch_pair_gemm_d<BN, true>(X, R, planes, plane_n, O_total, nbmaps, nboffs, K, P, gcol, Z, g_ts_phase_stamps, stream) \
      : launch_pair_gemm_d<BN, false>(X, R, planes, plane_n, O_total, nbmaps, nboffs, K, P, gcol, Z, g_ts_phase_stamps, stream))
  return bn == 96 ? TS_PD(96) : TS_PD(128);
#undef TS_PD
}

template <int BM, int BN, int WR, bool WT>
static int launch_pair_gemm_s(const float *X, int R, const float *W, int O_total, const int2 *nbmaps, const int *nboffs,
                              int K, int64_t P, int gcol, float *Z, hipStream_t stream) {
  const size_t lds = (size_t)3 * (BM * PS_AP + (WT ? BN * PS_AP : PS_BK * (BN + 8))) * 2;
  dim3 grid((unsigned)((ts_cdiv(P, BM) + K + 7) / 8 * 8), (unsigned)(O_total / BN));
  if (g_ts_phase_stamps && BM == 128 && BN >= 96) {
    constexpr bool PB = BM == 128 && BN >= 96;
    TS_REQUIRE((int64_t)grid.x * grid.y <= g_ts_phase_cap, TS_ERR_INVALID_ARGUMENT, "phase stamps: %lld workgroups, room for %lld",
               (long long)grid.x * grid.y, (long long)g_ts_phase_cap);
    TS_CHECK_HIP(hipMemsetAsync(g_ts_phase_stamps, 0, (size_t)g_ts_phase_cap * 16 * 8, stream), "phase stamps memset");
    pair_gemm_s_kernel<BM, BN, WR, WT, true, PB><<<grid, 256, lds, stream>>>(X, R, W, O_total, nbmaps, nboffs, K, gcol,
                                                                              Z, g_ts_phase_stamps);
  } else if (g_ts_conv_impl == 8)    // tiles in launch order (A/B of the XCD remap)
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
