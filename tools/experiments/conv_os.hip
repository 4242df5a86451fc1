// Output-stationary (Z-free) sparse convolution for the wide, shallow submanifold layers (up3 / up4 of MinkUNet: 3x3x3,
// 96 / 128 channels, strides 1 and 2): one launch replaces pair GEMM + gather-sum, no per-pair product matrix Z in HBM.
//
//   out[j, :] = sum_k  X[nbr[k, j], :] @ W_k          (forward;  nbr = the reference's `results` table, conv.py:160-166)
//   gx [i, :] = sum_k dY[nbr[K-1-k, i], :] @ W_k^T     (input gradient of a submanifold odd kernel: pair (i, j, k) of the
//                                                        rulebook <=> pair (j, i, K-1-k), so the same table serves both)
// replacing the gather -> GEMM -> scatter-add loop of convolution_cuda.cu:101-164 / :167-258.
//
// A workgroup (512 threads, 8 waves) owns TM consecutive output rows whose fp32 sums live in LDS for the whole kernel.
// It walks the K offsets in ascending weight order; per offset
//   * one wave compacts the tile's live rows (ballot + prefix popcount) into a list (source row, tile row) in LDS,
//     padded to 16-row MFMA blocks - 24 % of the (row, offset) slots are live, so the list, not the tile, is multiplied;
//   * all waves copy the pre-split bf16 planes of W_k (taseg_amd/planes.py, split_planes_kernel) global -> registers ->
//     LDS, the loads issued one offset ahead;
//   * work items = (16-row block of the list) x (half of the output columns), dealt round-robin to the waves: a lane
//     loads the 8 consecutive floats of ITS MFMA fragment straight from the gathered row (as pair_gemm_d_kernel), splits
//     them exactly into three bf16 numbers and issues the same six v_mfma_f32_16x16x32_bf16 per 32-deep slice in the same
//     order as the two-pass kernels; the first item's rows of the NEXT offset are in flight while this offset multiplies;
//   * the block's products are added into the LDS tile.  Within one offset a tile row receives at most one product row
//     (a voxel has one neighbour per offset) and offsets are separated by barriers: plain read-add-write, no atomics,
//     the additions of a row happen in ascending offset order - the order of gather_list_kernel, so the result equals
//     pair GEMM + gather-sum BIT FOR BIT (tests/test_gpu_conv_os.py).
// Two barriers per offset.  LDS: TM x (O + 4) floats + 3 planes of W_k + two lists (140 - 160 KB: one workgroup per CU).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ unsigned os_pk_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf2));   // v_cvt_pk_bf16_f32 (RNE)
}

// 8 floats -> three planes of 8 bf16 (x = h + m + l exactly; conv_pairs_s.hip)
__device__ __forceinline__ void os_split8(const f32x4 &v0, const f32x4 &v1, u32x4 &h, u32x4 &m, u32x4 &l) {
  const float a[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = a[2 * i], x1 = a[2 * i + 1];
    const unsigned hh = os_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(hh << 16), r1 = x1 - __uint_as_float(hh & 0xffff0000u);
    const unsigned mm = os_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(mm << 16), s1 = r1 - __uint_as_float(mm & 0xffff0000u);
    h[i] = hh;
    m[i] = mm;
    l[i] = os_pk_bf16(s0, s1);
  }
}

__device__ __forceinline__ bf8 os_frag_tr(const unsigned short *img, int pitch, int r0, int c0, int tq, int tp) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + (r0 + tq) * pitch + c0 + 4 * tp));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + (r0 + 4 + tq) * pitch + c0 + 4 * tp));
  return __builtin_bit_cast(bf8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// six of the nine partial products, smallest first - the order of TS_SPLIT_MMA in conv_pairs_s.hip
#define OS_SPLIT_MMA(ACC, A, B)                                                         \
  do {                                                                                  \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[2], (B)[0], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[0], (B)[2], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[1], (B)[1], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[1], (B)[0], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[0], (B)[1], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A)[0], (B)[0], ACC, 0, 0, 0);        \
  } while (0)

}  // namespace

// X [*, R] fp32 rows; planes: 3 x (K * R * O) bf16 in the weight's own [K, C_in, C_out] layout, `plane_n` elements apart;
// WT = false: C_in = R, C_out = O (forward);  WT = true: C_in = O, C_out = R (input gradient, W_k read transposed).
// nbr [K, n_rows]; krev: table row K-1-k goes with weight k.  out [n_rows, O] (+ addend [n_rows, O] when given).
template <int R, int O, bool WT, int TM>
__global__ __launch_bounds__(512, 1) void conv_os_kernel(const float *__restrict__ X, const unsigned short *__restrict__ Wp,
                                                         int64_t plane_n, const int *__restrict__ nbr, int64_t n_rows,
                                                         int K, int krev, float *__restrict__ out,
                                                         const float *__restrict__ addend, TsWgradReduce side, int dbg) {
  constexpr int S = R / 32;                         // 32-deep slices
  constexpr int OP = O + 4;                         // pitch of the fp32 tile
  constexpr int IMG_ROWS = WT ? O : R;              // rows of W_k as stored: C_in
  constexpr int IMG_COLS = WT ? R : O;              // C_out
  constexpr int BP = IMG_COLS + 8;                  // bf16 pitch of a plane image (16-byte multiple)
  constexpr int B_PLANE = IMG_ROWS * BP;
  constexpr int CH_ROW = IMG_COLS / 8;              // 16-byte chunks per image row
  constexpr int CH_PLANE = IMG_ROWS * CH_ROW;
  constexpr int CH_ALL = 3 * CH_PLANE;
  constexpr int W_IT = (CH_ALL + 511) / 512;
  constexpr int HALF = O / 2;                       // output columns of one work item
  constexpr int NI = HALF / 16;
  constexpr int PJ = TM / 64;                       // position registers of the compacting wave
  static_assert(R % 32 == 0 && O % 32 == 0 && TM % 64 == 0, "shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char os_smem[];
  float *outT = (float *)os_smem;                                         // [TM][OP]
  unsigned short *Wl = (unsigned short *)(outT + TM * OP);                // 3 x [IMG_ROWS][BP]
  int *srcl = (int *)(Wl + 3 * B_PLANE);                                  // [2][TM]
  int *rowl = srcl + 2 * TM;                                              // [2][TM]
  int *cnt = rowl + 2 * TM;                                               // [2]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int tq = r16 >> 2, tp = lane & 3;
  const int64_t j0 = (int64_t)blockIdx.x * TM;

  // the ordered weight-gradient sum of the block calls rides on this launch like on the gather-sum launch it replaces
  {
    const int64_t e = (int64_t)blockIdx.x * 512 + tid, step = (int64_t)gridDim.x * 512;
    for (int64_t i = e; i < (int64_t)side.K * side.cacb4; i += step) ts_wgrad_reduce_one(side, i);
  }
  for (int i = tid; i < TM * OP / 4; i += 512) ((f32x4 *)outT)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- helpers ------------------------------------------------------------------------------------------------
  int posreg[PJ];
  auto load_pos = [&](int kw) {          // by the wave that will compact offset kw
    const int kk = krev ? K - 1 - kw : kw;
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
      const int64_t row = j0 + 64 * j + lane;
      posreg[j] = row < n_rows ? nbr[(int64_t)kk * n_rows + row] : -1;
    }
  };
  auto build_list = [&](int buf) {       // one wave: live rows of the tile, ascending, padded to a multiple of 16
    int base = 0;
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
      const int p = posreg[j];
      const bool live = p >= 0;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(live);
      const int idx = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (live) {
        srcl[buf * TM + idx] = p;
        rowl[buf * TM + idx] = 64 * j + lane;
      }
      base += __builtin_popcountll(m);
    }
    const int padded = (base + 15) & ~15;
    if (base + lane < padded) {          // < 16 lanes
      srcl[buf * TM + base + lane] = 0;  // any valid row: its products are never added
      rowl[buf * TM + base + lane] = -1;
    }
    if (lane == 0) cnt[buf] = base;
  };
  u32x4 wreg[W_IT];
  auto load_w = [&](int kw) {
    if ((dbg & 1) && kw > 0) return;
    const unsigned short *src = Wp + (int64_t)kw * (R * O);
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int e = tid + it * 512;
      if (W_IT * 512 == CH_ALL || e < CH_ALL) {
        const int pl = e / CH_PLANE, c = e - pl * CH_PLANE;
        wreg[it] = *(const u32x4 *)(src + (int64_t)pl * plane_n + c * 8);
      }
    }
  };
  auto write_w = [&]() {
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int e = tid + it * 512;
      if (W_IT * 512 == CH_ALL || e < CH_ALL) {
        const int pl = e / CH_PLANE, c = e - pl * CH_PLANE;
        const int row = c / CH_ROW, c8 = (c - row * CH_ROW) * 8;
        *(u32x4 *)(Wl + pl * B_PLANE + row * BP + c8) = wreg[it];
      }
    }
  };
  struct AFrag {
    f32x4 v[S][2];
  };
  auto load_a = [&](AFrag &a, int buf, int blk) {
    if (dbg & 8) return;
    const int src = srcl[buf * TM + blk * 16 + r16];
    const float *p = X + (int64_t)src * R + 8 * g;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      a.v[s][0] = *(const f32x4 *)(p + 32 * s);
      a.v[s][1] = *(const f32x4 *)(p + 32 * s + 4);
    }
  };
  auto compute = [&](const AFrag &a, int buf, int blk, int half) {
    const bool live = rowl[buf * TM + blk * 16 + r16] >= 0;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[ni] = zero;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      u32x4 h, m, l;
      os_split8(live ? a.v[s][0] : zero, live ? a.v[s][1] : zero, h, m, l);
      bf8 af[3] = {__builtin_bit_cast(bf8, h), __builtin_bit_cast(bf8, m), __builtin_bit_cast(bf8, l)};
      if (dbg & 2) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int col = half * HALF + ni * 16;
        bf8 b[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          if (WT)
            b[p] = *(const bf8 *)&Wl[p * B_PLANE + (col + r16) * BP + 32 * s + 8 * g];
          else
            b[p] = os_frag_tr(Wl + p * B_PLANE, BP, 32 * s + 8 * g, col, tq, tp);
        }
        OS_SPLIT_MMA(acc[ni], af, b);
      }
    }
    // products of list rows 4g .. 4g+3 (columns col + r16) into the tile; a tile row gets one product row per offset and
    // the rows of a block are distinct: all reads first, then the adds, then the writes
    if (dbg & 4) return;
    const int4 rows = *(const int4 *)&rowl[buf * TM + blk * 16 + 4 * g];
    const int rr[4] = {rows.x, rows.y, rows.z, rows.w};
    float t[4][NI];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float *src = outT + max(rr[q], 0) * OP + half * HALF + r16;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) t[q][ni] = src[ni * 16];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (rr[q] >= 0) {
        float *dst = outT + rr[q] * OP + half * HALF + r16;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) dst[ni * 16] = t[q][ni] + acc[ni][q];
      }
    }
  };

  // ---- prologue: list 0, W_0 in registers, positions of offset 1 ------------------------------------------------
  if (wave == 0) {
    load_pos(0);
    build_list(0);
  }
  if (wave == 1 && K > 1) load_pos(1);
  load_w(0);
  __syncthreads();                       // list 0, zeroed tile
  AFrag anext = {};                      // first item of the coming offset
  {
    const int items0 = 2 * ((cnt[0] + 15) >> 4);
    if (wave < items0) load_a(anext, 0, wave >> 1);
  }

  for (int kw = 0; kw < K; ++kw) {
    const int buf = kw & 1;
    if (kw) __syncthreads();             // B1: every wave is done with offset kw - 1 (W image, list buf ^ 1 free)
    write_w();
    if (kw + 1 < K && wave == ((kw + 1) & 7)) build_list(buf ^ 1);
    __syncthreads();                     // B2: W_kw and list kw + 1 visible
    if (kw + 1 < K) load_w(kw + 1);
    if (kw + 2 < K && wave == ((kw + 2) & 7)) load_pos(kw + 2);
    const int items = 2 * ((cnt[buf] + 15) >> 4);
    AFrag cur = anext;
    if (kw + 1 < K) {
      const int items1 = 2 * ((cnt[buf ^ 1] + 15) >> 4);
      if (wave < items1) load_a(anext, buf ^ 1, wave >> 1);
    }
    for (int it = wave; it < items; it += 8) {
      AFrag nxt;
      const bool more = it + 8 < items;
      if (more) load_a(nxt, buf, (it + 8) >> 1);
      compute(cur, buf, it >> 1, it & 1);
      if (more) cur = nxt;
    }
  }
  __syncthreads();
  // ---- tile -> global rows (16-byte stores) ---------------------------------------------------------------------
  constexpr int V = O / 4;
  for (int i = tid; i < TM * V; i += 512) {
    const int r = i / V, c4 = (i - r * V) * 4;
    const int64_t j = j0 + r;
    if (j < n_rows) {
      f32x4 v = *(const f32x4 *)(outT + r * OP + c4);
      if (addend) {
        const f32x4 a = *(const f32x4 *)(addend + j * O + c4);
        v += a;
      }
      *(f32x4 *)(out + j * O + c4) = v;
    }
  }
}

static int g_ts_conv_os_dbg = 0;
// diagnostic (tools/os_probe.py --ablate): bit 0 no weight staging after the first offset, 1 no MFMAs, 2 no tile update,
// 3 no gathered rows - wrong results, timing only
extern "C" void ts_debug_conv_os(int32_t bits) { g_ts_conv_os_dbg = bits; }

template <int R, int O, bool WT, int TM>
static int launch_conv_os(const float *X, const unsigned short *planes, int64_t plane_n, const int *nbr, int64_t n_rows,
                          int K, int krev, float *out, const float *addend, const TsWgradReduce &side, hipStream_t stream) {
  constexpr int IMG_ROWS = WT ? O : R, IMG_COLS = WT ? R : O;
  constexpr size_t lds = (size_t)TM * (O + 4) * 4 + (size_t)3 * IMG_ROWS * (IMG_COLS + 8) * 2 + (size_t)4 * TM * 4 + 16;
  static_assert(lds <= 163840, "LDS budget");
  static bool attr_set = false;
  auto kern = conv_os_kernel<R, O, WT, TM>;
  if (!attr_set) {
    TS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                 "conv_os: LDS attribute");
    attr_set = true;
  }
  const unsigned grid = (unsigned)ts_cdiv(n_rows, TM);
  kern<<<grid, 512, lds, stream>>>(X, planes, plane_n, nbr, n_rows, K, krev, out, addend, side, g_ts_conv_os_dbg);
  TS_CHECK_LAUNCH("conv_os");
  return TS_OK;
}

// shapes the output-stationary kernel is built for (reduction width, output width)
bool ts_conv_os_shape_ok(int c_red, int c_out) {
  return (c_red == 96 && c_out == 96) || (c_red == 128 && c_out == 96) || (c_red == 96 && c_out == 128);
}

int ts_conv_os_ex(const float *feat, int32_t c_red, const void *planes, int64_t plane_n, int32_t K, int32_t c_out,
                  const int32_t *nbr, int64_t n_rows, int32_t wt, float *out, const float *addend,
                  const TsWgradReduce *side_job, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TsWgradReduce side = {};
  if (side_job) side = *side_job;
  TS_REQUIRE(K > 0 && n_rows >= 0 && ts_conv_os_shape_ok(c_red, c_out), TS_ERR_INVALID_ARGUMENT,
             "ts_conv_os: unsupported shape %d -> %d (K %d)", c_red, c_out, K);
  if (n_rows == 0) return TS_OK;
  TS_REQUIRE(feat && planes && nbr && out, TS_ERR_INVALID_ARGUMENT, "ts_conv_os: null pointer");
  TS_REQUIRE(((((uintptr_t)feat) | ((uintptr_t)planes) | ((uintptr_t)out) | ((uintptr_t)addend)) & 15) == 0 &&
                 plane_n >= (int64_t)K * c_red * c_out && plane_n % 8 == 0,
             TS_ERR_INVALID_ARGUMENT, "ts_conv_os: misaligned pointer or short planes");
  const unsigned short *pl = (const unsigned short *)planes;
  const int krev = wt ? 1 : 0;
#define TS_OS(R, O, WT, TM) launch_conv_os<R, O, WT, TM>(feat, pl, plane_n, nbr, n_rows, K, krev, out, addend, side, stream)
  if (c_red == 96 && c_out == 96) return wt ? TS_OS(96, 96, true, 192) : TS_OS(96, 96, false, 192);
  if (c_red == 128 && c_out == 96) return wt ? TS_OS(128, 96, true, 192) : TS_OS(128, 96, false, 192);
  return wt ? TS_OS(96, 128, true, 128) : TS_OS(96, 128, false, 128);
#undef TS_OS
}

// out[j] = sum_k feat[nbr[wt ? K-1-k : k, j]] @ (wt ? W_k^T : W_k) on the pre-split planes of W [K, C_in, C_out]
// (ts_conv_split_planes); wt = 1 is the input gradient of a submanifold odd-kernel convolution (see the header comment).
extern "C" int ts_conv_os(const float *feat, int32_t c_red, const void *planes, int64_t plane_n, int32_t K, int32_t c_out,
                          const int32_t *nbr, int64_t n_rows, int32_t wt, float *out, const float *addend,
                          ts_stream_t stream) {
  return ts_conv_os_ex(feat, c_red, planes, plane_n, K, c_out, nbr, n_rows, wt, out, addend, nullptr, stream);
}

extern "C" int32_t ts_conv_os_supported(int32_t c_red, int32_t c_out) { return ts_conv_os_shape_ok(c_red, c_out) ? 1 : 0; }
