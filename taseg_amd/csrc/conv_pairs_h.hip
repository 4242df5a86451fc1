// fp16-storage / fp32-accumulate form of the two-pass sparse convolution (the reference's default training mode is
// AMP: TS/torchsparse/nn/functional/conv.py:19 casts features and kernel to half, convolution_cuda.cu runs the
// per-offset GEMMs in half and accumulates its scatter in half; here every accumulation is fp32).
//
//   ts_cast_weights_f16      W f32 [K, Ci, Co] -> W16 [K, Ci, Co] and W16T [K, Co, Ci] (one pass; rows contiguous in
//                            the reduction dimension for forward (W16T) and dgrad (W16))
//   pair_gemm_h_kernel       Z[p, :] = X[g_p, :] @ W_k  with v_mfma_f32_16x16x32_f16 (16x the f32 MFMA rate: the
//                            kernel is bound by the gathered-row and Z traffic, half of the fp32 path's)
//   gather_sum_h_kernel      Y[j, :] = sum_k Z[pos[k, j], :], fp32 accumulation, half in / half out
// Tiling = pair_gemm_fast_kernel (128 pairs x BN columns per workgroup, tiles cut per offset, 32-deep slices,
// double-buffered LDS, one register stage).  Lane l of the MFMA holds A[row l & 15][k = 8 (l >> 4) + j] and
// B[k = 8 (l >> 4) + j][col l & 15], j = 0..7: with both operands stored row-major in the reduction index every
// fragment is one 16-byte LDS read.  The Z tile goes through LDS once more so that rows leave as 16-byte chunks.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define PH_BM 128
#define PH_BK 32            // halves per slice (64 bytes of every gathered row)
#define PH_AP (PH_BK + 8)   // LDS pitch in halves: 80 bytes = 20 dwords (4 mod 8)

__global__ __launch_bounds__(256) void cast_weights_f16_kernel(const float *__restrict__ w, int K, int ci, int co,
                                                               _Float16 *__restrict__ w16, _Float16 *__restrict__ w16t) {
  // one thread per (k, ci, co) element, co fastest: coalesced read and W16 write; the transposed write is strided
  // (weights are <= 2.6 M elements per layer and L2 resident)
  const int64_t total = (int64_t)K * ci * co;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; e < total; e += step) {
    const _Float16 h = (_Float16)w[e];
    if (w16) w16[e] = h;
    if (w16t) {
      const int64_t k = e / ((int64_t)ci * co);
      const int r = (int)(e - k * ci * co);
      const int i = r / co, o = r - i * co;
      w16t[(k * co + o) * ci + i] = h;
    }
  }
}

extern "C" int ts_cast_weights_f16(const float *w, int32_t K, int32_t c_in, int32_t c_out, void *w16, void *w16t,
                                   ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(K > 0 && c_in > 0 && c_out > 0, TS_ERR_INVALID_ARGUMENT, "ts_cast_weights_f16: bad sizes");
  TS_REQUIRE(w && (w16 || w16t), TS_ERR_INVALID_ARGUMENT, "ts_cast_weights_f16: null pointer");
  const int64_t total = (int64_t)K * c_in * c_out;
  const unsigned grid = (unsigned)std::min<int64_t>(ts_cdiv(total, 256), 1 << 16);
  cast_weights_f16_kernel<<<grid, 256, 0, stream>>>(w, K, c_in, c_out, (_Float16 *)w16, (_Float16 *)w16t);
  TS_CHECK_LAUNCH("ts_cast_weights_f16");
  return TS_OK;
}

// Half copies of many weights in one launch (16 per launch, like ts_conv_split_planes_batch): job.planes = W16 [K, Ci, Co],
// the layout both half pair GEMMs read.  taseg_amd/planes.py keeps these copies next to the parameters and refreshes the
// stale ones once per optimizer step; ts_conv_block_forward then skips its own cast (ts_conv_planes_hint(w, w16, ..)).
struct TsCastJobs {
  TsPlaneJob job[16];
};
__global__ __launch_bounds__(256) void cast_weights_f16_batch_kernel(TsCastJobs jobs) {
  const TsPlaneJob jb = jobs.job[blockIdx.y];
  const float *__restrict__ w = jb.w;
  _Float16 *__restrict__ w16 = (_Float16 *)jb.planes;
  const int64_t n8 = ((int64_t)jb.K * jb.c_in * jb.c_out) >> 3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const f32x4 a = *(const f32x4 *)(w + 8 * i), b = *(const f32x4 *)(w + 8 * i + 4);
    h8 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      o[q] = (_Float16)a[q];
      o[4 + q] = (_Float16)b[q];
    }
    *(h8 *)(w16 + 8 * i) = o;
  }
}

extern "C" int ts_cast_weights_f16_batch(const TsPlaneJob *jobs, int32_t n_jobs, ts_stream_t stream) {
  TS_REQUIRE(n_jobs >= 0 && (jobs || n_jobs == 0), TS_ERR_INVALID_ARGUMENT, "ts_cast_weights_f16_batch: bad arguments");
  for (int32_t j0 = 0; j0 < n_jobs; j0 += 16) {
    TsCastJobs chunk;
    const int cnt = std::min(16, n_jobs - j0);
    int64_t big = 0;
    for (int j = 0; j < cnt; ++j) {
      const TsPlaneJob &jb = jobs[j0 + j];
      TS_REQUIRE(jb.w && jb.planes && jb.K > 0 && jb.c_in > 0 && jb.c_out > 0 && ((int64_t)jb.c_in * jb.c_out) % 8 == 0 &&
                     ((((uintptr_t)jb.w) | ((uintptr_t)jb.planes)) & 15) == 0,
                 TS_ERR_INVALID_ARGUMENT,
                 "ts_cast_weights_f16_batch: job %d: null / misaligned pointer or C_in * C_out not a multiple of 8", j0 + j);
      chunk.job[j] = jb;
      big = std::max<int64_t>(big, (int64_t)jb.K * jb.c_in * jb.c_out);
    }
    dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(ts_cdiv(big / 8, 256), 128)), (unsigned)cnt);
    cast_weights_f16_batch_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(chunk);
    TS_CHECK_LAUNCH("ts_cast_weights_f16_batch");
  }
  return TS_OK;
}

// fragment of a [k][col] image: 8 consecutive k rows of column c0 + (lane & 15) through the transposing LDS load
// (lane 4q + p of a 16-lane group addresses row q, columns 4p .. 4p+3; as in wgrad_h_kernel below)
typedef __fp16 hv4t __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ h8 ph_frag_tr(const _Float16 *img, int pitch, int r0, int c0, int tq, int tp) {
  typedef hv4t __attribute__((address_space(3))) * lds_hv4;
  const hv4t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_hv4)(img + (r0 + tq) * pitch + c0 + 4 * tp));
  const hv4t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_hv4)(img + (r0 + 4 + tq) * pitch + c0 + 4 * tp));
  h8 v;
  v[0] = (_Float16)lo[0]; v[1] = (_Float16)lo[1]; v[2] = (_Float16)lo[2]; v[3] = (_Float16)lo[3];
  v[4] = (_Float16)hi[0]; v[5] = (_Float16)hi[1]; v[6] = (_Float16)hi[2]; v[7] = (_Float16)hi[3];
  return v;
}

// X [*, R] half rows; Z [P, O_total] half.  WT = true: Wr [K, O_total, R] (row = output column, contiguous in the
// reduction index - the input gradient reads the [K, C_in, C_out] half weight this way); WT = false: Wr [K, R, O_total]
// (the forward pass reads the same [K, C_in, C_out] weight in place, fragments through ds_read_b64_tr_b16 - no
// transposed copy of the weight is made).
template <int BN, int WR, bool WT>
__global__ __launch_bounds__(256, 2) void pair_gemm_h_kernel(const _Float16 *__restrict__ X, int R,
                                                          const _Float16 *__restrict__ Wr, int O_total,
                                                          const int2 *__restrict__ nbmaps,
                                                          const int *__restrict__ nboffs, int K, int gcol,
                                                          _Float16 *__restrict__ Z) {
  constexpr int WC = 4 / WR;
  constexpr int MI = (PH_BM / 16) / WR;
  constexpr int NI = (BN / 16) / WC;
  constexpr int A_HALVES = PH_BM * PH_AP;
  constexpr int BP = BN + 8;                        // pitch of the [k][col] weight image (WT = false)
  constexpr int B_HALVES = WT ? BN * PH_AP : PH_BK * BP;
  constexpr int A_IT = PH_BM * (PH_BK / 8) / 256;   // 16-byte chunks per thread per A slice (2)
  constexpr int B_IT = (BN * (PH_BK / 8) + 255) / 256;
  constexpr int ZP = BN + 8;                        // pitch of the Z tile image (halves)
  extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
  _Float16 *Abuf = smem_h;
  _Float16 *Bbuf = Abuf + 2 * A_HALVES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int tq = r16 >> 2, tp = lane & 3;
  const int wr = wave / WC, wc = wave % WC;
  const int o0 = blockIdx.y * BN;

  // tile -> (offset k, first pair, rows), as in pair_gemm_fast_kernel
  const int offv = nboffs[min(lane, K)];
  const int offn = nboffs[min(lane + 1, K)];
  int incl = lane < K ? (offn - offv + PH_BM - 1) / PH_BM : 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  const int tile = blockIdx.x;
  if (tile >= __builtin_amdgcn_readlane(incl, 63)) return;
  const int k = __builtin_popcountll(__builtin_amdgcn_ballot_w64(incl <= tile));
  const int t_in_k = tile - (k ? __builtin_amdgcn_readlane(incl, max(k - 1, 0)) : 0);
  const int p0 = __builtin_amdgcn_readlane(offv, k) + t_in_k * PH_BM;
  const int np = min(PH_BM, __builtin_amdgcn_readlane(offv, k + 1) - p0);

  // A slots: chunk (tid & 3) of tile row (tid >> 2) + 64 it
  const int arow0 = tid >> 2, acol = (tid & 3) << 3;
  const _Float16 *aptr[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int2 pr = nbmaps[p0 + min(arow0 + 64 * it, np - 1)];
    aptr[it] = X + (int64_t)(gcol ? pr.y : pr.x) * R + acol;
  }
  // B slots: WT: chunk (e & 3) of weight row (e >> 2); natural layout: chunk (e % (BN / 8)) of reduction row e / (BN / 8)
  const _Float16 *wk = WT ? Wr + ((int64_t)k * O_total + o0) * R : Wr + (int64_t)k * R * O_total + o0;
  int boff[B_IT], bdst[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int e = min(tid + it * 256, BN * 4 - 1);
    if (WT) {
      const int col = e >> 2, c8 = (e & 3) << 3;
      boff[it] = col * R + c8;
      bdst[it] = col * PH_AP + c8;
    } else {
      constexpr int q8 = BN >> 3;
      const int kk = e / q8, c8 = (e - kk * q8) << 3;
      boff[it] = kk * O_total + c8;
      bdst[it] = kk * BP + c8;
    }
  }

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  h8 ra[A_IT], rb[B_IT];
  auto load_regs = [&](int c0) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) ra[it] = *(const h8 *)(aptr[it] + c0);
    const _Float16 *wb = WT ? wk + c0 : wk + (int64_t)c0 * O_total;
#pragma unroll
    for (int it = 0; it < B_IT; ++it) rb[it] = *(const h8 *)(wb + boff[it]);
  };
  auto store_lds = [&](_Float16 *At, _Float16 *Bt) {
    const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int rr = arow0 + 64 * it;
      *(h8 *)&At[rr * PH_AP + acol] = rr < np ? ra[it] : zero;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it)
      if (B_IT * 256 == BN * 4 || tid + it * 256 < BN * 4) *(h8 *)&Bt[bdst[it]] = rb[it];
  };
  auto mma = [&](const _Float16 *At, const _Float16 *Bt) {
    h8 a[MI], b[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[mi] = *(const h8 *)&At[((wr * MI + mi) * 16 + r16) * PH_AP + 8 * g];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      b[ni] = WT ? *(const h8 *)&Bt[((wc * NI + ni) * 16 + r16) * PH_AP + 8 * g]
                 : ph_frag_tr(Bt, BP, 8 * g, (wc * NI + ni) * 16, tq, tp);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
  };

  load_regs(0);
  int t = 0;
  for (int c0 = 0; c0 < R; c0 += PH_BK, ++t) {
    _Float16 *At = Abuf + (t & 1) * A_HALVES, *Bt = Bbuf + (t & 1) * B_HALVES;
    store_lds(At, Bt);
    __syncthreads();
    if (c0 + PH_BK < R) load_regs(c0 + PH_BK);
    mma(At, Bt);
  }
  // Z tile: accumulators -> half -> LDS image [row][col] -> 16-byte chunks of whole rows
  __syncthreads();
  _Float16 *Zt = smem_h;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        Zt[((wr * MI + mi) * 16 + 4 * g + q) * ZP + (wc * NI + ni) * 16 + r16] = (_Float16)acc[mi][ni][q];
  __syncthreads();
  constexpr int CH = BN / 8;   // 16-byte chunks per row
  for (int e = tid; e < PH_BM * CH; e += 256) {
    const int row = e / CH, ch = e - row * CH;
    if (row < np) *(h8 *)(Z + (int64_t)(p0 + row) * O_total + o0 + ch * 8) = *(const h8 *)&Zt[row * ZP + ch * 8];
  }
}

template <int BN, int WR, bool WT>
static int launch_pair_gemm_h(const _Float16 *X, int R, const _Float16 *Wr, int O_total, const int2 *nbmaps,
                              const int *nboffs, int K, int64_t P, int gcol, _Float16 *Z, hipStream_t stream) {
  const size_t stage = (size_t)2 * (PH_BM * PH_AP + (WT ? BN * PH_AP : PH_BK * (BN + 8))) * 2;
  const size_t ztile = (size_t)PH_BM * (BN + 8) * 2;
  const size_t lds = std::max(stage, ztile);
  dim3 grid((unsigned)(ts_cdiv(P, PH_BM) + K), (unsigned)(O_total / BN));
  pair_gemm_h_kernel<BN, WR, WT><<<grid, 256, lds, stream>>>(X, R, Wr, O_total, nbmaps, nboffs, K, gcol, Z);
  TS_CHECK_LAUNCH("conv_pair_gemm_f16");
  return TS_OK;
}

static int pair_gemm_f16_any(const void *feat, int64_t n_rows, int32_t c_red, const void *w_rows, int32_t K,
                             const int32_t *nbmaps, const int32_t *nboffs, int64_t n_pairs, int32_t gather_col, void *z,
                             int32_t c_out, bool natural, ts_stream_t stream_);

extern "C" int ts_conv_pair_gemm_f16(const void *feat, int64_t n_rows, int32_t c_red, const void *w_rows, int32_t K,
                                     const int32_t *nbmaps, const int32_t *nboffs, int64_t n_pairs, int32_t gather_col,
                                     void *z, int32_t c_out, ts_stream_t stream_) {
  return pair_gemm_f16_any(feat, n_rows, c_red, w_rows, K, nbmaps, nboffs, n_pairs, gather_col, z, c_out, false, stream_);
}

// the same product with the weight in its natural layout w [K, c_red, c_out] (what the forward pass of a convolution
// has: kernel [K, C_in, C_out]); no transposed half copy needed
extern "C" int ts_conv_pair_gemm_f16_nat(const void *feat, int64_t n_rows, int32_t c_red, const void *w, int32_t K,
                                         const int32_t *nbmaps, const int32_t *nboffs, int64_t n_pairs,
                                         int32_t gather_col, void *z, int32_t c_out, ts_stream_t stream_) {
  return pair_gemm_f16_any(feat, n_rows, c_red, w, K, nbmaps, nboffs, n_pairs, gather_col, z, c_out, true, stream_);
}

static int pair_gemm_f16_any(const void *feat, int64_t n_rows, int32_t c_red, const void *w_rows, int32_t K,
                             const int32_t *nbmaps, const int32_t *nboffs, int64_t n_pairs, int32_t gather_col, void *z,
                             int32_t c_out, bool natural, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n_rows >= 0 && c_red > 0 && c_out > 0 && K > 0 && K <= 63 && n_pairs >= 0 && n_pairs < (1LL << 31),
             TS_ERR_INVALID_ARGUMENT, "ts_conv_pair_gemm_f16: bad sizes");
  TS_REQUIRE(c_red % PH_BK == 0 && c_out % 32 == 0, TS_ERR_UNSUPPORTED,
             "ts_conv_pair_gemm_f16: channel counts must be multiples of 32");
  if (n_pairs == 0) return TS_OK;
  TS_REQUIRE(feat && w_rows && nbmaps && nboffs && z, TS_ERR_INVALID_ARGUMENT, "ts_conv_pair_gemm_f16: null pointer");
  TS_REQUIRE(((((uintptr_t)feat) | ((uintptr_t)w_rows) | ((uintptr_t)z)) & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_conv_pair_gemm_f16: pointers must be 16-byte aligned");
  const _Float16 *x = (const _Float16 *)feat, *w = (const _Float16 *)w_rows;
  const int2 *nm = (const int2 *)nbmaps;
  const int gc = gather_col ? 1 : 0;
#define TS_PH(BN, WR)                                                                                              \
  (natural ? launch_pair_gemm_h<BN, WR, false>(x, c_red, w, c_out, nm, nboffs, K, n_pairs, gc, (_Float16 *)z, stream) \
           : launch_pair_gemm_h<BN, WR, true>(x, c_red, w, c_out, nm, nboffs, K, n_pairs, gc, (_Float16 *)z, stream))
  if (c_out % 128 == 0) return TS_PH(128, 2);
  if (c_out % 96 == 0) return TS_PH(96, 2);
  if (c_out % 64 == 0) return TS_PH(64, 2);
  return TS_PH(32, 4);
#undef TS_PH
}

// ------------------------------------------------------------------------------------- pass 2, half storage
template <int KT>
__global__ __launch_bounds__(256) void gather_sum_h_kernel(const _Float16 *__restrict__ Z, int C,
                                                           const int *__restrict__ pos, int K, int64_t n_rows,
                                                           _Float16 *__restrict__ out, TsWgradReduce side,
                                                           const _Float16 *__restrict__ addend) {
  const int c8n = C >> 3;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = n_rows * c8n, step = (int64_t)gridDim.x * blockDim.x;
  // side job (common.h): the ordered sum of the weight-gradient partials of the launch before this one
  for (int64_t i = e; i < (int64_t)side.K * side.cacb4; i += step) ts_wgrad_reduce_one(side, i);
  for (; e < total; e += step) {
    const int64_t j = e / c8n;
    const int c8 = (int)(e - j * c8n) << 3;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (KT > 0) {
      int p[KT > 0 ? KT : 1];
#pragma unroll
      for (int k = 0; k < KT; ++k) p[k] = pos[(int64_t)k * n_rows + j];
      h8 v[KT > 0 ? KT : 1];
#pragma unroll
      for (int k = 0; k < KT; ++k)
        v[k] = p[k] >= 0 ? __builtin_nontemporal_load((const h8 *)(Z + (int64_t)p[k] * C + c8)) : (h8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += (float)v[k][i];
    } else {
      for (int k = 0; k < K; ++k) {
        const int p = pos[(int64_t)k * n_rows + j];
        if (p >= 0) {
          const h8 v = *(const h8 *)(Z + (int64_t)p * C + c8);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] += (float)v[i];
        }
      }
    }
    if (addend) {
      const h8 a = *(const h8 *)(addend + j * C + c8);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += (float)a[i];
    }
    h8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (_Float16)acc[i];
    *(h8 *)(out + j * C + c8) = o;
  }
}

// The list form of pass 2 (gather_list_kernel, conv_pairs.hip) for half rows: a workgroup owns 256 / (C / 8) whole rows,
// half-waves compact the live positions of a row into LDS (lane = offset, ballot), a lane walks its row's list with R
// independent 16-byte loads per round.  float32 sums in ascending offset order, one rounding: the bits of
// gather_sum_h_kernel.  K <= 32, C >= 64.
template <int R>
__global__ __launch_bounds__(256) void gather_list_h_kernel(const _Float16 *__restrict__ Z, int C,
                                                            const int *__restrict__ pos, int K, int64_t n,
                                                            _Float16 *__restrict__ out, TsWgradReduce side,
                                                            const _Float16 *__restrict__ addend, int rpw, TsGatherEpilogue epi) {
  __shared__ int lst[64][33];
  __shared__ int cnt[64];
  const int tid = threadIdx.x;
  {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + tid, step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = e; i < (int64_t)side.K * side.cacb4; i += step) ts_wgrad_reduce_one(side, i);
  }
  const int64_t j0 = (int64_t)blockIdx.x * rpw;
  const int hw = tid >> 5, l = tid & 31;
  for (int base = 0; base < rpw; base += 8) {          // uniform trip count: the ballot below needs every lane
    const int r = base + hw;
    const int64_t j = j0 + r;
    int p = -1;
    if (r < rpw && j < n && l < K) p = pos[(int64_t)l * n + j];
    const bool live = p >= 0;
    const unsigned long long m64 = __builtin_amdgcn_ballot_w64(live);
    const unsigned m = (tid & 32) ? (unsigned)(m64 >> 32) : (unsigned)m64;
    if (live) lst[r][__builtin_popcount(m & ((1u << l) - 1u))] = p;
    if (l == 0 && r < rpw) cnt[r] = __builtin_popcount(m);
  }
  __syncthreads();
  const int c8n = C >> 3;
  const int r = tid / c8n;
  const int64_t j = j0 + r;
  if (r >= rpw || j >= n) return;
  const int c8 = (tid - r * c8n) << 3;
  const int m = cnt[r];
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int b = 0; b < m; b += R) {
    h8 v[R];
#pragma unroll
    for (int i = 0; i < R; ++i)
      v[i] = b + i < m ? __builtin_nontemporal_load((const h8 *)(Z + (int64_t)lst[r][b + i] * C + c8)) : (h8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += (float)v[i][q];
  }
  if (addend) {
    const h8 a = *(const h8 *)(addend + j * C + c8);
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] += (float)a[q];
  }
  if (epi.mean) {                        // evaluation block: bn_act_fwd_h_kernel's arithmetic on the fp32 sum
    h8 r;
    if (epi.residual) r = *(const h8 *)((const _Float16 *)epi.residual + j * C + c8);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = (acc[q] - epi.mean[c8 + q]) * epi.invstd[c8 + q] * epi.w[c8 + q] + epi.b[c8 + q];
      if (epi.residual) v += (float)r[q];
      if (epi.relu) v = fmaxf(v, 0.f);
      acc[q] = v;
    }
  }
  h8 o;
#pragma unroll
  for (int q = 0; q < 8; ++q) o[q] = (_Float16)acc[q];
  *(h8 *)(out + j * C + c8) = o;
}

extern "C" int ts_conv_gather_sum_f16(const void *z, int32_t c, const int32_t *pos, int32_t K, int64_t n_rows,
                                      int64_t n_pairs, void *out, ts_stream_t stream_) {
  return ts_conv_gather_sum_f16_ex(z, c, pos, K, n_rows, n_pairs, out, nullptr, nullptr, stream_);
}

int ts_conv_gather_sum_f16_ex(const void *z, int32_t c, const int32_t *pos, int32_t K, int64_t n_rows, int64_t n_pairs,
                              void *out, const TsWgradReduce *side_job, const void *addend, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TsWgradReduce side = {};
  if (side_job) side = *side_job;
  TS_REQUIRE(!side_job || (n_rows > 0 && side.part && side.dW && side.nboffs && side.chunk > 0 &&
                           ((((uintptr_t)side.part) | ((uintptr_t)side.dW)) & 15) == 0),
             TS_ERR_INVALID_ARGUMENT, "ts_conv_gather_sum_f16: bad side job");
  TS_REQUIRE(c > 0 && (c & 7) == 0 && K > 0 && n_rows >= 0 && n_pairs >= 0, TS_ERR_INVALID_ARGUMENT,
             "ts_conv_gather_sum_f16: bad sizes (C must be a multiple of 8)");
  if (n_rows == 0) return TS_OK;
  TS_REQUIRE(pos && out && (z || n_pairs == 0), TS_ERR_INVALID_ARGUMENT, "ts_conv_gather_sum_f16: null pointer");
  TS_REQUIRE(((((uintptr_t)z) | ((uintptr_t)out)) & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_conv_gather_sum_f16: pointers must be 16-byte aligned");
  const _Float16 *zz = (const _Float16 *)z;
  // default: live positions compacted in LDS first; TASEG_GATHER_POSITIONS=1 keeps the K-register form (A/B runs)
  const bool k_registers = ts_get_option(TS_OPT_GATHER_POSITIONS) != 0;
  if (K <= 32 && c >= 64 && c <= 2048 && !k_registers && g_ts_conv_impl != 1) {   // 32 channels: 64 rows per workgroup, the list build costs more than it saves
    const int rpw = 256 / (c >> 3);
    gather_list_h_kernel<8><<<(unsigned)ts_cdiv(n_rows, rpw), 256, 0, stream>>>(zz, c, pos, K, n_rows, (_Float16 *)out, side,
                                                                                (const _Float16 *)addend, rpw,
                                                                                TsGatherEpilogue{nullptr, nullptr, nullptr, nullptr, nullptr, 0});
    TS_CHECK_LAUNCH("ts_conv_gather_sum_f16 (list)");
    return TS_OK;
  }
  const int grid = (int)std::min<int64_t>(ts_cdiv(n_rows * (c / 8), 256), 1 << 20);
  if (K == 27)
    gather_sum_h_kernel<27><<<grid, 256, 0, stream>>>(zz, c, pos, K, n_rows, (_Float16 *)out, side, (const _Float16 *)addend);
  else if (K == 8)
    gather_sum_h_kernel<8><<<grid, 256, 0, stream>>>(zz, c, pos, K, n_rows, (_Float16 *)out, side, (const _Float16 *)addend);
  else
    gather_sum_h_kernel<0><<<grid, 256, 0, stream>>>(zz, c, pos, K, n_rows, (_Float16 *)out, side, (const _Float16 *)addend);
  TS_CHECK_LAUNCH("ts_conv_gather_sum_f16");
  return TS_OK;
}

int ts_conv_gather_sum_f16_epi(const void *z, int32_t c, const int32_t *pos, int32_t K, int64_t n_rows, int64_t n_pairs, void *out,
                               const TsGatherEpilogue &epi, ts_stream_t stream_) {
  const bool k_registers = ts_get_option(TS_OPT_GATHER_POSITIONS) != 0;
  if (!(K > 0 && K <= 32 && c >= 64 && c <= 2048 && (c & 7) == 0 && !k_registers && g_ts_conv_impl != 1 && n_rows > 0 && z && pos &&
        out && epi.mean && epi.invstd && epi.w && epi.b &&
        ((((uintptr_t)z) | ((uintptr_t)out) | ((uintptr_t)epi.residual)) & 15) == 0))
    return TS_ERR_UNSUPPORTED;
  const int rpw = 256 / (c >> 3);
  gather_list_h_kernel<8><<<(unsigned)ts_cdiv(n_rows, rpw), 256, 0, (hipStream_t)stream_>>>(
      (const _Float16 *)z, c, pos, K, n_rows, (_Float16 *)out, TsWgradReduce{}, nullptr, rpw, epi);
  TS_CHECK_LAUNCH("ts_conv_gather_sum_f16 (list + evaluation tail)");
  return TS_OK;
}

// ------------------------------------------------------------------------------------- weight gradient, half inputs
//   dW_k[ci, co] (fp32) = sum_{pairs p of k} A[pa_p, ci] * B[pb_p, co],   A, B half rows
// Chunking / flushing as wgrad_gemm_fast_kernel.  The reduction runs over the PAIR index, but both operands arrive
// pair-major (one gathered row per pair); they are staged as they come ([pair][channel] images, 16-byte chunks) and
// the MFMA fragments - 8 consecutive pairs of one channel per lane - are read with gfx950's transposing LDS load
// ds_read_b64_tr_b16: a 16-lane group reads a 4-pair x 16-channel block and every lane receives the 4 pairs of its
// channel (lane 4q + p of the group addresses pair row q, channels 4p .. 4p+3).
typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
#define WH_PS 64            // pairs per step (two 32-deep MFMA k-blocks)
#define WH_MAXCHUNK 1024

template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void wgrad_h_kernel(const _Float16 *__restrict__ A, int CA,
                                                      const _Float16 *__restrict__ B, int CB,
                                                      const int2 *__restrict__ nbmaps, const int *__restrict__ nboffs,
                                                      int K, int P, int col_a, int chunk, float *__restrict__ dW,
                                                      float *__restrict__ part) {
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int XP = TM + 8, YP = TN + 8;              // pitches in halves (16-byte multiples, 4 mod 8 dwords)
  constexpr int A_IT = (WH_PS * (TM / 8) + 255) / 256, B_IT = (WH_PS * (TN / 8) + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) _Float16 smem_w[];
  _Float16 *Xl = smem_w;                               // 2 x [64][XP]
  _Float16 *Yl = Xl + 2 * WH_PS * XP;                  // 2 x [64][YP]
  int *idxA = (int *)(Yl + 2 * WH_PS * YP), *idxB = idxA + WH_MAXCHUNK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
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
    int pn = st.p0 + WH_PS;
    if (pn < kend) {
      st.p0 = pn;
      st.np = min(WH_PS, kend - pn);
      return st;
    }
    pn = kend;
    for (++st.k; st.k < K && pn < p_end; ++st.k) {
      kend = min(off_at(st.k + 1), p_end);
      if (kend > pn) {
        st.p0 = pn;
        st.np = min(WH_PS, kend - pn);
        return st;
      }
    }
    st.k = K;
    return st;
  };
  const _Float16 *abase = A + ci0, *bbase = B + co0;
  h8 ra[A_IT], rb[B_IT];
  auto load_regs = [&](const WStep &st) {
    const int l0 = st.p0 - p_beg, last = st.np - 1;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int e = min(tid + it * 256, WH_PS * (TM / 8) - 1);
      const int pp = e / (TM / 8), c8 = (e - pp * (TM / 8)) << 3;
      ra[it] = *(const h8 *)(abase + (int64_t)idxA[l0 + min(pp, last)] * CA + c8);
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int e = min(tid + it * 256, WH_PS * (TN / 8) - 1);
      const int pp = e / (TN / 8), c8 = (e - pp * (TN / 8)) << 3;
      rb[it] = *(const h8 *)(bbase + (int64_t)idxB[l0 + min(pp, last)] * CB + c8);
    }
  };
  auto store_lds = [&](_Float16 *xl, _Float16 *yl, const WStep &st) {
    const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int e = tid + it * 256;
      const int pp = e / (TM / 8), c8 = (e - pp * (TM / 8)) << 3;
      if (e < WH_PS * (TM / 8)) *(h8 *)&xl[pp * XP + c8] = pp < st.np ? ra[it] : zero;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int e = tid + it * 256;
      const int pp = e / (TN / 8), c8 = (e - pp * (TN / 8)) << 3;
      if (e < WH_PS * (TN / 8)) *(h8 *)&yl[pp * YP + c8] = pp < st.np ? rb[it] : zero;
    }
  };
  // transposed fragment: 8 consecutive pairs (rows r0 .. r0+7 of the image) of channel c0 + r16
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  auto frag = [&](const _Float16 *img, int pitch, int r0, int c0) -> h8 {
    typedef hv4 __attribute__((address_space(3))) * lds_hv4;
    const hv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_hv4)(img + (r0 + tq) * pitch + c0 + 4 * tp));
    const hv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_hv4)(img + (r0 + 4 + tq) * pitch + c0 + 4 * tp));
    h8 v;
    v[0] = (_Float16)lo[0]; v[1] = (_Float16)lo[1]; v[2] = (_Float16)lo[2]; v[3] = (_Float16)lo[3];
    v[4] = (_Float16)hi[0]; v[5] = (_Float16)hi[1]; v[6] = (_Float16)hi[2]; v[7] = (_Float16)hi[3];
    return v;
  };

  WStep cur;
  cur.k = k0;
  cur.p0 = p_beg;
  cur.np = min(WH_PS, min(off_at(k0 + 1), p_end) - p_beg);
  __syncthreads();  // pair indices visible
  load_regs(cur);
  int buf = 0;
  while (cur.k < K) {
    _Float16 *xl = Xl + buf * (WH_PS * XP), *yl = Yl + buf * (WH_PS * YP);
    store_lds(xl, yl, cur);
    __syncthreads();
    const WStep nxt = advance(cur);
    if (nxt.k < K) load_regs(nxt);
#pragma unroll
    for (int kb = 0; kb < WH_PS; kb += 32) {
      h8 a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = frag(xl, XP, kb + 8 * g, (wr * MI + mi) * 16);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = frag(yl, YP, kb + 8 * g, (wc * NI + ni) * 16);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
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
    buf ^= 1;
  }
}

template <int TM, int TN>
static int launch_wgrad_h(const _Float16 *A, int CA, const _Float16 *B, int CB, const int2 *nbmaps, const int *nboffs,
                          int K, int col_a, int64_t n_pairs, float *dW, hipStream_t stream) {
  const int tiles = (CA / TM) * (CB / TN);
  const TsWgradPlan plan = ts_wgrad_plan(n_pairs, tiles, K, WH_PS, WH_MAXCHUNK);
  g_ts_wgrad_plan = plan;
  const int64_t chunk = plan.chunk;
  dim3 grid((unsigned)plan.n_chunks, tiles);
  const size_t lds = (size_t)2 * WH_PS * ((TM + 8) + (TN + 8)) * 2 + 2 * WH_MAXCHUNK * 4;
  static bool attr_set = false;
  if (!attr_set) {
    TS_CHECK_HIP(hipFuncSetAttribute((const void *)wgrad_h_kernel<TM, TN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024), "hipFuncSetAttribute");
    attr_set = true;
  }
  wgrad_h_kernel<TM, TN><<<grid, 256, lds, stream>>>(A, CA, B, CB, nbmaps, nboffs, K, (int)n_pairs, col_a, (int)chunk, dW,
                                                     g_ts_wgrad_part);
  TS_CHECK_LAUNCH("conv_wgrad_f16");
  return TS_OK;
}

extern "C" int ts_conv_wgrad_f16(const void *a_feat, int32_t c_a, const void *b_feat, int32_t c_b,
                                 const int32_t *nbmaps, const int32_t *nboffs, int32_t K, int32_t col_a,
                                 int64_t n_pairs, float *grad_kernel, ts_stream_t stream_) {
  return ts_conv_wgrad_f16_ex(a_feat, c_a, b_feat, c_b, nbmaps, nboffs, K, col_a, n_pairs, grad_kernel, 0, stream_);
}

// Deterministic form (partial tiles into `ws`, ordered sum): see ts_conv_wgrad_det.
extern "C" int ts_conv_wgrad_f16_det(const void *a_feat, int32_t c_a, const void *b_feat, int32_t c_b,
                                     const int32_t *nbmaps, const int32_t *nboffs, int32_t K, int32_t col_a,
                                     int64_t n_pairs, float *grad_kernel, void *ws, size_t ws_bytes, ts_stream_t stream_) {
  TS_REQUIRE(c_a > 0 && c_b > 0 && K > 0 && n_pairs >= 0, TS_ERR_INVALID_ARGUMENT, "ts_conv_wgrad_f16_det: bad sizes");
  TS_REQUIRE(ws && ws_bytes >= ts_conv_wgrad_workspace_bytes(n_pairs, c_a, c_b, K) && ((uintptr_t)ws & 15) == 0 &&
                 ((uintptr_t)grad_kernel & 15) == 0,
             TS_ERR_INVALID_ARGUMENT, "ts_conv_wgrad_f16_det: workspace too small or misaligned");
  if (n_pairs == 0) {
    TS_CHECK_HIP(hipMemsetAsync(grad_kernel, 0, (size_t)K * c_a * c_b * 4, (hipStream_t)stream_), "wgrad memset");
    return TS_OK;
  }
  g_ts_wgrad_part = (float *)ws;
  const int rc = ts_conv_wgrad_f16_ex(a_feat, c_a, b_feat, c_b, nbmaps, nboffs, K, col_a, n_pairs, grad_kernel, 1, stream_);
  g_ts_wgrad_part = nullptr;
  if (rc != TS_OK) return rc;
  const TsWgradReduce job = {(const float *)ws, nboffs, grad_kernel, K, g_ts_wgrad_plan.chunk, (int64_t)c_a * c_b / 4};
  return ts_wgrad_reduce(job, stream_);
}

int ts_conv_wgrad_f16_ex(const void *a_feat, int32_t c_a, const void *b_feat, int32_t c_b, const int32_t *nbmaps,
                         const int32_t *nboffs, int32_t K, int32_t col_a, int64_t n_pairs, float *grad_kernel,
                         int32_t already_zero, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(c_a > 0 && c_b > 0 && K > 0 && K <= 63 && n_pairs >= 0 && n_pairs < (1LL << 31), TS_ERR_INVALID_ARGUMENT,
             "ts_conv_wgrad_f16: bad sizes");
  TS_REQUIRE(c_a % 32 == 0 && c_b % 32 == 0, TS_ERR_UNSUPPORTED, "ts_conv_wgrad_f16: channel counts must be multiples of 32");
  TS_REQUIRE(grad_kernel && nboffs, TS_ERR_INVALID_ARGUMENT, "ts_conv_wgrad_f16: null pointer");
  if (!already_zero) TS_CHECK_HIP(hipMemsetAsync(grad_kernel, 0, (size_t)K * c_a * c_b * 4, stream), "wgrad memset");
  if (n_pairs == 0) return TS_OK;
  TS_REQUIRE(a_feat && b_feat && nbmaps, TS_ERR_INVALID_ARGUMENT, "ts_conv_wgrad_f16: null pointer");
  TS_REQUIRE(((((uintptr_t)a_feat) | ((uintptr_t)b_feat)) & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "ts_conv_wgrad_f16: pointers must be 16-byte aligned");
  const _Float16 *a = (const _Float16 *)a_feat, *b = (const _Float16 *)b_feat;
  const int2 *nm = (const int2 *)nbmaps;
  col_a = col_a ? 1 : 0;
  auto pick = [](int c) { return c % 128 == 0 ? 128 : c % 96 == 0 ? 96 : c % 64 == 0 ? 64 : 32; };
  const int tm = pick(c_a), tn = pick(c_b);
#define TS_WH(TM, TN) launch_wgrad_h<TM, TN>(a, c_a, b, c_b, nm, nboffs, K, col_a, n_pairs, grad_kernel, stream)
#define TS_WH_ROW(TM)                    \
  switch (tn) {                          \
    case 32: return TS_WH(TM, 32);       \
    case 64: return TS_WH(TM, 64);       \
    case 96: return TS_WH(TM, 96);       \
    default: return TS_WH(TM, 128);      \
  }
  switch (tm) {
    case 32: TS_WH_ROW(32)
    case 64: TS_WH_ROW(64)
    case 96: TS_WH_ROW(96)
    default: TS_WH_ROW(128)
  }
#undef TS_WH_ROW
#undef TS_WH
}
