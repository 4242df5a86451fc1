// Two-pass sparse convolution on the rulebook (the default forward / dgrad path).
//
//   pass 1  pair_gemm_kernel    Z[p, :] = X[g_p, :] @ W_{k(p)}  for all P pairs of the rulebook at once.
//           The pair list is ordered by (offset k, output row), so a tile of 128 consecutive pairs
//           almost always belongs to ONE offset: a plain dense 128 x BN x C_in MFMA GEMM whose A rows are
//           gathered by index while they are staged into LDS and whose B operand (W_k, <= 384 x 256 floats)
//           stays L2 resident.  No flops are spent on missing neighbours, no padding per (tile, offset),
//           every workgroup does the same amount of work.  Tiles that straddle an offset boundary
//           (<= K - 1 of them) run once per offset with the foreign rows zeroed.
//   pass 2  gather_sum_kernel   Y[j, :] = sum_k Z[pos[k, j], :]  - a streaming reduction: for a fixed k the
//           rows Z[pos[k, j]] of consecutive j are consecutive in memory.  Deterministic (k ascending),
//           no atomics, every output row written once.
//
// Reference algorithm: torchsparse backend/convolution/convolution_cuda.cu:101-164 runs, per offset, a
// gather kernel, a cuBLAS GEMM and a read-modify-write scatter kernel (3 K launches, host-synchronised).
// HBM traffic here: Z written once and read once (2 P C_out s bytes) + Y; the gathers hit L2 / Infinity Cache.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PG_BM 128
#define PG_BK 32
#define PG_AP (PG_BK + 4)

// BN = output columns per workgroup, WR = waves along the pair (row) dimension (WC = 4 / WR along columns)
template <int BN, int WR, bool WT>
__global__ __launch_bounds__(256) void pair_gemm_kernel(const float *__restrict__ X, int R,
                                                        const float *__restrict__ W, int O_total,
                                                        const int2 *__restrict__ nbmaps,
                                                        const int *__restrict__ nboffs, int K, int64_t P, int gcol,
                                                        float *__restrict__ Z) {
  constexpr int WC = 4 / WR;
  constexpr int MI = (PG_BM / 16) / WR;   // 16-row blocks per wave
  constexpr int NI = (BN / 16) / WC;      // 16-col blocks per wave
  constexpr int BP = BN + 4;              // pitch of the [k][col] weight tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *At = smem;                                       // [128][36]
  float *Bt = At + PG_BM * PG_AP;                         // !WT: [32][BN+4]   WT: [BN][36]
  int *rowidx = (int *)(Bt + (WT ? BN * PG_AP : PG_BK * BP));  // [128]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int wr = wave / WC, wc = wave % WC;
  const int64_t p0 = (int64_t)blockIdx.x * PG_BM;
  const int np = (int)min((int64_t)PG_BM, P - p0);
  const int o0 = blockIdx.y * BN;
  const int OT = min(BN, O_total - o0);
  const int O16 = (OT + 15) & ~15;

  if (tid < PG_BM) {
    int v = -1;
    if (tid < np) {
      int2 pr = nbmaps[p0 + tid];
      v = gcol ? pr.y : pr.x;
    }
    rowidx[tid] = v;
  }
  // offsets overlapping this tile (wave-uniform scalar scan of the K+1 prefix sums)
  int k_lo = 0, k_hi = 0;
  for (int k = 0; k < K; ++k) {
    int b = nboffs[k];
    if ((int64_t)b <= p0) k_lo = k;
    if ((int64_t)b <= p0 + np - 1) k_hi = k;
  }

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool x_vec = ((R & 3) == 0) && ((((uintptr_t)X) & 15) == 0);
  const bool w_vec = WT ? (((R & 3) == 0) && ((((uintptr_t)W) & 15) == 0))
                        : (((O_total & 3) == 0) && ((((uintptr_t)W) & 15) == 0));

  for (int k = k_lo; k <= k_hi; ++k) {
    const int s0 = max((int)((int64_t)nboffs[k] - p0), 0);
    const int s1 = min((int)((int64_t)nboffs[k + 1] - p0), np);
    if (s1 <= s0) continue;  // uniform
    for (int c0 = 0; c0 < R; c0 += PG_BK) {
      const int ck = min(PG_BK, R - c0);
      const int ck16 = (ck + 15) & ~15;
      __syncthreads();  // rowidx visible / previous slice's fragment reads finished
      // ---- A: gathered rows of this offset's segment, zero elsewhere
      for (int e = tid; e < PG_BM * (PG_BK / 4); e += 256) {
        int rr = e >> 3, c4 = (e & 7) << 2;
        if (c4 >= ck16) continue;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rr >= s0 && rr < s1) {
          const float *src = X + (int64_t)rowidx[rr] * R + c0 + c4;
          if (x_vec && c4 + 3 < ck) {
            v = *(const float4 *)src;
          } else {
            if (c4 + 0 < ck) v.x = src[0];
            if (c4 + 1 < ck) v.y = src[1];
            if (c4 + 2 < ck) v.z = src[2];
            if (c4 + 3 < ck) v.w = src[3];
          }
        }
        *(float4 *)&At[rr * PG_AP + c4] = v;
      }
      // ---- B: the W_k slice
      if (!WT) {
        const int q4 = O16 >> 2;
        const float *wk = W + ((int64_t)k * R + c0) * O_total + o0;
        for (int e = tid; e < ck16 * q4; e += 256) {
          int kk = e / q4, c4 = (e - kk * q4) << 2;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (kk < ck) {
            const float *src = wk + (int64_t)kk * O_total + c4;
            if (w_vec && c4 + 3 < OT) {
              v = *(const float4 *)src;
            } else {
              if (c4 + 0 < OT) v.x = src[0];
              if (c4 + 1 < OT) v.y = src[1];
              if (c4 + 2 < OT) v.z = src[2];
              if (c4 + 3 < OT) v.w = src[3];
            }
          }
          *(float4 *)&Bt[kk * BP + c4] = v;
        }
      } else {
        const float *wk = W + ((int64_t)k * O_total + o0) * R + c0;
        for (int e = tid; e < O16 * (PG_BK / 4); e += 256) {
          int col = e >> 3, c4 = (e & 7) << 2;
          if (c4 >= ck16) continue;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (col < OT) {
            const float *src = wk + (int64_t)col * R + c4;
            if (w_vec && c4 + 3 < ck) {
              v = *(const float4 *)src;
            } else {
              if (c4 + 0 < ck) v.x = src[0];
              if (c4 + 1 < ck) v.y = src[1];
              if (c4 + 2 < ck) v.z = src[2];
              if (c4 + 3 < ck) v.w = src[3];
            }
          }
          *(float4 *)&Bt[col * PG_AP + c4] = v;
        }
      }
      __syncthreads();
      // ---- MFMA.  k-slot permutation as in conv.hip: lane group g supplies reduction index 4 g + s in
      // step s, so A (and W^T) fragments are single 16-byte LDS reads.
      for (int j = 0; j < ck16; j += 16) {
        float4 a[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int mb = wr * MI + mi;
          a[mi] = *(const float4 *)&At[(mb * 16 + r16) * PG_AP + j + 4 * g];
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int nb = wc * NI + ni;
          if (nb * 16 >= O16) continue;  // uniform per wave
          float b0, b1, b2, b3;
          if (!WT) {
            const float *bp = &Bt[(j + 4 * g) * BP + nb * 16 + r16];
            b0 = bp[0];
            b1 = bp[BP];
            b2 = bp[2 * BP];
            b3 = bp[3 * BP];
          } else {
            const float4 b = *(const float4 *)&Bt[(nb * 16 + r16) * PG_AP + j + 4 * g];
            b0 = b.x;
            b1 = b.y;
            b2 = b.z;
            b3 = b.w;
          }
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const int mb = wr * MI + mi;
            if (mb * 16 >= s1 || mb * 16 + 16 <= s0) continue;  // row block outside this offset's segment
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi].x, b0, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi].y, b1, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi].z, b2, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi].w, b3, acc[mi][ni], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- Z rows (C/D map: col = lane & 15, row = 4 (lane >> 4) + reg)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = (wc * NI + ni) * 16 + r16;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = (wr * MI + mi) * 16 + 4 * g + q;
        if (row < np && col < OT) Z[(p0 + row) * O_total + o0 + col] = acc[mi][ni][q];
      }
    }
  }
}

template <int BN, int WR, bool WT>
static int launch_pair_gemm(const float *X, int R, const float *W, int O_total, const int2 *nbmaps, const int *nboffs,
                            int K, int64_t P, int gcol, float *Z, hipStream_t stream) {
  size_t lds = (size_t)(PG_BM * PG_AP + (WT ? BN * PG_AP : PG_BK * (BN + 4))) * 4 + PG_BM * 4;
  dim3 grid((unsigned)ts_cdiv(P, PG_BM), (unsigned)ts_cdiv(O_total, BN));
  pair_gemm_kernel<BN, WR, WT><<<grid, 256, lds, stream>>>(X, R, W, O_total, nbmaps, nboffs, K, P, gcol, Z);
  TS_CHECK_LAUNCH("conv_pair_gemm");
  return TS_OK;
}

extern "C" int ts_conv_pair_gemm(const float *feat, int64_t n_rows, int32_t c_in, const float *kernel, int32_t K,
                                 int32_t weight_transposed, const int32_t *nbmaps, const int32_t *nboffs,
                                 int64_t n_pairs, int32_t gather_col, float *z, int32_t c_out, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n_rows >= 0 && c_in > 0 && c_out > 0 && K > 0 && n_pairs >= 0, TS_ERR_INVALID_ARGUMENT,
             "ts_conv_pair_gemm: bad sizes");
  TS_REQUIRE(n_pairs < (1LL << 31), TS_ERR_UNSUPPORTED, "ts_conv_pair_gemm: too many pairs");
  if (n_pairs == 0) return TS_OK;
  TS_REQUIRE(feat && kernel && nbmaps && nboffs && z, TS_ERR_INVALID_ARGUMENT, "ts_conv_pair_gemm: null pointer");
  const int2 *nm = (const int2 *)nbmaps;
  const int gc = gather_col ? 1 : 0;
#define TS_PG(BN, WR)                                                                                             \
  (weight_transposed                                                                                              \
       ? launch_pair_gemm<BN, WR, true>(feat, c_in, kernel, c_out, nm, nboffs, K, n_pairs, gc, z, stream)         \
       : launch_pair_gemm<BN, WR, false>(feat, c_in, kernel, c_out, nm, nboffs, K, n_pairs, gc, z, stream))
  if (c_out <= 32) return TS_PG(32, 4);
  if (c_out <= 64) return TS_PG(64, 2);
  return TS_PG(128, 2);
#undef TS_PG
}

// ------------------------------------------------------------------------------------- pass 2
template <int VEC>
__global__ __launch_bounds__(256) void gather_sum_kernel(const float *__restrict__ Z, int C,
                                                         const int *__restrict__ pos, int K, int64_t n,
                                                         int64_t n_pairs, float *__restrict__ out) {
  const int cv = C / VEC;
  const int64_t total = n * cv;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; e < total; e += step) {
    const int64_t j = e / cv;
    const int c = (int)(e - j * cv) * VEC;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
      const int p = pos[(int64_t)k * n + j];
      if (p >= 0 && p < n_pairs) {
        if (VEC == 4) {
          const float4 f = *(const float4 *)(Z + (int64_t)p * C + c);
          acc[0] += f.x;
          acc[1 % VEC] += f.y;
          acc[2 % VEC] += f.z;
          acc[3 % VEC] += f.w;
        } else {
          acc[0] += Z[(int64_t)p * C + c];
        }
      }
    }
    if (VEC == 4) {
      *(float4 *)(out + j * C + c) = make_float4(acc[0], acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
    } else {
      out[j * C + c] = acc[0];
    }
  }
}

extern "C" int ts_conv_gather_sum(const float *z, int32_t c, const int32_t *pos, int32_t K, int64_t n_rows,
                                  int64_t n_pairs, float *out, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(c > 0 && K > 0 && n_rows >= 0 && n_pairs >= 0, TS_ERR_INVALID_ARGUMENT, "ts_conv_gather_sum: bad sizes");
  if (n_rows == 0) return TS_OK;
  TS_REQUIRE(pos && out && (z || n_pairs == 0), TS_ERR_INVALID_ARGUMENT, "ts_conv_gather_sum: null pointer");
  const bool vec = (c % 4 == 0) && ((((uintptr_t)z) & 15) == 0) && ((((uintptr_t)out) & 15) == 0);
  if (vec) {
    int grid = (int)std::min<int64_t>(ts_cdiv(n_rows * (c / 4), 256), 1 << 20);
    gather_sum_kernel<4><<<grid, 256, 0, stream>>>(z, c, pos, K, n_rows, n_pairs, out);
  } else {
    int grid = (int)std::min<int64_t>(ts_cdiv(n_rows * c, 256), 1 << 20);
    gather_sum_kernel<1><<<grid, 256, 0, stream>>>(z, c, pos, K, n_rows, n_pairs, out);
  }
  TS_CHECK_LAUNCH("conv_gather_sum");
  return TS_OK;
}

// ------------------------------------------------------------------------------------- weight gradient
//   dW_k[ci, co] = sum_{pairs p of k}  A[pa_p, ci] * B[pb_p, co]
// One workgroup = (offset k, chunk of its pairs, TM x TN tile of dW_k): a "TN" GEMM whose reduction runs over
// the pair list.  Both operands are gathered rows, staged 32 pairs at a time; each wave owns a
// (TM/2) x (TN/2) register tile so every LDS fragment feeds TM/32 or TN/32 MFMAs.  Partial tiles of the
// chunks are combined with float atomics (dW is small: the atomic bytes are ~1 / (chunk pairs / 2) of the flops).
#define WG_PS 32

template <int TM, int TN>
__global__ __launch_bounds__(256) void wgrad_gemm_kernel(const float *__restrict__ A, int CA,
                                                         const float *__restrict__ B, int CB,
                                                         const int2 *__restrict__ nbmaps,
                                                         const int *__restrict__ nboffs, int col_a, int pairs_per_wg,
                                                         float *__restrict__ dW) {
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int XP = TM + 4, YP = TN + 4;
  __shared__ __attribute__((aligned(16))) float Xl[WG_PS * XP];
  __shared__ __attribute__((aligned(16))) float Yl[WG_PS * YP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int k = blockIdx.y;
  const int beg = nboffs[k] + blockIdx.x * pairs_per_wg;
  const int end = min(nboffs[k + 1], beg + pairs_per_wg);
  if (beg >= end) return;  // uniform
  const int tiles_n = (CB + TN - 1) / TN;
  const int ci0 = (blockIdx.z / tiles_n) * TM, co0 = (blockIdx.z % tiles_n) * TN;
  const int ca = min(TM, CA - ci0), cb = min(TN, CB - co0);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool a_vec = ((CA & 3) == 0) && ((ci0 & 3) == 0) && ((((uintptr_t)A) & 15) == 0);
  const bool b_vec = ((CB & 3) == 0) && ((co0 & 3) == 0) && ((((uintptr_t)B) & 15) == 0);

  for (int p0 = beg; p0 < end; p0 += WG_PS) {
    const int np = min(WG_PS, end - p0);
    __syncthreads();
    for (int e = tid; e < WG_PS * (TM / 4); e += 256) {
      const int pp = e / (TM / 4), c4 = (e - pp * (TM / 4)) << 2;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pp < np && c4 < ca) {
        const int2 pr = nbmaps[p0 + pp];
        const float *src = A + (int64_t)(col_a ? pr.y : pr.x) * CA + ci0 + c4;
        if (a_vec && c4 + 3 < ca) {
          v = *(const float4 *)src;
        } else {
          if (c4 + 0 < ca) v.x = src[0];
          if (c4 + 1 < ca) v.y = src[1];
          if (c4 + 2 < ca) v.z = src[2];
          if (c4 + 3 < ca) v.w = src[3];
        }
      }
      *(float4 *)&Xl[pp * XP + c4] = v;
    }
    for (int e = tid; e < WG_PS * (TN / 4); e += 256) {
      const int pp = e / (TN / 4), c4 = (e - pp * (TN / 4)) << 2;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pp < np && c4 < cb) {
        const int2 pr = nbmaps[p0 + pp];
        const float *src = B + (int64_t)(col_a ? pr.x : pr.y) * CB + co0 + c4;
        if (b_vec && c4 + 3 < cb) {
          v = *(const float4 *)src;
        } else {
          if (c4 + 0 < cb) v.x = src[0];
          if (c4 + 1 < cb) v.y = src[1];
          if (c4 + 2 < cb) v.z = src[2];
          if (c4 + 3 < cb) v.w = src[3];
        }
      }
      *(float4 *)&Yl[pp * YP + c4] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < WG_PS; j += 16) {
      float a[MI][4], b[NI][4];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const float *ap = &Xl[(j + 4 * g) * XP + (wr * MI + mi) * 16 + r16];
        a[mi][0] = ap[0];
        a[mi][1] = ap[XP];
        a[mi][2] = ap[2 * XP];
        a[mi][3] = ap[3 * XP];
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const float *bp = &Yl[(j + 4 * g) * YP + (wc * NI + ni) * 16 + r16];
        b[ni][0] = bp[0];
        b[ni][1] = bp[YP];
        b[ni][2] = bp[2 * YP];
        b[ni][3] = bp[3 * YP];
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        if ((wr * MI + mi) * 16 >= ca) continue;  // uniform per wave
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if ((wc * NI + ni) * 16 >= cb) continue;
#pragma unroll
          for (int s = 0; s < 4; ++s)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][s], b[ni][s], acc[mi][ni], 0, 0, 0);
        }
      }
    }
  }
  float *dwk = dW + (int64_t)k * CA * CB;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int co = (wc * NI + ni) * 16 + r16;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ci = (wr * MI + mi) * 16 + 4 * g + q;
        if (ci < ca && co < cb) atomicAdd(&dwk[(int64_t)(ci0 + ci) * CB + co0 + co], acc[mi][ni][q]);
      }
    }
  }
}

template <int TM, int TN>
static int launch_wgrad(const float *A, int CA, const float *B, int CB, const int2 *nbmaps, const int *nboffs, int K,
                        int col_a, int64_t max_pairs, float *dW, hipStream_t stream) {
  const int tiles = (int)(ts_cdiv(CA, TM) * ts_cdiv(CB, TN));
  // ~1024 workgroups over the launch, chunks of at least 256 pairs
  int64_t chunks_per_k = std::max<int64_t>(1, 1024 / ((int64_t)K * tiles));
  int64_t ppw = ts_cdiv(max_pairs < 1 ? 1 : max_pairs, chunks_per_k);
  ppw = std::max<int64_t>(256, (ppw + WG_PS - 1) / WG_PS * WG_PS);
  const int nchunks = (int)ts_cdiv(max_pairs < 1 ? 1 : max_pairs, ppw);
  dim3 grid(nchunks, K, tiles);
  wgrad_gemm_kernel<TM, TN><<<grid, 256, 0, stream>>>(A, CA, B, CB, nbmaps, nboffs, col_a, (int)ppw, dW);
  TS_CHECK_LAUNCH("conv_wgrad");
  return TS_OK;
}

__global__ __launch_bounds__(256) void conv_wgrad_scalar_kernel(const float *__restrict__ A, int CA,
                                                                const float *__restrict__ B, int CB,
                                                                const int2 *__restrict__ nbmaps,
                                                                const int *__restrict__ nboffs, int col_a,
                                                                float *__restrict__ dW) {
  int k = blockIdx.y;
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= CA * CB) return;
  int ci = e / CB, co = e - ci * CB;
  float s = 0.f;
  for (int p = nboffs[k]; p < nboffs[k + 1]; ++p) {
    int2 pr = nbmaps[p];
    int ia = col_a ? pr.y : pr.x, ib = col_a ? pr.x : pr.y;
    s = fmaf(A[(int64_t)ia * CA + ci], B[(int64_t)ib * CB + co], s);
  }
  dW[(int64_t)k * CA * CB + e] = s;
}

extern "C" int ts_conv_wgrad(const float *a_feat, int32_t c_a, const float *b_feat, int32_t c_b,
                             const int32_t *nbmaps, const int32_t *nboffs, int32_t K, int32_t col_a,
                             int64_t max_pairs_per_offset, float *grad_kernel, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(c_a > 0 && c_b > 0 && K > 0 && max_pairs_per_offset >= 0, TS_ERR_INVALID_ARGUMENT,
             "ts_conv_wgrad: bad sizes");
  TS_REQUIRE(grad_kernel && nboffs, TS_ERR_INVALID_ARGUMENT, "ts_conv_wgrad: null pointer");
  TS_CHECK_HIP(hipMemsetAsync(grad_kernel, 0, (size_t)K * c_a * c_b * 4, stream), "wgrad memset");
  if (max_pairs_per_offset == 0) return TS_OK;
  TS_REQUIRE(a_feat && b_feat && nbmaps, TS_ERR_INVALID_ARGUMENT, "ts_conv_wgrad: null pointer");
  const int2 *nm = (const int2 *)nbmaps;
  col_a = col_a ? 1 : 0;
  if (g_ts_conv_impl == 1) {
    dim3 grid((unsigned)ts_cdiv((int64_t)c_a * c_b, 256), K);
    conv_wgrad_scalar_kernel<<<grid, 256, 0, stream>>>(a_feat, c_a, b_feat, c_b, nm, nboffs, col_a, grad_kernel);
    TS_CHECK_LAUNCH("conv_wgrad_scalar");
    return TS_OK;
  }
  const int cmax = std::max(c_a, c_b), cmin = std::min(c_a, c_b);
#define TS_WG(TM, TN) launch_wgrad<TM, TN>(a_feat, c_a, b_feat, c_b, nm, nboffs, K, col_a, max_pairs_per_offset, grad_kernel, stream)
  if (cmax <= 32) return TS_WG(32, 32);
  if (c_a <= 32) return TS_WG(32, 128);    // stem: C_in = 4 / 5
  if (c_b <= 32) return TS_WG(128, 32);
  if (cmax <= 64 || cmin <= 48) return TS_WG(64, 64);
  return TS_WG(128, 128);
#undef TS_WG
}
