// Sparse convolution on a neighbour table: output-stationary gather -> f32 MFMA -> write once,
// and the pair-list weight gradient.  gfx950 only (v_mfma_f32_16x16x4_f32: exact f32,
// bitwise an fmaf chain in k order, 157 TF/s dense peak).
//
// Reference algorithm (torchsparse backend/convolution/convolution_cuda.cu:53-278): for every
// kernel offset k: gather rows -> dense buffer -> cuBLAS GEMM -> scatter `+=`, one launch
// each, host-synchronised on nbsizes.  Here one launch per pass:
//
//   conv_nbr_kernel   one workgroup = BM consecutive output rows.  For each offset k the
//                     rows that have a neighbour are compacted (wave ballot), their input rows
//                     are gathered HBM/L2 -> LDS in CK-deep slices next to the matching W_k
//                     slice, multiplied with MFMA (M = compacted pairs, so no flops are spent
//                     on missing neighbours beyond padding to 16), and added into an LDS-resident
//                     [BM x C_out] accumulator tile.  Each output row is written to HBM exactly
//                     once: no atomics, no read-modify-write, bit-reproducible.
//                     The same kernel is dgrad (W_k^T, inverse table).
//   conv_wgrad_kernel one workgroup = (offset k, chunk of its pair list, slice of C_out):
//                     dW_k += X[in]^T dY[out] with the pair index as the MFMA reduction
//                     dimension; partial tiles are added with float atomics.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CV_CK 32          // reduction-slice depth staged per step
#define CV_AP (CV_CK + 4)  // LDS row pitch (floats) of K-major tiles: 16-B aligned, bank-skewed

int g_ts_conv_impl = 0;

extern "C" void ts_set_conv_impl(int32_t impl) { g_ts_conv_impl = impl; }

// ======================================================================================
// conv_nbr_kernel
// ======================================================================================
//   X   [n_in][R]            rows to gather (features or output gradients)
//   W   WT == false: [K][R][O_total]   B(kk, col) = W[k][kk][col]
//       WT == true : [K][O_total][R]   B(kk, col) = W[k][col][kk]
//   nbr [K][n_out]           row of X feeding output row j through offset k, or -1
//   Y   [n_out][O_total]     this launch writes columns [o0, o0 + OT)
template <int BM, int MAXU, bool WT>
__global__ __launch_bounds__(256) void conv_nbr_kernel(const float *__restrict__ X, int R,
                                                       const float *__restrict__ W,
                                                       const int *__restrict__ nbr, int64_t n_out, int K,
                                                       float *__restrict__ Y, int O_total, int o_tile,
                                                       int k_per_group) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: conditions on it become s_cbranch, not exec masks
  const int r16 = lane & 15, g = lane >> 4;
  const int o0 = blockIdx.y * o_tile;
  // blockIdx.z = group of kernel offsets handled by this workgroup (small problems are split
  // over offsets to fill the chip; the groups' tiles are then combined with float atomics)
  const int k_begin = blockIdx.z * k_per_group;
  const int k_end = min(K, k_begin + k_per_group);
  const int OT = min(o_tile, O_total - o0);
  const int O16 = (OT + 15) & ~15;
  const int OP = O16 + 4;
  const int NB = O16 >> 4;

  float *outT = smem;                                   // [BM][OP]
  float *At = outT + BM * OP;                           // [BM][CV_AP]
  float *Wt = At + BM * CV_AP;                          // fwd [CV_CK][OP] / dgrad [O16][CV_AP]
  const int wt_floats = WT ? O16 * CV_AP : CV_CK * OP;
  int *list_in = (int *)(Wt + wt_floats);               // [BM]
  int *list_row = list_in + BM;                         // [BM]
  int *wave_cnt = list_row + BM;                        // [4] + m

  const int64_t row0 = (int64_t)blockIdx.x * BM;
  const int nrows = (int)min((int64_t)BM, n_out - row0);

  for (int e = tid; e < BM * OP; e += 256) outT[e] = 0.f;

  // static unit -> (mb, nb) map of this wave: unit u = wave + 4 i, u = mb * NB + nb
  int u_mb[MAXU], u_nb[MAXU];
#pragma unroll
  for (int i = 0; i < MAXU; ++i) {
    int u = wave + 4 * i;
    u_mb[i] = u / NB;
    u_nb[i] = u - u_mb[i] * NB;
  }
  const bool x_vec = ((R & 3) == 0) && ((((uintptr_t)X) & 15) == 0);
  const bool w_vec = WT ? (((R & 3) == 0) && ((((uintptr_t)W) & 15) == 0))
                        : (((O_total & 3) == 0) && ((o0 & 3) == 0) && ((((uintptr_t)W) & 15) == 0));

  for (int k = k_begin; k < k_end; ++k) {
    __syncthreads();  // previous offset fully folded into outT; lists / tiles reusable
    // ---- compact the rows of this tile that have a neighbour at offset k
    int idx = -1, rank = 0;
    if (tid < BM) {
      if (tid < nrows) idx = nbr[(int64_t)k * n_out + row0 + tid];
      unsigned long long mk = __ballot(idx >= 0);
      if (lane == 0) wave_cnt[wave] = (int)__popcll(mk);
      rank = (int)__popcll(mk & ((1ULL << lane) - 1ULL));
    }
    __syncthreads();
    int m = 0;
#pragma unroll
    for (int w = 0; w < BM / 64; ++w) m += wave_cnt[w];
    if (m == 0) continue;  // uniform: every thread reads the same counters after the barrier
    if (idx >= 0) {
      int pos = rank;
      for (int w = 0; w < wave; ++w) pos += wave_cnt[w];
      list_in[pos] = idx;
      list_row[pos] = tid;
    }
    const int M16 = (m + 15) >> 4;
    const int mrows = M16 << 4;

    f32x4 acc[MAXU];
#pragma unroll
    for (int i = 0; i < MAXU; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int c0 = 0; c0 < R; c0 += CV_CK) {
      const int ck = min(CV_CK, R - c0);
      const int ck16 = (ck + 15) & ~15;
      __syncthreads();  // lists visible (first slice) / previous slice's MFMA reads done
      // ---- stage A: gathered rows, zero padded to mrows x ck16
      for (int e = tid; e < mrows * (CV_CK / 4); e += 256) {
        int rr = e >> 3, c4 = (e & 7) << 2;
        if (c4 >= ck16) continue;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rr < m) {
          const float *src = X + (int64_t)list_in[rr] * R + c0 + c4;
          if (x_vec && c4 + 3 < ck) {
            v = *(const float4 *)src;
          } else {
            if (c4 + 0 < ck) v.x = src[0];
            if (c4 + 1 < ck) v.y = src[1];
            if (c4 + 2 < ck) v.z = src[2];
            if (c4 + 3 < ck) v.w = src[3];
          }
        }
        *(float4 *)&At[rr * CV_AP + c4] = v;
      }
      // ---- stage the W_k slice
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
          *(float4 *)&Wt[kk * OP + c4] = v;
        }
      } else {
        const float *wk = W + ((int64_t)k * O_total + o0) * R + c0;
        for (int e = tid; e < O16 * (CV_CK / 4); e += 256) {
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
          *(float4 *)&Wt[col * CV_AP + c4] = v;
        }
      }
      __syncthreads();
      // ---- MFMA: acc[unit] += A[mb] (16 x ck16) * B[nb] (ck16 x 16)
      // k-slot permutation: in step s of a 16-deep block, lane group g supplies reduction
      // index 4 g + s to both operands, so A (and W^T) fragments are single 16-B LDS reads.
      for (int j = 0; j < ck16; j += 16) {
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
          if (u_mb[i] < M16) {
            const float4 a = *(const float4 *)&At[(u_mb[i] * 16 + r16) * CV_AP + j + 4 * g];
            float b0, b1, b2, b3;
            if (!WT) {
              const float *bp = &Wt[(j + 4 * g) * OP + u_nb[i] * 16 + r16];
              b0 = bp[0];
              b1 = bp[OP];
              b2 = bp[2 * OP];
              b3 = bp[3 * OP];
            } else {
              const float4 b = *(const float4 *)&Wt[(u_nb[i] * 16 + r16) * CV_AP + j + 4 * g];
              b0 = b.x;
              b1 = b.y;
              b2 = b.z;
              b3 = b.w;
            }
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b2, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b3, acc[i], 0, 0, 0);
          }
        }
      }
    }
    // ---- fold this offset's products into the output tile (C/D map: col = lane & 15,
    // row = 4 (lane >> 4) + reg).  Within one offset every output row appears at most once
    // and units are disjoint across waves, so plain LDS read-add-write is race free.
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
      if (u_mb[i] < M16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int prow = u_mb[i] * 16 + 4 * g + q;
          if (prow < m) {
            float *dst = &outT[list_row[prow] * OP + u_nb[i] * 16 + r16];
            *dst += acc[i][q];
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- write the tile once, coalesced (or add it, when the offsets were split over workgroups)
  if (gridDim.z > 1) {
    for (int e = tid; e < nrows * OT; e += 256) {
      int rr = e / OT, cc = e - rr * OT;
      float v = outT[rr * OP + cc];
      if (v != 0.f) atomicAdd(&Y[(row0 + rr) * O_total + o0 + cc], v);
    }
    return;
  }
  const bool y_vec = ((O_total & 3) == 0) && ((o0 & 3) == 0) && ((OT & 3) == 0) && ((((uintptr_t)Y) & 15) == 0);
  if (y_vec) {
    const int q4 = OT >> 2;
    for (int e = tid; e < nrows * q4; e += 256) {
      int rr = e / q4, c4 = (e - rr * q4) << 2;
      *(float4 *)&Y[(row0 + rr) * O_total + o0 + c4] = *(const float4 *)&outT[rr * OP + c4];
    }
  } else {
    for (int e = tid; e < nrows * OT; e += 256) {
      int rr = e / OT, cc = e - rr * OT;
      Y[(row0 + rr) * O_total + o0 + cc] = outT[rr * OP + cc];
    }
  }
}

template <int BM, int MAXU, bool WT>
static int launch_conv_nbr(const float *X, int R, const float *W, const int *nbr, int64_t n_out, int K, float *Y,
                           int O_total, int o_tile, int kgroups, hipStream_t stream) {
  const int O16 = (std::min(o_tile, O_total) + 15) & ~15;
  const int OP = O16 + 4;
  size_t lds = (size_t)(BM * OP + BM * CV_AP + (WT ? O16 * CV_AP : CV_CK * OP)) * 4 + (size_t)(2 * BM + 8) * 4;
  TS_REQUIRE(lds <= 160 * 1024, TS_ERR_UNSUPPORTED, "conv_nbr: LDS tile %zu B exceeds 160 KiB", lds);
  TS_REQUIRE((BM / 16) * (O16 / 16) <= 4 * MAXU, TS_ERR_UNSUPPORTED, "conv_nbr: tile does not fit the instantiation");
  auto kern = conv_nbr_kernel<BM, MAXU, WT>;
  static bool attr_set = false;
  if (!attr_set) {
    TS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                 "hipFuncSetAttribute");
    attr_set = true;
  }
  const int kpg = (K + kgroups - 1) / kgroups;
  kgroups = (K + kpg - 1) / kpg;
  if (kgroups > 1) TS_CHECK_HIP(hipMemsetAsync(Y, 0, (size_t)n_out * O_total * 4, stream), "conv_nbr memset");
  dim3 grid((unsigned)ts_cdiv(n_out, BM), (unsigned)ts_cdiv(O_total, o_tile), (unsigned)kgroups);
  kern<<<grid, 256, lds, stream>>>(X, R, W, nbr, n_out, K, Y, O_total, o_tile, kpg);
  TS_CHECK_LAUNCH("conv_nbr");
  return TS_OK;
}

// scalar cross-check: one thread per output element, offsets in order
__global__ __launch_bounds__(256) void conv_nbr_scalar_kernel(const float *__restrict__ X, int R,
                                                              const float *__restrict__ W, int wt,
                                                              const int *__restrict__ nbr, int64_t n_out, int K,
                                                              float *__restrict__ Y, int O) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_out * O) return;
  int64_t j = e / O;
  int o = (int)(e - j * O);
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    int i = nbr[(int64_t)k * n_out + j];
    if (i < 0) continue;
    const float *x = X + (int64_t)i * R;
    float s = 0.f;
    if (!wt) {
      const float *w = W + (int64_t)k * R * O + o;
      for (int r = 0; r < R; ++r) s = fmaf(x[r], w[(int64_t)r * O], s);
    } else {
      const float *w = W + ((int64_t)k * O + o) * R;
      for (int r = 0; r < R; ++r) s = fmaf(x[r], w[r], s);
    }
    acc += s;
  }
  Y[e] = acc;
}

extern "C" int ts_conv_nbr(const float *in_feat, int64_t n_in, int32_t c_in, const float *kernel, int32_t K,
                           int32_t weight_transposed, const int32_t *nbr, float *out_feat, int64_t n_out,
                           int32_t c_out, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n_in >= 0 && n_out >= 0 && c_in > 0 && c_out > 0 && K > 0, TS_ERR_INVALID_ARGUMENT,
             "ts_conv_nbr: bad sizes");
  if (n_out == 0) return TS_OK;
  TS_REQUIRE(kernel && nbr && out_feat && (in_feat || n_in == 0), TS_ERR_INVALID_ARGUMENT, "ts_conv_nbr: null pointer");
  TS_REQUIRE(n_out * (int64_t)c_out < (1LL << 40) && (int64_t)K * n_out < (1LL << 31), TS_ERR_UNSUPPORTED,
             "ts_conv_nbr: problem too large");
  if (g_ts_conv_impl == 1) {
    int64_t total = n_out * c_out;
    conv_nbr_scalar_kernel<<<(unsigned)ts_cdiv(total, 256), 256, 0, stream>>>(
        in_feat, c_in, kernel, weight_transposed ? 1 : 0, nbr, n_out, K, out_feat, c_out);
    TS_CHECK_LAUNCH("conv_nbr_scalar");
    return TS_OK;
  }
  // Tiling heuristic.  A workgroup owns (BM rows) x (o_tile columns) x (a group of offsets):
  //   - large problems: wide column tiles (gathered rows are reused across more columns);
  //   - small problems (deep U-Net levels: a few thousand rows): narrow column tiles and the offsets
  //     split over workgroups, so that >= ~768 workgroups exist and several are resident per CU
  //     (they hide each other's gather latency).
  const int c16 = (c_out + 15) & ~15;
  const int64_t tiles64 = ts_cdiv(n_out, 64);
  int o_tile;
  if (c16 <= 32) o_tile = c16;
  else if (c16 <= 64) o_tile = c16;
  else if (tiles64 * ts_cdiv(c16, 128) >= 1024) o_tile = (c16 % 128 == 0 || c16 > 256) ? 128 : (c16 <= 128 ? c16 : 64);
  else o_tile = (c16 % 64 == 0 || c16 > 128) ? 64 : c16;  // 96 stays one tile
  if (c16 > 64 && c16 <= 128 && c16 % 64 != 0) o_tile = c16;
  const int bm = (o_tile <= 64 && tiles64 >= 2048) ? 128 : 64;
  const int64_t wgs = ts_cdiv(n_out, bm) * ts_cdiv(c_out, o_tile);
  int kgroups = 1;
  if (wgs < 768 && K >= 4) {
    kgroups = (int)std::min<int64_t>(ts_cdiv(768, wgs), K >= 9 ? K / 3 : K / 2);
    if (kgroups < 1) kgroups = 1;
  }
  const bool wt = weight_transposed != 0;
#define TS_LAUNCH(BM, MAXU)                                                                                        \
  (wt ? launch_conv_nbr<BM, MAXU, true>(in_feat, c_in, kernel, nbr, n_out, K, out_feat, c_out, o_tile, kgroups,    \
                                        stream)                                                                    \
      : launch_conv_nbr<BM, MAXU, false>(in_feat, c_in, kernel, nbr, n_out, K, out_feat, c_out, o_tile, kgroups,   \
                                         stream))
  // MAXU = ceil((BM / 16) * (o_tile / 16) / 4)
  if (bm == 128) return o_tile <= 32 ? TS_LAUNCH(128, 4) : TS_LAUNCH(128, 8);
  if (o_tile <= 32) return TS_LAUNCH(64, 2);
  if (o_tile <= 64) return TS_LAUNCH(64, 4);
  if (o_tile <= 128) return TS_LAUNCH(64, 8);
  return TS_LAUNCH(64, 16);
#undef TS_LAUNCH
}

// ======================================================================================
// Reference-form entry points (explicit rulebook + host nbsizes)
// ======================================================================================
struct NbOffsArg {
  int v[66];
};
__global__ void write_nboffs_kernel(NbOffsArg a, int n, int *out) {
  int i = threadIdx.x;
  if (i < n) out[i] = a.v[i];
}

extern "C" size_t ts_convolution_workspace_bytes(int64_t n_in, int64_t n_out, int32_t c_in, int32_t c_out,
                                                 int32_t K) {
  (void)c_in;
  (void)c_out;
  int64_t rows = std::max<int64_t>(n_in, n_out);
  return ts_align_up((size_t)K * (size_t)(rows < 1 ? 1 : rows) * 4, 256) + 512;
}

static int stage_nboffs(const int32_t *nbsizes_host, int K, int *dev, hipStream_t stream, int64_t *max_pairs,
                        int64_t *total) {
  TS_REQUIRE(K <= 64, TS_ERR_UNSUPPORTED, "reference-form convolution supports kernel_volume <= 64");
  NbOffsArg a;
  int64_t run = 0, mx = 0;
  for (int k = 0; k < K; ++k) {
    TS_REQUIRE(nbsizes_host[k] >= 0, TS_ERR_INVALID_ARGUMENT, "negative nbsizes entry");
    a.v[k] = (int)run;
    run += nbsizes_host[k];
    mx = std::max<int64_t>(mx, nbsizes_host[k]);
  }
  TS_REQUIRE(run < (1LL << 31), TS_ERR_UNSUPPORTED, "too many pairs");
  a.v[K] = (int)run;
  write_nboffs_kernel<<<1, 128, 0, stream>>>(a, K + 1, dev);
  TS_CHECK_LAUNCH("write_nboffs");
  *max_pairs = mx;
  *total = run;
  return TS_OK;
}

extern "C" int ts_convolution_forward(const float *in_feat, int64_t n_in, int32_t c_in, float *out_feat,
                                      int64_t n_out, int32_t c_out, const float *kernel, int32_t K,
                                      const int32_t *nbmap, const int32_t *nbsizes_host, int32_t transpose, void *ws,
                                      size_t ws_bytes, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n_in >= 0 && n_out >= 0 && c_in > 0 && c_out > 0 && K > 0, TS_ERR_INVALID_ARGUMENT,
             "ts_convolution_forward: bad sizes");
  TS_REQUIRE(nbsizes_host, TS_ERR_INVALID_ARGUMENT, "ts_convolution_forward: null nbsizes");
  TS_REQUIRE(ws && ws_bytes >= ts_convolution_workspace_bytes(n_in, n_out, c_in, c_out, K), TS_ERR_WORKSPACE_TOO_SMALL,
             "ts_convolution_forward: workspace too small");
  if (n_out == 0) return TS_OK;
  int *nbr = (int *)ws;
  int *nboffs = (int *)((char *)ws + ts_align_up((size_t)K * std::max<int64_t>(std::max(n_in, n_out), 1) * 4, 256));
  int64_t mx, total;
  int rc = stage_nboffs(nbsizes_host, K, nboffs, stream, &mx, &total);
  if (rc != TS_OK) return rc;
  TS_REQUIRE(nbmap || total == 0, TS_ERR_INVALID_ARGUMENT, "ts_convolution_forward: null nbmap");
  if (total == 0) {
    TS_CHECK_HIP(hipMemsetAsync(nbr, 0xFF, (size_t)K * n_out * 4, stream), "nbr memset");
  } else {
    rc = ts_nbr_from_nbmaps(nbmap, nboffs, K, transpose ? 1 : 0, n_out, nbr, stream_);
    if (rc != TS_OK) return rc;
  }
  return ts_conv_nbr(in_feat, n_in, c_in, kernel, K, 0, nbr, out_feat, n_out, c_out, stream_);
}

extern "C" int ts_convolution_backward(const float *in_feat, int64_t n_in, int32_t c_in, float *grad_in,
                                       const float *grad_out, int64_t n_out, int32_t c_out, const float *kernel,
                                       float *grad_kernel, int32_t K, const int32_t *nbmap,
                                       const int32_t *nbsizes_host, int32_t transpose, void *ws, size_t ws_bytes,
                                       ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n_in >= 0 && n_out >= 0 && c_in > 0 && c_out > 0 && K > 0, TS_ERR_INVALID_ARGUMENT,
             "ts_convolution_backward: bad sizes");
  TS_REQUIRE(nbsizes_host && grad_kernel, TS_ERR_INVALID_ARGUMENT, "ts_convolution_backward: null pointer");
  TS_REQUIRE(ws && ws_bytes >= ts_convolution_workspace_bytes(n_in, n_out, c_in, c_out, K), TS_ERR_WORKSPACE_TOO_SMALL,
             "ts_convolution_backward: workspace too small");
  int *nbr = (int *)ws;
  int *nboffs = (int *)((char *)ws + ts_align_up((size_t)K * std::max<int64_t>(std::max(n_in, n_out), 1) * 4, 256));
  int64_t mx, total;
  int rc = stage_nboffs(nbsizes_host, K, nboffs, stream, &mx, &total);
  if (rc != TS_OK) return rc;
  TS_REQUIRE(nbmap || total == 0, TS_ERR_INVALID_ARGUMENT, "ts_convolution_backward: null nbmap");
  if (grad_in && n_in > 0) {
    if (total == 0) {
      TS_CHECK_HIP(hipMemsetAsync(nbr, 0xFF, (size_t)K * n_in * 4, stream), "nbr memset");
    } else {
      // inverse table: rows = input voxels, entries = output voxels
      rc = ts_nbr_from_nbmaps(nbmap, nboffs, K, transpose ? 0 : 1, n_in, nbr, stream_);
      if (rc != TS_OK) return rc;
    }
    rc = ts_conv_nbr(grad_out, n_out, c_out, kernel, K, 1, nbr, grad_in, n_in, c_in, stream_);
    if (rc != TS_OK) return rc;
  }
  (void)mx;
  return ts_conv_wgrad(in_feat, c_in, grad_out, c_out, nbmap, nboffs, K, transpose ? 1 : 0, total, grad_kernel, stream_);
}
