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
#include <stdlib.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PG_BM 128
#define PG_BK 32
#define PG_AP (PG_BK + 4)

// split-bf16 forms of the full-tile kernels (conv_pairs_s.hip); selected unless ts_set_conv_impl asks for the f32 MFMA
int ts_pair_gemm_split(const float *X, int R, const float *W, int O_total, const int2 *nbmaps, const int *nboffs, int K,
                       int64_t P, int gcol, float *Z, int bn, int wt, hipStream_t stream);
int ts_wgrad_split(const float *A, int CA, const float *B, int CB, const int2 *nbmaps, const int *nboffs, int K,
                   int col_a, int64_t n_pairs, float *dW, int tm, int tn, hipStream_t stream);
bool ts_pair_gemm_direct_ok(int bn);
int ts_pair_gemm_direct(const float *X, int R, const unsigned short *planes, int64_t plane_n, int O_total,
                        const int2 *nbmaps, const int *nboffs, int K, int64_t P, int gcol, float *Z, int bn, int wt,
                        hipStream_t stream);

// BN = output columns per workgroup, WR = waves along the pair (row) dimension (WC = 4 / WR along columns).
// Software pipeline: the (offset, C_in-slice) steps of a tile are flattened; the global loads of step s+1 are
// issued into registers right after the barrier of step s and land while its MFMAs run; LDS is double
// buffered so one barrier per step suffices.
struct PgStep {
  int k, s0, s1, c0;
};

template <int BN, int WR, bool WT>
__global__ __launch_bounds__(256, 2) void pair_gemm_kernel(const float *__restrict__ X, int R,
                                                        const float *__restrict__ W, int O_total,
                                                        const int2 *__restrict__ nbmaps,
                                                        const int *__restrict__ nboffs, int K, int64_t P, int gcol,
                                                        float *__restrict__ Z) {
  constexpr int WC = 4 / WR;
  constexpr int MI = (PG_BM / 16) / WR;   // 16-row blocks per wave
  constexpr int NI = (BN / 16) / WC;      // 16-col blocks per wave
  constexpr int BP = BN + 4;              // pitch of the [k][col] weight tile
  constexpr int A_FLOATS = PG_BM * PG_AP;
  constexpr int B_FLOATS = WT ? BN * PG_AP : PG_BK * BP;
  constexpr int A_IT = PG_BM * (PG_BK / 4) / 256;   // float4 per thread per A slice (4)
  constexpr int B_IT = (BN * (PG_BK / 4) + 255) / 256;  // float4 per thread per B slice
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Abuf = smem;                         // 2 x [128][36]
  float *Bbuf = Abuf + 2 * A_FLOATS;          // 2 x (!WT: [32][BN+4] | WT: [BN][36])
  int *rowidx = (int *)(Bbuf + 2 * B_FLOATS);  // [128]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: conditions on it become s_cbranch, not exec masks
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
  auto segment = [&](int k, int &s0, int &s1) {
    s0 = max((int)((int64_t)nboffs[k] - p0), 0);
    s1 = min((int)((int64_t)nboffs[k + 1] - p0), np);
  };
  auto advance = [&](PgStep st) -> PgStep {  // next (offset, slice); k > k_hi when exhausted
    st.c0 += PG_BK;
    if (st.c0 < R) return st;
    st.c0 = 0;
    for (++st.k; st.k <= k_hi; ++st.k) {
      segment(st.k, st.s0, st.s1);
      if (st.s1 > st.s0) break;
    }
    return st;
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool x_vec = ((R & 3) == 0) && ((((uintptr_t)X) & 15) == 0);
  const bool w_vec = WT ? (((R & 3) == 0) && ((((uintptr_t)W) & 15) == 0))
                        : (((O_total & 3) == 0) && ((((uintptr_t)W) & 15) == 0));

  float4 ra0[A_IT], rb0[B_IT], ra1[A_IT], rb1[B_IT];  // two register stages: loads run two steps ahead
  // ---- global -> registers for one step
  auto load_regs = [&](const PgStep &st, float4 (&ra)[A_IT], float4 (&rb)[B_IT]) {
    const int ck = min(PG_BK, R - st.c0);
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int e = tid + it * 256;
      const int rr = e >> 3, c4 = (e & 7) << 2;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rr >= st.s0 && rr < st.s1 && c4 < ck) {
        const float *src = X + (int64_t)rowidx[rr] * R + st.c0 + c4;
        if (x_vec && c4 + 3 < ck) {
          v = *(const float4 *)src;
        } else {
          if (c4 + 0 < ck) v.x = src[0];
          if (c4 + 1 < ck) v.y = src[1];
          if (c4 + 2 < ck) v.z = src[2];
          if (c4 + 3 < ck) v.w = src[3];
        }
      }
      ra[it] = v;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int e = tid + it * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!WT) {
        const int q4 = BN >> 2;
        const int kk = e / q4, c4 = (e - kk * q4) << 2;
        if (kk < ck && c4 < OT) {
          const float *src = W + ((int64_t)st.k * R + st.c0 + kk) * O_total + o0 + c4;
          if (w_vec && c4 + 3 < OT) {
            v = *(const float4 *)src;
          } else {
            if (c4 + 0 < OT) v.x = src[0];
            if (c4 + 1 < OT) v.y = src[1];
            if (c4 + 2 < OT) v.z = src[2];
            if (c4 + 3 < OT) v.w = src[3];
          }
        }
      } else {
        const int col = e >> 3, c4 = (e & 7) << 2;
        if (col < OT && c4 < ck) {
          const float *src = W + ((int64_t)st.k * O_total + o0 + col) * R + st.c0 + c4;
          if (w_vec && c4 + 3 < ck) {
            v = *(const float4 *)src;
          } else {
            if (c4 + 0 < ck) v.x = src[0];
            if (c4 + 1 < ck) v.y = src[1];
            if (c4 + 2 < ck) v.z = src[2];
            if (c4 + 3 < ck) v.w = src[3];
          }
        }
      }
      rb[it] = v;
    }
  };
  // ---- registers -> LDS buffer
  auto store_lds = [&](float *At, float *Bt, const float4 (&ra)[A_IT], const float4 (&rb)[B_IT]) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int e = tid + it * 256;
      *(float4 *)&At[(e >> 3) * PG_AP + ((e & 7) << 2)] = ra[it];
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int e = tid + it * 256;
      if (!WT) {
        const int q4 = BN >> 2;
        const int kk = e / q4, c4 = (e - kk * q4) << 2;
        if (kk < PG_BK) *(float4 *)&Bt[kk * BP + c4] = rb[it];
      } else {
        const int col = e >> 3, c4 = (e & 7) << 2;
        if (col < BN) *(float4 *)&Bt[col * PG_AP + c4] = rb[it];
      }
    }
  };

  // ---- MFMA over one staged slice.  k-slot permutation as in conv.hip: lane group g supplies reduction index
  // 4 g + s in step s, so A (and W^T) fragments are single 16-byte LDS reads.
  auto mma = [&](const float *At, const float *Bt, int ck16) {
    for (int j = 0; j < ck16; j += 16) {
      float a[MI][4], b[NI][4];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int mb = wr * MI + mi;
        const float4 v = *(const float4 *)&At[(mb * 16 + r16) * PG_AP + j + 4 * g];
        a[mi][0] = v.x;
        a[mi][1] = v.y;
        a[mi][2] = v.z;
        a[mi][3] = v.w;
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int nb = wc * NI + ni;
        if (!WT) {
          const float *bp = &Bt[(j + 4 * g) * BP + nb * 16 + r16];
          b[ni][0] = bp[0];
          b[ni][1] = bp[BP];
          b[ni][2] = bp[2 * BP];
          b[ni][3] = bp[3 * BP];
        } else {
          const float4 v = *(const float4 *)&Bt[(nb * 16 + r16) * PG_AP + j + 4 * g];
          b[ni][0] = v.x;
          b[ni][1] = v.y;
          b[ni][2] = v.z;
          b[ni][3] = v.w;
        }
      }
      // No per-block guards in the hot loop (they cost a branch per MFMA): rows outside the offset's
      // segment and columns beyond C_out were staged as zeros, so their blocks just add 0.  Reduction step
      // outermost: consecutive MFMAs hit different accumulators (a dependent v_mfma_f32_16x16x4_f32 chain
      // issues every 40 cycles instead of 32).
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][s4], b[ni][s4], acc[mi][ni], 0, 0, 0);
        }
      }
    }
  };

  // ---- pipeline: the loads of step t are issued at iteration t - 2 (two register stages), written to the LDS
  // buffer t % 2 at iteration t, consumed by the MFMAs of iteration t.  One barrier per step.
  PgStep cur;
  cur.k = k_lo - 1;
  cur.c0 = R;  // so that advance() finds the first non-empty offset
  cur.s0 = cur.s1 = 0;
  __syncthreads();  // rowidx visible
  cur = advance(cur);
  PgStep nxt = advance(cur);
  load_regs(cur, ra0, rb0);
  if (nxt.k <= k_hi) load_regs(nxt, ra1, rb1);
  while (true) {
    {  // even phase: `cur` sits in register stage 0, LDS buffer 0
      store_lds(Abuf, Bbuf, ra0, rb0);
      __syncthreads();
      PgStep nn = nxt;
      if (nxt.k <= k_hi) nn = advance(nxt);
      if (nxt.k <= k_hi && nn.k <= k_hi) load_regs(nn, ra0, rb0);
      mma(Abuf, Bbuf, (min(PG_BK, R - cur.c0) + 15) & ~15);
      cur = nxt;
      nxt = nn;
      if (cur.k > k_hi) break;
    }
    {  // odd phase: register stage 1, LDS buffer 1
      store_lds(Abuf + A_FLOATS, Bbuf + B_FLOATS, ra1, rb1);
      __syncthreads();
      PgStep nn = nxt;
      if (nxt.k <= k_hi) nn = advance(nxt);
      if (nxt.k <= k_hi && nn.k <= k_hi) load_regs(nn, ra1, rb1);
      mma(Abuf + A_FLOATS, Bbuf + B_FLOATS, (min(PG_BK, R - cur.c0) + 15) & ~15);
      cur = nxt;
      nxt = nn;
      if (cur.k > k_hi) break;
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

// Fast path of pass 1 for the shapes every layer but the stem takes: C_in % 32 == 0, C_out % BN == 0, 16-byte
// aligned operands, K <= 63.  Same MFMA tiling and summation order as pair_gemm_kernel (bit-identical Z), but
//  * tiles are cut per offset (tile t of offset k covers pairs nboffs[k] + 128 t ..), so no tile straddles an
//    offset boundary: every workgroup runs exactly C_in / 32 steps.  (With flat 128-pair tiles the ~K straddlers
//    run twice as long as the rest and set the duration of small layers.)  The grid is sized from the upper
//    bound ceil(P / 128) + K; the tile -> (offset, first pair) map comes from the K + 1 prefix sums held in one
//    VGPR (wave scan + ballot), surplus workgroups exit;
//  * the staging code has no per-element guards: a thread keeps the 4 gathered row pointers of its A slots in
//    registers for the whole tile (read straight from the rulebook), every global access is one unconditional
//    16-byte load, rows beyond the tile's pairs are zeroed when they are written to LDS (a select right after the
//    load would make the wave wait for the data before the MFMAs instead of after them).
template <int BN, int WR, bool WT>
__global__ __launch_bounds__(256, 2) void pair_gemm_fast_kernel(const float *__restrict__ X, int R,
                                                             const float *__restrict__ W, int O_total,
                                                             const int2 *__restrict__ nbmaps,
                                                             const int *__restrict__ nboffs, int K, int64_t P,
                                                             int gcol, float *__restrict__ Z) {
  constexpr int WC = 4 / WR;
  constexpr int MI = (PG_BM / 16) / WR;
  constexpr int NI = (BN / 16) / WC;
  constexpr int BP = BN + 4;
  constexpr int A_FLOATS = PG_BM * PG_AP;
  constexpr int B_FLOATS = WT ? BN * PG_AP : PG_BK * BP;
  constexpr int A_IT = PG_BM * (PG_BK / 4) / 256;  // 4
  constexpr int B_IT = BN * (PG_BK / 4) / 256;     // BN / 32, exact for BN in {32, 64, 96, 128}
  static_assert(BN % 32 == 0, "BN must be a multiple of 32");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Abuf = smem;
  float *Bbuf = Abuf + 2 * A_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int wr = wave / WC, wc = wave % WC;
  const int o0 = blockIdx.y * BN;

  // ---- tile -> (offset k, first pair, rows): lane l holds nboffs[l] and the tile count of offset l
  const int offv = nboffs[min(lane, K)];
  const int offn = nboffs[min(lane + 1, K)];
  int incl = lane < K ? (offn - offv + PG_BM - 1) / PG_BM : 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  const int tile = blockIdx.x;
  if (tile >= __builtin_amdgcn_readlane(incl, 63)) return;  // uniform: beyond the last tile
  const int k = __builtin_popcountll(__builtin_amdgcn_ballot_w64(incl <= tile));
  const int t_in_k = tile - (k ? __builtin_amdgcn_readlane(incl, max(k - 1, 0)) : 0);
  const int p0 = __builtin_amdgcn_readlane(offv, k) + t_in_k * PG_BM;
  const int np = min(PG_BM, __builtin_amdgcn_readlane(offv, k + 1) - p0);

  // gathered rows of this thread's A slots: slot it covers tile row (tid >> 3) + 32 it, floats 4 (tid & 7) ..+3
  const int arow0 = tid >> 3, acol = (tid & 7) << 2;
  const float *aptr[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int rr = arow0 + 32 * it;
    const int2 pr = nbmaps[p0 + min(rr, np - 1)];
    aptr[it] = X + (int64_t)(gcol ? pr.y : pr.x) * R + acol;
  }
  // this thread's B slots (element offsets relative to the slice base)
  int boff[B_IT], bdst[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int e = tid + it * 256;
    if (!WT) {
      constexpr int q4 = BN >> 2;
      const int kk = e / q4, c4 = (e - kk * q4) << 2;
      boff[it] = kk * O_total + c4;
      bdst[it] = kk * BP + c4;
    } else {
      const int col = e >> 3, c4 = (e & 7) << 2;
      boff[it] = col * R + c4;
      bdst[it] = col * PG_AP + c4;
    }
  }
  const float *wk = WT ? W + ((int64_t)k * O_total + o0) * R : W + (int64_t)k * R * O_total + o0;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 ra[A_IT], rb[B_IT];  // one register stage: the loads of slice c0 + 32 fly during the MFMAs of slice c0
  auto load_regs = [&](int c0) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) ra[it] = *(const f32x4 *)(aptr[it] + c0);
    const float *wb = WT ? wk + c0 : wk + (int64_t)c0 * O_total;
#pragma unroll
    for (int it = 0; it < B_IT; ++it) rb[it] = *(const f32x4 *)(wb + boff[it]);
  };
  auto store_lds = [&](float *At, float *Bt) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int rr = arow0 + 32 * it;
      *(f32x4 *)&At[rr * PG_AP + acol] = rr < np ? ra[it] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) *(f32x4 *)&Bt[bdst[it]] = rb[it];
  };
  // MFMA over one staged slice.  k-slot permutation as in conv.hip: lane group g supplies reduction index
  // 4 g + s in step s, so A (and W^T) fragments are single 16-byte LDS reads.  Reduction step outermost:
  // consecutive MFMAs hit different accumulators.
  auto mma = [&](const float *At, const float *Bt) {
#pragma unroll
    for (int j = 0; j < PG_BK; j += 16) {
      float a[MI][4], b[NI][4];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const float4 v = *(const float4 *)&At[((wr * MI + mi) * 16 + r16) * PG_AP + j + 4 * g];
        a[mi][0] = v.x;
        a[mi][1] = v.y;
        a[mi][2] = v.z;
        a[mi][3] = v.w;
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int nb = wc * NI + ni;
        if (!WT) {
          const float *bp = &Bt[(j + 4 * g) * BP + nb * 16 + r16];
          b[ni][0] = bp[0];
          b[ni][1] = bp[BP];
          b[ni][2] = bp[2 * BP];
          b[ni][3] = bp[3 * BP];
        } else {
          const float4 v = *(const float4 *)&Bt[(nb * 16 + r16) * PG_AP + j + 4 * g];
          b[ni][0] = v.x;
          b[ni][1] = v.y;
          b[ni][2] = v.z;
          b[ni][3] = v.w;
        }
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][s4], b[ni][s4], acc[mi][ni], 0, 0, 0);
        }
      }
    }
  };

  load_regs(0);
  for (int c0 = 0, t = 0; c0 < R; c0 += PG_BK, ++t) {
    float *At = Abuf + (t & 1) * A_FLOATS, *Bt = Bbuf + (t & 1) * B_FLOATS;
    store_lds(At, Bt);
    __syncthreads();   // LDS is double buffered: one barrier per step
    if (c0 + PG_BK < R) load_regs(c0 + PG_BK);
    mma(At, Bt);
  }
  float *zt = Z + (int64_t)p0 * O_total + o0;
  if (np == PG_BM) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          zt[(int64_t)((wr * MI + mi) * 16 + 4 * g + q) * O_total + (wc * NI + ni) * 16 + r16] = acc[mi][ni][q];
  } else {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = (wr * MI + mi) * 16 + 4 * g + q;
          if (row < np) zt[(int64_t)row * O_total + (wc * NI + ni) * 16 + r16] = acc[mi][ni][q];
        }
  }
}

// Persistent form of pair_gemm_fast_kernel: gridDim.x workgroups walk the tile list with stride gridDim.x and run
// ONE software pipeline across tile boundaries - the rulebook rows of the next tile are fetched while the current
// tile computes, its first slice is loaded during the current tile's last MFMA step, and the Z stores of a
// finished tile drain while the next one is already multiplying.  Same tiles, same MFMA order, same Z.
template <int BN, int WR, bool WT>
__global__ __launch_bounds__(256, 2) void pair_gemm_persist_kernel(const float *__restrict__ X, int R,
                                                                const float *__restrict__ W, int O_total,
                                                                const int2 *__restrict__ nbmaps,
                                                                const int *__restrict__ nboffs, int K, int64_t P,
                                                                int gcol, float *__restrict__ Z) {
  constexpr int WC = 4 / WR;
  constexpr int MI = (PG_BM / 16) / WR;
  constexpr int NI = (BN / 16) / WC;
  constexpr int BP = BN + 4;
  constexpr int A_FLOATS = PG_BM * PG_AP;
  constexpr int B_FLOATS = WT ? BN * PG_AP : PG_BK * BP;
  constexpr int A_IT = PG_BM * (PG_BK / 4) / 256;
  constexpr int B_IT = BN * (PG_BK / 4) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Abuf = smem;
  float *Bbuf = Abuf + 2 * A_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int wr = wave / WC, wc = wave % WC;
  const int o0 = blockIdx.y * BN;

  const int offv = nboffs[min(lane, K)];
  const int offn = nboffs[min(lane + 1, K)];
  int incl = lane < K ? (offn - offv + PG_BM - 1) / PG_BM : 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  const int total = __builtin_amdgcn_readlane(incl, 63);
  int tile = blockIdx.x;
  if (tile >= total) return;
  auto locate = [&](int t, int &k, int &p0, int &np) {
    k = __builtin_popcountll(__builtin_amdgcn_ballot_w64(incl <= t));
    const int t_in_k = t - (k ? __builtin_amdgcn_readlane(incl, max(k - 1, 0)) : 0);
    p0 = __builtin_amdgcn_readlane(offv, k) + t_in_k * PG_BM;
    np = min(PG_BM, __builtin_amdgcn_readlane(offv, k + 1) - p0);
  };

  const int arow0 = tid >> 3, acol = (tid & 7) << 2;
  int boff[B_IT], bdst[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int e = tid + it * 256;
    if (!WT) {
      constexpr int q4 = BN >> 2;
      const int kk = e / q4, c4 = (e - kk * q4) << 2;
      boff[it] = kk * O_total + c4;
      bdst[it] = kk * BP + c4;
    } else {
      const int col = e >> 3, c4 = (e & 7) << 2;
      boff[it] = col * R + c4;
      bdst[it] = col * PG_AP + c4;
    }
  }
  auto weights_of = [&](int k) { return WT ? W + ((int64_t)k * O_total + o0) * R : W + (int64_t)k * R * O_total + o0; };
  auto row_index = [&](int p0, int np, int it) {
    const int2 pr = nbmaps[p0 + min(arow0 + 32 * it, np - 1)];
    return gcol ? pr.y : pr.x;
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 ra[A_IT], rb[B_IT];
  int np_regs;   // rows of the tile whose slice sits in ra / rb
  auto load_regs = [&](const float *const (&ap)[A_IT], const float *wk, int c0) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) ra[it] = *(const f32x4 *)(ap[it] + c0);
    const float *wb = WT ? wk + c0 : wk + (int64_t)c0 * O_total;
#pragma unroll
    for (int it = 0; it < B_IT; ++it) rb[it] = *(const f32x4 *)(wb + boff[it]);
  };
  auto store_lds = [&](float *At, float *Bt) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int rr = arow0 + 32 * it;
      *(f32x4 *)&At[rr * PG_AP + acol] = rr < np_regs ? ra[it] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) *(f32x4 *)&Bt[bdst[it]] = rb[it];
  };
  auto mma = [&](const float *At, const float *Bt) {
#pragma unroll
    for (int j = 0; j < PG_BK; j += 16) {
      float a[MI][4], b[NI][4];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const float4 v = *(const float4 *)&At[((wr * MI + mi) * 16 + r16) * PG_AP + j + 4 * g];
        a[mi][0] = v.x;
        a[mi][1] = v.y;
        a[mi][2] = v.z;
        a[mi][3] = v.w;
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int nb = wc * NI + ni;
        if (!WT) {
          const float *bp = &Bt[(j + 4 * g) * BP + nb * 16 + r16];
          b[ni][0] = bp[0];
          b[ni][1] = bp[BP];
          b[ni][2] = bp[2 * BP];
          b[ni][3] = bp[3 * BP];
        } else {
          const float4 v = *(const float4 *)&Bt[(nb * 16 + r16) * PG_AP + j + 4 * g];
          b[ni][0] = v.x;
          b[ni][1] = v.y;
          b[ni][2] = v.z;
          b[ni][3] = v.w;
        }
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][s4], b[ni][s4], acc[mi][ni], 0, 0, 0);
        }
      }
    }
  };

  int k, p0, np;
  locate(tile, k, p0, np);
  const float *aptr[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) aptr[it] = X + (int64_t)row_index(p0, np, it) * R + acol;
  const float *wk = weights_of(k);
  np_regs = np;
  load_regs(aptr, wk, 0);
  int t = 0;
  for (;;) {
    const int nxt = tile + gridDim.x;
    const bool has_next = nxt < total;
    int kn = 0, p0n = 0, npn = 1, idxn[A_IT];
    if (has_next) {   // rulebook rows of the next tile: in flight during this tile's steps
      locate(nxt, kn, p0n, npn);
#pragma unroll
      for (int it = 0; it < A_IT; ++it) idxn[it] = row_index(p0n, npn, it);
    }
    const float *aptr_n[A_IT];
    const float *wk_n = wk;
    for (int c0 = 0; c0 < R; c0 += PG_BK, ++t) {
      float *At = Abuf + (t & 1) * A_FLOATS, *Bt = Bbuf + (t & 1) * B_FLOATS;
      store_lds(At, Bt);
      __syncthreads();
      if (c0 + PG_BK < R) {
        load_regs(aptr, wk, c0 + PG_BK);
      } else if (has_next) {   // last step of this tile: first slice of the next one
#pragma unroll
        for (int it = 0; it < A_IT; ++it) aptr_n[it] = X + (int64_t)idxn[it] * R + acol;
        wk_n = weights_of(kn);
        np_regs = npn;
        load_regs(aptr_n, wk_n, 0);
      }
      mma(At, Bt);
    }
    // Z rows of the finished tile (the stores drain while the next tile's MFMAs run)
    // (`ot` is laundered through an empty asm so the 64 store offsets are recomputed per tile instead of being
    // hoisted out of the tile loop, where they would occupy 64+ VGPRs for the whole kernel)
    int ot = O_total;
    asm volatile("" : "+s"(ot));
    float *zt = Z + (int64_t)p0 * ot + o0 + (wc * NI) * 16 + r16;
    const int row0 = wr * MI * 16 + 4 * g;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = row0 + mi * 16 + q;
        float *zr = zt + row * ot;
        if (row < np) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) zr[ni * 16] = acc[mi][ni][q];
        }
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (!has_next) break;
    tile = nxt;
    k = kn;
    p0 = p0n;
    np = npn;
    wk = wk_n;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) aptr[it] = aptr_n[it];
  }
}

template <int BN, int WR, bool WT>
static int launch_pair_gemm(const float *X, int R, const float *W, int O_total, const int2 *nbmaps, const int *nboffs,
                            int K, int64_t P, int gcol, float *Z, hipStream_t stream) {
  size_t lds = (size_t)2 * (PG_BM * PG_AP + (WT ? BN * PG_AP : PG_BK * (BN + 4))) * 4 + PG_BM * 4;
  auto kern = pair_gemm_kernel<BN, WR, WT>;
  static bool attr_set = false;
  if (!attr_set) {
    TS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                 "hipFuncSetAttribute");
    attr_set = true;
  }
  dim3 grid((unsigned)ts_cdiv(P, PG_BM), (unsigned)ts_cdiv(O_total, BN));
  const bool fast = (R % PG_BK == 0) && (O_total % BN == 0) && ((((uintptr_t)X) | ((uintptr_t)W)) & 15) == 0 &&
                    K <= 63 && g_ts_conv_impl != 2;
  if (fast && (g_ts_conv_impl == 0 || (g_ts_conv_impl >= 6 && g_ts_conv_impl <= 8) || g_ts_conv_impl == 11 || g_ts_conv_impl == 13))   // default: fp32 operands on the bf16 matrix pipe (conv_pairs_s.hip)
    return ts_pair_gemm_split(X, R, W, O_total, nbmaps, nboffs, K, P, gcol, Z, BN, WT ? 1 : 0, stream);
  if (fast) {
    static bool fattr_set = false;
    if (!fattr_set) {
      TS_CHECK_HIP(hipFuncSetAttribute((const void *)pair_gemm_fast_kernel<BN, WR, WT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "hipFuncSetAttribute");
      fattr_set = true;
    }
    dim3 fgrid((unsigned)(ts_cdiv(P, PG_BM) + K), grid.y);   // upper bound on sum_k ceil(n_k / 128)
    // measured on the MinkUNet layer shapes (tools/ab_pair_gemm.sh): the persistent pipeline wins 3-9 % with the
    // forward weight layout and loses with the transposed one (256 VGPRs there) -> forward only.
    // ts_set_conv_impl(3) = one workgroup per tile everywhere, (4) = persistent everywhere.
    if ((!WT && g_ts_conv_impl != 3) || g_ts_conv_impl == 4) {
      static bool pattr_set = false;
      if (!pattr_set) {
        TS_CHECK_HIP(hipFuncSetAttribute((const void *)pair_gemm_persist_kernel<BN, WR, WT>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "hipFuncSetAttribute");
        pattr_set = true;
      }
      dim3 pgrid(std::min<unsigned>(fgrid.x, std::max(1u, 512u / grid.y)), grid.y);   // 2 workgroups per CU
      pair_gemm_persist_kernel<BN, WR, WT><<<pgrid, 256, lds, stream>>>(X, R, W, O_total, nbmaps, nboffs, K, P, gcol, Z);
    } else
    pair_gemm_fast_kernel<BN, WR, WT><<<fgrid, 256, lds, stream>>>(X, R, W, O_total, nbmaps, nboffs, K, P, gcol, Z);
  } else {
    pair_gemm_kernel<BN, WR, WT><<<grid, 256, lds, stream>>>(X, R, W, O_total, nbmaps, nboffs, K, P, gcol, Z);
  }
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
  {   // pre-split planes of this weight left by the caller (one-shot, common.h): the direct-rows kernel where it wins
    const TsPlanesHint hint = g_ts_planes_hint;
    g_ts_planes_hint = TsPlanesHint{nullptr, nullptr, 0, 0, 0};
    const int bn = c_out <= 32 ? 32 : c_out <= 64 ? 64 : c_out % 96 == 0 ? 96 : 128;
    const bool shapes = weight_transposed ? (hint.c_in == c_out && hint.c_out == c_in) : (hint.c_in == c_in && hint.c_out == c_out);
    if (hint.w == kernel && hint.planes && hint.K == K && shapes && K <= 63 && c_in % 32 == 0 && c_out % bn == 0 &&
        ((((uintptr_t)feat) | ((uintptr_t)z) | ((uintptr_t)hint.planes)) & 15) == 0 && ts_pair_gemm_direct_ok(bn)) {
      const int64_t n = (int64_t)K * c_in * c_out;
      return ts_pair_gemm_direct(feat, c_in, hint.planes, n, c_out, nm, nboffs, K, n_pairs, gc, z, bn,
                                 weight_transposed ? 1 : 0, stream);
    }
  }
#define TS_PG(BN, WR)                                                                                             \
  (weight_transposed                                                                                              \
       ? launch_pair_gemm<BN, WR, true>(feat, c_in, kernel, c_out, nm, nboffs, K, n_pairs, gc, z, stream)         \
       : launch_pair_gemm<BN, WR, false>(feat, c_in, kernel, c_out, nm, nboffs, K, n_pairs, gc, z, stream))
  if (c_out <= 32) return TS_PG(32, 4);
  if (c_out <= 64) return TS_PG(64, 2);
  if (c_out % 96 == 0) return TS_PG(96, 2);  // 96 / 192 / 384 channels: three 16-col blocks per wave, no padding
  return TS_PG(128, 2);
#undef TS_PG
}

// ------------------------------------------------------------------------------------- pass 2
// KT > 0: kernel volume known at compile time (27, 8): all K position loads are issued first, then all row
// loads - up to K independent 16-byte loads in flight per lane.  KT == 0: generic loop.
// every Z row is read exactly once: non-temporal loads keep it out of the way of the position table and of the next
// kernel's operands in L2 (step 19.33 -> 19.10 ms)
typedef float ts_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ts_zload(const float4 *p) {
  const ts_f32x4 v = __builtin_nontemporal_load((const ts_f32x4 *)p);
  return make_float4(v[0], v[1], v[2], v[3]);
}
#define TS_ZLOAD(ptr) ts_zload(ptr)
template <int VEC, int KT>
__global__ __launch_bounds__(256) void gather_sum_kernel(const float *__restrict__ Z, int C,
                                                         const int *__restrict__ pos, int K, int64_t n,
                                                         int64_t n_pairs, float *__restrict__ out,
                                                         TsWgradReduce side, const float *__restrict__ addend) {
  const int cv = C / VEC;
  const int64_t total = n * cv;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  // side job (common.h): the ordered sum of the weight-gradient partials of the launch before this one
  for (int64_t i = e; i < (int64_t)side.K * side.cacb4; i += step) ts_wgrad_reduce_one(side, i);
  for (; e < total; e += step) {
    const int64_t j = e / cv;
    const int c = (int)(e - j * cv) * VEC;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    if (KT > 0 && VEC == 4) {
      int p[KT > 0 ? KT : 1];
#pragma unroll
      for (int k = 0; k < KT; ++k) p[k] = pos[(int64_t)k * n + j];
      float4 f[KT > 0 ? KT : 1];
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        f[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p[k] >= 0 && p[k] < n_pairs) f[k] = TS_ZLOAD((const float4 *)(Z + (int64_t)p[k] * C + c));
      }
#pragma unroll
      for (int k = 0; k < KT; ++k) {  // k ascending: fixed summation order
        acc[0] += f[k].x;
        acc[1 % VEC] += f[k].y;
        acc[2 % VEC] += f[k].z;
        acc[3 % VEC] += f[k].w;
      }
    } else {
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
    }
    if (addend) {          // one add per element after the sum over the offsets: the same bits as a separate a + b
      if (VEC == 4) {
        const float4 a = *(const float4 *)(addend + j * C + c);
        acc[0] += a.x;
        acc[1 % VEC] += a.y;
        acc[2 % VEC] += a.z;
        acc[3 % VEC] += a.w;
      } else {
        acc[0] += addend[j * C + c];
      }
    }
    if (VEC == 4) {
      *(float4 *)(out + j * C + c) = make_float4(acc[0], acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
    } else {
      out[j * C + c] = acc[0];
    }
  }
}

// Pass 2 with the live positions of a row compacted in LDS first: a workgroup owns 256 / (C / 4) whole rows; half-waves
// load the K positions of a row (lane = offset), ballot the live ones and leave them - ascending offset, the fixed
// summation order - as a list in LDS; then a lane walks its row's list with R independent 16-byte loads per round.
// Against gather_sum_kernel<4, K>: one position load per lane instead of K, only live rows requested, R + a few instead
// of 5 K registers (8 instead of 3 waves per SIMD).  Same additions in the same order: bit-identical sums.  K <= 32.
template <int R>
__global__ __launch_bounds__(256) void gather_list_kernel(const float *__restrict__ Z, int C,
                                                          const int *__restrict__ pos, int K, int64_t n,
                                                          int64_t n_pairs, float *__restrict__ out, TsWgradReduce side,
                                                          const float *__restrict__ addend, int rpw, TsGatherEpilogue epi) {
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
    const bool live = p >= 0 && p < n_pairs;
    const unsigned long long m64 = __builtin_amdgcn_ballot_w64(live);
    const unsigned m = (tid & 32) ? (unsigned)(m64 >> 32) : (unsigned)m64;
    if (live) lst[r][__builtin_popcount(m & ((1u << l) - 1u))] = p;
    if (l == 0 && r < rpw) cnt[r] = __builtin_popcount(m);
  }
  __syncthreads();
  const int cv = C >> 2;
  const int r = tid / cv;
  const int64_t j = j0 + r;
  if (r >= rpw || j >= n) return;
  const int c = (tid - r * cv) << 2;
  const int m = cnt[r];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b = 0; b < m; b += R) {
    float4 f[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      f[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b + i < m) f[i] = TS_ZLOAD((const float4 *)(Z + (int64_t)lst[r][b + i] * C + c));
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      acc.x += f[i].x;
      acc.y += f[i].y;
      acc.z += f[i].z;
      acc.w += f[i].w;
    }
  }
  if (addend) {
    const float4 a = *(const float4 *)(addend + j * C + c);
    acc.x += a.x;
    acc.y += a.y;
    acc.z += a.z;
    acc.w += a.w;
  }
  if (epi.mean) {                        // evaluation block: bn_act_fwd_kernel's arithmetic on the sum
    const float4 m = *(const float4 *)(epi.mean + c), s = *(const float4 *)(epi.invstd + c);
    const float4 ww = *(const float4 *)(epi.w + c), bb = *(const float4 *)(epi.b + c);
    acc.x = (acc.x - m.x) * s.x * ww.x + bb.x;
    acc.y = (acc.y - m.y) * s.y * ww.y + bb.y;
    acc.z = (acc.z - m.z) * s.z * ww.z + bb.z;
    acc.w = (acc.w - m.w) * s.w * ww.w + bb.w;
    if (epi.residual) {
      const float4 r = *(const float4 *)((const float *)epi.residual + j * C + c);
      acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
    }
    if (epi.relu) {
      acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    }
  }
  *(float4 *)(out + j * C + c) = acc;
}

static int launch_gather_list(const float *z, int c, const int *pos, int K, int64_t n_rows, int64_t n_pairs, float *out,
                              const TsWgradReduce &side, const float *addend, hipStream_t stream,
                              const TsGatherEpilogue &epi = TsGatherEpilogue{nullptr, nullptr, nullptr, nullptr, nullptr, 0}) {
  const int cv = c >> 2, rpw = 256 / cv;
  const unsigned grid = (unsigned)ts_cdiv(n_rows, rpw);
  // R = 4 / 8 / 12 / 16 measured within 3 % of each other on every layer (profiles/r02_v9_gather_forms_probe.txt)
  gather_list_kernel<8><<<grid, 256, 0, stream>>>(z, c, pos, K, n_rows, n_pairs, out, side, addend, rpw, epi);
  TS_CHECK_LAUNCH("conv_gather_sum (list)");
  return TS_OK;
}

extern "C" int ts_conv_gather_sum(const float *z, int32_t c, const int32_t *pos, int32_t K, int64_t n_rows,
                                  int64_t n_pairs, float *out, ts_stream_t stream_) {
  return ts_conv_gather_sum_ex(z, c, pos, K, n_rows, n_pairs, out, nullptr, nullptr, stream_);
}

int ts_conv_gather_sum_ex(const float *z, int32_t c, const int32_t *pos, int32_t K, int64_t n_rows, int64_t n_pairs,
                          float *out, const TsWgradReduce *side_job, const float *addend, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TsWgradReduce side = {};
  if (side_job) side = *side_job;
  TS_REQUIRE(!side_job || (n_rows > 0 && side.part && side.dW && side.nboffs && side.chunk > 0 &&
                           ((((uintptr_t)side.part) | ((uintptr_t)side.dW)) & 15) == 0),
             TS_ERR_INVALID_ARGUMENT, "ts_conv_gather_sum: bad side job");
  TS_REQUIRE(c > 0 && K > 0 && n_rows >= 0 && n_pairs >= 0, TS_ERR_INVALID_ARGUMENT, "ts_conv_gather_sum: bad sizes");
  if (n_rows == 0) return TS_OK;
  TS_REQUIRE(pos && out && (z || n_pairs == 0), TS_ERR_INVALID_ARGUMENT, "ts_conv_gather_sum: null pointer");
  const bool vec = (c % 4 == 0) && ((((uintptr_t)z) & 15) == 0) && ((((uintptr_t)out) & 15) == 0);
  // default: live positions compacted in LDS first (gather_list_kernel); TASEG_GATHER_POSITIONS=1 in the environment
  // keeps the K-register form (A/B runs)
  const bool k_registers = ts_get_option(TS_OPT_GATHER_POSITIONS) != 0;
  if (vec && K <= 32 && c >= 16 && c <= 1024 && !k_registers && g_ts_conv_impl != 1)
    return launch_gather_list(z, c, pos, K, n_rows, n_pairs, out, side, addend, stream);
  if (vec) {
    int grid = (int)std::min<int64_t>(ts_cdiv(n_rows * (c / 4), 256), 1 << 20);
    if (K == 27)
      gather_sum_kernel<4, 27><<<grid, 256, 0, stream>>>(z, c, pos, K, n_rows, n_pairs, out, side, addend);
    else if (K == 8)
      gather_sum_kernel<4, 8><<<grid, 256, 0, stream>>>(z, c, pos, K, n_rows, n_pairs, out, side, addend);
    else
      gather_sum_kernel<4, 0><<<grid, 256, 0, stream>>>(z, c, pos, K, n_rows, n_pairs, out, side, addend);
  } else {
    int grid = (int)std::min<int64_t>(ts_cdiv(n_rows * c, 256), 1 << 20);
    gather_sum_kernel<1, 0><<<grid, 256, 0, stream>>>(z, c, pos, K, n_rows, n_pairs, out, side, addend);
  }
  TS_CHECK_LAUNCH("conv_gather_sum");
  return TS_OK;
}

int ts_conv_gather_sum_epi(const float *z, int32_t c, const int32_t *pos, int32_t K, int64_t n_rows, int64_t n_pairs, float *out,
                           const TsGatherEpilogue &epi, ts_stream_t stream_) {
  const bool k_registers = ts_get_option(TS_OPT_GATHER_POSITIONS) != 0;
  const bool vec = (c % 4 == 0) && ((((uintptr_t)z) | ((uintptr_t)out) | ((uintptr_t)epi.residual) | ((uintptr_t)epi.mean) |
                                     ((uintptr_t)epi.invstd) | ((uintptr_t)epi.w) | ((uintptr_t)epi.b)) & 15) == 0;
  if (!(vec && K > 0 && K <= 32 && c >= 16 && c <= 1024 && !k_registers && g_ts_conv_impl != 1 && n_rows > 0 && z && pos && out &&
        epi.mean && epi.invstd && epi.w && epi.b))
    return TS_ERR_UNSUPPORTED;
  return launch_gather_list(z, c, pos, K, n_rows, n_pairs, out, TsWgradReduce{}, nullptr, (hipStream_t)stream_, epi);
}

// ------------------------------------------------------------------------------------- weight gradient
//   dW_k[ci, co] = sum_{pairs p of k}  A[pa_p, ci] * B[pb_p, co]
// One workgroup = (chunk of consecutive rulebook pairs, TM x TN tile of dW): a "TN" GEMM whose reduction runs
// over the pair list.  Chunks are cut from the flat list (equal work per workgroup; the grid is sized from P),
// a chunk that crosses an offset boundary flushes its accumulators there.  The chunk's pair indices are staged
// in LDS once; both operands are gathered rows, staged 32 pairs at a time; each wave owns a (TM/2) x (TN/2)
// register tile so every LDS fragment feeds TM/32 or TN/32 MFMAs.  Partial tiles are combined with float
// atomics (dW is small: atomic bytes are 1 / (chunk / 2) of the flops).
#define WG_PS 32
#define WG_MAXCHUNK 1024

template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void wgrad_gemm_kernel(const float *__restrict__ A, int CA,
                                                         const float *__restrict__ B, int CB,
                                                         const int2 *__restrict__ nbmaps,
                                                         const int *__restrict__ nboffs, int K, int P, int col_a,
                                                         int chunk, float *__restrict__ dW, float *__restrict__ part) {
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int XP = TM + 4, YP = TN + 4;
  __shared__ __attribute__((aligned(16))) float Xl[2 * WG_PS * XP];  // double buffered
  __shared__ __attribute__((aligned(16))) float Yl[2 * WG_PS * YP];
  __shared__ int idxA[WG_MAXCHUNK], idxB[WG_MAXCHUNK];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: conditions on it become s_cbranch, not exec masks
  const int r16 = lane & 15, g = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int p_beg = blockIdx.x * chunk;
  const int p_end = min(P, p_beg + chunk);
  if (p_beg >= p_end) return;  // uniform
  const int tiles_n = (CB + TN - 1) / TN;
  const int ci0 = (blockIdx.y / tiles_n) * TM, co0 = (blockIdx.y % tiles_n) * TN;
  const int ca = min(TM, CA - ci0), cb = min(TN, CB - co0);

  for (int t = tid; t < p_end - p_beg; t += 256) {
    const int2 pr = nbmaps[p_beg + t];
    idxA[t] = col_a ? pr.y : pr.x;
    idxB[t] = col_a ? pr.x : pr.y;
  }
  int k = 0;
  for (int kk = 0; kk < K; ++kk)
    if (nboffs[kk] <= p_beg) k = kk;  // offset containing the first pair (uniform scalar scan)

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool a_vec = ((CA & 3) == 0) && ((ci0 & 3) == 0) && ((((uintptr_t)A) & 15) == 0);
  const bool b_vec = ((CB & 3) == 0) && ((co0 & 3) == 0) && ((((uintptr_t)B) & 15) == 0);

  // flattened (offset, 32-pair step) sequence of this chunk; k == K when exhausted
  struct WStep {
    int k, p0, np;
  };
  auto advance = [&](WStep st) -> WStep {
    int kend = min(nboffs[st.k + 1], p_end);
    int pn = st.p0 + WG_PS;
    if (pn < kend) {
      st.p0 = pn;
      st.np = min(WG_PS, kend - pn);
      return st;
    }
    pn = kend;
    for (++st.k; st.k < K && pn < p_end; ++st.k) {
      kend = min(nboffs[st.k + 1], p_end);
      if (kend > pn) {
        st.p0 = pn;
        st.np = min(WG_PS, kend - pn);
        return st;
      }
    }
    st.k = K;
    return st;
  };
  constexpr int A_IT = WG_PS * (TM / 4) / 256, B_IT = WG_PS * (TN / 4) / 256;
  float4 ra[A_IT > 0 ? A_IT : 1], rb[B_IT > 0 ? B_IT : 1];
  auto load_regs = [&](const WStep &st) {
    const int l0 = st.p0 - p_beg;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int e = tid + it * 256;
      const int pp = e / (TM / 4), c4 = (e - pp * (TM / 4)) << 2;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pp < st.np && c4 < ca) {
        const float *src = A + (int64_t)idxA[l0 + pp] * CA + ci0 + c4;
        if (a_vec && c4 + 3 < ca) {
          v = *(const float4 *)src;
        } else {
          if (c4 + 0 < ca) v.x = src[0];
          if (c4 + 1 < ca) v.y = src[1];
          if (c4 + 2 < ca) v.z = src[2];
          if (c4 + 3 < ca) v.w = src[3];
        }
      }
      ra[it] = v;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int e = tid + it * 256;
      const int pp = e / (TN / 4), c4 = (e - pp * (TN / 4)) << 2;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pp < st.np && c4 < cb) {
        const float *src = B + (int64_t)idxB[l0 + pp] * CB + co0 + c4;
        if (b_vec && c4 + 3 < cb) {
          v = *(const float4 *)src;
        } else {
          if (c4 + 0 < cb) v.x = src[0];
          if (c4 + 1 < cb) v.y = src[1];
          if (c4 + 2 < cb) v.z = src[2];
          if (c4 + 3 < cb) v.w = src[3];
        }
      }
      rb[it] = v;
    }
  };
  auto store_lds = [&](float *xl, float *yl) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int e = tid + it * 256;
      const int pp = e / (TM / 4), c4 = (e - pp * (TM / 4)) << 2;
      *(float4 *)&xl[pp * XP + c4] = ra[it];
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int e = tid + it * 256;
      const int pp = e / (TN / 4), c4 = (e - pp * (TN / 4)) << 2;
      *(float4 *)&yl[pp * YP + c4] = rb[it];
    }
  };

  WStep cur;
  cur.k = k;
  cur.p0 = p_beg;
  cur.np = min(WG_PS, min(nboffs[k + 1], p_end) - p_beg);
  __syncthreads();  // pair indices visible
  load_regs(cur);
  int buf = 0;
  while (cur.k < K) {
    float *xl = Xl + buf * (WG_PS * XP), *yl = Yl + buf * (WG_PS * YP);
    store_lds(xl, yl);
    __syncthreads();
    const WStep nxt = advance(cur);
    if (nxt.k < K) load_regs(nxt);  // lands while this step's MFMAs run
#pragma unroll
    for (int j = 0; j < WG_PS; j += 16) {
      float a[MI][4], b[NI][4];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const float *ap = &xl[(j + 4 * g) * XP + (wr * MI + mi) * 16 + r16];
        a[mi][0] = ap[0];
        a[mi][1] = ap[XP];
        a[mi][2] = ap[2 * XP];
        a[mi][3] = ap[3 * XP];
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const float *bp = &yl[(j + 4 * g) * YP + (wc * NI + ni) * 16 + r16];
        b[ni][0] = bp[0];
        b[ni][1] = bp[YP];
        b[ni][2] = bp[2 * YP];
        b[ni][3] = bp[3 * YP];
      }
      // reduction step outermost (independent accumulators back to back); no per-block guards: channels
      // beyond C_a / C_b were staged as zeros
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][s], b[ni][s], acc[mi][ni], 0, 0, 0);
        }
      }
    }
    if (nxt.k != cur.k) {  // last step of this offset inside the chunk: flush its partial tile
      float *dwk = part ? part + ((int64_t)blockIdx.x + cur.k) * CA * CB : dW + (int64_t)cur.k * CA * CB;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int co = (wc * NI + ni) * 16 + r16;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ci = (wr * MI + mi) * 16 + 4 * g + q;
            if (ci < ca && co < cb) {
              float *dst = &dwk[(int64_t)(ci0 + ci) * CB + co0 + co];
              if (part)
                *dst = acc[mi][ni][q];      // slot (chunk + offset) of the partial buffer: plain store, summed in order later
              else
                atomicAdd(dst, acc[mi][ni][q]);
            }
          }
          acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    cur = nxt;
    buf ^= 1;
  }
}

// Fast path of the weight gradient for full tiles (C_a % TM == 0, C_b % TN == 0, 16-byte aligned rows): same
// chunking, staging order and MFMA order as wgrad_gemm_kernel, without per-element guards - every gathered row
// slice is one unconditional 16-byte load (pair slots beyond a short step re-read the step's last pair and are
// zeroed when they are written to LDS), the flush is unguarded.
template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void wgrad_gemm_fast_kernel(const float *__restrict__ A, int CA,
                                                              const float *__restrict__ B, int CB,
                                                              const int2 *__restrict__ nbmaps,
                                                              const int *__restrict__ nboffs, int K, int P,
                                                              int col_a, int chunk, float *__restrict__ dW,
                                                              float *__restrict__ part) {
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int XP = TM + 4, YP = TN + 4;
  constexpr int A_IT = TM / 32, B_IT = TN / 32;  // float4 slots per thread per 32-pair step
  __shared__ __attribute__((aligned(16))) float Xl[2 * WG_PS * XP];
  __shared__ __attribute__((aligned(16))) float Yl[2 * WG_PS * YP];
  __shared__ int idxA[WG_MAXCHUNK], idxB[WG_MAXCHUNK];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int p_beg = blockIdx.x * chunk;
  const int p_end = min(P, p_beg + chunk);
  if (p_beg >= p_end) return;  // uniform
  const int tiles_n = CB / TN;
  const int ci0 = (blockIdx.y / tiles_n) * TM, co0 = (blockIdx.y % tiles_n) * TN;

  for (int t = tid; t < p_end - p_beg; t += 256) {
    const int2 pr = nbmaps[p_beg + t];
    idxA[t] = col_a ? pr.y : pr.x;
    idxB[t] = col_a ? pr.x : pr.y;
  }
  // prefix sums in one VGPR, offset of the first pair by ballot (see pair_gemm_fast_kernel)
  const int offv = nboffs[min(lane, K)];
  const int k = __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < K && offv <= p_beg)) - 1;
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
    int pn = st.p0 + WG_PS;
    if (pn < kend) {
      st.p0 = pn;
      st.np = min(WG_PS, kend - pn);
      return st;
    }
    pn = kend;
    for (++st.k; st.k < K && pn < p_end; ++st.k) {
      kend = min(off_at(st.k + 1), p_end);
      if (kend > pn) {
        st.p0 = pn;
        st.np = min(WG_PS, kend - pn);
        return st;
      }
    }
    st.k = K;
    return st;
  };
  const float *abase = A + ci0, *bbase = B + co0;
  f32x4 ra[A_IT], rb[B_IT];
  auto load_regs = [&](const WStep &st) {
    const int l0 = st.p0 - p_beg, last = st.np - 1;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int e = tid + it * 256;
      const int pp = e / (TM / 4), c4 = (e - pp * (TM / 4)) << 2;
      ra[it] = *(const f32x4 *)(abase + (int64_t)idxA[l0 + min(pp, last)] * CA + c4);
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int e = tid + it * 256;
      const int pp = e / (TN / 4), c4 = (e - pp * (TN / 4)) << 2;
      rb[it] = *(const f32x4 *)(bbase + (int64_t)idxB[l0 + min(pp, last)] * CB + c4);
    }
  };
  auto store_lds = [&](float *xl, float *yl, const WStep &st) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int e = tid + it * 256;
      const int pp = e / (TM / 4), c4 = (e - pp * (TM / 4)) << 2;
      *(f32x4 *)&xl[pp * XP + c4] = pp < st.np ? ra[it] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int e = tid + it * 256;
      const int pp = e / (TN / 4), c4 = (e - pp * (TN / 4)) << 2;
      *(f32x4 *)&yl[pp * YP + c4] = pp < st.np ? rb[it] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };

  WStep cur;
  cur.k = k;
  cur.p0 = p_beg;
  cur.np = min(WG_PS, min(off_at(k + 1), p_end) - p_beg);
  __syncthreads();  // pair indices visible
  load_regs(cur);
  int buf = 0;
  while (cur.k < K) {
    float *xl = Xl + buf * (WG_PS * XP), *yl = Yl + buf * (WG_PS * YP);
    store_lds(xl, yl, cur);
    __syncthreads();
    const WStep nxt = advance(cur);
    if (nxt.k < K) load_regs(nxt);
#pragma unroll
    for (int j = 0; j < WG_PS; j += 16) {
      float a[MI][4], b[NI][4];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const float *ap = &xl[(j + 4 * g) * XP + (wr * MI + mi) * 16 + r16];
        a[mi][0] = ap[0];
        a[mi][1] = ap[XP];
        a[mi][2] = ap[2 * XP];
        a[mi][3] = ap[3 * XP];
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const float *bp = &yl[(j + 4 * g) * YP + (wc * NI + ni) * 16 + r16];
        b[ni][0] = bp[0];
        b[ni][1] = bp[YP];
        b[ni][2] = bp[2 * YP];
        b[ni][3] = bp[3 * YP];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][s], b[ni][s], acc[mi][ni], 0, 0, 0);
        }
      }
    }
    if (nxt.k != cur.k) {
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
static int launch_wgrad(const float *A, int CA, const float *B, int CB, const int2 *nbmaps, const int *nboffs, int K,
                        int col_a, int64_t n_pairs, float *dW, hipStream_t stream) {
  const int tiles = (int)(ts_cdiv(CA, TM) * ts_cdiv(CB, TN));
  // ~512 workgroups over the launch (one resident round at 2 per CU): every workgroup ends with a TM x TN tile of
  // float atomics (64 KB at 128 x 128), so chunks must be long - 128 .. 1024 pairs, multiples of the 32-pair step
  const TsWgradPlan plan = ts_wgrad_plan(n_pairs, tiles, K, WG_PS, WG_MAXCHUNK);
  g_ts_wgrad_plan = plan;
  const int64_t chunk = plan.chunk;
  dim3 grid((unsigned)plan.n_chunks, tiles);
  float *part = g_ts_wgrad_part;
  const bool fast = (CA % TM == 0) && (CB % TN == 0) && ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0 &&
                    K <= 63 && g_ts_conv_impl != 2;
  if (fast && (g_ts_conv_impl == 0 || (g_ts_conv_impl >= 6 && g_ts_conv_impl <= 8)))
    return ts_wgrad_split(A, CA, B, CB, nbmaps, nboffs, K, col_a, n_pairs, dW, TM, TN, stream);
  if (fast)
    wgrad_gemm_fast_kernel<TM, TN><<<grid, 256, 0, stream>>>(A, CA, B, CB, nbmaps, nboffs, K, (int)n_pairs, col_a,
                                                            (int)chunk, dW, part);
  else
    wgrad_gemm_kernel<TM, TN><<<grid, 256, 0, stream>>>(A, CA, B, CB, nbmaps, nboffs, K, (int)n_pairs, col_a,
                                                       (int)chunk, dW, part);
  TS_CHECK_LAUNCH("conv_wgrad");
  return TS_OK;
}

thread_local float *g_ts_wgrad_part = nullptr;
thread_local TsWgradPlan g_ts_wgrad_plan = {0, 0, 0};

// Stand-alone ordered sum (1x1x1 convolutions, the class head, blocks without an input gradient): few output elements
// (K * C_a * C_b / 4 float4 granules) but up to ~500 partial tiles each, so four thread groups sum a quarter of the
// chunk range each and the quarters are added in a fixed order - still one summation order per element.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(TsWgradReduce job) {
  __shared__ float4 red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + tx;
  const bool live = i < (int64_t)job.K * job.cacb4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    const int k = (int)(i / job.cacb4);
    const int64_t e = i - (int64_t)k * job.cacb4;
    const int lo = job.nboffs[k], hi = job.nboffs[k + 1];
    if (hi > lo) {
      const int c0 = lo / job.chunk, c1 = (hi - 1) / job.chunk;
      const int per = (c1 - c0 + 4) >> 2;
      const int a = c0 + ty * per, b = min(c1 + 1, a + per);
      const float4 *src = (const float4 *)job.part + ((int64_t)a + k) * job.cacb4 + e;
#pragma unroll 4
      for (int c = a; c < b; ++c, src += job.cacb4) {
        const float4 v = *src;
        acc.x += v.x;
        acc.y += v.y;
        acc.z += v.z;
        acc.w += v.w;
      }
    }
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && live) {
    float4 r = red[0][tx];
#pragma unroll
    for (int y = 1; y < 4; ++y) {
      r.x += red[y][tx].x;
      r.y += red[y][tx].y;
      r.z += red[y][tx].z;
      r.w += red[y][tx].w;
    }
    ((float4 *)job.dW)[i] = r;
  }
}

int ts_wgrad_reduce(const TsWgradReduce &job, ts_stream_t stream_) {
  const int64_t n = (int64_t)job.K * job.cacb4;
  if (n == 0) return TS_OK;
  wgrad_reduce_kernel<<<(unsigned)ts_cdiv(n, 64), 256, 0, (hipStream_t)stream_>>>(job);
  TS_CHECK_LAUNCH("wgrad_reduce");
  return TS_OK;
}

// The ordered sum in the order of the form that rides on another launch (ts_wgrad_reduce_one: chunk after chunk), as a launch of its
// own: what the weight gradient on a second stream uses, so that it leaves the bits of the riding form.
__global__ __launch_bounds__(256) void wgrad_reduce_seq_kernel(TsWgradReduce job) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (int64_t)job.K * job.cacb4) ts_wgrad_reduce_one(job, i);
}
int ts_wgrad_reduce_seq(const TsWgradReduce &job, ts_stream_t stream_) {
  const int64_t n = (int64_t)job.K * job.cacb4;
  if (n == 0) return TS_OK;
  wgrad_reduce_seq_kernel<<<(unsigned)ts_cdiv(n, 256), 256, 0, (hipStream_t)stream_>>>(job);
  TS_CHECK_LAUNCH("wgrad_reduce (chunk order)");
  return TS_OK;
}

// bytes of the partial buffer of a deterministic weight gradient: an upper bound over the tilings the launchers pick
// (the chunk length grows with the tile count, so one tile gives the most chunks; every kernel family steps by 32 pairs
// and caps chunks at 1024)
size_t ts_wgrad_partial_bytes(int64_t n_pairs, int32_t c_a, int32_t c_b, int32_t K) {
  return (size_t)ts_wgrad_plan(n_pairs, 1, K, 32, 1024).slots * c_a * c_b * sizeof(float);
}

extern "C" size_t ts_conv_wgrad_workspace_bytes(int64_t n_pairs, int32_t c_a, int32_t c_b, int32_t K) {
  return ts_align_up(ts_wgrad_partial_bytes(n_pairs, c_a, c_b, K), 256);
}

// Deterministic weight gradient: partial tiles into `ws`, then their ordered sum - run to run identical bits.
extern "C" int ts_conv_wgrad_det(const float *a_feat, int32_t c_a, const float *b_feat, int32_t c_b,
                                 const int32_t *nbmaps, const int32_t *nboffs, int32_t K, int32_t col_a,
                                 int64_t n_pairs, float *grad_kernel, void *ws, size_t ws_bytes, ts_stream_t stream_) {
  TS_REQUIRE(c_a > 0 && c_b > 0 && K > 0 && n_pairs >= 0, TS_ERR_INVALID_ARGUMENT, "ts_conv_wgrad_det: bad sizes");
  if (g_ts_conv_impl == 1 || ((int64_t)c_a * c_b) % 4 != 0 || ((uintptr_t)grad_kernel & 15) != 0)   // scalar cross-check kernel /
    return ts_conv_wgrad_ex(a_feat, c_a, b_feat, c_b, nbmaps, nboffs, K, col_a, n_pairs, grad_kernel, 0, stream_);   // odd shapes
  TS_REQUIRE(ws && ws_bytes >= ts_conv_wgrad_workspace_bytes(n_pairs, c_a, c_b, K) && ((uintptr_t)ws & 15) == 0,
             TS_ERR_INVALID_ARGUMENT, "ts_conv_wgrad_det: workspace too small");
  if (n_pairs == 0) {
    TS_CHECK_HIP(hipMemsetAsync(grad_kernel, 0, (size_t)K * c_a * c_b * 4, (hipStream_t)stream_), "wgrad memset");
    return TS_OK;
  }
  g_ts_wgrad_part = (float *)ws;
  const int rc = ts_conv_wgrad_ex(a_feat, c_a, b_feat, c_b, nbmaps, nboffs, K, col_a, n_pairs, grad_kernel, 1, stream_);
  g_ts_wgrad_part = nullptr;
  if (rc != TS_OK) return rc;
  const TsWgradReduce job = {(const float *)ws, nboffs, grad_kernel, K, g_ts_wgrad_plan.chunk, (int64_t)c_a * c_b / 4};
  return ts_wgrad_reduce(job, stream_);
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
                             int64_t n_pairs, float *grad_kernel, ts_stream_t stream_) {
  return ts_conv_wgrad_ex(a_feat, c_a, b_feat, c_b, nbmaps, nboffs, K, col_a, n_pairs, grad_kernel, 0, stream_);
}

int ts_conv_wgrad_ex(const float *a_feat, int32_t c_a, const float *b_feat, int32_t c_b, const int32_t *nbmaps,
                     const int32_t *nboffs, int32_t K, int32_t col_a, int64_t n_pairs, float *grad_kernel,
                     int32_t already_zero, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(c_a > 0 && c_b > 0 && K > 0 && n_pairs >= 0 && n_pairs < (1LL << 31), TS_ERR_INVALID_ARGUMENT,
             "ts_conv_wgrad: bad sizes");
  TS_REQUIRE(grad_kernel && nboffs, TS_ERR_INVALID_ARGUMENT, "ts_conv_wgrad: null pointer");
  if (!already_zero) TS_CHECK_HIP(hipMemsetAsync(grad_kernel, 0, (size_t)K * c_a * c_b * 4, stream), "wgrad memset");
  if (n_pairs == 0) return TS_OK;
  TS_REQUIRE(a_feat && b_feat && nbmaps, TS_ERR_INVALID_ARGUMENT, "ts_conv_wgrad: null pointer");
  const int2 *nm = (const int2 *)nbmaps;
  col_a = col_a ? 1 : 0;
  if (g_ts_conv_impl == 1) {
    dim3 grid((unsigned)ts_cdiv((int64_t)c_a * c_b, 256), K);
    conv_wgrad_scalar_kernel<<<grid, 256, 0, stream>>>(a_feat, c_a, b_feat, c_b, nm, nboffs, col_a, grad_kernel);
    TS_CHECK_LAUNCH("conv_wgrad_scalar");
    return TS_OK;
  }
  // tile per dimension: the smallest of {32, 64, 96, 128} that covers min(C, 128) without padding a 96-multiple
  auto pick = [](int c) { return c <= 32 ? 32 : c <= 64 ? 64 : (c % 96 == 0 ? 96 : 128); };
  const int tm = pick(c_a), tn = pick(c_b);
#define TS_WG(TM, TN) launch_wgrad<TM, TN>(a_feat, c_a, b_feat, c_b, nm, nboffs, K, col_a, n_pairs, grad_kernel, stream)
#define TS_WG_ROW(TM)                    \
  switch (tn) {                          \
    case 32: return TS_WG(TM, 32);       \
    case 64: return TS_WG(TM, 64);       \
    case 96: return TS_WG(TM, 96);       \
    default: return TS_WG(TM, 128);      \
  }
  switch (tm) {
    case 32: TS_WG_ROW(32)
    case 64: TS_WG_ROW(64)
    case 96: TS_WG_ROW(96)
    default: TS_WG_ROW(128)
  }
#undef TS_WG_ROW
#undef TS_WG
}
