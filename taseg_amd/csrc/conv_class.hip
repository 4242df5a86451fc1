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
// compute units of the current device (persistent launches size their grid from it); 256 when the query fails
static int ts_cu_count() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return 256;
  return cus;
}
static inline int64_t cg_npad(int64_t n) { return (n + CG_BM - 1) / CG_BM * CG_BM; }

extern "C" int64_t ts_conv_class_rows(int64_t n) { return n < 0 ? 0 : CG_GROUPS * cg_npad(n); }
extern "C" int64_t ts_conv_class_rows2(int64_t n, int32_t groups) { return (n < 0 || groups < 1) ? 0 : groups * cg_npad(n); }

// key = group * 512 + (511 - mask): inside a group, rows with more / higher neighbour bits first, rows without a neighbour in
// the group (and the padding up to a multiple of 128) last; value = row (-1: padding).  gk = offsets per group (<= 9).
__global__ __launch_bounds__(256) void class_keys_kernel(const int *__restrict__ nbr, int64_t n, int64_t npad, int groups, int gk,
                                                        unsigned short *__restrict__ keys, int *__restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= groups * npad) return;
  const int g = (int)(i / npad);
  const int64_t j = i - (int64_t)g * npad;
  unsigned bits = 0;
  if (j < n) {
#pragma unroll
    for (int kl = 0; kl < CG_GK; ++kl)
      if (kl < gk) bits |= (unsigned)(nbr[(int64_t)(gk * g + kl) * n + j] >= 0) << kl;
  }
  keys[i] = (unsigned short)(g * 512 + (511 - bits));
  vals[i] = j < n ? (int)j : -1;
}

// one 128-thread workgroup per tile of the sorted list: the neighbour table in sorted order (src), the position of every row in
// the list (pos; plans with pass 2) or the row of every list slot (rows; direct plans: the product is stored straight into the
// destination rows), the union mask of the tile (bit 30: the tile holds at least one real row)
__global__ __launch_bounds__(CG_BM) void class_fill_kernel(const int *__restrict__ nbr, int64_t n, int64_t npad, int groups,
                                                          int gk, const unsigned short *__restrict__ keys,
                                                          const int *__restrict__ vals, int *__restrict__ src,
                                                          int *__restrict__ pos, int *__restrict__ rows,
                                                          int *__restrict__ tile_mask) {
  __shared__ unsigned wmask[CG_BM / 64];
  const int64_t m_pad = groups * npad;
  const int64_t i = (int64_t)blockIdx.x * CG_BM + threadIdx.x;
  const int key = keys[i];
  const int g = key >> 9;
  const unsigned bits = 511u - (unsigned)(key & 511);
  const int j = vals[i];
  const bool live = j >= 0 && bits != 0;
#pragma unroll
  for (int kl = 0; kl < CG_GK; ++kl)
    if (kl < gk) src[(int64_t)kl * m_pad + i] = (live && ((bits >> kl) & 1)) ? nbr[(int64_t)(gk * g + kl) * n + j] : -1;
  if (pos && j >= 0) pos[(int64_t)g * n + j] = live ? (int)i : -1;
  if (rows) rows[i] = j;
  unsigned m = (live ? bits : 0u) | (j >= 0 ? (1u << 30) : 0u);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m |= __shfl_xor(m, d, 64);
  if ((threadIdx.x & 63) == 0) wmask[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) tile_mask[blockIdx.x] = (int)(wmask[0] | wmask[1]);
}

// live tiles, longest (most offsets) first: tile_info[t] = (group + 4 * tile, union mask); one workgroup (<= 70k tiles at the
// 3e6-voxel cap).  The order inside a length class follows the arrival of the LDS atomics - every tile owns its Z' rows, so
// the order of the list changes the schedule, never a result.  direct: tiles of rows WITHOUT any neighbour are listed too (last,
// with mask 0): their destination rows must be written (zeros).  (A list in two parts - the outer groups' tiles, then the centre
// group's, for the finish inside the product - cost the ordinary launch 0.1 ms per step: the phases filter the one list instead.)
__global__ __launch_bounds__(1024) void class_tiles_kernel(const int *__restrict__ tile_mask, int n_all, int64_t npad, int direct,
                                                           int2 *__restrict__ tile_info, int *__restrict__ n_tiles) {
  __shared__ int cnt[CG_GK + 1], base[CG_GK + 1];
  if (threadIdx.x <= CG_GK) cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < n_all; t += 1024) {
    const int m = tile_mask[t];
    if ((m & 511) || (direct && (m >> 30))) atomicAdd(&cnt[__builtin_popcount(m & 511)], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0, steps = 0;
    for (int c = CG_GK; c >= 0; --c) {
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
    if ((m & 511) || (direct && (m >> 30))) {
      const int at = atomicAdd(&base[__builtin_popcount(m & 511)], 1);
      tile_info[at] = make_int2(t / tiles_per_group + 4 * t, m & 511);
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

// nbr [K][n] = input row feeding destination row j through offset k, or -1 (ts_build_kmap's `nbr` table; for the transposed
// direction of a strided map the table of the inverse map) -> src [K / groups][m_pad], tile_info [m_pad / 128] (x, y) pairs,
// n_tiles [2] = (listed tiles, their (tile, offset) steps), and either pos [groups][n] (pass 2 adds the groups' rows) or - direct
// plans, groups == 1 - rows [m_pad] (destination row of every list slot, -1 = padding);  m_pad = ts_conv_class_rows2(n, groups)
extern "C" int ts_conv_class_plan(const int32_t *nbr, int64_t n, int32_t K, int32_t groups, int32_t *src, int32_t *tile_info,
                                  int32_t *n_tiles, int32_t *pos, int32_t *rows, void *ws, size_t ws_bytes, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(groups >= 1 && groups <= CG_GROUPS && K > 0 && K % groups == 0 && K / groups <= CG_GK, TS_ERR_UNSUPPORTED,
             "ts_conv_class_plan: K offsets must split into 1-3 groups of <= 9 (got K %d, groups %d)", K, groups);
  TS_REQUIRE(n > 0 && n < (1LL << 28), TS_ERR_INVALID_ARGUMENT, "ts_conv_class_plan: bad n");
  TS_REQUIRE(nbr && src && tile_info && n_tiles && ws, TS_ERR_INVALID_ARGUMENT, "ts_conv_class_plan: null pointer");
  TS_REQUIRE((pos != nullptr) != (rows != nullptr), TS_ERR_INVALID_ARGUMENT,
             "ts_conv_class_plan: exactly one of pos (pass-2 plan) and rows (direct plan) must be given");
  TS_REQUIRE(!rows || groups == 1, TS_ERR_INVALID_ARGUMENT, "ts_conv_class_plan: a direct plan has one group");
  TS_REQUIRE(ws_bytes >= ts_conv_class_plan_workspace_bytes(n), TS_ERR_WORKSPACE_TOO_SMALL, "ts_conv_class_plan: workspace too small");
  const int gk = K / groups;
  const int64_t npad = cg_npad(n), m = groups * npad;
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
  class_keys_kernel<<<(unsigned)ts_cdiv(m, 256), 256, 0, stream>>>(nbr, n, npad, groups, gk, k0, v0);
  TS_CHECK_LAUNCH("ts_conv_class_plan/keys");
  TS_CHECK_HIP(rocprim::radix_sort_pairs(p, sort_bytes, k0, k1, v0, v1, (size_t)m, 0, 11, stream), "ts_conv_class_plan/sort");
  class_fill_kernel<<<(unsigned)(m / CG_BM), CG_BM, 0, stream>>>(nbr, n, npad, groups, gk, k1, v1, src, pos, rows, tmask);
  TS_CHECK_LAUNCH("ts_conv_class_plan/fill");
  class_tiles_kernel<<<1, 1024, 0, stream>>>(tmask, (int)(m / CG_BM), npad, rows ? 1 : 0, (int2 *)tile_info, n_tiles);
  TS_CHECK_LAUNCH("ts_conv_class_plan/tiles");
  return TS_OK;
}

// Direct plan of the ONE-PAIR-PER-DESTINATION direction of a strided map (destination = its fine / input rows: the transposed
// forward and the strided input gradient) straight from the rulebook, without a sort: the pairs are ordered by offset already, so
// slot p = pair p, rows[p] = its input row, src[k(p)][p] = its output row - tiles of 128 consecutive pairs hold one offset (two
// where they straddle an offset boundary).  Needs every destination row in exactly one pair (n_pairs == n: 2x2x2 / stride-2 maps,
// SURVEY App. A); 2 launches instead of the ~10 of the sorted builder.
__global__ __launch_bounds__(CG_BM) void class_fill_pairs_kernel(const int2 *__restrict__ nbmaps, const int *__restrict__ nboffs,
                                                                int K, int64_t n_pairs, int64_t m_pad, int *__restrict__ src,
                                                                int *__restrict__ rows, int *__restrict__ tile_mask) {
  __shared__ unsigned wmask[CG_BM / 64];
  const int64_t i = (int64_t)blockIdx.x * CG_BM + threadIdx.x;
  int k = -1;
  int2 pr = make_int2(-1, -1);
  if (i < n_pairs) {
    pr = nbmaps[i];
#pragma unroll
    for (int kk = 0; kk < CG_GK; ++kk)
      if (kk < K && i >= nboffs[kk]) k = kk;
  }
#pragma unroll
  for (int kl = 0; kl < CG_GK; ++kl)
    if (kl < K) src[(int64_t)kl * m_pad + i] = (kl == k) ? pr.y : -1;
  rows[i] = k >= 0 ? pr.x : -1;
  unsigned m = k >= 0 ? ((1u << k) | (1u << 30)) : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m |= __shfl_xor(m, d, 64);
  if ((threadIdx.x & 63) == 0) wmask[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) tile_mask[blockIdx.x] = (int)(wmask[0] | wmask[1]);
}

extern "C" int ts_conv_class_plan_pairs(const int32_t *nbmaps, const int32_t *nboffs, int32_t K, int64_t n_pairs, int32_t *src,
                                        int32_t *tile_info, int32_t *n_tiles, int32_t *rows, void *ws, size_t ws_bytes,
                                        ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(K > 0 && K <= CG_GK, TS_ERR_UNSUPPORTED, "ts_conv_class_plan_pairs: 1 .. 9 offsets (got %d)", K);
  TS_REQUIRE(n_pairs > 0 && n_pairs < (1LL << 28), TS_ERR_INVALID_ARGUMENT, "ts_conv_class_plan_pairs: bad pair count");
  TS_REQUIRE(nbmaps && nboffs && src && tile_info && n_tiles && rows && ws, TS_ERR_INVALID_ARGUMENT,
             "ts_conv_class_plan_pairs: null pointer");
  const int64_t m = cg_npad(n_pairs);
  TS_REQUIRE(ws_bytes >= ts_align_up((size_t)(m / CG_BM) * 4, 256), TS_ERR_WORKSPACE_TOO_SMALL,
             "ts_conv_class_plan_pairs: workspace too small");
  int *tmask = (int *)ws;
  class_fill_pairs_kernel<<<(unsigned)(m / CG_BM), CG_BM, 0, stream>>>((const int2 *)nbmaps, nboffs, K, n_pairs, m, src, rows, tmask);
  TS_CHECK_LAUNCH("ts_conv_class_plan_pairs/fill");
  class_tiles_kernel<<<1, 1024, 0, stream>>>(tmask, (int)(m / CG_BM), m, 1, (int2 *)tile_info, n_tiles);
  TS_CHECK_LAUNCH("ts_conv_class_plan_pairs/tiles");
  return TS_OK;
}

// inverse table of a kernel map for the plans of its transposed direction: nbr_t[k][i] = output row fed by input row i through
// offset k (nbmaps[pos_in[k][i]].y) or -1
__global__ __launch_bounds__(256) void class_nbr_t_kernel(const int *__restrict__ pos_in, const int2 *__restrict__ nbmaps,
                                                         int64_t total, int *__restrict__ nbr_t) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int p = pos_in[e];
  nbr_t[e] = p >= 0 ? nbmaps[p].y : -1;
}

extern "C" int ts_conv_nbr_transposed(const int32_t *pos_in, const int32_t *nbmaps, int32_t K, int64_t n_in, int32_t *nbr_t,
                                      ts_stream_t stream) {
  TS_REQUIRE(K > 0 && n_in >= 0, TS_ERR_INVALID_ARGUMENT, "ts_conv_nbr_transposed: bad sizes");
  if (n_in == 0) return TS_OK;
  TS_REQUIRE(pos_in && nbmaps && nbr_t, TS_ERR_INVALID_ARGUMENT, "ts_conv_nbr_transposed: null pointer");
  const int64_t total = (int64_t)K * n_in;
  class_nbr_t_kernel<<<(unsigned)ts_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(pos_in, (const int2 *)nbmaps, total, nbr_t);
  TS_CHECK_LAUNCH("ts_conv_nbr_transposed");
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

// finish of a three-group plan inside the product (common.h, TsClassFinish): phase 0 = every listed tile, Z' rows (pass 2 follows);
// phase 1 = the tiles of the groups without the centre offset, Z' rows; phase 2 = the tiles of the centre group gfin (every
// output row is in exactly one of them), the RESULT rows: out[r] = ((z_0[r] +) own sums (+ z_2[r])) (+ addend[r]), r = the slot's
// own row (its centre neighbour, offset klc of the group) - pass 2's additions in pass 2's order.  Both phases walk the one tile
// list and skip the other phase's tiles.
struct CgFinish {
  const int *pos;
  const void *addend;
  void *out;
  int64_t n;
  int phase, klc, gfin;
};

// X [n, R] fp32 rows; W [K, R, O_total] (WT = false: forward) or [K, O_total, R] (WT = true: the input gradient multiplies with
// the transposed slice of the MIRRORED offset); Zp [m_pad, O_total].  grid (upper bound of the tile count, O_total / BN).
// gk offsets per group (k = gk * group + kl); mirror: the transposed product takes the slice of offset K-1-k (submanifold maps: the
// map is its own transpose with the offsets reversed) instead of k (direct plans of a strided map: built for that direction);
// rows != NULL (direct plans): list slot i is stored into row rows[i] of Zp (= the result itself), -1 = padding.
// Workgroups with blockIdx.x >= tile_blocks (launched when side.K > 0) form the ordered sum of the weight-gradient partials of the
// launch before this one (common.h) - the job that rides on pass 2 where there is one.
template <int BN, int WR, bool WT>
__global__ __launch_bounds__(256, 2) void class_gemm_kernel(const float *__restrict__ X, int R, const float *__restrict__ W,
                                                           int O_total, const int *__restrict__ src, int64_t m_pad,
                                                           const int2 *__restrict__ tile_info,
                                                           const int *__restrict__ n_tiles, int K, int gk, int mirror,
                                                           const int *__restrict__ rows, float *__restrict__ Zp,
                                                           TsWgradReduce side, int tile_blocks, CgFinish fin) {
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
  constexpr int A_PLANES = 3;                                  // h, m, l of the gathered slice
  extern __shared__ __attribute__((aligned(16))) unsigned short smem_cg[];
  unsigned short *Ap = smem_cg;                                // planes [128][CG_AP]
  unsigned short *Bp = Ap + A_PLANES * A_PLANE;                // 3 planes of the weight slice

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int tq = r16 >> 2, tp = lane & 3;
  const int wr = wave / WC, wc = wave % WC;
  const int o0 = blockIdx.y * BN;
  int tile = (int)blockIdx.x;                   // launch order = list order (longest first): every XCD gets tiles of every length
  if (tile >= tile_blocks) {                    // side job: ordered sum of the weight-gradient partials
    if (blockIdx.y == 0) {
      const int64_t step = (int64_t)(gridDim.x - tile_blocks) * 256;
      for (int64_t i = (int64_t)(tile - tile_blocks) * 256 + tid; i < (int64_t)side.K * side.cacb4; i += step)
        ts_wgrad_reduce_one(side, i);
    }
    return;
  }
  if (tile >= *n_tiles) return;
  const int2 info = tile_info[tile];
  const int grp = __builtin_amdgcn_readfirstlane(info.x) & 3;
  if (fin.phase && (grp == fin.gfin) != (fin.phase == 2)) return;      // the other phase's tile
  int mask = __builtin_amdgcn_readfirstlane(info.y);
  const int64_t row0 = (int64_t)(__builtin_amdgcn_readfirstlane(info.x) >> 2) * BM;
  if (mask == 0) {                              // direct plans: a tile of rows without any neighbour - their result is zero
    for (int e = tid; e < BM * (BN / 4); e += 256) {
      const int r = e / (BN / 4), c4 = e - r * (BN / 4);
      const int dst = rows ? rows[row0 + r] : (int)(row0 + r);
      if (dst >= 0) *(f32x4 *)(Zp + (int64_t)dst * O_total + o0 + 4 * c4) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    return;
  }

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
  auto bind = [&](int kl) {             // operand pointers of group offset kl: k = gk grp + kl
    const int k = gk * grp + kl;
    const int kw = (WT && mirror) ? (K - 1 - k) : k;
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
      const f32x4 v0 = rlive[it] ? ra[it][0] : zero, v1 = rlive[it] ? ra[it][1] : zero;
      unsigned short *dst = Ap + rr * CG_AP + acol;
      u32x4 h, m, l;
      cg_split8(v0, v1, h, m, l);
      *(u32x4 *)dst = h;
      *(u32x4 *)(dst + A_PLANE) = m;
      *(u32x4 *)(dst + 2 * A_PLANE) = l;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      if (B_IT * 256 == B_CHUNKS || tid + it * 256 < B_CHUNKS) {
        unsigned short *dst = Bp + bdst[it];
        u32x4 h, m, l;
        cg_split8(rb[it][0], rb[it][1], h, m, l);
        *(u32x4 *)dst = h;
        *(u32x4 *)(dst + B_PLANE) = m;
        *(u32x4 *)(dst + 2 * B_PLANE) = l;
      }
    }
  };
  auto mma = [&]() {
    bf8 a[MI][A_PLANES];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int p = 0; p < A_PLANES; ++p)
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
  if (rows) {                                   // direct plan: the sums ARE the result rows
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int dst = rows[row0 + (wr * MI + mi) * 16 + 4 * g + q];
        if (dst >= 0) {
          float *zr = Zp + (int64_t)dst * O_total + o0 + r16;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) zr[(wc * NI + ni) * 16] = tot[mi][ni][q];
        }
      }
    return;
  }
  if (fin.phase == 2) {                         // the centre group: result rows, through an LDS image of the tile's sums so that
    // the other groups' rows, the addend and the result move as 16-byte pieces of whole rows
    constexpr int ZF = BN + 4;
    float *Zt = (float *)smem_cg;
    __syncthreads();                            // every wave has read the last slice
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          Zt[((wr * MI + mi) * 16 + 4 * g + q) * ZF + (wc * NI + ni) * 16 + r16] = tot[mi][ni][q];
    __syncthreads();
    const int *crow = src + (int64_t)fin.klc * m_pad + row0;
    float *out = (float *)fin.out;
    const float *add = (const float *)fin.addend;
    constexpr int C4 = BN / 4;
    for (int e = tid; e < BM * C4; e += 256) {
      const int row = e / C4, c4 = e - row * C4;
      const int r = crow[row];
      if (r < 0) continue;
      const int p0 = fin.pos[r], p2 = fin.pos[2 * fin.n + r];
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p0 >= 0) v += *(const f32x4 *)(Zp + (int64_t)p0 * O_total + o0 + 4 * c4);
      v += *(const f32x4 *)&Zt[row * ZF + 4 * c4];
      if (p2 >= 0) v += *(const f32x4 *)(Zp + (int64_t)p2 * O_total + o0 + 4 * c4);
      if (add) v += *(const f32x4 *)(add + (int64_t)r * O_total + o0 + 4 * c4);
      *(f32x4 *)(out + (int64_t)r * O_total + o0 + 4 * c4) = v;
    }
    return;
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

struct CgArgs {          // what a class-GEMM launch takes beyond operands and result
  const int *src;
  int64_t m_pad;
  const int2 *tile_info;
  const int *n_tiles;
  int K, gk, mirror;
  const int *rows;
  TsWgradReduce side;
  CgFinish fin;          // phase 0: plain launch
  int groups;
};
static int64_t cg_tile_bound(const CgArgs &a) { return a.m_pad / CG_BM; }      // tiles a launch can meet (every phase walks the list)

template <int BN, int WR, bool WT>
static int launch_class(const float *X, int R, const float *W, int O_total, const CgArgs &a, float *Zp, hipStream_t stream) {
  const size_t stage = (size_t)(3 * CG_BM * CG_AP + 3 * (WT ? BN * CG_AP : CG_BK * (BN + 8))) * 2;
  const size_t lds = a.fin.phase == 2 ? std::max(stage, (size_t)CG_BM * (BN + 4) * 4) : stage;      // phase 2: the image of the sums
  const unsigned tiles = (unsigned)cg_tile_bound(a);
  dim3 grid(tiles + (a.side.K > 0 ? 64u : 0u), (unsigned)(O_total / BN));
  class_gemm_kernel<BN, WR, WT><<<grid, 256, lds, stream>>>(X, R, W, O_total, a.src, a.m_pad, a.tile_info, a.n_tiles, a.K, a.gk,
                                                            a.mirror, a.rows, Zp, a.side, (int)tiles, a.fin);
  TS_CHECK_LAUNCH("ts_conv_class_gemm");
  return TS_OK;
}
static int cg_tile_columns(int c_out) { return c_out % 128 == 0 ? 128 : c_out % 96 == 0 ? 96 : c_out % 64 == 0 ? 64 : c_out % 32 == 0 ? 32 : 0; }

extern "C" int32_t ts_conv_class_supported(int32_t c_red, int32_t c_out) {
  return c_red > 0 && c_red % CG_BK == 0 && cg_tile_columns(c_out) != 0;
}

static int cg_check(const char *what, int K, int groups, int c_red, int c_out, int64_t m_pad, const void *a, const void *b,
                    const void *c, const void *src, const void *ti, const void *nt) {
  TS_REQUIRE(groups >= 1 && groups <= CG_GROUPS && K > 0 && K % groups == 0 && K / groups <= CG_GK, TS_ERR_UNSUPPORTED,
             "%s: K offsets must split into 1-3 groups of <= 9 (got K %d, groups %d)", what, K, groups);
  TS_REQUIRE(ts_conv_class_supported(c_red, c_out), TS_ERR_UNSUPPORTED,
             "%s: C_red must be a multiple of 32 and C_out of 32 (got %d, %d)", what, c_red, c_out);
  TS_REQUIRE(a && b && c && src && ti && nt, TS_ERR_INVALID_ARGUMENT, "%s: null pointer", what);
  TS_REQUIRE(m_pad > 0 && m_pad % ((int64_t)groups * CG_BM) == 0 && m_pad < (1LL << 31), TS_ERR_INVALID_ARGUMENT,
             "%s: m_pad must be ts_conv_class_rows2(n, groups)", what);
  TS_REQUIRE(((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0, TS_ERR_INVALID_ARGUMENT,
             "%s: pointers must be 16-byte aligned", what);
  return TS_OK;
}

// Where the finish inside the product pays.  It trades pass 2 (one launch, 4 row moves per output row) for a second launch of the
// product (2 row moves, but two launch tails instead of one).  Alone on the device (profiles/r04_class_finish_probe.txt) that is a
// gain from ~60k rows in half storage (stride-1 96 -> 96: 105 -> 89 us) and only on the largest maps in fp32 (178k rows: 233 -> 224
// us; 84k rows: level; 30k rows: a third slower); inside the training step, with the staging stream's kernels beside it, the
// nuScenes autocast step got 0.2 ms SLOWER with it (16.94 -> 17.17 ms, profiles/r04_ab_class_finish.txt).  So the block calls keep
// pass 2 (thresholds 0 = never); TS_OPT_CLASS_FINISH_ROWS / TS_OPT_CLASS_FINISH_ROWS_HALF (rows from which the finish is taken) switch
// it on for A/B runs, and ts_conv_class_conv is there for callers that want the convolution in one call.
extern "C" int32_t ts_conv_class_finish_pays(int64_t n, int32_t half) {
  const int64_t lim = ts_get_option(half ? TS_OPT_CLASS_FINISH_ROWS_HALF : TS_OPT_CLASS_FINISH_ROWS);
  return lim > 0 && n >= lim;
}

// shared by the fp32 and the half entry: the finish descriptor of a call (checked), and the two launches it turns into
static int cg_finish(const char *what, const TsClassFinish *fin, int K, int groups, const int32_t *rows, CgFinish &out) {
  out = CgFinish{nullptr, nullptr, nullptr, 0, 0, 0, 0};
  if (!fin) return TS_OK;
  TS_REQUIRE(groups == CG_GROUPS && !rows && (K / 2) / (K / groups) == 1, TS_ERR_INVALID_ARGUMENT,
             "%s: the finish inside the product needs a three-group pass-2 plan", what);
  TS_REQUIRE(fin->pos && fin->out && fin->n > 0 && (((uintptr_t)fin->out) & 15) == 0 && (((uintptr_t)fin->addend) & 15) == 0,
             TS_ERR_INVALID_ARGUMENT, "%s: finish: null / misaligned pointer", what);
  out = CgFinish{fin->pos, fin->addend, fin->out, fin->n, 1, (K / 2) % (K / groups), (K / 2) / (K / groups)};
  return TS_OK;
}

// library-internal form (csrc/block.hip): + the weight-gradient sum riding on the launch, + the finish inside the product
int ts_conv_class_gemm_ex(const float *feat, int32_t c_red, const float *kernel, int32_t K, int32_t groups, int32_t c_out,
                          const int32_t *src, int64_t m_pad, const int32_t *tile_info, const int32_t *n_tiles, int32_t wt,
                          int32_t mirror, const int32_t *rows, float *zp, const TsWgradReduce *side, const TsClassFinish *fin,
                          ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int rc = cg_check("ts_conv_class_gemm", K, groups, c_red, c_out, m_pad, feat, kernel, zp, src, tile_info, n_tiles);
  if (rc != TS_OK) return rc;
  CgArgs a = {src, m_pad, (const int2 *)tile_info, n_tiles, K, K / groups, mirror ? 1 : 0, rows, {}, {}, groups};
  {
    const int frc = cg_finish("ts_conv_class_gemm", fin, K, groups, rows, a.fin);
    if (frc != TS_OK) return frc;
  }
#define CG_GO(BN, WR)                                                                  \
  (wt ? launch_class<BN, WR, true>(feat, c_red, kernel, c_out, a, zp, stream)          \
      : launch_class<BN, WR, false>(feat, c_red, kernel, c_out, a, zp, stream))
  // phase 1 (or the one plain launch), then - finish inside the product - phase 2, which also carries the side job
  for (int phase = a.fin.phase; phase <= (fin ? 2 : 0); ++phase) {
    a.fin.phase = phase;
    a.side = (side && phase != 1) ? *side : TsWgradReduce{};
    int r;
    switch (cg_tile_columns(c_out)) {
      case 128: r = CG_GO(128, 2); break;
      case 96: r = CG_GO(96, 2); break;
      case 64: r = CG_GO(64, 2); break;
      default: r = CG_GO(32, 4); break;
    }
    if (r != TS_OK) return r;
  }
  return TS_OK;
#undef CG_GO
}

// zp = pass 1 of the convolution on the plan (wt = 0: feat = input rows, kernel [K, c_red, c_out]; wt = 1: the transposed product,
// feat = output-gradient rows [n, c_red], kernel [K, c_out, c_red] as stored, result columns = c_out = C_in; mirror = 1 takes the
// slice of offset K-1-k: submanifold maps).  rows == NULL: zp [m_pad, c_out], one row per (destination row, group) - pass 2 is
// ts_conv_gather_sum(zp, pos, groups); rows != NULL (direct plan, groups == 1): zp [n, c_out] IS the result.
extern "C" int ts_conv_class_gemm(const float *feat, int32_t c_red, const float *kernel, int32_t K, int32_t groups, int32_t c_out,
                                  const int32_t *src, int64_t m_pad, const int32_t *tile_info, const int32_t *n_tiles, int32_t wt,
                                  int32_t mirror, const int32_t *rows, float *zp, ts_stream_t stream) {
  return ts_conv_class_gemm_ex(feat, c_red, kernel, K, groups, c_out, src, m_pad, tile_info, n_tiles, wt, mirror, rows, zp, nullptr,
                               nullptr, stream);
}


// The whole convolution on a three-group plan: out [n, c_out] = sum over the offsets (+ addend), as ts_conv_class_gemm followed by
// ts_conv_gather_sum(zp, pos, 3) - the same bits - in two launches of the product kernel and without the centre group's Z' rows
// (zp [m_pad, c_out] is scratch for the other two groups' rows).
extern "C" int ts_conv_class_conv(const float *feat, int32_t c_red, const float *kernel, int32_t K, int32_t groups, int32_t c_out,
                                  const int32_t *src, int64_t m_pad, const int32_t *tile_info, const int32_t *n_tiles, int32_t wt,
                                  int32_t mirror, const int32_t *pos, int64_t n, const float *addend, float *zp, float *out,
                                  ts_stream_t stream) {
  const TsClassFinish fin = {pos, n, out, addend};
  return ts_conv_class_gemm_ex(feat, c_red, kernel, K, groups, c_out, src, m_pad, tile_info, n_tiles, wt, mirror, nullptr, zp, nullptr,
                               &fin, stream);
}

// ------------------------------------------------------------------------------------------- half storage
// The same walk for IEEE-half rows (torch.autocast; conv_pairs_h.hip): one v_mfma_f32_16x16x32_f16 per product, fp32 sums,
// Z' rows rounded to half ONCE per (row, group) - the two-pass form rounds every pair's Z row.
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

// Whole rows, persistent workgroups.  A step = (tile, offset of the tile's mask, RC-column chunk of the rows).  (The first form
// of this kernel walked 32-column slices like the fp32 kernel above, one tile per workgroup: ONE slice - 8 KB of gathered rows per
// workgroup - in flight, a workgroup living for the 3.7 offsets of its tile, every workgroup paying the chain tile entry -> row
// indices -> rows -> LDS before its first MFMA and every slice a fabric round trip: 3.2 TB/s on the stride-1 96 -> 96 layer,
// profiles/r04_class_h_whole_rows_probe.txt.  Without a split there is too little arithmetic per byte to hide that.)  Here
//   * the WHOLE [128 x RC] block of gathered rows and the [RC x BN] weight block of step s + 1 are in flight (registers) while
//     step s multiplies from LDS (16-byte pieces of a row on consecutive lanes: full lines per request), the row indices of the
//     offset after that are fetched a step earlier still, the entry of the next tile a tile ahead;
//   * a workgroup walks the tile list (tile = round * G + its slot, the slot order reversed every other round: the list is sorted
//     longest first), so the stream of steps does not stop at a tile's end - the finished tile's sums leave through LDS while the
//     next tile's first block is already in flight.
// One accumulator set across the offsets of a tile (the sums are rounded to half once, at the end; the fp32 kernel's second set
// buys nothing at half precision).  Same Z' / direct-row stores, same side job.
struct CgCursor {          // walks (tile, offset) in list order for one workgroup; everything wave-uniform
  int tile, round, slot, G, n_tiles;
  int rest;                // offsets of the current tile not handed out yet
  int row0, grp;
  int2 ahead;              // entry of the tile after the current one (loaded a tile ahead)
};
struct CgStep {
  int row0, grp, kl, last;  // last: the tile's last offset
};

template <int BN, int WR, bool WT, int RC>
__global__ __launch_bounds__(256, 2) void class_gemm_h_kernel(
    const _Float16 *__restrict__ X, int R, const _Float16 *__restrict__ W, int O_total, const int *__restrict__ src, int64_t m_pad,
    const int2 *__restrict__ tile_info, const int *__restrict__ n_tiles_p, int K, int gk, int mirror, const int *__restrict__ rows,
    _Float16 *__restrict__ Zp, TsWgradReduce side, int G, CgFinish fin) {
  constexpr int BM = CG_BM;
  constexpr int WC = 4 / WR;
  constexpr int MI = (BM / 16) / WR;
  constexpr int NI = (BN / 16) / WC;
  constexpr int AP = RC + 8;                       // pitch of the row block (and of the WT weight block) in halves
  constexpr int BP = BN + 8;
  constexpr int A_HALVES = BM * AP;
  constexpr int RQ = RC / 8;                       // 16-byte pieces per row chunk
  constexpr int A_IT = BM * RQ / 256;
  constexpr int B_CHUNKS = RC * BN / 8;
  constexpr int B_IT = (B_CHUNKS + 255) / 256;
  constexpr int KB = RC / 32;
  constexpr int ZP = BN + 8;
  constexpr int CH = BN / 8;
  extern __shared__ __attribute__((aligned(16))) _Float16 smem_cgh[];
  _Float16 *At = smem_cgh;
  _Float16 *Bt = At + A_HALVES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int tq = r16 >> 2, tp = lane & 3;
  const int wr = wave / WC, wc = wave % WC;
  const int o0 = blockIdx.y * BN;
  if ((int)blockIdx.x >= G) {                   // side job: ordered sum of the weight-gradient partials
    if (blockIdx.y == 0) {
      const int64_t step = (int64_t)(gridDim.x - G) * 256;
      for (int64_t i = (int64_t)((int)blockIdx.x - G) * 256 + tid; i < (int64_t)side.K * side.cacb4; i += step)
        ts_wgrad_reduce_one(side, i);
    }
    return;
  }

  CgCursor cur;
  cur.G = G;
  cur.slot = (int)blockIdx.x;
  cur.round = 0;
  cur.n_tiles = __builtin_amdgcn_readfirstlane(*n_tiles_p);
  cur.tile = cur.slot;
  cur.rest = 0;
  cur.row0 = cur.grp = 0;
  cur.ahead = make_int2(0, 0);
  auto tile_of = [&](int round) { return round * G + ((round & 1) ? G - 1 - cur.slot : cur.slot); };
  auto load_entry = [&](int t) -> int2 {
    int2 e = make_int2(0, 0);
    if (t < cur.n_tiles) e = tile_info[t];
    return make_int2(__builtin_amdgcn_readfirstlane(e.x), __builtin_amdgcn_readfirstlane(e.y));
  };
  auto zero_tile = [&](int row0) {               // direct plans: a tile of rows without any neighbour - their result is zero
    const ch8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = tid; e < BM * CH; e += 256) {
      const int r = e / CH, c8 = e - r * CH;
      const int dst = rows ? rows[row0 + r] : (int)(row0 + r);
      if (dst >= 0) *(ch8 *)(Zp + (int64_t)dst * O_total + o0 + 8 * c8) = zero;
    }
  };
  bool opened = false;
  auto next_step = [&](CgStep &st) -> bool {     // false: this workgroup's list is finished
    while (cur.rest == 0) {
      int2 e;
      if (!opened) {
        e = load_entry(cur.tile);
        opened = true;
      } else {
        cur.round += 1;
        cur.tile = tile_of(cur.round);
        e = cur.ahead;
      }
      if (cur.tile >= cur.n_tiles) return false;
      cur.ahead = load_entry(tile_of(cur.round + 1));
      cur.grp = e.x & 3;
      cur.row0 = (e.x >> 2) * BM;
      cur.rest = e.y;
      if (fin.phase && (cur.grp == fin.gfin) != (fin.phase == 2)) cur.rest = 0;      // the other phase's tile
      else if (cur.rest == 0) zero_tile(cur.row0);
    }
    st.kl = __builtin_ctz(cur.rest);
    cur.rest &= cur.rest - 1;
    st.row0 = cur.row0;
    st.grp = cur.grp;
    st.last = cur.rest == 0;
    return true;
  };

  // piece e = it * 256 + tid of the row block: row e / RQ, 16-byte piece e % RQ (consecutive lanes = consecutive pieces of a row)
  int arow[A_IT], acol[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int e = it * 256 + tid;
    arow[it] = e / RQ;
    acol[it] = (e - arow[it] * RQ) << 3;
  }
  int boff[B_IT], bdst[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int e = min(tid + it * 256, B_CHUNKS - 1);
    if (WT) {
      const int col = e / RQ, c8 = (e - col * RQ) << 3;
      boff[it] = col * R + c8;
      bdst[it] = col * AP + c8;
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

  int sa[A_IT], sb[A_IT];                // input rows of the offset being loaded / of the offset after it
  auto fetch = [&](const CgStep &st, int (&dst)[A_IT]) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) dst[it] = src[(int64_t)st.kl * m_pad + st.row0 + arow[it]];
  };
  ch8 ra[A_IT], rb[B_IT];
  auto load_regs = [&](const CgStep &st, int c0) {
    const ch8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      ra[it] = zero;
      if (sa[it] >= 0) ra[it] = *(const ch8 *)(X + (int64_t)sa[it] * R + c0 + acol[it]);
    }
    const int k = gk * st.grp + st.kl;
    const int kw = (WT && mirror) ? (K - 1 - k) : k;
    const _Float16 *wb = WT ? W + ((int64_t)kw * O_total + o0) * R + c0 : W + ((int64_t)kw * R + c0) * O_total + o0;
#pragma unroll
    for (int it = 0; it < B_IT; ++it) rb[it] = *(const ch8 *)(wb + boff[it]);
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) *(ch8 *)&At[arow[it] * AP + acol[it]] = ra[it];
#pragma unroll
    for (int it = 0; it < B_IT; ++it)
      if (B_IT * 256 == B_CHUNKS || tid + it * 256 < B_CHUNKS) *(ch8 *)&Bt[bdst[it]] = rb[it];
  };
  auto mma = [&]() {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      ch8 a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = *(const ch8 *)&At[((wr * MI + mi) * 16 + r16) * AP + 32 * kb + 8 * g];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        b[ni] = WT ? *(const ch8 *)&Bt[((wc * NI + ni) * 16 + r16) * AP + 32 * kb + 8 * g]
                   : cgh_frag_tr(Bt, BP, 32 * kb + 8 * g, (wc * NI + ni) * 16, tq, tp);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
  };
  // the finished tile: sums -> half -> LDS image [row][col] -> 16-byte chunks of whole rows (Z' rows, or the destination rows of a
  // direct plan); called between two barriers' worth of LDS use (every wave has read the last block)
  auto flush_tile = [&](int row0) {
    _Float16 *Zt = smem_cgh;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          Zt[((wr * MI + mi) * 16 + 4 * g + q) * ZP + (wc * NI + ni) * 16 + r16] = (_Float16)acc[mi][ni][q];
          acc[mi][ni][q] = 0.f;
        }
    __syncthreads();
    if (fin.phase == 2) {                 // the centre group: result rows (pass 2's additions in pass 2's order, fp32, one rounding)
      const int *crow = src + (int64_t)fin.klc * m_pad + row0;
      _Float16 *out = (_Float16 *)fin.out;
      const _Float16 *add = (const _Float16 *)fin.addend;
      for (int e = tid; e < BM * CH; e += 256) {
        const int row = e / CH, ch = e - row * CH;
        const int r = crow[row];
        if (r < 0) continue;
        const int p0 = fin.pos[r], p2 = fin.pos[2 * fin.n + r];
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
        if (p0 >= 0) {
          const ch8 z = *(const ch8 *)(Zp + (int64_t)p0 * O_total + o0 + ch * 8);
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] += (float)z[i];
        }
        {
          const ch8 z = *(const ch8 *)&Zt[row * ZP + ch * 8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] += (float)z[i];
        }
        if (p2 >= 0) {
          const ch8 z = *(const ch8 *)(Zp + (int64_t)p2 * O_total + o0 + ch * 8);
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] += (float)z[i];
        }
        if (add) {
          const ch8 z = *(const ch8 *)(add + (int64_t)r * O_total + o0 + ch * 8);
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] += (float)z[i];
        }
        ch8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (_Float16)v[i];
        *(ch8 *)(out + (int64_t)r * O_total + o0 + ch * 8) = o;
      }
      return;
    }
    for (int e = tid; e < BM * CH; e += 256) {
      const int row = e / CH, ch = e - row * CH;
      const int64_t dst = rows ? (int64_t)rows[row0 + row] : (int64_t)row0 + row;
      if (dst >= 0) *(ch8 *)(Zp + dst * O_total + o0 + ch * 8) = *(const ch8 *)&Zt[row * ZP + ch * 8];
    }
  };

  CgStep sl, sf, sm;                      // the step being loaded, the one whose row indices are fetched, the one multiplying
  if (!next_step(sl)) return;
  fetch(sl, sa);
  bool has_f = next_step(sf);
  if (has_f) fetch(sf, sb);
  load_regs(sl, 0);
  int c0 = 0;
  while (true) {
    store_lds();                           // the block of this step (waits for its loads)
    __syncthreads();
    sm = sl;
    const bool tile_done = sm.last && c0 + RC >= R;
    bool more = true;
    if (c0 + RC < R) {
      c0 += RC;
    } else if (has_f) {
      c0 = 0;
      sl = sf;
#pragma unroll
      for (int it = 0; it < A_IT; ++it) sa[it] = sb[it];
      has_f = next_step(sf);
      if (has_f) fetch(sf, sb);
    } else {
      more = false;
    }
    if (more) load_regs(sl, c0);           // the next step's block, in flight behind this step's MFMAs (and the tile's flush)
    mma();
    __syncthreads();                       // every wave has read this step's block
    if (tile_done) {
      flush_tile(sm.row0);
      __syncthreads();
    }
    if (!more) break;
  }
}

template <int BN, int WR, bool WT, int RC>
static int launch_class_h(const _Float16 *X, int R, const _Float16 *W, int O_total, const CgArgs &a, _Float16 *Zp,
                           hipStream_t stream) {
  const size_t stage = (size_t)(CG_BM * (RC + 8) + (WT ? BN * (RC + 8) : RC * (BN + 8))) * 2;
  const size_t ztile = (size_t)CG_BM * (BN + 8) * 2;
  const size_t lds = std::max(stage, ztile);
  const int64_t tiles = cg_tile_bound(a);
  static const int cus = ts_cu_count();
  // one resident workgroup per slot the kernel can hold on the chip (registers and LDS of THIS instantiation), each walking the list
  static const int per_cu = [&] {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, class_gemm_h_kernel<BN, WR, WT, RC>, 256, lds) != hipSuccess || nb < 1) nb = 1;
    return nb;
  }();
  const int ny = O_total / BN;
  const int G = (int)std::max<int64_t>(1, std::min<int64_t>(tiles, (int64_t)cus * per_cu / ny));
  dim3 grid((unsigned)G + (a.side.K > 0 ? 64u : 0u), (unsigned)ny);
  class_gemm_h_kernel<BN, WR, WT, RC><<<grid, 256, lds, stream>>>(X, R, W, O_total, a.src, a.m_pad, a.tile_info, a.n_tiles, a.K,
                                                                    a.gk, a.mirror, a.rows, Zp, a.side, G, a.fin);
  TS_CHECK_LAUNCH("ts_conv_class_gemm_f16 (whole rows)");
  return TS_OK;
}
// row chunk of the deep-prefetch form: the largest of 128 / 96 / 64 / 32 that divides the reduction width
static int cgh_row_chunk(int c_red) { return c_red % 128 == 0 ? 128 : c_red % 96 == 0 ? 96 : c_red % 64 == 0 ? 64 : c_red % 32 == 0 ? 32 : 0; }

int ts_conv_class_gemm_f16_ex(const void *feat, int32_t c_red, const void *w, int32_t K, int32_t groups, int32_t c_out,
                              const int32_t *src, int64_t m_pad, const int32_t *tile_info, const int32_t *n_tiles, int32_t wt,
                              int32_t mirror, const int32_t *rows, void *zp, const TsWgradReduce *side, const TsClassFinish *fin,
                              ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int chk = cg_check("ts_conv_class_gemm_f16", K, groups, c_red, c_out, m_pad, feat, w, zp, src, tile_info, n_tiles);
  if (chk != TS_OK) return chk;
  CgArgs a = {src, m_pad, (const int2 *)tile_info, n_tiles, K, K / groups, mirror ? 1 : 0, rows, {}, {}, groups};
  {
    const int frc = cg_finish("ts_conv_class_gemm_f16", fin, K, groups, rows, a.fin);
    if (frc != TS_OK) return frc;
  }
  const _Float16 *x = (const _Float16 *)feat, *wh = (const _Float16 *)w;
  const int rc = cgh_row_chunk(c_red);
#define CGH2_RC(BN, WR, RC)                                                                      \
  (wt ? launch_class_h<BN, WR, true, RC>(x, c_red, wh, c_out, a, (_Float16 *)zp, stream)          \
      : launch_class_h<BN, WR, false, RC>(x, c_red, wh, c_out, a, (_Float16 *)zp, stream))
#define CGH2_GO(BN, WR)                                                                          \
  (rc == 128 ? CGH2_RC(BN, WR, 128) : rc == 96 ? CGH2_RC(BN, WR, 96) : rc == 64 ? CGH2_RC(BN, WR, 64) : CGH2_RC(BN, WR, 32))
  for (int phase = a.fin.phase; phase <= (fin ? 2 : 0); ++phase) {
    a.fin.phase = phase;
    a.side = (side && phase != 1) ? *side : TsWgradReduce{};
    int r;
    switch (cg_tile_columns(c_out)) {
      case 128: r = CGH2_GO(128, 2); break;
      case 96: r = CGH2_GO(96, 2); break;
      case 64: r = CGH2_GO(64, 2); break;
      default: r = CGH2_GO(32, 4); break;
    }
    if (r != TS_OK) return r;
  }
  return TS_OK;
#undef CGH2_GO
#undef CGH2_RC
}

// half rows: feat / zp IEEE half, w = the half weight [K, C_in, C_out] as stored (wt = 0: forward, read in place through the
// transposing LDS load; wt = 1: the transposed product, rows of the (mirrored) offset's slice = result columns)
extern "C" int ts_conv_class_gemm_f16(const void *feat, int32_t c_red, const void *w, int32_t K, int32_t groups, int32_t c_out,
                                      const int32_t *src, int64_t m_pad, const int32_t *tile_info, const int32_t *n_tiles,
                                      int32_t wt, int32_t mirror, const int32_t *rows, void *zp, ts_stream_t stream) {
  return ts_conv_class_gemm_f16_ex(feat, c_red, w, K, groups, c_out, src, m_pad, tile_info, n_tiles, wt, mirror, rows, zp, nullptr,
                                   nullptr, stream);
}

// ts_conv_class_conv for IEEE-half rows (fp32 sums, one rounding of the result)
extern "C" int ts_conv_class_conv_f16(const void *feat, int32_t c_red, const void *w, int32_t K, int32_t groups, int32_t c_out,
                                      const int32_t *src, int64_t m_pad, const int32_t *tile_info, const int32_t *n_tiles,
                                      int32_t wt, int32_t mirror, const int32_t *pos, int64_t n, const void *addend, void *zp,
                                      void *out, ts_stream_t stream) {
  const TsClassFinish fin = {pos, n, out, addend};
  return ts_conv_class_gemm_f16_ex(feat, c_red, w, K, groups, c_out, src, m_pad, tile_info, n_tiles, wt, mirror, nullptr, zp, nullptr,
                                   &fin, stream);
}
