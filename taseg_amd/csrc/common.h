// Shared helpers for libtaseg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>

#include "../../include/taseg_hip.h"

#define TS_WAVE 64

void ts_set_error(const char *fmt, ...);
extern int g_ts_conv_impl;  // 0 = MFMA kernels, 1 = scalar cross-check kernels (ts_set_conv_impl)

#define TS_REQUIRE(cond, code, ...)  \
  do {                               \
    if (!(cond)) {                   \
      ts_set_error(__VA_ARGS__);     \
      return (code);                 \
    }                                \
  } while (0)

#define TS_CHECK_LAUNCH(what)                                            \
  do {                                                                   \
    hipError_t e_ = hipGetLastError();                                   \
    if (e_ != hipSuccess) {                                              \
      ts_set_error("%s: launch failed: %s", what, hipGetErrorString(e_)); \
      return TS_ERR_LAUNCH_FAILED;                                       \
    }                                                                    \
  } while (0)

#define TS_CHECK_HIP(expr, what)                                      \
  do {                                                                \
    hipError_t e_ = (expr);                                           \
    if (e_ != hipSuccess) {                                           \
      ts_set_error("%s: %s", what, hipGetErrorString(e_));            \
      return TS_ERR_LAUNCH_FAILED;                                    \
    }                                                                 \
  } while (0)

static inline size_t ts_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static inline int64_t ts_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// FNV-1a-64 over the four coordinates, folded to 60 bits
// (reference: backend/hash/hash_cuda.cu:15-20).
__device__ __forceinline__ uint64_t ts_fnv60(int x, int y, int z, int b) {
  uint64_t h = 14695981039346656037ULL;
  h ^= (uint32_t)x;
  h *= 1099511628211ULL;
  h ^= (uint32_t)y;
  h *= 1099511628211ULL;
  h ^= (uint32_t)z;
  h *= 1099511628211ULL;
  h ^= (uint32_t)b;
  h *= 1099511628211ULL;
  return (h >> 60) ^ (h & 0x0FFFFFFFFFFFFFFFULL);
}

// ---- open-addressing hash table {uint64 key -> int32 value} -------------------
// keys[cap] (empty = ~0), vals[cap] (init INT_MAX-ish so atomicMin keeps the
// smallest position for duplicate keys).  cap is a power of two >= 2*n.
#define TS_EMPTY_KEY 0xFFFFFFFFFFFFFFFFULL

struct TsTable {
  unsigned long long *keys;
  int *vals;
  uint32_t mask;
};

static inline uint32_t ts_table_capacity(int64_t n) {
  uint64_t cap = 1024;
  while (cap < (uint64_t)(2 * n + 2)) cap <<= 1;
  return (uint32_t)cap;
}
static inline size_t ts_table_bytes(int64_t n) {
  size_t cap = ts_table_capacity(n);
  return ts_align_up(cap * 8, 256) + ts_align_up(cap * 4, 256);
}

__device__ __forceinline__ uint32_t ts_slot0(uint64_t h, uint32_t mask) {
  // fold the high half in: the low FNV bits alone are fine, this is cheap insurance
  uint64_t m = h * 0x9E3779B97F4A7C15ULL;
  return (uint32_t)(m >> 32) & mask;
}

__device__ __forceinline__ void ts_table_insert(const TsTable &t, uint64_t key, int val) {
  uint32_t s = ts_slot0(key, t.mask);
  for (uint32_t probe = 0; probe <= t.mask; ++probe) {
    unsigned long long prev = atomicCAS(&t.keys[s], (unsigned long long)TS_EMPTY_KEY,
                                        (unsigned long long)key);
    if (prev == TS_EMPTY_KEY || prev == key) {
      atomicMin(&t.vals[s], val);
      return;
    }
    s = (s + 1) & t.mask;
  }
}

__device__ __forceinline__ int ts_table_find(const TsTable &t, uint64_t key) {
  uint32_t s = ts_slot0(key, t.mask);
  for (uint32_t probe = 0; probe <= t.mask; ++probe) {
    unsigned long long k = t.keys[s];
    if (k == key) return t.vals[s];
    if (k == TS_EMPTY_KEY) return -1;
    s = (s + 1) & t.mask;
  }
  return -1;
}

// N look-ups at once: the first-slot key loads are independent and in flight together, then the value loads of the hits;
// only a key whose first slot holds ANOTHER key (load factor <= 0.5) walks on sequentially.  Same results as N calls of
// ts_table_find, which costs 2 dependent round trips per key, one key after the other.
template <int N>
__device__ __forceinline__ void ts_table_find_n(const TsTable &t, const unsigned long long (&want)[N], bool valid, int (&r)[N]) {
  uint32_t slot[N];
  unsigned long long got[N];
#pragma unroll
  for (int i = 0; i < N; ++i) slot[i] = ts_slot0(want[i], t.mask);
#pragma unroll
  for (int i = 0; i < N; ++i) got[i] = valid ? t.keys[slot[i]] : TS_EMPTY_KEY;
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = (got[i] == want[i]) ? t.vals[slot[i]] : -1;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (got[i] != want[i] && got[i] != TS_EMPTY_KEY) {
      uint32_t s = (slot[i] + 1) & t.mask;
      for (uint32_t probe = 0; probe < t.mask; ++probe) {
        const unsigned long long kk = t.keys[s];
        if (kk == want[i]) {
          r[i] = t.vals[s];
          break;
        }
        if (kk == TS_EMPTY_KEY) break;
        s = (s + 1) & t.mask;
      }
    }
  }
}

// One launch that fills up to TS_FILL_MAX buffers with a 32-bit pattern each (pointers 4-byte aligned, sizes multiples of
// 4 bytes; empty segments are skipped) - the index-plan builders clear their tables and counters with it instead of one
// hipMemsetAsync per buffer.
#define TS_FILL_MAX 8
struct TsFillSeg {
  void *p;
  size_t bytes;
  uint32_t word;
};
int ts_fill_segments(const TsFillSeg *segs, int n, hipStream_t stream);

// Carve a table out of a workspace and reset it on `stream`.  `extra` (n_extra <= TS_FILL_MAX - 2 segments) is cleared by
// the same launch.
int ts_table_init(TsTable *t, int64_t n, void *ws, size_t ws_bytes, hipStream_t stream, size_t *used,
                  const TsFillSeg *extra = nullptr, int n_extra = 0);

// ---- deterministic weight gradient ------------------------------------------------------------------------------
// A weight-gradient workgroup = (chunk c of consecutive rulebook pairs, TM x TN tile).  Instead of adding its partial
// tile of offset k into dW_k with float atomics (order = whatever workgroup finishes first), it stores the tile into
// slot (c + k) of a partial buffer [slots][C_a][C_b] - the pairs are ordered by offset, so (c, k) -> c + k is one to
// one - and dW_k = sum of the slots c0(k) + k .. c1(k) + k in ascending c: a fixed summation order, plain stores, no
// fill of dW.  The sum is formed by `wgrad_reduce_kernel` or, in the fused block calls, on the side of the gather-sum
// launch that follows anyway (no extra launch).
struct TsWgradPlan {
  int chunk;          // pairs per workgroup along the list
  int n_chunks;       // ceil(n_pairs / chunk)
  int64_t slots;      // n_chunks + K
};
// workgroups a weight-gradient launch aims for (TS_OPT_WGRAD_WGS; default 512 = two resident rounds of 256 CUs).  Round 6 measured
// 256 (chunks of the strides-4 .. 16 layers twice as long, half the partial tiles) on the whole step and per family
// (profiles/r06_wgrad_wgs_ab.txt): the bs-2 steps gain 1 % (fp32 15.08 -> 14.94 ms, autocast 9.47 -> 9.36) because the gradient
// then leaves more of the chip to the input gradient beside it on the second stream - but the family ALONE gets slower (3.51 ->
// 3.94 ms per step, 0.45 -> 0.40 of HBM: fewer workgroups cost the mid-size layers more than the partial tiles they save) and the
// large batches lose (mask distillation at bs 6: 97 -> 104 ms).  512 stays.  One target for every mode either way: the target fixes
// the summation order, and the bits must not depend on the stream a gradient runs on.
static inline int ts_wgrad_wgs() {
  const int64_t v = ts_get_option(TS_OPT_WGRAD_WGS);
  return v ? (int)v : 512;
}
static inline TsWgradPlan ts_wgrad_plan(int64_t n_pairs, int tiles, int K, int step, int max_chunk) {
  int64_t chunk = ts_cdiv(n_pairs * tiles, ts_wgrad_wgs());
  chunk = std::min<int64_t>(max_chunk, std::max<int64_t>(128, (chunk + step - 1) / step * step));
  TsWgradPlan p;
  p.chunk = (int)chunk;
  p.n_chunks = (int)ts_cdiv(std::max<int64_t>(n_pairs, 1), chunk);
  p.slots = (int64_t)p.n_chunks + K;
  return p;
}
struct TsWgradReduce {       // everything the ordered sum needs (by value into the kernels)
  const float *part;         // [slots][cacb]
  const int *nboffs;         // [K + 1]
  float *dW;                 // [K][cacb]
  int K, chunk;
  int64_t cacb4;             // C_a * C_b / 4 (float4 granules; C_a * C_b is a multiple of 4 on this path)
};
extern thread_local float *g_ts_wgrad_part;    // != nullptr: the next weight-gradient launch stores partial tiles here
extern thread_local TsWgradPlan g_ts_wgrad_plan;   // ... and leaves the plan it used here
int ts_wgrad_reduce(const TsWgradReduce &job, ts_stream_t stream);
int ts_wgrad_reduce_seq(const TsWgradReduce &job, ts_stream_t stream);   // ... in the riding form's order (chunk after chunk)
size_t ts_wgrad_partial_bytes(int64_t n_pairs, int32_t c_a, int32_t c_b, int32_t K);

__device__ __forceinline__ void ts_wgrad_reduce_one(const TsWgradReduce &job, int64_t i) {
  const int k = (int)(i / job.cacb4);
  const int64_t e = i - (int64_t)k * job.cacb4;
  const int lo = job.nboffs[k], hi = job.nboffs[k + 1];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (hi > lo) {
    const int c0 = lo / job.chunk, c1 = (hi - 1) / job.chunk;
    const float4 *src = (const float4 *)job.part + ((int64_t)c0 + k) * job.cacb4 + e;
#pragma unroll 4
    for (int c = c0; c <= c1; ++c, src += job.cacb4) {      // ascending chunk: the fixed summation order
      const float4 v = *src;
      acc.x += v.x;
      acc.y += v.y;
      acc.z += v.z;
      acc.w += v.w;
    }
  }
  ((float4 *)job.dW)[i] = acc;
}

// Communicator sentinels of ts_bn_sync_* / ts_conv_block_* (include/taseg_hip.h): the CALLER owns the all-reduce and splits the
// call in two around it - PRE stops after the local sums, POST starts behind the all-reduce.
#define TS_COMM_CALLER_PRE ((void *)1)
#define TS_COMM_CALLER_POST ((void *)2)

// ---- pre-split weight planes (conv_pairs_s.hip) -------------------------------------------------------------------
struct TsPlanesHint {
  const float *w;                  // the weight the planes were split from
  const unsigned short *planes;    // [3][K * c_in * c_out] bf16: h | m | l, each in W's own layout
  int K, c_in, c_out;
};
extern thread_local TsPlanesHint g_ts_planes_hint;   // one-shot: set by ts_conv_planes_hint, cleared by the call that reads it

// class-sorted implicit GEMM (csrc/conv_class.hip), library-internal forms: + the ordered weight-gradient sum riding on the launch,
// + the finish of a three-group plan inside the product (fin != NULL, rows == NULL, groups == 3): two launches - the groups
// without the centre offset write their Z' rows, then the centre group's tiles (every row is in one: a row is its own centre
// neighbour) add the other groups' rows of their output rows in pass 2's order, the addend, and store the RESULT rows - no pass 2,
// a third of Z' never written; the same bits as class GEMM + ts_conv_gather_sum_ex.
struct TsClassFinish {
  const int32_t *pos;        // [3][n] position table of the plan
  int64_t n;                 // output rows
  void *out;                 // [n, c_out] result
  const void *addend;        // optional [n, c_out]
};
int ts_conv_class_gemm_ex(const float *feat, int32_t c_red, const float *kernel, int32_t K, int32_t groups, int32_t c_out,
                          const int32_t *src, int64_t m_pad, const int32_t *tile_info, const int32_t *n_tiles, int32_t wt,
                          int32_t mirror, const int32_t *rows, float *zp, const TsWgradReduce *side, const TsClassFinish *fin,
                          ts_stream_t stream);
int ts_conv_class_gemm_f16_ex(const void *feat, int32_t c_red, const void *w, int32_t K, int32_t groups, int32_t c_out,
                              const int32_t *src, int64_t m_pad, const int32_t *tile_info, const int32_t *n_tiles, int32_t wt,
                              int32_t mirror, const int32_t *rows, void *zp, const TsWgradReduce *side, const TsClassFinish *fin,
                              ts_stream_t stream);

// Library-internal forms used by the fused block calls (csrc/block.hip): the gather-sum pass can form the ordered sum
// of the weight-gradient partials on the side (saves the reduce launch), and the atomic form of the weight gradient
// can be told that its output is already zero.
// addend (optional, [n_rows, c] like out): out = (sum_k z rows) + addend - the gradient that reaches the block's input
// along its other path (a residual connection) lands in the same store instead of a separate add launch
int ts_conv_gather_sum_ex(const float *z, int32_t c, const int32_t *pos, int32_t K, int64_t n_rows, int64_t n_pairs,
                          float *out, const TsWgradReduce *side, const float *addend, ts_stream_t stream);
int ts_conv_gather_sum_f16_ex(const void *z, int32_t c, const int32_t *pos, int32_t K, int64_t n_rows, int64_t n_pairs,
                              void *out, const TsWgradReduce *side, const void *addend, ts_stream_t stream);
// Pass 2 with the evaluation block's elementwise tail in its store: out = act((sum - mean) invstd w + b [+ residual]) - the
// arithmetic of bn_act_fwd_kernel on the fp32 sum (half storage: the sum is not rounded to half in between).  Returns
// TS_ERR_UNSUPPORTED (and launches nothing) where the list form of pass 2 does not apply: the caller then runs the two launches.
struct TsGatherEpilogue {
  const float *mean, *invstd, *w, *b;      // [c]
  const void *residual;                    // optional [n_rows, c]
  int relu;
};
int ts_conv_gather_sum_epi(const float *z, int32_t c, const int32_t *pos, int32_t K, int64_t n_rows, int64_t n_pairs, float *out,
                           const TsGatherEpilogue &epi, ts_stream_t stream);
int ts_conv_gather_sum_f16_epi(const void *z, int32_t c, const int32_t *pos, int32_t K, int64_t n_rows, int64_t n_pairs, void *out,
                               const TsGatherEpilogue &epi, ts_stream_t stream);
int ts_conv_wgrad_ex(const float *a_feat, int32_t c_a, const float *b_feat, int32_t c_b, const int32_t *nbmaps,
                     const int32_t *nboffs, int32_t K, int32_t col_a, int64_t n_pairs, float *grad_kernel,
                     int32_t already_zero, ts_stream_t stream);
int ts_conv_wgrad_f16_ex(const void *a_feat, int32_t c_a, const void *b_feat, int32_t c_b, const int32_t *nbmaps,
                         const int32_t *nboffs, int32_t K, int32_t col_a, int64_t n_pairs, float *grad_kernel,
                         int32_t already_zero, ts_stream_t stream);
