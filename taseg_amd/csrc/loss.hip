// Lovasz-softmax on the device around ONE sort (reference: tools/utils/common/lovasz_losses.py:158-227, the authors'
// per-class loop with classes = 'present'; taseg_amd/pcseg/loss/lovasz.py states the batched form this follows).
//
//   ts_lovasz_errors   err[c, p] = |fg(c, p) - probas[p, c]| for the rows that count (0 otherwise), class-major so that
//                      the caller's sort runs along the contiguous dimension
//   (caller)           errors_sorted, perm = sort(err, dim = 1, descending)
//   ts_lovasz_grad     from the sorted order: foreground prefix sums (tile counts -> scan of the tile totals -> tile-local
//                      scan), the Lovasz gradient g_i = J_i - J_(i-1) of the Jaccard extension, the per-class dot products
//                      sum_i errors_sorted_i * g_i, the mean over the classes present, and - in the same pass - the
//                      gradient w.r.t. the probabilities scattered back through the permutation:
//                          d loss / d probas[perm_i, c] = sign_i * g_i * present_c / #present,
//                      sign_i = -1 on foreground, +1 elsewhere, 0 where the error is 0 (torch's sign(0)).
// The prefix sums are exact small integers in fp32 like the reference's cumsums; the summation order of the dot product
// is fixed (tile by tile), so the value is run-to-run deterministic.
#include "common.h"

#define LV_TILE 1024          // elements per workgroup (256 threads x 4); LV_BIG x as many in the calls with the byte label table
#define LV_BIG_PER 2      // elements per thread in the large calls

__global__ __launch_bounds__(256) void lovasz_errors_kernel(const float *__restrict__ probas,
                                                            const int64_t *__restrict__ labels, int64_t ignore,
                                                            int64_t P, int C, float *__restrict__ err) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int64_t lab = labels[p];
  const bool valid = lab != ignore;
  for (int c = 0; c < C; ++c) {
    const float pr = probas[p * C + c];
    const float fg = (valid && lab == c) ? 1.f : 0.f;
    err[(int64_t)c * P + p] = valid ? fabsf(fg - pr) : 0.f;
  }
}

extern "C" int ts_lovasz_errors(const float *probas, const int64_t *labels, int64_t ignore, int64_t n_points,
                                int32_t n_classes, float *errors, ts_stream_t stream) {
  TS_REQUIRE(n_points >= 0 && n_classes > 0, TS_ERR_INVALID_ARGUMENT, "ts_lovasz_errors: bad sizes");
  if (n_points == 0) return TS_OK;
  TS_REQUIRE(probas && labels && errors, TS_ERR_INVALID_ARGUMENT, "ts_lovasz_errors: null pointer");
  lovasz_errors_kernel<<<(unsigned)ts_cdiv(n_points, 256), 256, 0, (hipStream_t)stream>>>(probas, labels, ignore, n_points,
                                                                                          n_classes, errors);
  TS_CHECK_LAUNCH("ts_lovasz_errors");
  return TS_OK;
}

__device__ __forceinline__ int lv_fg(const int64_t *__restrict__ labels, int64_t ignore, int64_t src, int c) {
  const int64_t lab = labels[src];
  return (lab != ignore && lab == c) ? 1 : 0;
}
// Large clouds (the dense image loss of TIAF: 4.9 M rows x 20 classes): the label of every element of every class row is looked up
// through the sort permutation - 98 M random reads.  From the int64 labels (39 MB) they miss L2; a byte table (4.9 MB: 255 = ignored /
// no class) stays in it.  Built once per call in the workspace.
__device__ __forceinline__ int lv_fg(const uint8_t *__restrict__ labels, int64_t, int64_t src, int c) { return labels[src] == c ? 1 : 0; }

__global__ __launch_bounds__(256) void lovasz_pack_labels_kernel(const int64_t *__restrict__ labels, int64_t ignore, int64_t P,
                                                                 uint8_t *__restrict__ lab8) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int64_t lab = labels[p];
  lab8[p] = (lab != ignore && lab >= 0 && lab < 255) ? (uint8_t)lab : (uint8_t)255;
}
#define LV_BYTE_LABELS_FROM (1 << 20)   // points from which the byte table is used (below, the int64 labels fit L2 themselves)

// foreground count of every tile of every class
template <typename LabT, int PER>
__global__ __launch_bounds__(256) void lovasz_tile_count_kernel(const int64_t *__restrict__ perm,
                                                                const LabT *__restrict__ labels, int64_t ignore,
                                                                int64_t P, int tiles, int *__restrict__ tile_fg) {
  __shared__ int red[4];
  const int c = blockIdx.y, t = blockIdx.x;
  const int64_t base = (int64_t)t * (256 * PER) + threadIdx.x * PER;
  int cnt = 0;
#pragma unroll
  for (int u = 0; u < PER; ++u)
    if (base + u < P) cnt += lv_fg(labels, ignore, perm[(int64_t)c * P + base + u], c);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) tile_fg[c * tiles + t] = red[0] + red[1] + red[2] + red[3];
}

// one workgroup per class: exclusive scan of its tile counts and the class total
__global__ __launch_bounds__(256) void lovasz_tile_scan_kernel(int *__restrict__ tile_fg, int tiles, float *__restrict__ gts) {
  __shared__ int carry;
  __shared__ int wsum[4];
  const int c = blockIdx.x;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int t0 = 0; t0 < tiles; t0 += 256) {
    const int t = t0 + threadIdx.x;
    const int v = t < tiles ? tile_fg[c * tiles + t] : 0;
    int inc = v;                                       // inclusive scan over the 256 threads
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(inc, d, 64);
      if ((threadIdx.x & 63) >= d) inc += up;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += wsum[w];
    if (t < tiles) tile_fg[c * tiles + t] = before + inc - v;      // exclusive prefix
    __syncthreads();
    if (threadIdx.x == 255) carry = before + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) gts[c] = (float)carry;
}

// number of classes that occur (gts > 0), the same in every caller
__device__ __forceinline__ float lv_present(const float *__restrict__ gts, int C) {
  const int lane = threadIdx.x & 63;
  const unsigned long long m = __builtin_amdgcn_ballot_w64(lane < C && gts[min(lane, C - 1)] > 0.f);
  return (float)__builtin_popcountll(m);
}

// tile-local scan + Lovasz gradient + partial dot product + gradient scatter
template <typename LabT, int PER>
__global__ __launch_bounds__(256) void lovasz_grad_kernel(const float *__restrict__ errors_sorted,
                                                          const int64_t *__restrict__ perm,
                                                          const LabT *__restrict__ labels, int64_t ignore, int64_t P,
                                                          int C, int tiles, const int *__restrict__ tile_base,
                                                          const float *__restrict__ gts,
                                                          float *__restrict__ tile_loss, float *__restrict__ dprob, int class_major) {
  __shared__ int wsum[4];
  __shared__ float wred[4];
  const int c = blockIdx.y, t = blockIdx.x;
  const int64_t i0 = (int64_t)t * (256 * PER) + threadIdx.x * PER;
  int64_t src[PER];
  int fg[PER];
  float es[PER];
  int mine = 0;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const bool in = i0 + u < P;
    src[u] = in ? perm[(int64_t)c * P + i0 + u] : 0;
    fg[u] = in ? lv_fg(labels, ignore, src[u], c) : 0;
    es[u] = in ? errors_sorted[(int64_t)c * P + i0 + u] : 0.f;
    mine += fg[u];
  }
  int inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(inc, d, 64);
    if ((threadIdx.x & 63) >= d) inc += up;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  int before = tile_base[c * tiles + t];
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += wsum[w];
  int cum = before + inc - mine;                          // foreground in front of this thread's first element
  const float g = gts[c];
  const float np = lv_present(gts, C);
  const float scale = (g > 0.f && np > 0.f) ? 1.f / np : 0.f;     // present_c / #present
  float part = 0.f;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    if (i0 + u < P) {
      const float rank = (float)(i0 + u + 1);
      const float cum_prev = (float)cum;
      cum += fg[u];
      const float cum_i = (float)cum;
      // jaccard_i = 1 - (gts - cum_i) / (gts + rank - cum_i); jaccard_(i-1) likewise with rank - 1 (0 in front of the row)
      const float j_i = 1.f - (g - cum_i) / (g + (rank - cum_i));
      const float j_p = (i0 + u == 0) ? 0.f : 1.f - (g - cum_prev) / (g + ((rank - 1.f) - cum_prev));
      const float gr = j_i - j_p;
      part += es[u] * gr;
      const float sign = es[u] > 0.f ? (fg[u] ? -1.f : 1.f) : 0.f;
      // the scatter through the sort permutation: 4-byte writes at random rows.  Row-major [P, C] they are spread over the whole
      // array (393 MB for TIAF's dense image loss: each a read-modify-write of a 64-byte sector in HBM, 3.1 ms); class-major
      // [C, P] the class in flight writes one 4 P-byte slab, which the memory-side cache holds
      dprob[class_major ? (int64_t)c * P + src[u] : src[u] * C + c] = sign * gr * scale;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) part += __shfl_down(part, d, 64);
  if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) tile_loss[c * tiles + t] = (wred[0] + wred[1]) + (wred[2] + wred[3]);
}

// loss = sum over the classes present of (their tile partials: 32 lanes per class take the tiles lane, lane + 32, ...
// in order, then the lanes are added in lane order) / #present.  One workgroup of 1024 threads: 32 classes x 32 lanes
// per pass.
__global__ __launch_bounds__(1024) void lovasz_finish_kernel(const float *__restrict__ tile_loss, const float *__restrict__ gts,
                                                             int C, int tiles, float *__restrict__ loss) {
  __shared__ float cls[64];
  const int lane = threadIdx.x & 31, slot = threadIdx.x >> 5;
  for (int c0 = 0; c0 < C; c0 += 32) {
    const int c = c0 + slot;
    float v = 0.f;
    if (c < C && gts[c] > 0.f)
      for (int t = lane; t < tiles; t += 32) v += tile_loss[c * tiles + t];
    // lanes of one class sit in one half-wave: add them in lane order
    float tot = 0.f;
    for (int l = 0; l < 32; ++l) tot += __shfl(v, (threadIdx.x & 32) + l, 64);
    if (lane == 0 && c < C) cls[c] = tot;
  }
  __syncthreads();
  const float np = lv_present(gts, C);          // a ballot: every lane of the wave takes part
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int k = 0; k < C; ++k) s += cls[k];
    *loss = np > 0.f ? s / np : 0.f;
  }
}

extern "C" size_t ts_lovasz_workspace_bytes(int64_t n_points, int32_t n_classes) {
  const size_t tiles = (size_t)ts_cdiv(std::max<int64_t>(n_points, 1), 256 * LV_BIG_PER < LV_TILE ? 256 * LV_BIG_PER : LV_TILE);
  return ts_align_up(tiles * n_classes * 4, 256) * 2 + ts_align_up((size_t)n_classes * 4 + 4, 256) +
         (n_points >= LV_BYTE_LABELS_FROM ? ts_align_up((size_t)n_points, 256) : 0);
}

// errors_sorted / perm [C, P] = sort(errors, descending) of ts_lovasz_errors' output; loss [1]; grad_probas [P, C], or [C, P] with
// class_major != 0 (every element written).  ws >= ts_lovasz_workspace_bytes.
extern "C" int ts_lovasz_grad(const float *errors_sorted, const int64_t *perm, const int64_t *labels, int64_t ignore,
                              int64_t n_points, int32_t n_classes, float *loss, float *grad_probas, int32_t class_major, void *ws,
                              size_t ws_bytes, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(n_points > 0 && n_classes > 0 && n_classes <= 64, TS_ERR_INVALID_ARGUMENT,
             "ts_lovasz_grad: need points and 1 .. 64 classes");
  TS_REQUIRE(errors_sorted && perm && labels && loss && grad_probas && ws, TS_ERR_INVALID_ARGUMENT,
             "ts_lovasz_grad: null pointer");
  TS_REQUIRE(ws_bytes >= ts_lovasz_workspace_bytes(n_points, n_classes), TS_ERR_INVALID_ARGUMENT,
             "ts_lovasz_grad: workspace too small");
  const bool big = n_points >= LV_BYTE_LABELS_FROM;
  const int tiles = (int)ts_cdiv(n_points, big ? 256 * LV_BIG_PER : LV_TILE);
  char *p = (char *)ws;
  int *tile_fg = (int *)p;
  p += ts_align_up((size_t)tiles * n_classes * 4, 256);
  float *tile_loss = (float *)p;
  p += ts_align_up((size_t)tiles * n_classes * 4, 256);
  float *gts = (float *)p;
  p += ts_align_up((size_t)n_classes * 4 + 4, 256);
  dim3 grid((unsigned)tiles, (unsigned)n_classes);
  if (big) {
    uint8_t *lab8 = (uint8_t *)p;
    lovasz_pack_labels_kernel<<<(unsigned)ts_cdiv(n_points, 256), 256, 0, stream>>>(labels, ignore, n_points, lab8);
    lovasz_tile_count_kernel<uint8_t, LV_BIG_PER><<<grid, 256, 0, stream>>>(perm, lab8, ignore, n_points, tiles, tile_fg);
    lovasz_tile_scan_kernel<<<n_classes, 256, 0, stream>>>(tile_fg, tiles, gts);
    lovasz_grad_kernel<uint8_t, LV_BIG_PER><<<grid, 256, 0, stream>>>(errors_sorted, perm, lab8, ignore, n_points, n_classes, tiles, tile_fg, gts,
                                                          tile_loss, grad_probas, class_major);
  } else {
    lovasz_tile_count_kernel<int64_t, 4><<<grid, 256, 0, stream>>>(perm, labels, ignore, n_points, tiles, tile_fg);
    lovasz_tile_scan_kernel<<<n_classes, 256, 0, stream>>>(tile_fg, tiles, gts);
    lovasz_grad_kernel<int64_t, 4><<<grid, 256, 0, stream>>>(errors_sorted, perm, labels, ignore, n_points, n_classes, tiles, tile_fg, gts,
                                                          tile_loss, grad_probas, class_major);
  }
  lovasz_finish_kernel<<<1, 1024, 0, stream>>>(tile_loss, gts, n_classes, tiles, loss);
  TS_CHECK_LAUNCH("ts_lovasz_grad");
  return TS_OK;
}

// ------------------------------------------------------------------------------------------------------
// Cross entropy (ignore_index, label smoothing, reduction 'mean': torch/nn/functional.py cross_entropy as the reference's
// nn.CrossEntropyLoss uses it, pcseg/loss/__init__.py:40-44) fused with the softmax the Lovasz term needs:
//   ts_softmax_ce_forward   one pass over the logits: probas [P, C], the Lovasz error matrix [C, P] (ts_lovasz_errors'
//                           output) and, per workgroup, the partial sums (sum_valid -logp[y], sum_valid -sum_c logp_c, #valid)
//   ts_ce_lovasz_finish     loss = w_ce ((1 - eps) nll / n + eps smooth / (n C)) + w_lov lovasz       (partials in order)
//   ts_ce_lovasz_backward   d loss / d logits from the saved probas, the labels and d lovasz / d probas:
//                           go [ w_ce valid / n ((1 - eps)(p - onehot) + eps (p - 1 / C)) + w_lov p (dp - sum_c p_c dp_c) ]
// C <= 32; one thread per row.
#define CE_MAXC 32

// A workgroup's 256 rows are one contiguous piece of a [P, C] array: it moves between HBM and LDS as whole 256-byte wave rows
// (lane = consecutive float) and a thread reads / writes ITS row in LDS (pitch C | 1: odd, no bank conflicts).  One thread per row
// straight from global memory - lanes 4 C bytes apart - ran at 0.8 TB/s on the dense image loss of TIAF (4.9 M rows).
// CT: the class count at compile time (0: any count up to 32 at run time).
#define CE_ROWS 256
template <int CT>
__device__ __forceinline__ void ce_rows_in(const float *__restrict__ g, int n, int C, float *__restrict__ tile) {
  const int Cc = CT ? CT : C, pitch = Cc | 1;
  for (int i = threadIdx.x; i < n; i += CE_ROWS) tile[(i / Cc) * pitch + i % Cc] = g[i];
}
template <int CT>
__device__ __forceinline__ void ce_rows_out(const float *__restrict__ tile, int n, int C, float *__restrict__ g) {
  const int Cc = CT ? CT : C, pitch = Cc | 1;
  for (int i = threadIdx.x; i < n; i += CE_ROWS) g[i] = tile[(i / Cc) * pitch + i % Cc];
}

template <int CT>
__global__ __launch_bounds__(256) void softmax_ce_fwd_kernel(const float *__restrict__ logits,
                                                             const int64_t *__restrict__ labels, int64_t ignore,
                                                             int64_t P, int C_, float *__restrict__ probas,
                                                             float *__restrict__ err, double *__restrict__ part) {
  __shared__ float tile[CE_ROWS * (CE_MAXC + 1)];
  __shared__ double red[3][4];
  const int C = CT ? CT : C_, pitch = C | 1;
  const int64_t row0 = (int64_t)blockIdx.x * CE_ROWS;
  const int rows = (int)min((int64_t)CE_ROWS, P - row0);
  ce_rows_in<CT>(logits + row0 * C, rows * C, C, tile);
  __syncthreads();
  const int64_t p = row0 + threadIdx.x;
  float *mine = tile + threadIdx.x * pitch;
  double nll = 0.0, smooth = 0.0, cnt = 0.0;
  if (p < P) {
    float v[CE_MAXC], raw[CE_MAXC];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < CE_MAXC; ++c)
      if (c < C) {
        raw[c] = mine[c];
        mx = fmaxf(mx, raw[c]);
      }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < CE_MAXC; ++c)
      if (c < C) {
        v[c] = expf(raw[c] - mx);
        se += v[c];
      }
    const float lse = logf(se), inv = 1.f / se;
    const int64_t lab = labels[p];
    const bool valid = lab != ignore;
    float sum_logp = 0.f, picked = 0.f;
#pragma unroll
    for (int c = 0; c < CE_MAXC; ++c)
      if (c < C) {
        const float lp = (raw[c] - mx) - lse;
        sum_logp += lp;
        if (lab == c) picked = lp;
        const float pr = v[c] * inv;
        mine[c] = pr;                                      // (this thread's row: leaves with the workgroup below)
        const float fg = (valid && lab == c) ? 1.f : 0.f;
        err[(int64_t)c * P + p] = valid ? fabsf(fg - pr) : 0.f;
      }
    if (valid) {
      // a label that is neither the ignore index nor a class: torch's CrossEntropyLoss (the reference,
      // pcseg/loss/__init__.py:40-44) device-asserts; here the row poisons the sum - the loss comes out NaN instead of
      // silently training on a wrong term (a mis-mapped raw id such as 255)
      nll = (lab >= 0 && lab < C) ? -(double)picked : (double)NAN;
      smooth = -(double)sum_logp;
      cnt = 1.0;
    }
  }
  __syncthreads();
  ce_rows_out<CT>(tile, rows * C, C, probas + row0 * C);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    nll += __shfl_down(nll, d, 64);
    smooth += __shfl_down(smooth, d, 64);
    cnt += __shfl_down(cnt, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = nll;
    red[1][threadIdx.x >> 6] = smooth;
    red[2][threadIdx.x >> 6] = cnt;
  }
  __syncthreads();
  if (threadIdx.x < 3) part[(int64_t)blockIdx.x * 3 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) +
                                                                      (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

extern "C" int ts_softmax_ce_forward(const float *logits, const int64_t *labels, int64_t ignore, int64_t n_points,
                                     int32_t n_classes, float *probas, float *errors, double *partials,
                                     ts_stream_t stream) {
  TS_REQUIRE(n_points > 0 && n_classes > 0 && n_classes <= CE_MAXC, TS_ERR_INVALID_ARGUMENT,
             "ts_softmax_ce_forward: need points and 1 .. 32 classes");
  TS_REQUIRE(logits && labels && probas && errors && partials, TS_ERR_INVALID_ARGUMENT, "ts_softmax_ce_forward: null pointer");
  const unsigned grid = (unsigned)ts_cdiv(n_points, CE_ROWS);
#define CE_FWD(CT_) softmax_ce_fwd_kernel<CT_><<<grid, 256, 0, (hipStream_t)stream>>>(logits, labels, ignore, n_points, n_classes, probas, errors, partials)
  switch (n_classes) {            // (the class counts of the reference's configurations at compile time: divisions by a constant)
    case 20: CE_FWD(20); break;
    case 19: CE_FWD(19); break;
    case 17: CE_FWD(17); break;
    default: CE_FWD(0);
  }
#undef CE_FWD
  TS_CHECK_LAUNCH("ts_softmax_ce_forward");
  return TS_OK;
}

// out[0] = total loss, out[1] = cross entropy, out[2] = lovasz, out[3] = number of rows that count
__global__ __launch_bounds__(256) void ce_lovasz_finish_kernel(const double *__restrict__ part, int64_t blocks, int C,
                                                               float smoothing, float w_ce, float w_lov,
                                                               const float *__restrict__ lovasz, float *__restrict__ out) {
  __shared__ double red[3][256];
  double a = 0.0, b = 0.0, n = 0.0;
  for (int64_t i = threadIdx.x; i < blocks; i += 256) {      // fixed assignment, fixed order
    a += part[i * 3];
    b += part[i * 3 + 1];
    n += part[i * 3 + 2];
  }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = b;
  red[2][threadIdx.x] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    double nll = 0.0, sm = 0.0, cnt = 0.0;
    for (int k = 0; k < 256; ++k) {
      nll += red[0][k];
      sm += red[1][k];
      cnt += red[2][k];
    }
    // no row counts: 0 / 0 = NaN, as torch's mean over nothing
    const double ce = (1.0 - (double)smoothing) * (nll / cnt) + (double)smoothing * (sm / (cnt * C));
    const float lov = lovasz ? *lovasz : 0.f;
    out[0] = w_ce * (float)ce + w_lov * lov;
    out[1] = (float)ce;
    out[2] = lov;
    out[3] = (float)cnt;
  }
}

extern "C" int ts_ce_lovasz_finish(const double *partials, int64_t n_points, int32_t n_classes, float smoothing, float w_ce,
                                   float w_lov, const float *lovasz, float *out4, ts_stream_t stream) {
  TS_REQUIRE(n_points > 0 && n_classes > 0 && partials && out4, TS_ERR_INVALID_ARGUMENT, "ts_ce_lovasz_finish: bad arguments");
  ce_lovasz_finish_kernel<<<1, 256, 0, (hipStream_t)stream>>>(partials, ts_cdiv(n_points, 256), n_classes, smoothing, w_ce,
                                                              w_lov, lovasz, out4);
  TS_CHECK_LAUNCH("ts_ce_lovasz_finish");
  return TS_OK;
}

template <int CT>
__global__ __launch_bounds__(256) void ce_lovasz_bwd_kernel(const float *__restrict__ probas,
                                                            const int64_t *__restrict__ labels, int64_t ignore,
                                                            const float *__restrict__ dprob, const float *__restrict__ out4,
                                                            const float *__restrict__ grad_out, int64_t P, int C_,
                                                            float smoothing, float w_ce, float w_lov,
                                                            float *__restrict__ dlogits) {
  __shared__ float tile[CE_ROWS * (CE_MAXC + 1)];
  const int C = CT ? CT : C_, pitch = C | 1;
  const int64_t row0 = (int64_t)blockIdx.x * CE_ROWS;
  const int rows = (int)min((int64_t)CE_ROWS, P - row0);
  const int64_t p = row0 + threadIdx.x;
  float *mine = tile + threadIdx.x * pitch;
  float pr[CE_MAXC], dp[CE_MAXC];
  // probas in and the result out through the LDS image (see softmax_ce_fwd_kernel)
  ce_rows_in<CT>(probas + row0 * C, rows * C, C, tile);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < CE_MAXC; ++c)
    if (c < C) pr[c] = p < P ? mine[c] : 0.f;
  __syncthreads();
  float dot = 0.f;                                         // d lovasz / d probas: CLASS-major [C, P] (ts_lovasz_grad with class_major):
#pragma unroll                                             // lane = consecutive row, a coalesced read per class
  for (int c = 0; c < CE_MAXC; ++c)
    if (c < C) {
      dp[c] = (dprob && p < P) ? dprob[(int64_t)c * P + p] : 0.f;
      dot += pr[c] * dp[c];
    }
  if (p < P) {
    const float go = *grad_out;
    const float n = out4[3];
    const int64_t lab = labels[p];
    const float kce = (lab != ignore) ? go * w_ce / n : 0.f;
    const float klov = go * w_lov, invc = 1.f / (float)C;
#pragma unroll
    for (int c = 0; c < CE_MAXC; ++c)
      if (c < C) {
        const float onehot = lab == c ? 1.f : 0.f;
        const float ce = (1.f - smoothing) * (pr[c] - onehot) + smoothing * (pr[c] - invc);
        // a label outside [0, C) that is not the ignore index poisons the row's gradient like it poisons the forward loss (the
        // reference device-asserts, R/pcseg/loss/__init__.py:40-44 -> F.cross_entropy): the optimizer must not step on it
        const bool bad = lab != ignore && (lab < 0 || lab >= C);
        mine[c] = bad ? __builtin_nanf("") : kce * ce + klov * pr[c] * (dp[c] - dot);
      }
  }
  __syncthreads();
  ce_rows_out<CT>(tile, rows * C, C, dlogits + row0 * C);
}

extern "C" int ts_ce_lovasz_backward(const float *probas, const int64_t *labels, int64_t ignore, const float *grad_probas,
                                     const float *out4, const float *grad_out, int64_t n_points, int32_t n_classes,
                                     float smoothing, float w_ce, float w_lov, float *grad_logits, ts_stream_t stream) {
  TS_REQUIRE(n_points > 0 && n_classes > 0 && n_classes <= CE_MAXC, TS_ERR_INVALID_ARGUMENT,
             "ts_ce_lovasz_backward: need points and 1 .. 32 classes");
  TS_REQUIRE(probas && labels && out4 && grad_out && grad_logits, TS_ERR_INVALID_ARGUMENT, "ts_ce_lovasz_backward: null pointer");
  const unsigned grid = (unsigned)ts_cdiv(n_points, CE_ROWS);
#define CE_BWD(CT_) ce_lovasz_bwd_kernel<CT_><<<grid, 256, 0, (hipStream_t)stream>>>(probas, labels, ignore, grad_probas, out4, grad_out, n_points, n_classes, smoothing, w_ce, w_lov, grad_logits)
  switch (n_classes) {
    case 20: CE_BWD(20); break;
    case 19: CE_BWD(19); break;
    case 17: CE_BWD(17); break;
    default: CE_BWD(0);
  }
#undef CE_BWD
  TS_CHECK_LAUNCH("ts_ce_lovasz_backward");
  return TS_OK;
}
