/*
 * taseg_hip.h — C ABI of libtaseg_hip.so, the MI355X (gfx950) backend for the
 * TASeg / OpenPCSeg sparse-convolution hot path.
 *
 * Every entry point below replaces one function of the reference's native
 * backend (the pybind module `torchsparse.backend`, mit-han-lab/torchsparse
 * v1.4.0 vendored at /root/reference/package/torchsparse.zip; paths below are
 * relative to the zip member `torchsparse/`) or one tensor-op sequence of its
 * Python rulebook construction.  The reference interface each one replaces is
 * cited as file:line.
 *
 * Conventions
 *   - plain C: device pointers + sizes, no torch / ATen types;
 *   - all pointers are DEVICE pointers unless the name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *     nothing here synchronises the host, allocates or frees device memory:
 *     scratch comes in through (`ws`, `ws_bytes`), sized by the matching
 *     `*_workspace_bytes()` query;
 *   - return value: TS_OK (0) or a negative TS_ERR_* code; ts_last_error()
 *     returns a thread-local message for the last failure (the reference
 *     throws std::invalid_argument, e.g. convolution_cuda.cu:57-59);
 *   - coordinates are int32 [N,4] rows (x, y, z, batch)  (utils/collate.py:26-31);
 *   - features are row-major float32 [N, C].
 */
#ifndef TASEG_HIP_H_
#define TASEG_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TS_OK 0
#define TS_ERR_INVALID_ARGUMENT (-1)
#define TS_ERR_WORKSPACE_TOO_SMALL (-2)
#define TS_ERR_LAUNCH_FAILED (-3)
#define TS_ERR_UNSUPPORTED (-4)

typedef void *ts_stream_t;

/* Library identity: "taseg_hip <version> gfx950". */
const char *ts_version(void);
/* Thread-local message describing the last non-TS_OK return. */
const char *ts_last_error(void);

/* Tuning values of the library (process-wide; the host side's options object - taseg_amd/options.py - pushes its fields when it
 * loads the library; the library itself reads no environment variable).  0 is every key's default.
 *   TS_OPT_GATHER_POSITIONS        != 0: pass 2 with K position registers per lane instead of LDS lists
 *   TS_OPT_WGRAD_WGS               workgroup target of the weight gradient's chunking (64 .. 8192; 0 = 512)
 *   TS_OPT_EVAL_TAIL_SEPARATE      != 0: the evaluation block's BatchNorm tail as a launch of its own
 *   TS_OPT_CLASS_FINISH_ROWS[_HALF] rows from which a class plan's finish runs inside the product (0 = never)
 *   TS_OPT_DEBUG_BN_ABLATE         timing diagnostics of csrc/bn.hip (bit field)
 *   TS_OPT_KMAP_FULL_PROBE         != 0: submanifold kernel maps on all 27 probes
 * ts_set_option returns TS_ERR_INVALID_ARGUMENT for an unknown key. */
enum {
  TS_OPT_GATHER_POSITIONS = 0,
  TS_OPT_WGRAD_WGS = 1,
  TS_OPT_EVAL_TAIL_SEPARATE = 2,
  TS_OPT_CLASS_FINISH_ROWS = 3,
  TS_OPT_CLASS_FINISH_ROWS_HALF = 4,
  TS_OPT_DEBUG_BN_ABLATE = 5,
  TS_OPT_KMAP_FULL_PROBE = 6,
  TS_OPT_COUNT = 7
};
int ts_set_option(int32_t key, int64_t value);
int64_t ts_get_option(int32_t key);

/* ------------------------------------------------------------------------ */
/* 1. The ten entry points of torchsparse.backend                           */
/*    (backend/pybind_cuda.cpp:18-39)                                       */
/* ------------------------------------------------------------------------ */

/* hash_cuda (backend/hash/hash_cuda.cu:10-23,67-73):
 * out[i] = fold60(FNV-1a-64 over (uint32)x,y,z,b).  coords [n,4], out [n]. */
int ts_hash(const int32_t *coords, int64_t n, int64_t *out, ts_stream_t stream);

/* kernel_hash_cuda (backend/hash/hash_cuda.cu:27-55,75-84):
 * out[k*n + i] = hash(x+ox[k], y+oy[k], z+oz[k], b).  offsets [K,3], out [K,n].
 * The row's own batch index is used (CUDA semantics, hash_cuda.cu:46), not
 * row 0's as in the CPU twin (hash_cpu.cpp:29). */
int ts_kernel_hash(const int32_t *coords, int64_t n, const int32_t *offsets,
                   int32_t n_offsets, int64_t *out, ts_stream_t stream);

/* hash_query_cuda (backend/others/query_cuda.cu:9-56):
 * out[q] = ref_idx[j] + 1 where ref_hash[j] == query[q], else 0.
 * ref_idx == NULL means arange(n_ref) (nn/functional/query.py:18-20).
 * Duplicate reference keys resolve to the smallest j (the CPU twin's
 * dense_hash_map::insert keeps the first, others/query_cpu.cpp:20-24).
 * Keys must not be -1 (reserved as the empty slot). */
size_t ts_hash_query_workspace_bytes(int64_t n_ref);
int ts_hash_query(const int64_t *query, int64_t n_query, const int64_t *ref_hash,
                  const int64_t *ref_idx, int64_t n_ref, int64_t *out, void *ws,
                  size_t ws_bytes, ts_stream_t stream);

/* count_cuda (backend/others/count_cuda.cu:10-31):
 * out[idx[i]]++ for idx[i] >= 0; out [n_out] is zeroed here first. */
int ts_count(const int32_t *idx, int64_t n, int32_t *out, int64_t n_out,
             ts_stream_t stream);

/* voxelize_forward_cuda (backend/voxelize/voxelize_cuda.cu:12-25,43-60):
 * out[idx[i], :] += feat[i, :] / counts[idx[i]]; out [m,c] is zeroed here. */
int ts_voxelize_forward(const float *feat, const int32_t *idx, const int32_t *counts,
                        int64_t n, int32_t c, int64_t m, float *out,
                        ts_stream_t stream);
/* voxelize_backward_cuda (voxelize_cuda.cu:28-41,62-80):
 * grad_feat[i, :] = grad_out[idx[i], :] / counts[idx[i]] (0 for skipped rows). */
int ts_voxelize_backward(const float *grad_out, const int32_t *idx,
                         const int32_t *counts, int64_t n, int32_t c, int64_t m,
                         float *grad_feat, ts_stream_t stream);

/* devoxelize_forward_cuda (backend/devoxelize/devoxelize_cuda.cu:11-33,61-78):
 * out[i,:] = sum_{k<8} w[i,k] * feat[idx[i,k],:]  (idx<0 contributes 0). */
int ts_devoxelize_forward(const float *feat, const int32_t *idx, const float *weight,
                          int64_t n, int32_t c, int64_t m, float *out,
                          ts_stream_t stream);
/* devoxelize_backward_cuda (devoxelize_cuda.cu:37-57,82-98): the exact adjoint,
 * grad_feat[idx[i,k],:] += w[i,k] * grad_out[i,:]; grad_feat [m,c] zeroed here. */
int ts_devoxelize_backward(const float *grad_out, const int32_t *idx,
                           const float *weight, int64_t n, int32_t c, int64_t m,
                           float *grad_feat, ts_stream_t stream);

/* convolution_forward_cuda (backend/convolution/convolution_cuda.cu:53-165):
 * out[o,:] = sum_k sum_{(i,o) in map_k} in[i,:] @ kernel[k]   (out is overwritten).
 * nbmap [P,2] int32 rows (in_idx, out_idx) grouped by offset k in order;
 * nbsizes_host [K] int32 on the HOST, exactly as the reference passes it
 * (nn/functional/conv.py:56).  transpose swaps the two map columns
 * (convolution_cuda.cu:21,34).  kernel is [K, c_in, c_out]. */
size_t ts_convolution_workspace_bytes(int64_t n_in, int64_t n_out, int32_t c_in,
                                      int32_t c_out, int32_t kernel_volume);
int ts_convolution_forward(const float *in_feat, int64_t n_in, int32_t c_in,
                           float *out_feat, int64_t n_out, int32_t c_out,
                           const float *kernel, int32_t kernel_volume,
                           const int32_t *nbmap, const int32_t *nbsizes_host,
                           int32_t transpose, void *ws, size_t ws_bytes,
                           ts_stream_t stream);
/* convolution_backward_cuda (convolution_cuda.cu:167-278):
 * grad_in = adjoint wrt in_feat, grad_kernel[k] = gather(in)^T @ gather(grad_out);
 * both outputs are overwritten.  grad_in may be NULL (input needs no grad). */
int ts_convolution_backward(const float *in_feat, int64_t n_in, int32_t c_in,
                            float *grad_in, const float *grad_out, int64_t n_out,
                            int32_t c_out, const float *kernel, float *grad_kernel,
                            int32_t kernel_volume, const int32_t *nbmap,
                            const int32_t *nbsizes_host, int32_t transpose, void *ws,
                            size_t ws_bytes, ts_stream_t stream);

/* ------------------------------------------------------------------------ */
/* 2. Rulebook construction (replaces the Python tensor-op sequences)        */
/* ------------------------------------------------------------------------ */

/* spdownsample (nn/functional/downsample.py:25-51), the branch taken when
 * stride[k] in {1, kernel_size[k]}:  c' = trunc(c / s) * s per axis, then the
 * unique rows in lexicographic (b, x, y, z) order (torch.unique(dim=0) of the
 * [b,x,y,z] permutation), returned as (x, y, z, b).
 * out_coords has room for n rows; *out_count (device int32) gets the number
 * of unique rows.  Supported range: 0 <= b < 1024, -2^17 <= x,y,z < 2^17
 * (checked on device; *out_count = -1 on violation). */
size_t ts_downsample_workspace_bytes(int64_t n);
int ts_downsample(const int32_t *coords, int64_t n, int32_t sx, int32_t sy, int32_t sz,
                  int32_t *out_coords, int32_t *out_count, void *ws, size_t ws_bytes,
                  ts_stream_t stream);

/* torch.unique(int64) + position lookup as used by initial_voxelize
 * (pcseg/model/segmentor/voxel/minkunet/utils.py:16-18):
 * uniq = ascending unique values of keys (non-negative, < 2^62),
 * inverse[i] = position of keys[i] in uniq (either output may be NULL),
 * *out_count (device int32) = number of unique values. */
size_t ts_unique_workspace_bytes(int64_t n);
int ts_unique_i64(const int64_t *keys, int64_t n, int64_t *uniq, int32_t *inverse,
                  int32_t *out_count, void *ws, size_t ws_bytes, ts_stream_t stream);

/* Kernel map construction: F.sphash(in) + F.sphash(out, offsets) +
 * F.sphashquery + the nonzero()/sum() compaction
 * (nn/functional/conv.py:156-176).
 *   nbr      [K, n_out] int32 : index of the input voxel at out_coords[j]+offsets[k], or -1
 *                               (== `results` of conv.py:166);
 *   nbr_t    [K, n_in]  int32 : the inverse table (output index j per (k, input i), or -1);
 *                               may be NULL;
 *   nbmaps   [K*n_out, 2] int32 capacity: rows (in_idx, out_idx) ordered by (k, out_idx)
 *                               (== conv.py:169-173); may be NULL together with nboffs;
 *   nbsizes  [K] int32        : hits per offset (== conv.py:168);
 *   nboffs   [K+1] int32      : exclusive prefix of nbsizes (nboffs[K] = P). */
size_t ts_build_kmap_workspace_bytes(int64_t n_in, int64_t n_out, int32_t n_offsets);
/*   ts_build_kmap_sym: the same tables, bit for bit, for a SUBMANIFOLD map (one coordinate set, K odd, offsets[K - 1 - k] =
 *   -offsets[k]: get_kernel_offsets of an odd kernel, nn/utils/kernel.py:11-32) on HALF the probes - offset k of voxel j finding r is
 *   offset K - 1 - k of r finding j.  Needs unique coordinates; a duplicate is detected on the device: nboffs[K] then reads -1 and
 *   the caller builds the map with ts_build_kmap.  Workspace: ts_build_kmap_workspace_bytes(n, n, K). */
/*   pos_out  [K, n_out] int32 : row of nbmaps holding the pair (k, j), or -1  (may be NULL);
 *   pos_in   [K, n_in]  int32 : row of nbmaps holding the pair of input i at offset k, or -1 (may be NULL). */
int ts_build_kmap(const int32_t *in_coords, int64_t n_in, const int32_t *out_coords,
                  int64_t n_out, const int32_t *offsets, int32_t n_offsets, int32_t *nbr,
                  int32_t *nbr_t, int32_t *nbmaps, int32_t *nbsizes, int32_t *nboffs,
                  int32_t *pos_out, int32_t *pos_in, void *ws, size_t ws_bytes,
                  ts_stream_t stream);
int ts_build_kmap_sym(const int32_t *coords, int64_t n, const int32_t *offsets, int32_t K, int32_t *nbr, int32_t *nbr_t, int32_t *nbmaps,
                      int32_t *nbsizes, int32_t *nboffs, int32_t *pos_out, int32_t *pos_in, void *ws, size_t ws_bytes, ts_stream_t stream);

/* Neighbour table from an explicit rulebook (the reference-form entry points
 * above go through this): nbr[k, col_out] = col_in for every pair of offset k.
 * nboffs [K+1] device prefix of the pair counts. */
int ts_nbr_from_nbmaps(const int32_t *nbmaps, const int32_t *nboffs, int32_t n_offsets,
                       int32_t col_in, int64_t n_rows, int32_t *nbr, ts_stream_t stream);

/* calc_ti_weights + the 8-corner lookup of voxel_to_point
 * (pcseg/.../minkunet/utils.py:72-82, nn/functional/devoxelize.py:10-48):
 * for point p (float [n,4] = x,y,z,b) and voxel grid `vox_coords` at stride s,
 * idx[i,k] = voxel at floor(p/s)*s + off_k (off = get_kernel_offsets(2, s), x
 * outermost) or -1, weight[i,k] = trilinear weight, masked and renormalised. */
size_t ts_trilinear_workspace_bytes(int64_t n_vox);
int ts_trilinear_map(const float *points, int64_t n_points, const int32_t *vox_coords,
                     int64_t n_vox, int32_t stride, int32_t *idx, float *weight, void *ws,
                     size_t ws_bytes, ts_stream_t stream);

/* ------------------------------------------------------------------------ */
/* 3. Sparse convolution on the neighbour table (the production kernels)     */
/* ------------------------------------------------------------------------ */

/* Output-stationary gather -> MFMA -> write-once convolution.
 *   out[j, :] = sum_k  in[nbr[k, j], :] @ W_k          (rows with nbr<0 skipped)
 * weight_transposed == 0: W_k = kernel[k]      ([c_in, c_out], forward);
 * weight_transposed == 1: W_k = kernel[k]^T    (kernel is [K, c_out, c_in]; this
 *                         is dgrad: in = grad_out, nbr = the inverse table).
 * out is overwritten; every row j < n_out is written exactly once. */
int ts_conv_nbr(const float *in_feat, int64_t n_in, int32_t c_in, const float *kernel,
                int32_t kernel_volume, int32_t weight_transposed, const int32_t *nbr,
                float *out_feat, int64_t n_out, int32_t c_out, ts_stream_t stream);

/* Weight gradient:  grad_kernel[k] = sum_{pairs p of k} a[pa_p, :]^T b[pb_p, :]
 * with (pa, pb) = nbmaps columns (col_a, 1-col_a); a is [*, c_a], b is [*, c_b],
 * grad_kernel [K, c_a, c_b] is overwritten (zeroed then accumulated).
 * n_pairs = nboffs[K], the rulebook length, known to the host (it sizes the grid). */
int ts_conv_wgrad(const float *a_feat, int32_t c_a, const float *b_feat, int32_t c_b,
                  const int32_t *nbmaps, const int32_t *nboffs, int32_t kernel_volume,
                  int32_t col_a, int64_t n_pairs, float *grad_kernel,
                  ts_stream_t stream);
/* The same weight gradient with run-to-run identical bits (the reference's grad_kernel[k] = in_buf^T . out_grad_buf,
 * convolution_cuda.cu:259-263, is one GEMM per offset and therefore deterministic; ts_conv_wgrad combines the partial
 * tiles of its pair chunks with float atomics): every workgroup stores its partial tile into `ws`, a second launch adds
 * the tiles of each offset in ascending chunk order.  ws >= ts_conv_wgrad_workspace_bytes(...), 16-byte aligned.
 * ts_conv_wgrad_f16_det: IEEE-half operands, float32 accumulation and result (the AMP path). */
size_t ts_conv_wgrad_workspace_bytes(int64_t n_pairs, int32_t c_a, int32_t c_b, int32_t kernel_volume);
int ts_conv_wgrad_det(const float *a_feat, int32_t c_a, const float *b_feat, int32_t c_b, const int32_t *nbmaps,
                      const int32_t *nboffs, int32_t kernel_volume, int32_t col_a, int64_t n_pairs, float *grad_kernel,
                      void *ws, size_t ws_bytes, ts_stream_t stream);
int ts_conv_wgrad_f16_det(const void *a_feat, int32_t c_a, const void *b_feat, int32_t c_b, const int32_t *nbmaps,
                          const int32_t *nboffs, int32_t kernel_volume, int32_t col_a, int64_t n_pairs,
                          float *grad_kernel, void *ws, size_t ws_bytes, ts_stream_t stream);

/* Two-pass convolution on the rulebook itself (the default forward / dgrad path):
 *   pass 1  ts_conv_pair_gemm :  z[p, :] = feat[g_p, :] @ W_{k(p)}   for every pair p of nbmaps,
 *           g_p = nbmaps[p][gather_col], k(p) = the offset whose [nboffs[k], nboffs[k+1]) holds p;
 *           one dense MFMA GEMM over all pairs (row gather fused into the A-operand load);
 *   pass 2  ts_conv_gather_sum:  out[j, :] = sum_k z[pos[k, j], :]   (pos = pos_out or pos_in of
 *           ts_build_kmap; -1 entries skipped), k ascending - deterministic, no atomics.
 * Together they equal ts_conv_nbr / the reference's gather -> GEMM -> scatter
 * (convolution_cuda.cu:101-164) with the K per-offset launches collapsed into one.
 * weight_transposed as in ts_conv_nbr.  z has n_pairs rows of c_out floats (caller-allocated). */
int ts_conv_pair_gemm(const float *feat, int64_t n_rows, int32_t c_in, const float *kernel,
                      int32_t kernel_volume, int32_t weight_transposed, const int32_t *nbmaps,
                      const int32_t *nboffs, int64_t n_pairs, int32_t gather_col, float *z,
                      int32_t c_out, ts_stream_t stream);
int ts_conv_gather_sum(const float *z, int32_t c, const int32_t *pos, int32_t kernel_volume,
                       int64_t n_rows, int64_t n_pairs, float *out, ts_stream_t stream);

/* BatchNorm over the [n, c] feature matrix of a sparse tensor (the reference runs nn.BatchNorm1d / nn.SyncBatchNorm on
 * conv outputs, pcseg/.../minkunet/minkunet.py:23-29).  c must be a multiple of 4 (<= 1024), pointers 16-byte aligned. */
/* mean[c], invstd[c] from sums = double [2, c] (sum x, sum x^2; e.g. the all-reduced pack of ts_bn_sync_stats) over
 * `total` rows (*total_dev if non-NULL - the all-reduced count of SyncBN - else total_host) and the nn.BatchNorm
 * momentum update of the running buffers (unbiased variance; either may be NULL). */
int ts_bn_finalize(const double *sums, const double *total_dev, double total_host, int32_t c, float eps,
                   float momentum, float *running_mean, float *running_var, float *mean,
                   float *invstd, ts_stream_t stream);

/* Fused elementwise halves of a conv block (the reference chains BatchNorm -> [+ shortcut] -> ReLU as separate
 * modules, minkunet.py:42-51,117-129: one full pass over [n, c] each):
 *   ts_bn_act_forward          out = act((x - mean) * invstd * weight + bias [+ residual]),  act = relu if relu != 0;
 *                              mask (optional, uint8 [n * c / 4]) gets the 4 sign bits of every float4 of `out`
 *   ts_bn_act_backward         g = grad_out * (out > 0) through that mask (all ones if mask == NULL); with
 *                              sums = double [2, c] (sum g, sum g (x - mean); ts_bn_sync_backward_reduce, all-reduced):
 *                              grad_x = (g - sums[0]/N - (x - mean) invstd^2 sums[1]/N) invstd weight,
 *                              grad_residual = g (optional); N = *total_dev if non-NULL else total_host. */
int ts_bn_act_forward(const float *x, const float *residual, const float *mean, const float *invstd,
                      const float *weight, const float *bias, int64_t n, int32_t c, int32_t relu,
                      float *out, uint8_t *mask, ts_stream_t stream);
int ts_bn_act_backward(const float *grad_out, const uint8_t *mask, const float *x, const float *mean,
                       const float *invstd, const float *weight, const double *sums,
                       const double *total_dev, double total_host, int64_t n, int32_t c, float *grad_x,
                       float *grad_residual, ts_stream_t stream);

/* Single-process training BatchNorm (+ residual) (+ ReLU) in ONE call per direction - what nn.BatchNorm1d in
 * training mode followed by the residual add and ReLU of the MinkUNet blocks computes (R minkunet.py:42-51,
 * 110-129; torch/nn/functional.py batch_norm).  Reductions go through <= 512 float partial slices in `ws`
 * (ts_bn_train_workspace_bytes(c) bytes, 16-byte aligned; stream-ordered scratch) that a finish kernel adds in
 * double: no memset, no atomics, results independent of workgroup scheduling.
 *   forward : mean/invstd [C] (saved for backward), running_mean/var updated in place and
 *             *num_batches_tracked incremented (each may be NULL),
 *             out = act((x - mean) invstd weight + bias [+ residual]), mask = 4-bit ReLU sign per float4 (relu only)
 *   backward: grad_x, grad_residual (optional, = masked grad_out), grad_weight [C], grad_bias [C] (optional)
 * SyncBatchNorm keeps using the split entry points above (the all-reduce sits between reduction and apply). */
size_t ts_bn_train_workspace_bytes(int32_t c);
int ts_bn_act_train_forward(const float *x, const float *residual, const float *weight, const float *bias,
                            float *running_mean, float *running_var, int64_t *num_batches_tracked, int64_t n,
                            int32_t c, float eps, float momentum, int32_t relu, float *mean, float *invstd,
                            float *out, uint8_t *mask, void *ws, size_t ws_bytes, ts_stream_t stream);
int ts_bn_act_train_backward(const float *grad_out, const uint8_t *mask, const float *x, const float *mean,
                             const float *invstd, const float *weight, int64_t n, int32_t c, float *grad_x,
                             float *grad_residual, float *grad_weight, float *grad_bias, void *ws, size_t ws_bytes,
                             ts_stream_t stream);

/* SyncBatchNorm reductions (the all-reduce over ranks happens between these and the elementwise entry points):
 *   ts_bn_sync_stats            pack[0..C) = sum x, pack[C..2C) = sum x^2, pack[2C] = n     (double [2C + 1])
 *   ts_bn_sync_backward_reduce  sums[0..C) = sum g, sums[C..2C) = sum g (x - mean), g = grad_out masked by the ReLU
 *                               mask when given; grad_weight / grad_bias [C] = this rank's parameter gradients
 * ws as for ts_bn_act_train_*. */
int ts_bn_sync_stats(const float *x, int64_t n, int32_t c, double *pack, void *ws, size_t ws_bytes, ts_stream_t stream);
int ts_bn_sync_backward_reduce(const float *grad_out, const uint8_t *mask, const float *x, const float *mean,
                               const float *invstd, int64_t n, int32_t c, double *sums, float *grad_weight,
                               float *grad_bias, void *ws, size_t ws_bytes, ts_stream_t stream);

/* Trilinear devoxelisation backward over runs of points that share their 8-corner index tuple (the points of one
 * interpolation cell).  Same result as ts_devoxelize_backward up to float summation order; the corner
 * contributions of a run are accumulated in registers and added to grad_feat once per run.
 *   ts_devox_order             order[n] = permutation grouping equal tuples (radix sort on 8 * first present corner
 *                              voxel + corner number; n_vox < 2^28); ws >= ts_devox_order_workspace_bytes(n)
 *   ts_devoxelize_backward_runs  grad_feat[m, c] (zeroed here) += runs; `order` may be NULL (natural order);
 *                              c % 4 == 0, 16-byte aligned rows */
/*   ts_devox_csr               inverse of a trilinear map: offsets[n_vox + 1] / entries[8 n] int32, the slots
 *                              (point * 8 + corner) with a non-zero weight onto voxel v in ascending order;
 *                              ws >= ts_devox_csr_workspace_bytes(n)
 *   ts_devoxelize_backward_csr grad_feat[v] = sum over the slots of v of weight[slot] * grad_out[point]: every row written
 *                              once, no fill, no atomics, fixed summation order (c % 4 == 0, 16-byte aligned rows) */
/*   ts_devoxelize_forward_ld / ts_devoxelize_backward_runs_ld / ts_devoxelize_backward_csr_ld: the same with the point-side
 *                              matrix `ld` floats per row (>= c, multiple of 4): the three devoxelisations of a
 *                              MinkUNet pass write into / read from column blocks of one [N, 480] matrix (no torch.cat of
 *                              z1 | z2 | z3 before the class head, no contiguous copies of its gradient slices) */
int ts_devoxelize_forward_ld(const float *feat, const int32_t *idx, const float *weight, int64_t n, int32_t c, int64_t m,
                             float *out, int64_t out_ld, ts_stream_t stream);
int ts_devoxelize_backward_runs_ld(const float *grad_out, int64_t go_ld, const int32_t *idx, const float *weight,
                                   const int32_t *order, int64_t n, int32_t c, int64_t m, float *grad_feat,
                                   ts_stream_t stream);
int ts_devoxelize_backward_csr_ld(const float *grad_out, int64_t go_ld, const float *weight, const int32_t *offsets,
                                  const int32_t *entries, int64_t n, int32_t c, int64_t m, float *grad_feat,
                                  ts_stream_t stream);

/* Cell-reduced form of the same adjoint for coarse strides (stride 16: ~700 contributions per voxel).  No reference
 * counterpart beyond devoxelize_backward_kernel (backend/devoxelize/devoxelize_cuda.cu:37-57), whose per-point
 * atomicAdds it replaces.  Plan (coordinates only, built once per batch):
 *   order      = ts_devox_order(idx): points of the same interpolation cell next to each other
 *   flags      = ts_devox_segments(idx, order, n, max_len): 1 where a segment (equal 8-corner tuple, <= max_len points)
 *                starts; seg_start = positions of the ones, then n
 *   offsets / entries = ts_devox_csr on the segments' tuples [n_seg, 8] with weight 1 on the present corners
 * ts_devoxelize_backward_cells_ld: stage 1 sums the weighted gradient rows of every segment per corner into part
 * [n_seg * 8, c] (every gradient row is read once), stage 2 gathers those rows per voxel along the inverse map.  Fixed
 * summation order, no atomics. */
int ts_devox_segments(const int32_t *idx, const int32_t *order, int64_t n, int32_t max_len, int32_t *flags,
                      ts_stream_t stream);
int ts_devoxelize_backward_cells_ld(const float *grad_out, int64_t go_ld, const float *weight, const int32_t *order,
                                    const int32_t *seg_start, int64_t n_seg, const int32_t *offsets,
                                    const int32_t *entries, int64_t n, int32_t c, int64_t m, float *part,
                                    float *grad_feat, ts_stream_t stream);
size_t ts_devox_csr_workspace_bytes(int64_t n);
int ts_devox_csr(const int32_t *idx, const float *weight, int64_t n, int64_t n_vox, int32_t *offsets, int32_t *entries,
                 void *ws, size_t ws_bytes, ts_stream_t stream);
int ts_devoxelize_backward_csr(const float *grad_out, const float *weight, const int32_t *offsets, const int32_t *entries,
                               int64_t n, int32_t c, int64_t m, float *grad_feat, ts_stream_t stream);
size_t ts_devox_order_workspace_bytes(int64_t n);
int ts_devox_order(const int32_t *idx, int64_t n, int64_t n_vox, int32_t *order, void *ws, size_t ws_bytes,
                   ts_stream_t stream);
int ts_devoxelize_backward_runs(const float *grad_out, const int32_t *idx, const float *weight, const int32_t *order,
                                int64_t n, int32_t c, int64_t m, float *grad_feat, ts_stream_t stream);
/* Half-storage forms for the AMP path (the reference's devoxelize op runs under `custom_fwd(cast_inputs=torch.half)`,
 * nn/functional/devoxelize.py:54): `void *` rows are IEEE half, leading dimensions in elements, sums in float32 and ONE
 * rounding at the store - the bits of the float32 entry points between a `.float()` of the inputs and a `.half()` of the
 * result, without those two passes.  C % 4 == 0, rows 8-byte aligned; `part` stays float32. */
int ts_devoxelize_forward_f16_ld(const void *feat, const int32_t *idx, const float *weight, int64_t n, int32_t c, int64_t m,
                                 void *out, int64_t out_ld, ts_stream_t stream);
int ts_devoxelize_backward_csr_f16_ld(const void *grad_out, int64_t go_ld, const float *weight, const int32_t *offsets,
                                      const int32_t *entries, int64_t n, int32_t c, int64_t m, void *grad_feat,
                                      ts_stream_t stream);
int ts_devoxelize_backward_cells_f16_ld(const void *grad_out, int64_t go_ld, const float *weight, const int32_t *order,
                                        const int32_t *seg_start, int64_t n_seg, const int32_t *offsets,
                                        const int32_t *entries, int64_t n, int32_t c, int64_t m, float *part,
                                        void *grad_feat, ts_stream_t stream);

/* TIAF image -> point gather (R pcseg/model/segmentor/voxel/minkunet/unet2d.py:180-214) on the NCHW stack:
 *   out[n, c] = feat[first_frame(b_n) + row_n / H, c, (row_n % H) >> shift, col_n >> shift]
 * feat [T, C, H >> shift, W >> shift] f32 (the stacked camera frames of all samples), pix [n, 2] f32 = (row, col) in
 * the sample's tall (T_b * H, W) image (truncated like the reference's `.long()`), pbatch [n] sample index,
 * frame_end [n_batch] cumulative frame counts (`offset_img`), shift = 0 (full scale) or 2 (the 1/4-scale map,
 * `row // 4, col // 4`).
 *   ts_image_plan            once per batch and scale: the points in RASTER order of their pixels - perm [n] (point index of the
 *                            i-th point of that order; stable: equal pixels keep index order), paddr [n] (pixel address
 *                            (frame * hs + r) * ws + c of that point, -1 if it falls outside its sample's frames: *err is then set
 *                            to 1 - the reference raises), run [n] (points on that pixel if i is the first of them, else 0)
 *   ts_image_gather_forward  out [n, C] rows in the original point order from one map of that scale (hw = hs * ws); lanes run
 *                            along the points of the raster order, one plane at a time, rows leave through an LDS transpose
 *   ts_image_gather_backward the adjoint as a segmented sum over the same order - no atomics, run-to-run identical: grad_feat
 *                            (n_feat = T * C * hw elements) += per-pixel sums, touching only pixels that have points;
 *                            accumulate == 0 zero-fills grad_feat first, != 0 adds into the gradient the map already has */
size_t ts_image_plan_workspace_bytes(int64_t n_pts);
int ts_image_plan(const float *pix, const int32_t *pbatch, const int32_t *frame_end, int64_t n_pts, int32_t n_batch, int32_t T, int32_t H,
                  int32_t W, int32_t shift, int32_t *perm, int32_t *paddr, int32_t *run, int32_t *err, void *ws, size_t ws_bytes,
                  ts_stream_t stream);
int ts_image_gather_forward(const float *feat, int32_t C, int64_t hw, const int32_t *perm, const int32_t *paddr, int64_t n_pts, float *out,
                            ts_stream_t stream);
int ts_image_gather_backward(const float *grad_out, int32_t C, int64_t hw, const int32_t *perm, const int32_t *paddr, const int32_t *run,
                             int64_t n_pts, float *grad_feat, int64_t n_feat, int32_t accumulate, ts_stream_t stream);

/* The same gather on a CHANNELS-LAST stack - feat [T, hs, ws, C] in memory, the layout the reference indexes (unet2d.py:183-187:
 * `permute(0, 2, 3, 1)` then row indexing) and the one the fp16 2-D convolutions produce: a pixel is one contiguous row of
 * C * elem_bytes bytes at row index paddr (ts_image_plan's pixel address), moved in 16-byte pieces by the threads of one row.
 *   ts_image_gather_rows_forward   out [n, C] in the original point order; a pure move: elements of elem_bytes = 1 / 2 / 4 / 8 bytes
 *   ts_image_gather_rows_backward  grad_feat [T, hs, ws, C] (n_feat elements) += per-pixel sums of grad_out [n, C]; half != 0: IEEE
 *                                  half rows and map (fp32 accumulation over the points of a pixel, one rounding into the map);
 *                                  every pixel row has one owner thread group: no atomics, run-to-run identical;
 *                                  accumulate as in ts_image_gather_backward */
int ts_image_gather_rows_forward(const void *feat, int32_t C, int32_t elem_bytes, const int32_t *perm, const int32_t *paddr, int64_t n_pts,
                                 void *out, ts_stream_t stream);
int ts_image_gather_rows_backward(const void *grad_out, int32_t C, int32_t half, const int32_t *perm, const int32_t *paddr,
                                  const int32_t *run, int64_t n_pts, void *grad_feat, int64_t n_feat, int32_t accumulate,
                                  ts_stream_t stream);

/* AvgPool2d(kernel 3, stride 2, padding 1, count_include_pad) of UNet2D's encoder blocks (R pcseg/model/segmentor/voxel/minkunet/
 * unet2d.py:58-62: `nn.AvgPool2d(kernel_size=kernel_size, stride=2, padding=1)`, kernel_size (3, 3)) on a channels-last stack:
 * x [T, H, W, C] -> y [T, Ho, Wo, C], Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1; fp32 accumulation, sum / 9.  half != 0: IEEE half.
 * The backward writes every element of grad_x (a gather over the <= 4 windows that hold a pixel: no atomics). */
int ts_avgpool3s2_rows_forward(const void *x, int32_t T, int32_t H, int32_t W, int32_t C, int32_t half, void *y, ts_stream_t stream);
int ts_avgpool3s2_rows_backward(const void *grad_y, int32_t T, int32_t H, int32_t W, int32_t C, int32_t half, void *grad_x,
                                ts_stream_t stream);

/* LeakyReLU -> BatchNorm2d (training) of UNet2D's blocks (unet2d.py:24-30,71,108: `bn(act(conv(x)))`, `nn.LeakyReLU()` slope 0.01) on
 * a channels-last stack seen as rows [N = T*H*W, C]: partial sums -> finish (double) -> elementwise pass, per direction.
 *   forward   out = (leaky(x) - mean) * invstd * weight + bias (+ residual [N, C], may be NULL: the block's `skip + y`, unet2d.py:31,64);
 *             mean / invstd [C] (statistics of the ACTIVATED values) kept for the backward; running_mean / running_var (momentum,
 *             unbiased variance) and num_batches_tracked updated when not NULL
 *   backward  grad_x with respect to the LeakyReLU's input, grad_weight / grad_bias [C] (may be NULL)
 * half != 0: IEEE half rows (C % 8 == 0), else float (C % 4 == 0); C <= 1024; ws >= ts_bn_train_workspace_bytes(c). */
int ts_leaky_bn_train_forward(const void *x, const float *weight, const float *bias, float *running_mean, float *running_var,
                              int64_t *num_batches_tracked, int64_t n, int32_t c, float eps, float momentum, float slope, int32_t half,
                              float *mean, float *invstd, const void *residual, void *out, void *ws, size_t ws_bytes, ts_stream_t stream);
int ts_leaky_bn_train_backward(const void *grad_out, const void *x, const float *mean, const float *invstd, const float *weight,
                               int64_t n, int32_t c, float slope, int32_t half, void *grad_x, float *grad_weight, float *grad_bias,
                               void *ws, size_t ws_bytes, ts_stream_t stream);

/* Dense 3 x 3 convolution, 32 -> 32 channels, stride 1, padding = dilation (1 or 2), on channels-last IEEE-half rows - the
 * full-resolution layers of UNet2D (unet2d.py:10-31,41-60: `nn.Conv2d(32, 32, (3, 3), padding=1)` and `dilation=2, padding=2`).
 *   ts_conv3x3c32_pack   the nn.Conv2d weight [co][ci][ky][kx] (half, any strides) -> the MFMA operand form the kernel keeps in
 *                        registers (ts_conv3x3c32_packed_bytes() bytes); mode 0: forward, mode 1: data gradient (taps mirrored,
 *                        channels swapped)
 *   ts_conv3x3c32_rows   y [T, H, W, 32] = conv(x [T, H, W, 32]) + bias [32] (float, may be NULL); with a mode-1 pack and
 *                        x = grad_y it returns grad_x.  fp32 accumulation, one rounding to half. */
size_t ts_conv3x3c32_packed_bytes(void);
int ts_conv3x3c32_pack(const void *weight, int64_t s_co, int64_t s_ci, int64_t s_ky, int64_t s_kx, int32_t mode, void *packed,
                       ts_stream_t stream);
int ts_conv3x3c32_rows(const void *x, const void *packed, const float *bias, int32_t T, int32_t H, int32_t W, int32_t dilation, void *y,
                       ts_stream_t stream);
/*   ts_conv3x3c32_wgrad  grad_weight [co][ci][ky][kx] (half, the element strides of the weight it belongs to) = sum over pixels of
 *                        x[t, y + (ky - 1) D, x + (kx - 1) D, ci] grad_y[t, y, x, co]: partial [9][32][32] sums per workgroup (fp32, LDS
 *                        transposing reads feed the MFMAs), then their sum in index order - no atomics, run-to-run identical.
 *                        grad_bias [32] (float, may be NULL): the column sums of grad_y, in the same pass.
 *                        ws >= ts_conv3x3c32_wgrad_workspace_bytes(). */
size_t ts_conv3x3c32_wgrad_workspace_bytes(void);
int ts_conv3x3c32_wgrad(const void *x, const void *grad_y, int32_t T, int32_t H, int32_t W, int32_t dilation, void *grad_weight,
                        int64_t s_co, int64_t s_ci, int64_t s_ky, int64_t s_kx, float *grad_bias, void *ws, size_t ws_bytes,
                        ts_stream_t stream);

/* The 1 x 1, 32 -> 32 channel layers in front of UNet2D's blocks with the LeakyReLU behind them (unet2d.py:10-12,41-43:
 * `act1(conv1(x))`) on channels-last IEEE-half rows: y [n_pixels, 32] = LeakyReLU_slope(x W^T + bias) (leaky = 0: no activation;
 * with a mode-1 pack that is the data gradient).  ts_conv1x1c32_wgrad: weight [co][ci] and bias gradient from x and the gradient g at
 * the layer's output - the 3 x 3 weight gradient's pass, centre tap (same workspace). */
size_t ts_conv1x1c32_packed_bytes(void);
int ts_conv1x1c32_pack(const void *weight, int64_t s_co, int64_t s_ci, int32_t mode, void *packed, ts_stream_t stream);
int ts_conv1x1c32_rows(const void *x, const void *packed, const float *bias, int64_t n_pixels, int32_t leaky, float slope, void *y,
                       ts_stream_t stream);
int ts_conv1x1c32_wgrad(const void *x, const void *g, int32_t T, int32_t H, int32_t W, void *grad_weight, int64_t s_co, int64_t s_ci,
                        float *grad_bias, void *ws, size_t ws_bytes, ts_stream_t stream);

/* The same convolution (stride 1, padding 1, no dilation) with the channel counts opened up - UpBlock.conv1 of the decoder
 * (unet2d.py:81-115: 96 -> 96 channels at 1/2 scale, 56 -> 96 at full scale): input channels a multiple of 8 up to 96, output
 * channels a multiple of 8.  A persistent workgroup owns one 32-channel output block, its 9 x C_in/16 weight fragments in registers.
 *   ts_conv3x3_rows_packed_bytes(k, m)  bytes of the packed operand for k reduction channels (the channels of the call's input)
 *                                       and m result channels; 0 = the kernel does not take the layer
 *   ts_conv3x3_rows_pack                weight [c_out][c_in][3][3] (half, element strides) -> packed; mode 0 forward
 *                                       (k, m) = (c_in, c_out), mode 1 data gradient (k, m) = (c_out, c_in)
 *   ts_conv3x3_rows                     y [T, H, W, y_channels] = conv(x [T, H, W, x_channels]) + bias (float, may be NULL) */
size_t ts_conv3x3_rows_packed_bytes(int32_t k_channels, int32_t m_channels);
int ts_conv3x3_rows_pack(const void *weight, int32_t c_out, int32_t c_in, int64_t s_co, int64_t s_ci, int64_t s_ky, int64_t s_kx,
                         int32_t mode, void *packed, ts_stream_t stream);
int ts_conv3x3_rows(const void *x, int32_t x_channels, const void *packed, const float *bias, int32_t T, int32_t H, int32_t W, void *y,
                    int32_t y_channels, ts_stream_t stream);
/*   ts_conv3x3_wgrad                    grad_weight [c_out][c_in][3][3] (half, the weight's element strides) and grad_bias [c_out]
 *                                       (float, may be NULL) of the same layers (channel counts multiples of 8 up to 96) from x and
 *                                       grad_y: the 9 x c_in x c_out fp32 sums in the registers of persistent workgroups (96 x 96 in two
 *                                       passes over the input channels), partials added in a fixed order - run-to-run identical.
 *                                       ws >= ts_conv3x3_wgrad_workspace_bytes(c_in, c_out) (0: layer not taken). */
size_t ts_conv3x3_wgrad_workspace_bytes(int32_t c_in, int32_t c_out);
int ts_conv3x3_wgrad(const void *x, int32_t c_in, const void *grad_y, int32_t c_out, int32_t T, int32_t H, int32_t W, void *grad_weight,
                     int64_t s_co, int64_t s_ci, int64_t s_ky, int64_t s_kx, float *grad_bias, void *ws, size_t ws_bytes, ts_stream_t stream);

/* UpBlock's entry on channels-last rows (unet2d.py:98-108: PixelShuffle(2), Dropout2d, torch.cat((upA, skip), dim=1), Dropout2d) in one
 * pass:  cat [T, 2h, 2w, C/4 + Cs] = concat(PixelShuffle(2)(x [T, h, w, C]), skip [T, 2h, 2w, Cs]) * scale [T, C/4 + Cs]
 * (float, may be NULL: the two dropout masks folded into one factor per frame and channel).  half != 0: IEEE half (C a multiple of
 * 32, Cs of 8), else float (16 / 4).  _backward is the adjoint: grad_x and grad_skip from grad_cat with the same scale. */
int ts_shuffle_cat_rows_forward(const void *x, const void *skip, const float *scale, int32_t T, int32_t h, int32_t w, int32_t C, int32_t Cs,
                                int32_t half, void *cat, ts_stream_t stream);
int ts_shuffle_cat_rows_backward(const void *grad_cat, const float *scale, int32_t T, int32_t h, int32_t w, int32_t C, int32_t Cs,
                                 int32_t half, void *grad_x, void *grad_skip, ts_stream_t stream);

/* ---- fp16 storage / fp32 accumulation (the reference trains under AMP: conv.py:19 `custom_fwd(cast_inputs=half)`).
 * `void *` operands are IEEE half arrays.  Channel counts must be multiples of 32, K <= 63.
 *   ts_cast_weights_f16      w f32 [K, Ci, Co] -> w16 [K, Ci, Co] and / or w16t [K, Co, Ci] (either may be NULL)
 *   ts_conv_pair_gemm_f16    z[p, :] = feat[nbmaps[p][gather_col], :] @ W_k(p); `w_rows` [K, c_out, c_red] holds, per
 *                            offset, one row per OUTPUT column contiguous in the reduction index: w16t for the forward
 *                            pass (c_red = Ci), w16 for the input gradient (c_red = Co, c_out = Ci)
 *   ts_conv_gather_sum_f16   out[j, :] = sum_k z[pos[k, j], :]  (fp32 accumulation, k ascending) */
int ts_cast_weights_f16(const float *w, int32_t K, int32_t c_in, int32_t c_out, void *w16, void *w16t,
                        ts_stream_t stream);
int ts_conv_pair_gemm_f16(const void *feat, int64_t n_rows, int32_t c_red, const void *w_rows, int32_t K,
                          const int32_t *nbmaps, const int32_t *nboffs, int64_t n_pairs, int32_t gather_col, void *z,
                          int32_t c_out, ts_stream_t stream);
/*   ts_conv_pair_gemm_f16_nat  ts_conv_pair_gemm_f16 with the weight in its natural layout w [K, c_red, c_out] (the forward
 *                              pass reads the half copy of kernel [K, C_in, C_out] in place; fragments come from the
 *                              transposing LDS load, no transposed copy of the weight exists) */
int ts_conv_pair_gemm_f16_nat(const void *feat, int64_t n_rows, int32_t c_red, const void *w, int32_t K,
                              const int32_t *nbmaps, const int32_t *nboffs, int64_t n_pairs, int32_t gather_col, void *z,
                              int32_t c_out, ts_stream_t stream);

int ts_conv_gather_sum_f16(const void *z, int32_t c, const int32_t *pos, int32_t K, int64_t n_rows, int64_t n_pairs,
                           void *out, ts_stream_t stream);
/*   ts_conv_wgrad_f16        grad_kernel[k] (fp32 [K, c_a, c_b], zeroed here) = sum_pairs a[pa]^T b[pb], half rows in,
 *                            fp32 accumulation (MFMA fragments via ds_read_b64_tr_b16), chunk partials by float atomics */
int ts_conv_wgrad_f16(const void *a_feat, int32_t c_a, const void *b_feat, int32_t c_b, const int32_t *nbmaps,
                      const int32_t *nboffs, int32_t K, int32_t col_a, int64_t n_pairs, float *grad_kernel,
                      ts_stream_t stream);

/* Half-storage forms of ts_bn_act_train_*: x / residual / out / grad_* are IEEE half [n, c] (c % 8 == 0), the mask is
 * one byte per 8 elements; statistics, affine parameters, their gradients and all arithmetic are fp32. */
int ts_bn_act_train_forward_f16(const void *x, const void *residual, const float *weight, const float *bias,
                                float *running_mean, float *running_var, int64_t *num_batches_tracked, int64_t n,
                                int32_t c, float eps, float momentum, int32_t relu, float *mean, float *invstd,
                                void *out, uint8_t *mask, void *ws, size_t ws_bytes, ts_stream_t stream);
int ts_bn_act_train_backward_f16(const void *grad_out, const uint8_t *mask, const void *x, const float *mean,
                                 const float *invstd, const float *weight, int64_t n, int32_t c, void *grad_x,
                                 void *grad_residual, float *grad_weight, float *grad_bias, void *ws, size_t ws_bytes,
                                 ts_stream_t stream);

/* Flat-buffer optimizer step of the training loop (R/train.py:399-417: GradScaler.unscale_ -> clip_grad_norm_ ->
 * SGD step -> GradScaler.update), decided on the device - no host read.  `state` = float[8] on the device:
 * [0] loss scale, [1] growth tracker, [2] gradient multiplier (clip coefficient / scale), [3] skip flag, [4] total norm.
 *   ts_sgd_grad_stats  *sumsq += sum g^2 (double), *nonfinite |= any non-finite        (call once per bucket)
 *   ts_sgd_decide      norm of the unscaled gradients, clip coefficient min(1, max_norm / (norm + 1e-6)), skip flag,
 *                      loss-scale update (amp != 0); clears sumsq / nonfinite for the next step
 *   ts_sgd_apply       d = g * state[2] + wd * p; m = first_step ? d : momentum * m + d; p -= lr * m   (skipped if flagged) */
int ts_sgd_grad_stats(const float *grad, int64_t n, double *sumsq, int32_t *nonfinite, ts_stream_t stream);
int ts_sgd_decide(double *sumsq, int32_t *nonfinite, float *state, float max_norm, float growth, float backoff,
                  int32_t growth_interval, int32_t amp, ts_stream_t stream);
int ts_sgd_apply(float *param, const float *grad, float *momentum_buf, int64_t n, const float *state, float lr,
                 float momentum, float weight_decay, int32_t first_step, ts_stream_t stream);

/* Half-storage forms of the SyncBatchNorm halves (ts_bn_sync_stats, ts_bn_act_forward, ts_bn_sync_backward_reduce,
 * ts_bn_act_backward): activations / gradients IEEE half [n, c] (c % 8 == 0), mask one byte per 8 elements, everything
 * else as in the fp32 forms.  ts_bn_act_backward_f16 needs 2 c floats of 16-byte aligned scratch. */
int ts_bn_sync_stats_f16(const void *x, int64_t n, int32_t c, double *pack, void *ws, size_t ws_bytes, ts_stream_t stream);
int ts_bn_act_forward_f16(const void *x, const void *residual, const float *mean, const float *invstd,
                          const float *weight, const float *bias, int64_t n, int32_t c, int32_t relu, void *out,
                          uint8_t *mask, ts_stream_t stream);
int ts_bn_sync_backward_reduce_f16(const void *grad_out, const uint8_t *mask, const void *x, const float *mean,
                                   const float *invstd, int64_t n, int32_t c, double *sums, float *grad_weight,
                                   float *grad_bias, void *ws, size_t ws_bytes, ts_stream_t stream);
int ts_bn_act_backward_f16(const void *grad_out, const uint8_t *mask, const void *x, const float *mean,
                           const float *invstd, const float *weight, const double *sums, const double *total_dev,
                           double total_host, int64_t n, int32_t c, void *grad_x, void *grad_residual, void *ws,
                           size_t ws_bytes, ts_stream_t stream);

/* SyncBatchNorm with the statistics all-reduce issued by the library on the CALLER'S stream (csrc/rccl.hip): RCCL is
 * bound at run time with dlopen (ts_rccl_load: path of the librccl.so the host framework already uses), the library
 * owns one communicator per SyncBatchNorm group (ts_rccl_unique_id on one rank, the 128-byte id distributed by the
 * host, ts_rccl_comm_init collectively), and one call per direction runs local sliced sums -> ncclAllReduce of
 * [2C + 1] / [2C] doubles -> elementwise pass.  Replaces nn.SyncBatchNorm (minkunet.py:23-25; torch/nn/modules/
 * _functions.py SyncBatchNorm) whose collectives go through the process group's own stream.
 * pack [2C + 1] / sums [2C]: double scratch; pack[2C] afterwards holds the global row count (= total_dev of the
 * backward call).  half = 1: IEEE-half activations / gradients.  ws as ts_bn_train_workspace_bytes(c). */
int ts_rccl_load(const char *librccl_path);
int ts_rccl_unique_id(void *id128);
int ts_rccl_comm_init(const void *id128, int32_t nranks, int32_t rank, void **comm);
int ts_rccl_comm_destroy(void *comm);
int ts_rccl_allreduce_f64(void *comm, double *buf, int64_t count, ts_stream_t stream);
int ts_bn_sync_forward(void *comm, const void *x, const void *residual, const float *weight, const float *bias,
                       float *running_mean, float *running_var, int64_t *num_batches_tracked, int64_t n, int32_t c,
                       float eps, float momentum,
                       int32_t relu, int32_t half, double *pack, float *mean, float *invstd, void *out, uint8_t *mask,
                       void *ws, size_t ws_bytes, ts_stream_t stream);
int ts_bn_sync_backward(void *comm, const void *grad_out, const uint8_t *mask, const void *x, const float *mean,
                        const float *invstd, const float *weight, const double *total_dev, int64_t n, int32_t c,
                        int32_t half, double *sums, void *grad_x, void *grad_residual, float *grad_weight,
                        float *grad_bias, void *ws, size_t ws_bytes, ts_stream_t stream);

/* One call per direction for conv -> BatchNorm(train) [+ residual] [-> ReLU], the unit of the MinkUNet family
 * (minkunet.py:31-129: BasicConvolutionBlock, BasicDeconvolutionBlock, the two halves of ResidualBlock).  Chains the
 * launches of ts_conv_pair_gemm / ts_conv_gather_sum (or the _f16 forms, half = 1) and ts_bn_act_train_* (comm == NULL)
 * or ts_bn_sync_* (comm = a ts_rccl_comm_init communicator) on `stream`; Z, the gradient w.r.t. the convolution output
 * and the transposed half weight live in `ws` (ts_conv_block_workspace_bytes).  Arguments: csrc/block.hip.
 * SyncBatchNorm with the all-reduce run by the CALLER (torch.distributed on its own communicator): comm = (void *)1 makes
 * the call stop after the local sums (forward: convolution + `pack`; backward: `sums`), the caller all-reduces that buffer
 * over its ranks and calls again with the same arguments and comm = (void *)2, which runs what is left (forward: statistics +
 * elementwise pass; backward: elementwise pass + weight gradient + input gradient).  ts_bn_sync_forward / _backward take the
 * same two values.
 *
 * `opts` (may be NULL) names everything a block call may use beyond the rulebook - explicitly, per call (round 4: it replaces
 * the thread-local one-shot hints ts_conv_class_hint / ts_conv_block_addend_hint and, for the block calls, ts_conv_planes_hint;
 * nothing a call leaves behind can steer another call):
 *   fwd_plan / dgrad_plan  class-sorted plans of THIS block's kernel map (ts_conv_class_plan) for the forward product and the
 *                          input gradient, or NULL for pair GEMM + pass 2.  A plan is used only if it fits the call: K, its row
 *                          count = the rows the product writes, channel counts the class kernels take, and map_id == the
 *                          call's nboffs pointer (the identity of the kernel map it was built from) - otherwise the two passes.
 *                          Plans with `rows` (direct, one group: the 2x2x2 strided / transposed maps) store the result rows
 *                          themselves: no Z, no pass 2.  Submanifold 3x3x3 maps use ONE plan for both directions (mirror = 1:
 *                          the input gradient takes the slice of offset K-1-k).
 *   planes                 fp32: the pre-split bf16 planes of `kernel` (ts_conv_split_planes), kept in step by the caller
 *   w16_current            half: the w16 buffer already holds the cast of `kernel` (the call casts nothing)
 *   addend                 backward only: [n_dgrad_rows, c_in] in grad_feat's storage type, 16-byte aligned, added in the
 *                          grad_feat store (the gradient reaching the block's input along a residual connection,
 *                          minkunet.py:117-129) - one rounding like a separate sum of the two */
typedef struct TsClassPlan {
  const int32_t *src, *tile_info, *n_tiles;
  const int32_t *pos;      /* [groups][n]: pass-2 plans */
  const int32_t *rows;     /* [m_pad]: direct plans (groups == 1) */
  int64_t n, m_pad;
  int64_t z_rows;          /* host copy of 128 * listed tiles (profile records only), or 0 */
  int32_t K, groups, mirror;
  const void *map_id;      /* nboffs pointer of the kernel map the plan was built from */
} TsClassPlan;
typedef struct TsConvBlockOpts {
  const TsClassPlan *fwd_plan, *dgrad_plan;
  const void *planes;
  int32_t w16_current;
  const void *addend;
  /* backward only: the weight gradient on a second stream.  wgrad_stream != NULL: the gradient w.r.t. the convolution output is
   * written into wgrad_ws (>= ts_conv_block_wgrad_ws_bytes) instead of the call's own workspace, and the weight gradient - partial
   * tiles + ordered sum, deterministic as before - is enqueued on wgrad_stream behind an event of the caller's stream; the input
   * gradient follows on the caller's stream without waiting for it.  wgrad_slot (0 .. 7) names the ring slot wgrad_ws belongs to:
   * a later call with the same slot first waits (on the caller's stream) for the slot's previous weight gradient.  grad_kernel is
   * complete on wgrad_stream: the caller joins the streams (ts_stream_join) before it reads the weight gradients. */
  ts_stream_t wgrad_stream;
  void *wgrad_ws;
  size_t wgrad_ws_bytes;
  int32_t wgrad_slot;
  /* wgrad_deferred != 0: the call only marks the point on its stream where the output gradient exists and leaves the weight
   * gradient to a ts_conv_block_wgrad_side call with the same arguments - from any host thread, so that its launches need not
   * come from the thread that issues the step */
  int32_t wgrad_deferred;
  /* natural != 0: a 1x1x1 convolution on the identity rulebook (K = 1, pair p = (p, p), n_pairs = n_out: conv.py:135-140's
   * `feats.matmul(weight)` as a block) - the pair GEMM's rows ARE the result rows: the product is written straight into the
   * convolution output / the input gradient, no Z, no pass 2, no position table read (pos may be any valid pointer); the shortcut
   * branch of a residual block (minkunet.py:105-111) as one call per direction.  Not combined with class plans or an addend. */
  int32_t natural;
} TsConvBlockOpts;
size_t ts_conv_block_workspace_bytes(int64_t n_pairs, int64_t n_rows_max, int32_t c_in, int32_t c_out, int32_t K,
                                     int32_t half);
size_t ts_conv_block_wgrad_ws_bytes(int64_t n_pairs, int64_t n_out, int32_t c_in, int32_t c_out, int32_t K, int32_t half);
/* `waiter` waits for everything enqueued on `other` so far (one event; no host synchronisation) */
int ts_stream_join(ts_stream_t waiter, ts_stream_t other);
int ts_conv_block_wgrad_side(const void *feat, int64_t n_feat_rows, int32_t c_in, int32_t K, const int32_t *nbmaps,
                             const int32_t *nboffs, int64_t n_pairs, int32_t wgrad_col_a, int64_t n_out, int32_t c_out, int32_t half,
                             float *grad_kernel, int32_t chunk_order, void *wgrad_ws, size_t wgrad_ws_bytes, int32_t slot,
                             ts_stream_t side_stream);
int ts_set_device(int32_t device);      /* hipSetDevice for a host thread the caller created */
/* The evaluation tail of the segmentors for a whole batch (R/pcseg/model/segmentor/voxel/minkunet/minkunet.py:435-455, minkunet_ms.py:
 * 433-458: per scene `out[scene mask][inverse_map of the scene]`, arg-max) without the per-scene loop of boolean masks:
 *   ts_scene_counts  counts [3][n_scenes] int64 = rows per scene of the voxel / point / label batch-index arrays (int32, `stride`
 *                    elements apart: the last column of an [N, 4] coordinate matrix is stride 4); *flags: bit 0 an index outside
 *                    0 .. n_scenes - 1, bit 1 an array that is not grouped by scene in ascending order (the caller then sorts).
 *                    counts and flags are zero-filled by the call.  n_scenes <= 64.
 *   ts_unvoxelise    for arrays grouped by scene: mapped [n_pts, C] (same type as logits; may be NULL) = logits[start of the point's
 *                    scene + inv[p], :], pred [n_pts] int64 (may be NULL) = index of the row's first maximum; *flags |= 4 where inv[p]
 *                    lies outside its scene (the reference's indexing raises).  counts = ts_scene_counts' first n_scenes entries. */
int ts_scene_counts(const int32_t *b_vox, int64_t stride_vox, int64_t n_vox, const int32_t *b_pts, int64_t stride_pts, int64_t n_pts,
                    const int32_t *b_lab, int64_t stride_lab, int64_t n_lab, int32_t n_scenes, int64_t *counts, int32_t *flags,
                    ts_stream_t stream);
int ts_unvoxelise(const void *logits, int32_t half, int32_t C, const int64_t *counts, int32_t n_scenes, const int32_t *b_pts,
                  int64_t stride_pts, const int64_t *inv, int64_t n_pts, void *mapped, int64_t *pred, int32_t *flags, ts_stream_t stream);
/* Column concatenation / slicing of row-major feature matrices - torchsparse.cat (TS/torchsparse/operators.py:10-17: torch.cat of
 * the feature matrices along dim 1, the skip connections of the decoder, R/pcseg/model/segmentor/voxel/minkunet/minkunet.py:403-416)
 * and its gradient, for callers that chain block calls without tensors (the stage programs of csrc/fastpath/stage_program.h).
 * Widths and pitches in BYTES (any element size); 16-byte vector copies when every pointer, pitch and width allows, else 4 / 2 / 1.
 *   ts_cat_cols   dst[r] = a[r] | b[r]   (dst pitch = a_bytes + b_bytes), one launch
 *   ts_copy_cols  dst[r, 0:width] = src[r, offset : offset + width]   (src rows src_pitch apart, dst rows dst_pitch apart) */
int ts_cat_cols(const void *a, int64_t a_bytes, const void *b, int64_t b_bytes, int64_t rows, void *dst, ts_stream_t stream);
int ts_copy_cols(const void *src, int64_t src_pitch, int64_t offset, int64_t width, int64_t rows, void *dst, int64_t dst_pitch,
                 ts_stream_t stream);
int ts_conv_block_forward(const void *feat, int64_t n_feat_rows, int32_t c_in, const float *kernel, int32_t K,
                          const int32_t *nbmaps, const int32_t *nboffs, int64_t n_pairs, int32_t gather_col,
                          const int32_t *pos, int64_t n_out, int32_t c_out, const void *residual, const float *bn_weight,
                          const float *bn_bias, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                          float eps, float momentum, int32_t relu, int32_t half, void *comm, double *pack, void *conv_out,
                          float *mean, float *invstd, void *out, uint8_t *mask, void *w16, const TsConvBlockOpts *opts,
                          void *ws, size_t ws_bytes, ts_stream_t stream);
/* evaluation form (module in eval mode, no graph): out = act((conv(feat) - mean) * invstd * bn_weight + bn_bias [+ residual]) with the
 * caller's mean / invstd (running_mean, 1 / sqrt(running_var + eps)); the same convolution as the training forward (opts->fwd_plan,
 * planes, w16_current), then one elementwise pass - no statistics, nothing kept for a backward pass */
int ts_conv_block_eval(const void *feat, int64_t n_feat_rows, int32_t c_in, const float *kernel, int32_t K,
                       const int32_t *nbmaps, const int32_t *nboffs, int64_t n_pairs, int32_t gather_col, const int32_t *pos,
                       int64_t n_out, int32_t c_out, const void *residual, const float *bn_weight, const float *bn_bias,
                       const float *mean, const float *invstd, int32_t relu, int32_t half, void *out, void *w16,
                       const TsConvBlockOpts *opts, void *ws, size_t ws_bytes, ts_stream_t stream);
int ts_conv_block_backward(const void *grad_out, const uint8_t *mask, const void *conv_out, const float *mean,
                           const float *invstd, const float *bn_weight, const double *total_dev, void *comm, double *sums,
                           int64_t n_out, int32_t c_out, int32_t half, const void *feat, int64_t n_feat_rows, int32_t c_in,
                           const void *weights, int32_t K, const int32_t *nbmaps, const int32_t *nboffs, int64_t n_pairs,
                           int32_t dgrad_gather_col, const int32_t *pos_dgrad, int64_t n_dgrad_rows, int32_t wgrad_col_a,
                           void *grad_feat, void *grad_residual, float *grad_kernel, float *grad_bn_weight,
                           float *grad_bn_bias, const TsConvBlockOpts *opts, void *ws, size_t ws_bytes, ts_stream_t stream);

/* Per-launch timing inside ts_conv_block_*: while enabled, every pair-GEMM / gather-sum / weight-gradient launch of the
 * block calls is bracketed by HIP events on the caller's stream.  ts_prof_collect waits for them and writes one record
 * of 9 doubles per launch: kind (0 pair GEMM, 1 gather-sum, 2 weight gradient), milliseconds, pairs, c_red, c_out, K,
 * rows, bytes per element, weight-transposed flag; returns the number of records (-1: HIP error).  ts_prof_reserve
 * creates events ahead of time (two per bracketed launch) so that none is created inside a timed region. */
void ts_prof_enable(int32_t on);
int ts_prof_reserve(int64_t n_events);
int64_t ts_prof_collect(double *records, int64_t capacity);
/* microseconds an event pair measures with nothing between its two records (median of `reps` pairs back to back on
 * `stream`): the bracket's own share of every per-launch figure above */
int ts_prof_empty_bracket_us(int32_t reps, ts_stream_t stream, double *out_us);

/* Debug / cross-check implementation selector: 0 = MFMA kernels (default; full-tile fp32 GEMMs run on the bf16 matrix
 * pipe through the exact three-way operand split of csrc/conv_pairs_s.hip), 5 = the same with v_mfma_f32_16x16x4_f32,
 * 1 = scalar reference kernels (one thread per output element, atomics),
 * 2 = MFMA kernels with the guarded generic staging code even where the unguarded full-tile variants apply,
 * 3 / 4 = full-tile pair GEMM with one workgroup per tile everywhere / persistent workgroups everywhere
 *         (default: persistent for the forward weight layout only),
 * 11 / 13 = with pre-split weight planes at hand (ts_conv_planes_hint) take the direct-rows pair GEMM on 96-column tiles
 *         as well / never (default: on 128-column tiles). */
void ts_set_conv_impl(int32_t impl);

/* Lovasz-softmax with classes = 'present' around ONE sort (R/tools/utils/common/lovasz_losses.py:158-227: lovasz_softmax_flat,
 * lovasz_grad, flatten_probas' ignore handling), forward value and gradient in one pass.
 *   ts_lovasz_errors  errors[c, p] = |(labels[p] == c) - probas[p, c]| where labels[p] != ignore, else 0 (class-major [C, P])
 *   caller            errors_sorted, perm = sort(errors, dim 1, descending)            (perm int64, as torch.sort returns it)
 *   ts_lovasz_grad    loss[0] = mean over the classes present of dot(errors_sorted_c, lovasz_grad(fg_sorted_c));
 *                     grad_probas[p, c] = d loss / d probas[p, c] (every element written; rows with the ignored label 0), stored
 *                     [P, C], or [C, P] when class_major != 0 (what ts_ce_lovasz_backward reads: the scatter through perm then
 *                     stays inside one class's 4 P bytes at a time)
 * probas [P, C] float32 row-major, labels [P] int64 (`ignore` = a value no label takes when nothing is ignored),
 * C <= 64.  ws >= ts_lovasz_workspace_bytes(P, C).  Rows with the ignored label keep their place with zero error and
 * zero foreground (same value as dropping them, no host read of their number). */
int ts_lovasz_errors(const float *probas, const int64_t *labels, int64_t ignore, int64_t n_points, int32_t n_classes,
                     float *errors, ts_stream_t stream);
size_t ts_lovasz_workspace_bytes(int64_t n_points, int32_t n_classes);
int ts_lovasz_grad(const float *errors_sorted, const int64_t *perm, const int64_t *labels, int64_t ignore, int64_t n_points,
                   int32_t n_classes, float *loss, float *grad_probas, int32_t class_major, void *ws, size_t ws_bytes,
                   ts_stream_t stream);

/* The segmentors' training loss, CE + Lovasz (R/pcseg/loss/__init__.py:40-44, 118-133: nn.CrossEntropyLoss(ignore_index,
 * label_smoothing) + lovasz_softmax(softmax(logits), ignore)), around the one sort:
 *   ts_softmax_ce_forward  one pass over logits [P, C] (C <= 32): probas [P, C], the Lovasz error matrix [C, P] (what
 *                          ts_lovasz_errors writes) and partials [ceil(P / 256), 3] doubles (sum -logp[y], sum -sum_c logp_c, rows)
 *   ts_ce_lovasz_finish    out4 = { w_ce ce + w_lov lovasz, ce, lovasz, rows that count }; ce = (1 - eps) nll / n + eps smooth / (n C)
 *                          (torch's label-smoothed mean over the rows whose label is not ignored); lovasz = ts_lovasz_grad's
 *                          loss scalar (device pointer) or NULL
 *   ts_ce_lovasz_backward  grad_logits [P, C] from probas, labels, grad_probas (= ts_lovasz_grad's in CLASS-MAJOR form [C, P], or
 *                          NULL) and the upstream gradient scalar (device pointer) */
int ts_softmax_ce_forward(const float *logits, const int64_t *labels, int64_t ignore, int64_t n_points, int32_t n_classes,
                          float *probas, float *errors, double *partials, ts_stream_t stream);
int ts_ce_lovasz_finish(const double *partials, int64_t n_points, int32_t n_classes, float smoothing, float w_ce, float w_lov,
                        const float *lovasz, float *out4, ts_stream_t stream);
int ts_ce_lovasz_backward(const float *probas, const int64_t *labels, int64_t ignore, const float *grad_probas,
                          const float *out4, const float *grad_out, int64_t n_points, int32_t n_classes, float smoothing,
                          float w_ce, float w_lov, float *grad_logits, ts_stream_t stream);

/* Pre-split weight planes for the fp32 pair GEMMs (csrc/conv_pairs_s.hip) - an optional accelerator of
 * ts_conv_pair_gemm (one-shot hint below) and ts_conv_block_forward / _backward (TsConvBlockOpts.planes), no counterpart in the reference (its
 * convolution_forward_cuda, backend/convolution/convolution_cuda.cu:101-164, multiplies fp32 operands in cuBLAS).
 * The fp32 kernels evaluate a product on the bf16 matrix pipe through the exact split x = h + m + l of both operands;
 * for a weight that split is the same in every workgroup of every launch until the optimizer changes the weight.
 *   planes: 3 * K * c_in * c_out bf16 (16-byte aligned): h | m | l, each in the layout of W [K, c_in, c_out] (the forward
 *           product and the input gradient read the same planes); c_in * c_out % 8 == 0.
 *   ts_conv_split_planes[_batch]  write them (one launch per 16 weights): call after every update of the weight.
 *   ts_conv_planes_hint           one-shot and per thread: the NEXT ts_conv_pair_gemm made by this thread may read
 *           `planes` in place of `w` if its weight pointer is `w` and its shapes are (K, c_in, c_out); that call clears
 *           the hint whether it used it or not (the block calls take their planes as an argument and clear any hint).  Results are bit-identical with and without planes; keeping them in
 *           step with the weight is the caller's contract (taseg_amd/planes.py does it for the modules). */
typedef struct TsPlaneJob {
  const float *w;
  void *planes;
  int32_t K, c_in, c_out;
} TsPlaneJob;
int ts_conv_split_planes(const float *w, int32_t K, int32_t c_in, int32_t c_out, void *planes, ts_stream_t stream);
int ts_conv_split_planes_batch(const TsPlaneJob *jobs, int32_t n_jobs, ts_stream_t stream);
void ts_conv_planes_hint(const float *w, const void *planes, int32_t K, int32_t c_in, int32_t c_out);
/* The half-storage counterpart: job.planes = w16 [K, c_in, c_out] IEEE half (what ts_cast_weights_f16 writes), 16 weights
 * per launch.  A ts_conv_block_forward(half = 1) call with opts->w16_current takes its w16 argument as already cast and
 * launches no cast of its own (torch.autocast casts the weight in every call: conv.py:19). */
int ts_cast_weights_f16_batch(const TsPlaneJob *jobs, int32_t n_jobs, ts_stream_t stream);

/* Class-sorted implicit GEMM (csrc/conv_class.hip) - pass 1 of a convolution with the sums of up to nine offsets kept in the
 * accumulators.  The K offsets are cut into `groups` groups of K / groups <= 9; per group the destination rows are sorted by
 * their neighbour mask and cut into 128-row tiles; a workgroup owns a tile and walks the offsets of the tile's union mask.
 * Same product as convolution_forward_cuda / convolution_backward_cuda (backend/convolution/convolution_cuda.cu:101-278) on the
 * rulebook ts_build_kmap gave; another summation order than ts_conv_pair_gemm + ts_conv_gather_sum (1e-6-close, deterministic).
 *   submanifold 3x3x3 maps: groups = 3 (one z-plane of the kernel each): Z' has one row per (output row, z-plane) instead of one
 *     per rulebook pair (about 2 N instead of 6.5 N rows on a LiDAR scan), pass 2 is ts_conv_gather_sum with K = 3 and the
 *     plan's position table `pos`.  The map is its own transpose with the offsets reversed, so the SAME plan serves the input
 *     gradient (wt = 1, mirror = 1: slice K-1-k, transposed).
 *   2x2x2 strided maps (and their transposed use, convolution_cuda.cu:21,34): groups = 1, a DIRECT plan (`rows` instead of `pos`):
 *     all offsets of a destination row sit in one tile, the sums ARE the result and are stored straight into the destination
 *     rows - no Z, no pass 2.  Two plans per map: destination = coarse rows (nbr = the map's own table: strided forward,
 *     transposed input gradient) and destination = fine rows (nbr = ts_conv_nbr_transposed: transposed forward, strided input
 *     gradient; every fine row has exactly one pair); mirror = 0.  Rows without any neighbour are written as zeros.
 *   ts_conv_class_rows2(n, groups)  m_pad = groups * roundup(n, 128): slots of src / rows of Z' (ts_conv_class_rows: groups = 3)
 *   ts_conv_class_plan      nbr [K][n] -> src [K / groups][m_pad] (input row of (group offset, slot) or -1), tile_info
 *                           [m_pad / 128][2], n_tiles [2] (device: listed tiles, and their (tile, offset) steps - 128 * steps
 *                           row-products against the rulebook's P pairs says what the plan costs: mask-sorted LiDAR rows give
 *                           ~1.1 P, rows with unrelated masks up to 3.7 P), and exactly one of pos [groups][n] (row of Z' per
 *                           (group, destination) or -1) and rows [m_pad] (destination row per slot or -1; groups == 1)
 *   ts_conv_nbr_transposed  nbr_t [K][n_in] of a kernel map from its pos_in table and rulebook
 *   ts_conv_class_gemm      wt = 0: feat = input rows, kernel [K, c_red, c_out]; wt = 1: the transposed product (feat = output
 *                           gradients [*, c_red], kernel [K, c_out, c_red] as stored; mirror selects the slice K-1-k);
 *                           rows == NULL: zp [m_pad, c_out]; rows != NULL: zp [n, c_out] = the result
 *   ts_conv_class_conv      the WHOLE convolution on a three-group plan, out [n, c_out] (+ addend [n, c_out], optional): every row
 *                           is its own centre neighbour, so the centre group's tiles hold every output row exactly once - they
 *                           run as a second launch of the product, add the other two groups' Z' rows of their rows (pass 2's
 *                           additions in pass 2's order: the same bits as ts_conv_class_gemm + ts_conv_gather_sum) and store the
 *                           result rows; no pass-2 launch, the centre group's third of Z' is never written.  zp [m_pad, c_out]
 *                           is scratch for the outer groups' rows. */
int64_t ts_conv_class_rows(int64_t n);
int64_t ts_conv_class_rows2(int64_t n, int32_t groups);
size_t ts_conv_class_plan_workspace_bytes(int64_t n);
int ts_conv_class_plan(const int32_t *nbr, int64_t n, int32_t K, int32_t groups, int32_t *src, int32_t *tile_info,
                       int32_t *n_tiles, int32_t *pos, int32_t *rows, void *ws, size_t ws_bytes, ts_stream_t stream);
/* the direct plan of a strided map's one-pair-per-destination direction (destination = its input rows) straight from the rulebook,
 * without a sort (slot p = pair p; needs n_pairs == number of destination rows); ws >= 4 * ceil(n_pairs / 128) bytes */
int ts_conv_class_plan_pairs(const int32_t *nbmaps, const int32_t *nboffs, int32_t K, int64_t n_pairs, int32_t *src,
                             int32_t *tile_info, int32_t *n_tiles, int32_t *rows, void *ws, size_t ws_bytes, ts_stream_t stream);
int ts_conv_nbr_transposed(const int32_t *pos_in, const int32_t *nbmaps, int32_t K, int64_t n_in, int32_t *nbr_t,
                           ts_stream_t stream);
int32_t ts_conv_class_supported(int32_t c_red, int32_t c_out);
int ts_conv_class_gemm(const float *feat, int32_t c_red, const float *kernel, int32_t K, int32_t groups, int32_t c_out,
                       const int32_t *src, int64_t m_pad, const int32_t *tile_info, const int32_t *n_tiles, int32_t wt,
                       int32_t mirror, const int32_t *rows, float *zp, ts_stream_t stream);
/* does finishing inside the product beat pass 2 on a map of n rows (measured thresholds; what the block calls use) */
int32_t ts_conv_class_finish_pays(int64_t n, int32_t half);
int ts_conv_class_conv(const float *feat, int32_t c_red, const float *kernel, int32_t K, int32_t groups, int32_t c_out,
                       const int32_t *src, int64_t m_pad, const int32_t *tile_info, const int32_t *n_tiles, int32_t wt,
                       int32_t mirror, const int32_t *pos, int64_t n, const float *addend, float *zp, float *out,
                       ts_stream_t stream);
/* half storage (torch.autocast): feat / zp / out / addend IEEE half, w = the half weight [K, C_in, C_out] as stored */
int ts_conv_class_gemm_f16(const void *feat, int32_t c_red, const void *w, int32_t K, int32_t groups, int32_t c_out,
                           const int32_t *src, int64_t m_pad, const int32_t *tile_info, const int32_t *n_tiles, int32_t wt,
                           int32_t mirror, const int32_t *rows, void *zp, ts_stream_t stream);
int ts_conv_class_conv_f16(const void *feat, int32_t c_red, const void *w, int32_t K, int32_t groups, int32_t c_out,
                           const int32_t *src, int64_t m_pad, const int32_t *tile_info, const int32_t *n_tiles, int32_t wt,
                           int32_t mirror, const int32_t *pos, int64_t n, const void *addend, void *zp, void *out,
                           ts_stream_t stream);

/* Diagnostic: while `stamps` (device memory, 16 x uint64 per workgroup, `capacity` workgroups) is set, the 96- / 128-
 * column fp32 pair GEMMs run an instrumented instantiation whose workgroups leave shader-clock stamps of their phases
 * (tools/phase_probe.py); NULL switches back to the product kernels. */
void ts_debug_phase_stamps(unsigned long long *stamps, int64_t capacity);

/* ------------------------------------------------------------------------ */
/* 4. Host-side data stage moved on device                                   */
/* ------------------------------------------------------------------------ */

/* Dataset voxelisation, first half (pcseg/data/dataset/semantickitti/semantickitti_voxel.py:119-120,
 * semantickitti_voxel_ms.py:127-130,151): c = int32(round_half_even(p / voxel_size)) in float32, then
 * c -= shift.  points [n, point_stride] float32 (x, y, z first); batch_idx [n] int32 or NULL (= 0).
 * shift_in [n_batch,3] given: subtract it.  shift_in NULL: the per-batch minimum is computed into
 * mins_out [n_batch,3] and subtracted (pc_ -= pc_.min(0)).  out_coords [n,4] = (x, y, z, b). */
int ts_voxel_coords(const float *points, int64_t n, int32_t point_stride, float voxel_size,
                    const int32_t *batch_idx, int32_t n_batch, const int32_t *shift_in,
                    int32_t *mins_out, int32_t *out_coords, ts_stream_t stream);

/* Dataset voxelisation, second half: sparse_quantize(coords, return_index, return_inverse)
 * (torchsparse utils/quantize.py:9-46 == np.unique of the ravel key).  coords [n,4] int32 (x,y,z,b):
 *   out_index   [<=n] : for voxel v (ascending (b,x,y,z)) the index of its FIRST point in input order;
 *   out_inverse [n]   : voxel id of every point (global over the batch);
 *   *out_count        : number of voxels (-1 if a coordinate is outside [-2^17, 2^17) or b >= 1024). */
/* out [n_seg, 3] = per-segment minimum of (x, y, z) of points [n, point_stride] whose segment index is seg[i] (int64, 0 ..
 * n_seg - 1 <= 63): the current scan's minimum every fused cloud of a batch is clamped to (semantickitti_voxel_ms.py:121-124),
 * all samples in one launch */
int ts_segment_min3(const float *points, int64_t n, int32_t point_stride, const int64_t *seg, int32_t n_seg, float *out,
                    ts_stream_t stream);
/* The batched data stage (csrc/stage.hip, taseg_amd/data/stage.py::voxelize_batch_ms): the per-point decisions and the layout of
 * the fused clouds of a WHOLE batch (semantickitti_ms.py:303-308, semantickitti_voxel_ms.py:121-212; nuscenes_ms.py:320-328).
 *   ts_stage_keep_flags   keep[i] = pre_keep[i] (optional) AND table[scan_idx[i]][cls[i] (neg_col where cls[i] < 0)] AND
 *                         xyz(i) >= lo[sample] ; sample[i] = sample_of_scan[scan_idx[i]]  (table [n_scans, table_cols] bytes)
 *   ts_stage_layout       the fused clouds sample-major, current scan first: pts [n_cur + n_kept, f], lab, sample (int64 and
 *                         int32), is_cur, from the current points (cur_b ascending, cur_start [B + 1] their cumulative counts), the
 *                         history rows and the ascending indices idx [n_kept] of the kept ones (kept_start [B + 1] cumulative)
 *   ts_stage_split_voxels after ts_sparse_quantize on the batch: vox [m, 4] = coords4[index], start [B + 1] first voxel of every
 *                         sample, offset [B] cumulative voxel counts, inverse_local[i] = inverse[i] - start[row_sample[i]] */
int ts_stage_keep_flags(const float *points, int64_t n, int32_t point_stride, const uint8_t *pre_keep, const int32_t *scan_idx,
                        const int64_t *cls, const uint8_t *table, int32_t n_scans, int32_t table_cols, int32_t neg_col,
                        const int64_t *sample_of_scan, const float *lo, int32_t n_samples, uint8_t *keep, int64_t *sample,
                        ts_stream_t stream);
int ts_stage_layout(const float *cur, const int64_t *cur_lab, const int64_t *cur_b, int64_t n_cur, const float *hist,
                    const int64_t *hist_lab, const int64_t *hist_b, const int64_t *idx, int64_t n_kept, int32_t f,
                    const int64_t *cur_start, const int64_t *kept_start, float *pts, int64_t *lab, int64_t *sample, int32_t *sample32,
                    uint8_t *is_cur, ts_stream_t stream);
int ts_stage_split_voxels(const int32_t *coords4, const int32_t *index, int64_t m, const int32_t *inverse,
                          const int64_t *row_sample, int64_t n, int32_t n_samples, int32_t *vox, int64_t *start, int32_t *offset,
                          int64_t *inverse_local, ts_stream_t stream);
size_t ts_quantize_workspace_bytes(int64_t n);
int ts_sparse_quantize(const int32_t *coords, int64_t n, int32_t *out_index, int32_t *out_inverse,
                       int32_t *out_count, void *ws, size_t ws_bytes, ts_stream_t stream);

/* fuse_multi_scan (pcseg/data/dataset/semantickitti/semantickitti_ms.py:403-417):
 * p' = ((p . R_t^T + t_t) - t_0) . R_0 for one history scan, in float32 with the
 * reference's summation order; points [n,4] (x,y,z,intensity), pose/pose0 are
 * 4x4 row-major float32 on the device; out [n,4]. */
int ts_fuse_scan(const float *points, int64_t n, const float *pose0, const float *pose,
                 float *out, ts_stream_t stream);
/* The same transform for the concatenated history scans of a sample: point i uses poses[scan_idx[i]]
 * (poses [n_scans, 4, 4] row-major).  Bit-identical to n_scans calls of ts_fuse_scan. */
int ts_fuse_scans(const float *points, const int32_t *scan_idx, int64_t n, const float *pose0, const float *poses,
                  int32_t n_scans, float *out, ts_stream_t stream);
/* the same for the history scans of a whole BATCH of samples: pose0s [n_scans, 4, 4] = the current-frame pose of the sample the
 * scan belongs to (one launch per batch instead of one per sample) */
int ts_fuse_scans_batch(const float *points, const int32_t *scan_idx, int64_t n, const float *pose0s, const float *poses,
                        int32_t n_scans, float *out, ts_stream_t stream);

/* nuScenes multi-scan fuse (pcseg/data/dataset/nuscenes/nuscenes_ms.py:280-318 per selected sweep, :348-373
 * transform_point): for point i of the concatenated sweeps, with s = sweep_idx[i] and params[s] = 28 doubles
 *   { A[9], a[3], flagA, B[9], b[3], flagB, dt, pad }:
 *   keep[i] = !(|x| < 1.0 && |y| < 1.5) on the RAW coordinates          (ego-box filter, :288 / :306)
 *   flagA:  p = f32(p . A^T); p = f32(p + a)                            (sweep -> its keyframe's lidar frame, :307-309)
 *   flagB:  p = f32(p . B + b)                                          (-> the current lidar frame, transform_point)
 *   out[i] = (p, intensity, f32(dt))                                    (:289 / :310)
 * Arithmetic in float64 with the fused-multiply-add chain numpy's matmul (dgemm) performs, rounded to float32 where
 * the reference stores into its float32 array.  points / out [n,5] float32; keep [n] uint8. */
int ts_fuse_sweeps(const float *points, const int32_t *sweep_idx, int64_t n, const double *params, int32_t n_sweeps,
                   float *out, uint8_t *keep, ts_stream_t stream);

/* TIAF camera projection of one scan (pcseg/data/dataset/semantickitti/semantickitti_ms_mm.py:411-461 get_fov_points,
 * the per-point part): for point i = (x, y, z, .)
 *   uvz = P . (x, y, z, 1)            P = P2 . Tr, 3x4 row-major float64; float64 FMA chain in k order (numpy's dgemm)
 *   keep[i] = x > 0  &&  0 < u < img_w  &&  0 < v < img_h  (u = uvz0 / uvz2, v = uvz1 / uvz2, float64)
 *                    &&  (int)v < crop_h  &&  (int)u < crop_w
 *   pix[i]  = ( float((int)v) + row_offset , float((int)u) )        row_offset = crop_h * frame index in the image stack
 * points [n,4] float32; pix [n,2] float32 (undefined where keep == 0); keep [n] uint8. */
int ts_project_fov(const float *points, int64_t n, const double *proj, int32_t img_w, int32_t img_h, int32_t crop_h,
                   int32_t crop_w, float row_offset, float *pix, uint8_t *keep, ts_stream_t stream);

/* nuScenes TIAF camera projection (pcseg/data/dataset/nuscenes/nuscenes_ms_mm.py:349-398 get_fov_points, the per-point
 * part): lidar frame -> ego -> global -> camera ego -> camera through the calibrated-sensor and ego-pose records of the lidar
 * and camera sample_data entries, pinhole projection with nuscenes-devkit's view_points, half-resolution pixel.
 *   cam[57] float64 = { M1[9] t1[3]  M2[9] t2[3]  t3[3] M3[9]  t4[3] M4[9]  K[9] }   (row-major 3x3; M3 / M4 = the TRANSPOSED
 *                       rotations of the camera's ego pose / calibrated sensor, as the reference applies them)
 *   p = M1 x + t1;  p = M2 p + t2;  p = M3 (p - t3);  p = M4 (p - t4)        every product an FMA chain in k order (dgemm)
 *   (u, v) = f32( (K p)[0,1] / (K p)[2] )
 *   keep[i] = p_z > 0  &&  0 < u < img_w  &&  0 < v < img_h  &&  ((int)v >> 1) >= crop_top
 *   pix[i]  = ( float(((int)v >> 1) - crop_top) + row_offset , float((int)u >> 1) )    row_offset = HEIGHT * image index
 * points [n,4] float32; pix [n,2] float32 (undefined where keep == 0); keep [n] uint8. */
int ts_project_cam(const float *points, int64_t n, const double *cam, int32_t img_w, int32_t img_h, int32_t crop_top,
                   float row_offset, float *pix, uint8_t *keep, ts_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TASEG_HIP_H_ */
