// One host call per direction for the unit the MinkUNet family is made of: sparse conv -> BatchNorm (training mode)
// [+ residual] [-> ReLU]   (reference: BasicConvolutionBlock / BasicDeconvolutionBlock / ResidualBlock,
// R/pcseg/model/segmentor/voxel/minkunet/minkunet.py:31-129; per block the reference dispatches conv3d, BatchNorm1d /
// SyncBatchNorm, the shortcut add and ReLU as separate autograd nodes and ~10 launches with a host synchronisation).
//
// No new arithmetic: the entry points chain the launches of ts_conv_pair_gemm / ts_conv_gather_sum /
// ts_bn_act_train_* (or their half-storage and SyncBatchNorm forms) on the caller's stream and place the
// intermediates that never leave the call (Z, the BatchNorm input gradient, the transposed half weight) in the
// caller's workspace.  What they remove is host work: the step of a 2-scan batch is ~1200 launches, and with one
// Python -> C crossing and one autograd node per block instead of four / two the host side of the step drops below
// the device side on slow hosts too.
#include <sched.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

static inline size_t blk_align(size_t x) { return ts_align_up(x, 256); }

// scratch for either direction of one block (bytes); esize = 2 for half storage
extern "C" size_t ts_conv_block_workspace_bytes(int64_t n_pairs, int64_t n_rows_max, int32_t c_in, int32_t c_out,
                                                int32_t K, int32_t half) {
  const size_t es = half ? 2 : 4;
  const size_t cmax = (size_t)std::max(c_in, c_out);
  size_t total = blk_align((size_t)n_pairs * cmax * es);            // Z (forward: c_out wide, backward: c_in wide)
  total += blk_align((size_t)n_rows_max * cmax * es);               // gradient w.r.t. the convolution output
  if (half) total += blk_align((size_t)K * c_in * c_out * 2);       // W16T of the forward pass
  total += blk_align(ts_bn_train_workspace_bytes(std::max(c_in, c_out)));
  total += blk_align(ts_wgrad_partial_bytes(n_pairs, c_in, c_out, K));   // weight-gradient partial tiles (backward)
  return total;
}

// the part of the backward call's scratch that the weight gradient on a second stream reads and writes (TsConvBlockOpts):
// the gradient w.r.t. the convolution output, then the partial tiles
extern "C" size_t ts_conv_block_wgrad_ws_bytes(int64_t n_pairs, int64_t n_out, int32_t c_in, int32_t c_out, int32_t K, int32_t half) {
  const size_t es = half ? 2 : 4;
  return blk_align((size_t)n_out * (size_t)std::max(c_in, c_out) * es) + blk_align(ts_wgrad_partial_bytes(n_pairs, c_in, c_out, K));
}

// `waiter` waits for everything enqueued on `other` so far.  The events come from a small ring that lives as long as the library:
// until round 6 every call created an event, recorded it, made the waiter wait and DESTROYED it right away - legal by the API's
// letter (resources are released when the wait has completed), but the one place where a missed wait would show is exactly where an
// intermittent failure did: two rank processes on one card with the weight gradients on a second stream, where a gradient bucket's
// all-reduce reads the slots right behind this join (tests/test_gpu_dist.py::test_second_stream_under_the_bucket_reducer...: once a
// crashed rank, once gradients that were not the one-stream run's).  A wait on an event that still exists has no such question
// mark; re-recording a ring entry while an older wait on it is pending is fine (a wait refers to the record at the time of the call).
extern "C" int ts_stream_join(ts_stream_t waiter, ts_stream_t other) {
  if (waiter == other) return TS_OK;
  constexpr int RING = 64;
  static hipEvent_t ring[RING];
  static std::atomic<int> made{0};
  static std::atomic<unsigned> next{0};
  static std::mutex mu;
  if (made.load(std::memory_order_acquire) == 0) {
    std::lock_guard<std::mutex> lock(mu);
    if (made.load(std::memory_order_relaxed) == 0) {
      for (int i = 0; i < RING; ++i) TS_CHECK_HIP(hipEventCreateWithFlags(&ring[i], hipEventDisableTiming), "ts_stream_join: event");
      made.store(1, std::memory_order_release);
    }
  }
  hipEvent_t ev = ring[next.fetch_add(1) % RING];
  std::lock_guard<std::mutex> lock(mu);          // (record + wait of one entry as a pair: two host threads join streams here)
  TS_CHECK_HIP(hipEventRecord(ev, (hipStream_t)other), "ts_stream_join: record");
  TS_CHECK_HIP(hipStreamWaitEvent((hipStream_t)waiter, ev, 0), "ts_stream_join: wait");
  return TS_OK;
}

// events of the weight-gradient ring (TsConvBlockOpts.wgrad_slot): ready[s] = the slot's output gradient exists (caller's stream),
// done[s] = the slot's weight gradient has finished (second stream)
static hipEvent_t g_wg_ready[8], g_wg_done[8];
static std::atomic<bool> g_wg_used[8];          // done[s] has been recorded at least once
static std::atomic<int> g_wg_owed[8];           // deferred form: the slot's weight gradient has been promised but not enqueued yet
static int wg_events() {
  static bool made = false;
  if (!made) {
    for (int i = 0; i < 8; ++i) {
      TS_CHECK_HIP(hipEventCreateWithFlags(&g_wg_ready[i], hipEventDisableTiming), "weight-gradient ring: event");
      TS_CHECK_HIP(hipEventCreateWithFlags(&g_wg_done[i], hipEventDisableTiming), "weight-gradient ring: event");
    }
    made = true;
  }
  return TS_OK;
}

#define TS_TRY(expr)              \
  do {                            \
    const int rc_ = (expr);       \
    if (rc_ != TS_OK) return rc_; \
  } while (0)

// ---- per-launch timing of the convolution kernels inside the block calls (bench.py's roofline figures) --------------
// While recording is on, every pair-GEMM / gather-sum / weight-gradient launch issued by ts_conv_block_* is bracketed by
// two HIP events on the caller's stream (from a pool created on first use).  ts_prof_collect synchronises the events
// and returns one record per launch: {kind (0 pair GEMM, 1 gather-sum, 2 weight gradient), milliseconds, pairs,
// c_red, c_out, K, rows (pair GEMM / weight gradient: rows of the gathered matrix; gather-sum: rows written), bytes per
// element, weight transposed (weight gradient: rows of the second operand; gather-sum: bytes of the weight-gradient sum
// it carries on the side)}.  Off (the default): one predictable branch per launch.
namespace {
struct ProfRec {
  int kind;
  hipEvent_t e0, e1;
  double meta[7];
};
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_prof_pool;
bool g_prof_on = false;
std::mutex g_prof_mutex;          // the weight gradients may be launched by a second host thread (ts_conv_block_wgrad_side)

hipEvent_t prof_event() {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  if (!g_prof_pool.empty()) {
    hipEvent_t e = g_prof_pool.back();
    g_prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

struct ProfScope {
  bool live = false;
  ProfRec rec;
  hipStream_t stream;
  ProfScope(int kind, ts_stream_t s, double pairs, double c_red, double c_out, double k, double rows, double esize, double wt)
      : stream((hipStream_t)s) {
    if (!g_prof_on) return;
    rec.kind = kind;
    rec.e0 = prof_event();
    rec.e1 = prof_event();
    if (!rec.e0 || !rec.e1) return;
    const double m[7] = {pairs, c_red, c_out, k, rows, esize, wt};
    for (int i = 0; i < 7; ++i) rec.meta[i] = m[i];
    live = hipEventRecord(rec.e0, stream) == hipSuccess;
  }
  ~ProfScope() {
    if (live && hipEventRecord(rec.e1, stream) == hipSuccess) {
      std::lock_guard<std::mutex> lock(g_prof_mutex);
      g_prof.push_back(rec);
    }
  }
};
}  // namespace

extern "C" void ts_prof_enable(int32_t on) { g_prof_on = on != 0; }

// create `n_events` events ahead of time (hipEventCreate costs ~15 us; the pool otherwise grows inside the first
// recorded steps)
extern "C" int ts_prof_reserve(int64_t n_events) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  while ((int64_t)g_prof_pool.size() < n_events) {
    hipEvent_t e = nullptr;
    TS_CHECK_HIP(hipEventCreate(&e), "hipEventCreate");
    g_prof_pool.push_back(e);
  }
  return TS_OK;
}

// records [capacity][9] doubles out; returns the number of records written (and forgets them), or -1 on a HIP error
extern "C" int64_t ts_prof_collect(double *records, int64_t capacity) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  int64_t n = 0;
  for (const ProfRec &r : g_prof) {
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return -1;
    if (n < capacity && records) {
      double *o = records + n * 9;
      o[0] = r.kind;
      o[1] = ms;
      for (int i = 0; i < 7; ++i) o[2 + i] = r.meta[i];
      ++n;
    }
    g_prof_pool.push_back(r.e0);
    g_prof_pool.push_back(r.e1);
  }
  g_prof.clear();
  return n;
}

// what an event pair measures with NOTHING between its two records (microseconds, median of `reps` pairs recorded back
// to back on `stream`): the part of every bracketed launch's elapsed time that is the bracket's own - bench.py subtracts
// it from the per-launch figures and reports it next to them
extern "C" int ts_prof_empty_bracket_us(int32_t reps, ts_stream_t stream_, double *out_us) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(reps > 0 && reps <= 4096 && out_us, TS_ERR_INVALID_ARGUMENT, "ts_prof_empty_bracket_us: bad arguments");
  std::vector<hipEvent_t> ev(2 * (size_t)reps, nullptr);
  for (auto &e : ev) {
    e = prof_event();
    TS_REQUIRE(e, TS_ERR_LAUNCH_FAILED, "ts_prof_empty_bracket_us: hipEventCreate failed");
  }
  for (auto &e : ev) TS_CHECK_HIP(hipEventRecord(e, stream), "hipEventRecord");
  TS_CHECK_HIP(hipEventSynchronize(ev.back()), "hipEventSynchronize");
  std::vector<double> us;
  for (int i = 0; i < reps; ++i) {
    float ms = 0.f;
    TS_CHECK_HIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]), "hipEventElapsedTime");
    us.push_back(1e3 * ms);
  }
  std::sort(us.begin(), us.end());
  *out_us = us[us.size() / 2];
  for (auto &e : ev) g_prof_pool.push_back(e);
  return TS_OK;
}

// Everything a block call may use beyond the rulebook arrives in TsConvBlockOpts (include/taseg_hip.h) - nothing is taken
// from, or left in, thread-local state: a stale ts_conv_planes_hint of the calling thread is cleared on entry and the planes
// of THIS call are armed right before its one fp32 pair GEMM.
struct PlanesScope {
  TsPlanesHint h;
  PlanesScope(const float *w, const void *planes, int K, int c_in, int c_out)
      : h{planes ? w : nullptr, (const unsigned short *)planes, K, c_in, c_out} {
    g_ts_planes_hint = TsPlanesHint{nullptr, nullptr, 0, 0, 0};
  }
  void arm() const { g_ts_planes_hint = h; }
  ~PlanesScope() { g_ts_planes_hint = TsPlanesHint{nullptr, nullptr, 0, 0, 0}; }
};

// may this call run a product on `p`?  K offsets over `n_dest` destination rows, plan built from THIS kernel map (map_id), shapes
// the class kernels take; pass-2 plans also need Z' to fit where Z would have gone
static bool plan_fits(const TsClassPlan *p, int32_t K, int64_t n_dest, int32_t c_red, int32_t c_out, int64_t n_pairs,
                      const int32_t *nboffs) {
  if (!p || !p->src || !p->tile_info || !p->n_tiles || g_ts_conv_impl != 0) return false;
  if (p->K != K || p->n != n_dest || p->map_id != (const void *)nboffs || !ts_conv_class_supported(c_red, c_out)) return false;
  if (p->groups < 1 || K % p->groups != 0 || p->m_pad != ts_conv_class_rows2(n_dest, p->groups)) return false;
  if (p->rows) return p->groups == 1 && !p->pos;
  return p->pos && p->m_pad <= n_pairs;
}
static double plan_z_rows(const TsClassPlan *p) { return (double)(p->z_rows > 0 ? p->z_rows : p->m_pad); }
// a three-group pass-2 plan finishes inside the product (conv_class.hip, TsClassFinish)
static bool plan_finishes(const TsClassPlan *p, int32_t half) {
  return p && !p->rows && p->pos && p->groups == 3 && p->K == 27 && ts_conv_class_finish_pays(p->n, half);
}
// ... and moves these Z' rows: the two outer groups' rows are written and read back once (the centre group holds every row)
static double plan_z_moves(const TsClassPlan *p, int64_t n) {
  return 2.0 * std::max(0.0, plan_z_rows(p) - (double)((n + 127) / 128 * 128));
}

// The convolution of a block: conv_out [n_out, c_out] = sum over the rulebook, on the class plan of opts (where it fits) or as pair
// GEMM + pass 2 through z.  Shared by the training forward and the evaluation forward.
// tail (evaluation block only): where the convolution ends in the list form of pass 2, that launch applies the block's elementwise
// tail in its store and writes tail_out instead of conv_out; *tail_done says whether it did.
static int block_conv(const void *feat, int64_t n_feat_rows, int32_t c_in, const float *kernel, int32_t K, const int32_t *nbmaps,
                      const int32_t *nboffs, int64_t n_pairs, int32_t gather_col, const int32_t *pos, int64_t n_out, int32_t c_out,
                      int32_t half, void *conv_out, void *w16, const TsConvBlockOpts &o, const PlanesScope &planes, void *z,
                      ts_stream_t stream, const TsGatherEpilogue *tail = nullptr, void *tail_out = nullptr, bool *tail_done = nullptr) {
  // pass 2 of this convolution, with the tail where it applies
  auto pass2 = [&](const void *zz, const int32_t *table, int32_t kk, int64_t zrows) -> int {
    if (tail && tail_out) {
      const int rc = half ? ts_conv_gather_sum_f16_epi(zz, c_out, table, kk, n_out, zrows, tail_out, *tail, stream)
                          : ts_conv_gather_sum_epi((const float *)zz, c_out, table, kk, n_out, zrows, (float *)tail_out, *tail, stream);
      if (rc == TS_OK) {
        if (tail_done) *tail_done = true;
        return TS_OK;
      }
      if (rc != TS_ERR_UNSUPPORTED) return rc;
    }
    return half ? ts_conv_gather_sum_f16(zz, c_out, table, kk, n_out, zrows, conv_out, stream)
                : ts_conv_gather_sum((const float *)zz, c_out, table, kk, n_out, zrows, (float *)conv_out, stream);
  };
  if (o.natural) {
    // 1x1x1 convolution on the identity rulebook: the pair GEMM's rows are the result rows
    TS_REQUIRE(K == 1 && n_pairs == n_out && n_feat_rows == n_out, TS_ERR_INVALID_ARGUMENT,
               "ts_conv_block: a natural call needs K = 1 and one pair per row (%d offsets, %lld pairs, %lld / %lld rows)", K,
               (long long)n_pairs, (long long)n_feat_rows, (long long)n_out);
    if (half && !o.w16_current) TS_TRY(ts_cast_weights_f16(kernel, K, c_in, c_out, w16, nullptr, stream));
    ProfScope ps(0, stream, (double)n_pairs, c_in, c_out, K, (double)n_feat_rows, half ? 2 : 4, 0);
    if (half) return ts_conv_pair_gemm_f16_nat(feat, n_feat_rows, c_in, w16, K, nbmaps, nboffs, n_pairs, gather_col, conv_out, c_out, stream);
    return ts_conv_pair_gemm((const float *)feat, n_feat_rows, c_in, kernel, K, 0, nbmaps, nboffs, n_pairs, gather_col,
                             (float *)conv_out, c_out, stream);
  }
  const TsClassPlan *cp = plan_fits(o.fwd_plan, K, n_out, c_in, c_out, n_pairs, nboffs) ? o.fwd_plan : nullptr;
  {
    // half storage: one half copy in the kernel's own layout serves both passes (the forward reads it through the transposing
    // LDS load, the input gradient directly); the cast is skipped when the caller keeps w16 in step with the weight itself
    if (half && !o.w16_current) TS_TRY(ts_cast_weights_f16(kernel, K, c_in, c_out, w16, nullptr, stream));
    const double es_d = half ? 2 : 4;
    if (cp) {
      // class-sorted implicit GEMM: the sums of a group of offsets stay in the accumulators; a direct plan (2x2x2 maps) stores
      // the result rows themselves, a pass-2 plan leaves <= 3 rows of Z' per output
      // a three-group pass-2 plan finishes the sum in the product (two launches, no pass 2); other pass-2 plans leave Z' to pass 2
      const bool fused = plan_finishes(cp, half);
      void *dst = cp->rows ? conv_out : z;
      const TsClassFinish fin = {cp->pos, n_out, conv_out, nullptr};
      {
        // (last field, kind 3: rows of Z' a pass-2 plan writes, or minus the result rows a direct plan writes; kind 4: the Z' rows
        // the finish-in-the-product form moves - written by the two outer groups, read back by the centre group)
        ProfScope ps(fused ? 4 : 3, stream, (double)n_pairs, c_in, c_out, K, (double)n_feat_rows, es_d,
                     cp->rows ? -(double)n_out : fused ? plan_z_moves(cp, n_out) : plan_z_rows(cp));
        if (half)
          TS_TRY(ts_conv_class_gemm_f16_ex(feat, c_in, w16, K, cp->groups, c_out, cp->src, cp->m_pad, cp->tile_info, cp->n_tiles,
                                           0, 0, cp->rows, dst, nullptr, fused ? &fin : nullptr, stream));
        else
          TS_TRY(ts_conv_class_gemm_ex((const float *)feat, c_in, kernel, K, cp->groups, c_out, cp->src, cp->m_pad, cp->tile_info,
                                       cp->n_tiles, 0, 0, cp->rows, (float *)dst, nullptr, fused ? &fin : nullptr, stream));
      }
      if (!cp->rows && !fused) {
        ProfScope ps(1, stream, plan_z_rows(cp), 0, c_out, cp->groups, (double)n_out, es_d, 0);
        TS_TRY(pass2(z, cp->pos, cp->groups, cp->m_pad));
      }
    } else if (half) {
      {
        ProfScope ps(0, stream, (double)n_pairs, c_in, c_out, K, (double)n_feat_rows, 2, 0);
        TS_TRY(ts_conv_pair_gemm_f16_nat(feat, n_feat_rows, c_in, w16, K, nbmaps, nboffs, n_pairs, gather_col, z, c_out, stream));
      }
      {
        ProfScope ps(1, stream, (double)n_pairs, 0, c_out, K, (double)n_out, 2, 0);
        TS_TRY(pass2(z, pos, K, n_pairs));
      }
    } else {
      {
        ProfScope ps(0, stream, (double)n_pairs, c_in, c_out, K, (double)n_feat_rows, 4, 0);
        planes.arm();
        TS_TRY(ts_conv_pair_gemm((const float *)feat, n_feat_rows, c_in, kernel, K, 0, nbmaps, nboffs, n_pairs, gather_col,
                                 (float *)z, c_out, stream));
      }
      {
        ProfScope ps(1, stream, (double)n_pairs, 0, c_out, K, (double)n_out, 4, 0);
        TS_TRY(pass2(z, pos, K, n_pairs));
      }
    }
  }
  return TS_OK;
}

// out = act(BN(conv(feat)) [+ residual]).
//   feat [n_feat_rows, c_in]; kernel fp32 [K, c_in, c_out]; rulebook (nbmaps, nboffs, n_pairs) with the gathered column
//   `gather_col` and the position table pos [K, n_out] of the rows being produced (pos_out, or pos_in for a transposed
//   convolution); conv_out [n_out, c_out] is kept for the backward pass (BatchNorm input), as are mean / invstd / mask
//   and - half storage - w16 [K, c_in, c_out].  comm != NULL: SyncBatchNorm on that communicator (pack [2 c_out + 1]
//   doubles; pack[2 c_out] = global row count afterwards).
extern "C" int ts_conv_block_forward(const void *feat, int64_t n_feat_rows, int32_t c_in, const float *kernel, int32_t K,
                                     const int32_t *nbmaps, const int32_t *nboffs, int64_t n_pairs, int32_t gather_col,
                                     const int32_t *pos, int64_t n_out, int32_t c_out, const void *residual,
                                     const float *bn_weight, const float *bn_bias, float *running_mean,
                                     float *running_var, int64_t *num_batches_tracked, float eps, float momentum,
                                     int32_t relu, int32_t half, void *comm, double *pack, void *conv_out, float *mean,
                                     float *invstd, void *out, uint8_t *mask, void *w16, const TsConvBlockOpts *opts, void *ws,
                                     size_t ws_bytes, ts_stream_t stream) {
  const TsConvBlockOpts none = {};
  const TsConvBlockOpts &o = opts ? *opts : none;
  PlanesScope planes(kernel, half ? nullptr : o.planes, K, c_in, c_out);
  TS_REQUIRE(n_pairs > 0 && n_out > 0 && c_in > 0 && c_out > 0 && K > 0, TS_ERR_INVALID_ARGUMENT,
             "ts_conv_block_forward: bad sizes");
  TS_REQUIRE(ws && ws_bytes >= ts_conv_block_workspace_bytes(n_pairs, n_out, c_in, c_out, K, half), TS_ERR_INVALID_ARGUMENT,
             "ts_conv_block_forward: workspace too small");
  TS_REQUIRE(!half || w16, TS_ERR_INVALID_ARGUMENT, "ts_conv_block_forward: half storage needs the w16 buffer");
  const size_t es = half ? 2 : 4;
  const size_t cmax = (size_t)std::max(c_in, c_out);
  char *p = (char *)ws;
  void *z = p;
  p += blk_align((size_t)n_pairs * cmax * es);
  p += blk_align((size_t)n_out * cmax * es);
  if (half) p += blk_align((size_t)K * c_in * c_out * 2);
  void *bn_ws = p;
  const size_t bn_ws_bytes = ts_bn_train_workspace_bytes(std::max(c_in, c_out));
  if (comm == TS_COMM_CALLER_POST) {
    // second half of a split call: conv_out and the all-reduced pack exist, statistics + elementwise pass are left
  } else {
    TS_TRY(block_conv(feat, n_feat_rows, c_in, kernel, K, nbmaps, nboffs, n_pairs, gather_col, pos, n_out, c_out, half, conv_out,
                      w16, o, planes, z, stream));
  }
  if (comm)
    return ts_bn_sync_forward(comm, conv_out, residual, bn_weight, bn_bias, running_mean, running_var, num_batches_tracked,
                              n_out, c_out, eps, momentum, relu, half, pack, mean, invstd, out, mask, bn_ws, bn_ws_bytes,
                              stream);
  if (half)
    return ts_bn_act_train_forward_f16(conv_out, residual, bn_weight, bn_bias, running_mean, running_var,
                                       num_batches_tracked, n_out, c_out, eps, momentum, relu, mean, invstd, out, mask,
                                       bn_ws, bn_ws_bytes, stream);
  return ts_bn_act_train_forward((const float *)conv_out, (const float *)residual, bn_weight, bn_bias, running_mean,
                                 running_var, num_batches_tracked, n_out, c_out, eps, momentum, relu, mean, invstd,
                                 (float *)out, mask, bn_ws, bn_ws_bytes, stream);
}

// Evaluation form of the block (module in eval mode, minkunet.py:435-455 / train.py:452-540: running statistics, no graph):
// out = act((conv(feat) - mean) * invstd * bn_weight + bn_bias [+ residual]) with the caller's mean / invstd (running_mean,
// 1 / sqrt(running_var + eps)) - convolution as in the training forward (class plans included), then ONE elementwise pass; the
// convolution output lives in the workspace.
extern "C" int ts_conv_block_eval(const void *feat, int64_t n_feat_rows, int32_t c_in, const float *kernel, int32_t K,
                                  const int32_t *nbmaps, const int32_t *nboffs, int64_t n_pairs, int32_t gather_col,
                                  const int32_t *pos, int64_t n_out, int32_t c_out, const void *residual, const float *bn_weight,
                                  const float *bn_bias, const float *mean, const float *invstd, int32_t relu, int32_t half,
                                  void *out, void *w16, const TsConvBlockOpts *opts, void *ws, size_t ws_bytes,
                                  ts_stream_t stream) {
  const TsConvBlockOpts none = {};
  const TsConvBlockOpts &o = opts ? *opts : none;
  PlanesScope planes(kernel, half ? nullptr : o.planes, K, c_in, c_out);
  TS_REQUIRE(n_pairs > 0 && n_out > 0 && c_in > 0 && c_out > 0 && K > 0, TS_ERR_INVALID_ARGUMENT, "ts_conv_block_eval: bad sizes");
  TS_REQUIRE(ws && ws_bytes >= ts_conv_block_workspace_bytes(n_pairs, n_out, c_in, c_out, K, half), TS_ERR_INVALID_ARGUMENT,
             "ts_conv_block_eval: workspace too small");
  TS_REQUIRE(!half || w16, TS_ERR_INVALID_ARGUMENT, "ts_conv_block_eval: half storage needs the w16 buffer");
  TS_REQUIRE(mean && invstd && bn_weight && bn_bias && out, TS_ERR_INVALID_ARGUMENT, "ts_conv_block_eval: null pointer");
  const size_t es = half ? 2 : 4;
  const size_t cmax = (size_t)std::max(c_in, c_out);
  char *p = (char *)ws;
  void *z = p;
  p += blk_align((size_t)n_pairs * cmax * es);
  void *conv_out = p;
  // the elementwise tail rides on pass 2 where the convolution ends in its list form (csrc/conv_pairs*.hip: one launch and one
  // round trip of the convolution output less; TASEG_EVAL_TAIL_IN_PASS2=0 keeps the separate pass)
  const bool tail_in_pass2 = ts_get_option(TS_OPT_EVAL_TAIL_SEPARATE) == 0;
  const TsGatherEpilogue tail = {mean, invstd, bn_weight, bn_bias, residual, relu ? 1 : 0};
  bool tail_done = false;
  TS_TRY(block_conv(feat, n_feat_rows, c_in, kernel, K, nbmaps, nboffs, n_pairs, gather_col, pos, n_out, c_out, half, conv_out, w16, o,
                    planes, z, stream, tail_in_pass2 ? &tail : nullptr, out, &tail_done));
  if (tail_done) return TS_OK;
  if (half)
    return ts_bn_act_forward_f16(conv_out, residual, mean, invstd, bn_weight, bn_bias, n_out, c_out, relu, out, nullptr, stream);
  return ts_bn_act_forward((const float *)conv_out, (const float *)residual, mean, invstd, bn_weight, bn_bias, n_out, c_out, relu,
                           (float *)out, nullptr, stream);
}

// Backward of ts_conv_block_forward.  weights = the fp32 kernel, or (half storage) the w16 buffer the forward filled.
// dgrad_gather_col / pos_dgrad [K, n_dgrad_rows] / wgrad_col_a select the rulebook columns exactly like
// torchsparse's ConvolutionFunction.backward (conv.py:72-119; transposed swaps them).  grad_feat may be NULL (first
// layer).  grad_residual (may be NULL) receives the masked output gradient.
extern "C" int ts_conv_block_backward(const void *grad_out, const uint8_t *mask, const void *conv_out, const float *mean,
                                      const float *invstd, const float *bn_weight, const double *total_dev, void *comm,
                                      double *sums, int64_t n_out, int32_t c_out, int32_t half, const void *feat,
                                      int64_t n_feat_rows, int32_t c_in, const void *weights, int32_t K,
                                      const int32_t *nbmaps, const int32_t *nboffs, int64_t n_pairs,
                                      int32_t dgrad_gather_col, const int32_t *pos_dgrad, int64_t n_dgrad_rows,
                                      int32_t wgrad_col_a, void *grad_feat, void *grad_residual, float *grad_kernel,
                                      float *grad_bn_weight, float *grad_bn_bias, const TsConvBlockOpts *opts, void *ws,
                                      size_t ws_bytes, ts_stream_t stream) {
  const TsConvBlockOpts none = {};
  const TsConvBlockOpts &o = opts ? *opts : none;
  PlanesScope planes((const float *)weights, half ? nullptr : o.planes, K, c_in, c_out);
  const void *addend = o.addend;                // added into grad_feat's store
  TS_REQUIRE(n_pairs > 0 && n_out > 0 && c_in > 0 && c_out > 0 && K > 0, TS_ERR_INVALID_ARGUMENT,
             "ts_conv_block_backward: bad sizes");
  TS_REQUIRE(!addend || (grad_feat && (((uintptr_t)addend) & 15) == 0), TS_ERR_INVALID_ARGUMENT,
             "ts_conv_block_backward: an addend needs grad_feat and 16-byte alignment");
  TS_REQUIRE(!o.natural || (K == 1 && n_pairs == n_out && n_dgrad_rows == n_out && n_feat_rows == n_out && !addend),
             TS_ERR_INVALID_ARGUMENT, "ts_conv_block_backward: a natural call needs K = 1, one pair per row and no addend");
  TS_REQUIRE(ws && ws_bytes >= ts_conv_block_workspace_bytes(n_pairs, n_out, c_in, c_out, K, half), TS_ERR_INVALID_ARGUMENT,
             "ts_conv_block_backward: workspace too small");
  const size_t es = half ? 2 : 4;
  const size_t cmax = (size_t)std::max(c_in, c_out);
  char *p = (char *)ws;
  void *z = p;
  p += blk_align((size_t)n_pairs * cmax * es);
  void *grad_conv = p;
  p += blk_align((size_t)n_out * cmax * es);
  if (half) p += blk_align((size_t)K * c_in * c_out * 2);
  void *bn_ws = p;
  const size_t bn_ws_bytes = ts_bn_train_workspace_bytes(std::max(c_in, c_out));
  if (comm == TS_COMM_CALLER_PRE) {
    // first half of a split call: this rank's sums of the BatchNorm backward, which the caller all-reduces; nothing else is
    // touched (the second half writes the gradient w.r.t. the convolution output - into the ring slot where the weight gradient
    // goes to the second stream)
    return ts_bn_sync_backward(comm, grad_out, mask, conv_out, mean, invstd, bn_weight, total_dev, n_out, c_out, half, sums, grad_conv,
                               grad_residual, grad_bn_weight, grad_bn_bias, bn_ws, bn_ws_bytes, stream);
  }
  // the weight gradient on a second stream (TsConvBlockOpts): its operands live in the caller's ring slot, not in ws
  const bool det_ok = grad_kernel && ((int64_t)c_in * c_out) % 4 == 0 && (((uintptr_t)grad_kernel) & 15) == 0 && g_ts_conv_impl != 1;
  const bool side_on = o.wgrad_stream && o.wgrad_stream != stream && o.wgrad_ws && !comm && det_ok;
  // a caller that has queued the weight gradient for another thread must not find it done here as well
  TS_REQUIRE(!(o.wgrad_stream && o.wgrad_deferred) || side_on, TS_ERR_INVALID_ARGUMENT,
             "ts_conv_block_backward: this call cannot defer its weight gradient (SyncBatchNorm communicator, odd C_in * C_out, "
             "misaligned grad_kernel, the scalar cross-check implementation, or the caller's own stream as second stream)");
  float *side_part = nullptr;
  if (side_on) {
    TS_REQUIRE(o.wgrad_slot >= 0 && o.wgrad_slot < 8 && (((uintptr_t)o.wgrad_ws) & 255) == 0 &&
                   o.wgrad_ws_bytes >= ts_conv_block_wgrad_ws_bytes(n_pairs, n_out, c_in, c_out, K, half),
               TS_ERR_INVALID_ARGUMENT, "ts_conv_block_backward: weight-gradient ring slot / scratch");
    TS_TRY(wg_events());
    grad_conv = o.wgrad_ws;
    side_part = (float *)((char *)o.wgrad_ws + blk_align((size_t)n_out * cmax * es));
    // the slot's previous weight gradient (another layer's) must have read its operands before they are overwritten
    // deferred form: until the slot's last weight gradient has been ENQUEUED (by the caller's other thread) its done event says
    // nothing - wait for that on the host (bounded; normally it happened several blocks ago)
    for (int spin = 0; g_wg_owed[o.wgrad_slot].load(std::memory_order_acquire) != 0; ++spin) {
      if (spin > 64) sched_yield();
      if (spin > 20000000) {               // seconds: the promise was abandoned (its backward call failed after making it)
        g_wg_owed[o.wgrad_slot].store(0, std::memory_order_release);
        break;
      }
    }
    // (usually long finished: a query is cheaper than a wait in the stream)
    if (g_wg_used[o.wgrad_slot].load() && hipEventQuery(g_wg_done[o.wgrad_slot]) != hipSuccess) {
      (void)hipGetLastError();             // "not ready" is an answer, not an error for the launch checks that follow
      TS_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, g_wg_done[o.wgrad_slot], 0), "ring wait");
    }
  }
  if (comm) {
    TS_TRY(ts_bn_sync_backward(comm, grad_out, mask, conv_out, mean, invstd, bn_weight, total_dev, n_out, c_out, half, sums,
                               grad_conv, grad_residual, grad_bn_weight, grad_bn_bias, bn_ws, bn_ws_bytes, stream));
  } else if (half) {
    TS_TRY(ts_bn_act_train_backward_f16(grad_out, mask, conv_out, mean, invstd, bn_weight, n_out, c_out, grad_conv,
                                        grad_residual, grad_bn_weight, grad_bn_bias, bn_ws, bn_ws_bytes, stream));
  } else {
    TS_TRY(ts_bn_act_train_backward((const float *)grad_out, mask, (const float *)conv_out, mean, invstd, bn_weight, n_out,
                                    c_out, (float *)grad_conv, (float *)grad_residual, grad_bn_weight, grad_bn_bias, bn_ws,
                                    bn_ws_bytes, stream));
  }
  // Weight gradient first, as partial tiles (deterministic form, common.h): their ordered sum then rides on the launch that
  // finishes the input gradient (pass 2, or the direct class GEMM) - no reduce launch, no fill of grad_kernel, no float
  // atomics.  A block without an input gradient (the first layer) sums with a launch of its own.
  const double es_d = half ? 2 : 4;
  float *part = side_on ? side_part : (float *)(((char *)bn_ws) + blk_align(bn_ws_bytes));
  const bool det = det_ok;
  TsWgradReduce job = {};
  const ts_stream_t main_stream = stream;
  const bool deferred = side_on && o.wgrad_deferred;
  if (side_on) {                         // from here to the end of the weight gradient: the second stream
    TS_CHECK_HIP(hipEventRecord(g_wg_ready[o.wgrad_slot], (hipStream_t)main_stream), "ring record");
    if (deferred) {
      g_wg_owed[o.wgrad_slot].store(1, std::memory_order_release);      // the caller launches it: ts_conv_block_wgrad_side
    } else {
      TS_CHECK_HIP(hipStreamWaitEvent((hipStream_t)o.wgrad_stream, g_wg_ready[o.wgrad_slot], 0), "ring wait");
      stream = o.wgrad_stream;
    }
  }
  if (grad_kernel && !deferred) {
    ProfScope ps(2, stream, (double)n_pairs, c_in, c_out, K, (double)n_feat_rows, es_d, (double)n_out);
    if (det) g_ts_wgrad_part = part;
    int rc;
    if (half)
      rc = ts_conv_wgrad_f16_ex(feat, c_in, grad_conv, c_out, nbmaps, nboffs, K, wgrad_col_a, n_pairs, grad_kernel,
                                det ? 1 : 0, stream);
    else
      rc = ts_conv_wgrad_ex((const float *)feat, c_in, (const float *)grad_conv, c_out, nbmaps, nboffs, K, wgrad_col_a,
                            n_pairs, grad_kernel, det ? 1 : 0, stream);
    g_ts_wgrad_part = nullptr;
    if (rc != TS_OK) return rc;
    if (det) job = TsWgradReduce{part, nboffs, grad_kernel, K, g_ts_wgrad_plan.chunk, (int64_t)c_in * c_out / 4};
  }
  if (side_on && !deferred) {            // the ordered sum follows on the second stream; the slot is free when it has run
    // (blocks without an input gradient sum with the stand-alone kernel's order on either path)
    TS_TRY((grad_feat && n_dgrad_rows > 0 && !o.natural) ? ts_wgrad_reduce_seq(job, stream) : ts_wgrad_reduce(job, stream));
    TS_CHECK_HIP(hipEventRecord(g_wg_done[o.wgrad_slot], (hipStream_t)stream), "ring record");
    g_wg_used[o.wgrad_slot].store(true);
    stream = main_stream;
  }
  const bool ride = det && grad_feat && n_dgrad_rows > 0 && !side_on && !o.natural;     // (a natural call has no pass 2 to ride on)
  const TsClassPlan *cp = (grad_feat && !o.natural && plan_fits(o.dgrad_plan, K, n_dgrad_rows, c_out, c_in, n_pairs, nboffs) &&
                           !(o.dgrad_plan->rows && addend))
                              ? o.dgrad_plan
                              : nullptr;
  // the ordered weight-gradient sum riding on a launch reads its partial tiles and writes grad_kernel: real bytes
  // ... and so is the shortcut's gradient when it is added in the store (one more read of [n_dgrad_rows, c_in])
  const double side_bytes = (ride ? 4.0 * c_in * c_out * ((double)g_ts_wgrad_plan.slots + K) : 0.0) +
                            (addend ? es_d * (double)n_dgrad_rows * c_in : 0.0);
  if (cp) {
    // input gradient on the class plan: gy rows through W_k^T (mirror: W_{K-1-k}^T), the sums of a group of offsets in the
    // accumulators; a direct plan writes grad_feat itself and carries the weight-gradient sum
    const bool fused = plan_finishes(cp, half);
    void *dst = cp->rows ? grad_feat : z;
    const TsClassFinish fin = {cp->pos, n_dgrad_rows, grad_feat, addend};
    {
      ProfScope ps(fused ? 4 : 3, stream, (double)n_pairs, c_out, c_in, K, (double)n_out, es_d,
                   cp->rows ? -(double)n_dgrad_rows : fused ? plan_z_moves(cp, n_dgrad_rows) : plan_z_rows(cp));
      const TsWgradReduce *side = ((cp->rows || fused) && ride) ? &job : nullptr;
      if (half)
        TS_TRY(ts_conv_class_gemm_f16_ex(grad_conv, c_out, weights, K, cp->groups, c_in, cp->src, cp->m_pad, cp->tile_info,
                                         cp->n_tiles, 1, cp->mirror, cp->rows, dst, side, fused ? &fin : nullptr, stream));
      else
        TS_TRY(ts_conv_class_gemm_ex((const float *)grad_conv, c_out, (const float *)weights, K, cp->groups, c_in, cp->src,
                                     cp->m_pad, cp->tile_info, cp->n_tiles, 1, cp->mirror, cp->rows, (float *)dst, side,
                                     fused ? &fin : nullptr, stream));
    }
    if (!cp->rows && !fused) {
      ProfScope ps(1, stream, plan_z_rows(cp), 0, c_in, cp->groups, (double)n_dgrad_rows, es_d, side_bytes);
      if (half)
        TS_TRY(ts_conv_gather_sum_f16_ex(z, c_in, cp->pos, cp->groups, n_dgrad_rows, cp->m_pad, grad_feat, ride ? &job : nullptr,
                                         addend, stream));
      else
        TS_TRY(ts_conv_gather_sum_ex((const float *)z, c_in, cp->pos, cp->groups, n_dgrad_rows, cp->m_pad, (float *)grad_feat,
                                     ride ? &job : nullptr, (const float *)addend, stream));
    }
  } else if (grad_feat && o.natural) {
    // identity rulebook: the rows of the product through W^T are the input gradient's rows
    ProfScope ps(0, stream, (double)n_pairs, c_out, c_in, K, (double)n_out, es_d, 1);
    if (half)
      TS_TRY(ts_conv_pair_gemm_f16(grad_conv, n_out, c_out, weights, K, nbmaps, nboffs, n_pairs, dgrad_gather_col, grad_feat, c_in,
                                   stream));
    else
      TS_TRY(ts_conv_pair_gemm((const float *)grad_conv, n_out, c_out, (const float *)weights, K, 1, nbmaps, nboffs, n_pairs,
                               dgrad_gather_col, (float *)grad_feat, c_in, stream));
  } else if (grad_feat) {
    {
      ProfScope ps(0, stream, (double)n_pairs, c_out, c_in, K, (double)n_out, es_d, 1);
      if (half)
        TS_TRY(ts_conv_pair_gemm_f16(grad_conv, n_out, c_out, weights, K, nbmaps, nboffs, n_pairs, dgrad_gather_col, z, c_in,
                                     stream));
      else {
        planes.arm();
        TS_TRY(ts_conv_pair_gemm((const float *)grad_conv, n_out, c_out, (const float *)weights, K, 1, nbmaps, nboffs,
                                 n_pairs, dgrad_gather_col, (float *)z, c_in, stream));
      }
    }
    {
      ProfScope ps(1, stream, (double)n_pairs, 0, c_in, K, (double)n_dgrad_rows, es_d, side_bytes);
      if (half)
        TS_TRY(ts_conv_gather_sum_f16_ex(z, c_in, pos_dgrad, K, n_dgrad_rows, n_pairs, grad_feat, ride ? &job : nullptr,
                                         addend, stream));
      else
        TS_TRY(ts_conv_gather_sum_ex((const float *)z, c_in, pos_dgrad, K, n_dgrad_rows, n_pairs, (float *)grad_feat,
                                     ride ? &job : nullptr, (const float *)addend, stream));
    }
  }
  if (det && !ride && !side_on) TS_TRY(ts_wgrad_reduce(job, stream));
  return TS_OK;
}

// The weight gradient a ts_conv_block_backward call with opts->wgrad_deferred left out: waits (on side_stream) for the ring slot's
// output gradient, launches the partial tiles and their ordered sum there, marks the slot.  May be called from another host thread
// than the backward call (that is the point: the launches leave the thread that issues the step); the arguments are the backward
// call's own.  chunk_order: the backward call had an input gradient (the ordered sum then keeps the order of the riding form).
extern "C" int ts_conv_block_wgrad_side(const void *feat, int64_t n_feat_rows, int32_t c_in, int32_t K, const int32_t *nbmaps,
                                        const int32_t *nboffs, int64_t n_pairs, int32_t wgrad_col_a, int64_t n_out, int32_t c_out,
                                        int32_t half, float *grad_kernel, int32_t chunk_order, void *wgrad_ws, size_t wgrad_ws_bytes,
                                        int32_t slot, ts_stream_t side_stream) {
  TS_REQUIRE(slot >= 0 && slot < 8 && wgrad_ws && grad_kernel && feat && side_stream &&
                 wgrad_ws_bytes >= ts_conv_block_wgrad_ws_bytes(n_pairs, n_out, c_in, c_out, K, half),
             TS_ERR_INVALID_ARGUMENT, "ts_conv_block_wgrad_side: bad arguments");
  const size_t es = half ? 2 : 4;
  const size_t cmax = (size_t)std::max(c_in, c_out);
  const void *grad_conv = wgrad_ws;
  float *part = (float *)((char *)wgrad_ws + blk_align((size_t)n_out * cmax * es));
  int rc = TS_OK;
  {
    TS_CHECK_HIP(hipStreamWaitEvent((hipStream_t)side_stream, g_wg_ready[slot], 0), "ring wait");
    ProfScope ps(2, side_stream, (double)n_pairs, c_in, c_out, K, (double)n_feat_rows, half ? 2.0 : 4.0, (double)n_out);
    g_ts_wgrad_part = part;
    if (half)
      rc = ts_conv_wgrad_f16_ex(feat, c_in, grad_conv, c_out, nbmaps, nboffs, K, wgrad_col_a, n_pairs, grad_kernel, 1, side_stream);
    else
      rc = ts_conv_wgrad_ex((const float *)feat, c_in, (const float *)grad_conv, c_out, nbmaps, nboffs, K, wgrad_col_a, n_pairs,
                            grad_kernel, 1, side_stream);
    g_ts_wgrad_part = nullptr;
  }
  if (rc == TS_OK) {
    const TsWgradReduce job = {part, nboffs, grad_kernel, K, g_ts_wgrad_plan.chunk, (int64_t)c_in * c_out / 4};
    // (kind 5: the ordered sum of the partial tiles as a launch of its own; last field = its bytes, `rows` = 1 for the chunk order
    // of the riding form, wgrad_reduce_seq_kernel)
    ProfScope ps(5, side_stream, 0, c_in, c_out, K, chunk_order ? 1.0 : 0.0, 4, 4.0 * c_in * c_out * ((double)g_ts_wgrad_plan.slots + K));
    rc = chunk_order ? ts_wgrad_reduce_seq(job, side_stream) : ts_wgrad_reduce(job, side_stream);
  }
  if (hipEventRecord(g_wg_done[slot], (hipStream_t)side_stream) != hipSuccess && rc == TS_OK) rc = TS_ERR_LAUNCH_FAILED;
  g_wg_used[slot].store(true);
  g_wg_owed[slot].store(0, std::memory_order_release);
  return rc;
}

extern "C" int ts_set_device(int32_t device) {
  TS_CHECK_HIP(hipSetDevice(device), "ts_set_device");
  return TS_OK;
}

// ---- column concatenation / slicing (torchsparse.cat, operators.py:10-17, and its gradient) for the stage programs -------------
// One thread per VEC-byte piece of a destination row; rows of the decoder's concatenations are 128-1536 bytes wide.
template <typename V>
__global__ void cat_cols_kernel(const V *__restrict__ a, int wa, const V *__restrict__ b, int wb, int64_t rows, V *__restrict__ dst) {
  const int w = wa + wb;
  const int64_t n = rows * w;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / w;
    const int c = (int)(i - r * w);
    dst[i] = c < wa ? a[r * wa + c] : b[r * wb + (c - wa)];
  }
}
template <typename V>
__global__ void copy_cols_kernel(const V *__restrict__ src, int64_t src_pitch, int w, int64_t rows, V *__restrict__ dst, int64_t dst_pitch) {
  const int64_t n = rows * w;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / w;
    const int c = (int)(i - r * w);
    dst[r * dst_pitch + c] = src[r * src_pitch + c];
  }
}
static inline int cols_vec(uintptr_t bits) { return (bits & 15) == 0 ? 16 : (bits & 3) == 0 ? 4 : (bits & 1) == 0 ? 2 : 1; }
static inline unsigned cols_grid(int64_t n) { return (unsigned)std::min<int64_t>(std::max<int64_t>(ts_cdiv(n, 256), 1), 256 * 16); }

extern "C" int ts_cat_cols(const void *a, int64_t a_bytes, const void *b, int64_t b_bytes, int64_t rows, void *dst, ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(a && b && dst && a_bytes > 0 && b_bytes > 0 && rows >= 0 && a_bytes + b_bytes < (1 << 30), TS_ERR_INVALID_ARGUMENT,
             "ts_cat_cols: bad arguments");
  if (rows == 0) return TS_OK;
  const int v = cols_vec((uintptr_t)a | (uintptr_t)b | (uintptr_t)dst | (uintptr_t)a_bytes | (uintptr_t)b_bytes);
  const int64_t n = rows * ((a_bytes + b_bytes) / v);
  if (v == 16)
    cat_cols_kernel<uint4><<<cols_grid(n), 256, 0, stream>>>((const uint4 *)a, (int)(a_bytes / 16), (const uint4 *)b, (int)(b_bytes / 16), rows, (uint4 *)dst);
  else if (v == 4)
    cat_cols_kernel<uint32_t><<<cols_grid(n), 256, 0, stream>>>((const uint32_t *)a, (int)(a_bytes / 4), (const uint32_t *)b, (int)(b_bytes / 4), rows, (uint32_t *)dst);
  else if (v == 2)
    cat_cols_kernel<uint16_t><<<cols_grid(n), 256, 0, stream>>>((const uint16_t *)a, (int)(a_bytes / 2), (const uint16_t *)b, (int)(b_bytes / 2), rows, (uint16_t *)dst);
  else
    cat_cols_kernel<uint8_t><<<cols_grid(n), 256, 0, stream>>>((const uint8_t *)a, (int)a_bytes, (const uint8_t *)b, (int)b_bytes, rows, (uint8_t *)dst);
  TS_CHECK_LAUNCH("ts_cat_cols");
  return TS_OK;
}

extern "C" int ts_copy_cols(const void *src, int64_t src_pitch, int64_t offset, int64_t width, int64_t rows, void *dst, int64_t dst_pitch,
                            ts_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TS_REQUIRE(src && dst && width > 0 && offset >= 0 && offset + width <= src_pitch && width <= dst_pitch && rows >= 0 && width < (1 << 30),
             TS_ERR_INVALID_ARGUMENT, "ts_copy_cols: bad arguments");
  if (rows == 0) return TS_OK;
  const char *s = (const char *)src + offset;
  const int v = cols_vec((uintptr_t)s | (uintptr_t)dst | (uintptr_t)src_pitch | (uintptr_t)dst_pitch | (uintptr_t)width);
  const int64_t n = rows * (width / v);
  if (v == 16)
    copy_cols_kernel<uint4><<<cols_grid(n), 256, 0, stream>>>((const uint4 *)s, src_pitch / 16, (int)(width / 16), rows, (uint4 *)dst, dst_pitch / 16);
  else if (v == 4)
    copy_cols_kernel<uint32_t><<<cols_grid(n), 256, 0, stream>>>((const uint32_t *)s, src_pitch / 4, (int)(width / 4), rows, (uint32_t *)dst, dst_pitch / 4);
  else if (v == 2)
    copy_cols_kernel<uint16_t><<<cols_grid(n), 256, 0, stream>>>((const uint16_t *)s, src_pitch / 2, (int)(width / 2), rows, (uint16_t *)dst, dst_pitch / 2);
  else
    copy_cols_kernel<uint8_t><<<cols_grid(n), 256, 0, stream>>>((const uint8_t *)s, src_pitch, (int)width, rows, (uint8_t *)dst, dst_pitch);
  TS_CHECK_LAUNCH("ts_copy_cols");
  return TS_OK;
}
