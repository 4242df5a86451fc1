// Native autograd node for conv -> BatchNorm(train) [+ residual] [-> ReLU] (the unit of the MinkUNet family,
// R/pcseg/model/segmentor/voxel/minkunet/minkunet.py:31-129).  Same backend calls as the Python Function
// taseg_amd.torchsparse.nn.functional._ConvBlock (ts_conv_block_forward / ts_conv_block_backward of libtaseg_hip.so,
// bound with dlopen - this file contains no device code); what it removes is interpreter work: the Python node costs
// ~53 us (forward) + ~85 us (backward) of host time per block, 55 blocks per step, about half of it tensor
// bookkeeping and argument marshalling around ~25 us of launches.  PyTorch supplies memory, streams and the autograd
// graph only.
#include <dlfcn.h>
#include <c10/core/Stream.h>
#include <torch/csrc/autograd/engine.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>
#include <torch/extension.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>

#include "taseg_hip.h"

namespace {

struct Api {
  void *handle = nullptr;
  decltype(&ts_conv_block_workspace_bytes) workspace_bytes = nullptr;
  decltype(&ts_conv_block_forward) forward = nullptr;
  decltype(&ts_conv_block_backward) backward = nullptr;
  decltype(&ts_conv_block_eval) eval = nullptr;
  decltype(&ts_conv_block_wgrad_ws_bytes) wgrad_ws_bytes = nullptr;
  decltype(&ts_stream_join) stream_join = nullptr;
  decltype(&ts_conv_block_wgrad_side) wgrad_side = nullptr;
  decltype(&ts_set_device) set_device = nullptr;
  decltype(&ts_last_error) last_error = nullptr;
  decltype(&ts_downsample_workspace_bytes) downsample_ws = nullptr;
  decltype(&ts_downsample) downsample = nullptr;
  decltype(&ts_build_kmap_workspace_bytes) build_kmap_ws = nullptr;
  decltype(&ts_build_kmap) build_kmap = nullptr;
  decltype(&ts_build_kmap_sym) build_kmap_sym = nullptr;
  decltype(&ts_trilinear_workspace_bytes) trilinear_ws = nullptr;
  decltype(&ts_trilinear_map) trilinear_map = nullptr;
  decltype(&ts_devox_order_workspace_bytes) devox_order_ws = nullptr;
  decltype(&ts_devox_order) devox_order = nullptr;
  decltype(&ts_cat_cols) cat_cols = nullptr;
  decltype(&ts_copy_cols) copy_cols = nullptr;
  decltype(&ts_get_option) get_option = nullptr;
} api;

void check(int rc, const char *what) {
  TORCH_CHECK(rc == 0, what, " failed (code ", rc, "): ", api.last_error ? api.last_error() : "?");
}

// stream-ordered scratch, one growing buffer per (device, stream) like taseg_amd._lib.workspace
std::mutex ws_mutex;
std::unordered_map<int64_t, at::Tensor> ws_pool;

at::Tensor workspace(size_t nbytes, const at::Tensor &like, int64_t stream) {
  std::lock_guard<std::mutex> lock(ws_mutex);
  const int64_t key = stream * 64 + like.get_device();
  auto it = ws_pool.find(key);
  if (it == ws_pool.end() || (size_t)it->second.numel() < nbytes) {
    const int64_t cap = std::max<int64_t>((int64_t)(nbytes * 1.5), 1 << 20);
    ws_pool[key] = at::empty({cap}, like.options().dtype(at::kByte));
    it = ws_pool.find(key);
  }
  return it->second;
}

// Weight gradients on a second stream (TsConvBlockOpts.wgrad_stream, csrc/block.hip): per device one stream from torch's pool and
// a ring of scratch buffers, one per slot, that hold the output gradient and the partial tiles of the layer the slot was last given
// to; the backend orders the ring with events.  The gradients are complete on the second stream: a callback queued on the autograd
// engine joins it into the node's stream when the backward pass ends (every consumer of p.grad comes after that).  Off unless
// set_wgrad_stream(true) (taseg_amd.fast: TASEG_WGRAD_STREAM; never with gradient buckets, whose hooks read p.grad mid-pass).
constexpr int WG_SLOTS = 8;
struct WgSide {
  bool on = false;
  int64_t raw = 0;                       // hipStream_t of the second stream (torch.cuda.Stream.cuda_stream)
  int64_t stream_id = 0, device_index = 0, device_type = 0;   // ... and what c10::Stream::unpack3 rebuilds it from (record_stream)
  at::Tensor ring[WG_SLOTS];
  int next = 0;
};
std::mutex wg_mutex;
WgSide wg_side;
std::atomic<bool> wg_join_queued{false};

// The launches of the second stream come from a thread of their own: the node's thread only marks the point where a block's
// output gradient exists (one event record inside ts_conv_block_backward) and queues the rest - wait for that event, partial
// tiles, ordered sum, the slot's done event: ts_conv_block_wgrad_side - here.  A job keeps the tensors it reads and writes alive
// until it has been enqueued (record_stream covers them from then on).
struct WgWorker {
  std::mutex m;
  std::condition_variable cv, idle;
  std::deque<std::function<void()>> q;
  int busy = 0;
  bool stop = false, started = false;
  std::string error;
  std::thread th;
  void start(int device) {
    if (started) return;
    started = true;
    th = std::thread([this, device] {
      if (api.set_device) api.set_device(device);
      for (;;) {
        std::function<void()> job;
        {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [this] { return stop || !q.empty(); });
          if (q.empty()) return;
          job = std::move(q.front());
          q.pop_front();
          busy = 1;
        }
        try {
          job();
        } catch (const std::exception &e) {
          std::lock_guard<std::mutex> lk(m);
          if (error.empty()) error = e.what();
        }
        job = nullptr;                   // (drops the tensors it held)
        {
          std::lock_guard<std::mutex> lk(m);
          busy = 0;
          if (q.empty()) idle.notify_all();
        }
      }
    });
  }
  void push(std::function<void()> job) {
    {
      std::lock_guard<std::mutex> lk(m);
      q.push_back(std::move(job));
    }
    cv.notify_one();
  }
  void drain() {                         // every queued launch has been enqueued on the second stream
    std::unique_lock<std::mutex> lk(m);
    idle.wait(lk, [this] { return q.empty() && busy == 0; });
    if (!error.empty()) {
      const std::string e = error;
      error.clear();
      TORCH_CHECK(false, "conv_block: weight gradient on the second stream: ", e);
    }
  }
  ~WgWorker() {
    if (!started) return;
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};
WgWorker wg_worker;

// raw == 0 switches the second stream off
void set_wgrad_stream(int64_t raw, int64_t stream_id, int64_t device_index, int64_t device_type) {
  if (wg_worker.started) wg_worker.drain();
  std::lock_guard<std::mutex> lock(wg_mutex);
  wg_side.on = raw != 0;
  wg_side.raw = raw;
  wg_side.stream_id = stream_id;
  wg_side.device_index = device_index;
  wg_side.device_type = device_type;
  if (wg_side.on) wg_worker.start((int)device_index);
}

// The caller is about to read weight gradients mid-pass (a gradient bucket whose all-reduce is launched while the backward pass is
// still running, taseg_amd/parallel.py): everything queued for the second stream so far is enqueued, and `main_raw` waits for it.
void join_wgrad_stream(int64_t main_raw) {
  if (!wg_side.on || !wg_worker.started) return;
  // (called from a Python hook: the worker thread may need the interpreter lock to let go of a tensor a finished job held)
  py::gil_scoped_release nogil;
  wg_worker.drain();
  check(api.stream_join((ts_stream_t)main_raw, (ts_stream_t)wg_side.raw), "ts_stream_join");
}

// SyncBatchNorm over torch.distributed: process groups registered from Python (register_group), addressed by index.  The
// node then splits its backend call around ProcessGroup::allreduce on the group's own communicator (csrc/block.hip: comm
// sentinels 1 / 2) - the C++ counterpart of functional._ConvBlock's split path, without the interpreter in between.
std::mutex group_mutex;
std::vector<c10::intrusive_ptr<c10d::ProcessGroup>> groups;

void group_sum(int64_t id, at::Tensor buf) {
  c10::intrusive_ptr<c10d::ProcessGroup> pg;
  {
    std::lock_guard<std::mutex> lock(group_mutex);
    TORCH_CHECK(id >= 0 && id < (int64_t)groups.size(), "conv_block: unknown process group id ", id);
    pg = groups[id];
  }
  std::vector<at::Tensor> v{std::move(buf)};
  c10d::AllreduceOptions opts;
  opts.asyncOp = false;                // as dist.all_reduce(async_op=False): ProcessGroupNCCL issues it on the CURRENT stream
  auto work = pg->allreduce(v, opts);
  if (work) work->wait();              // (ProcessGroupNCCL hands back no work object for a current-stream collective)
}

void *const COMM_PRE = (void *)1, *const COMM_POST = (void *)2;

// host-time accounting of the node (diagnostic, tools/block_host_probe.py): time inside the backend calls (= issuing the launches)
// against the whole of forward() / backward()
struct HostClock {
  std::atomic<int64_t> ns_fwd{0}, ns_fwd_api{0}, ns_bwd{0}, ns_bwd_api{0}, n_fwd{0}, n_bwd{0};
} host_clock;
inline int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

inline void *ptr(const at::Tensor &t) { return t.defined() ? t.data_ptr() : nullptr; }
inline void *optr(const c10::optional<at::Tensor> &t) { return (t.has_value() && t->defined()) ? t->data_ptr() : nullptr; }

// A class plan of the block's kernel map as it crosses from Python: tensors (src, tile_info, n_tiles, pos | rows) and
// meta (n, m_pad, z_rows, K, groups, mirror, direct); empty vectors = no plan.  map_id = the map's nboffs (plans are attributes
// of the KernelMap object they were built from, taseg_amd/torchsparse/nn/functional.py).
struct PlanRef {
  TsClassPlan plan;
  bool live = false;
  PlanRef(const std::vector<at::Tensor> &t, const std::vector<int64_t> &m, const at::Tensor &nboffs) {
    if (t.size() != 4 || m.size() != 7) return;
    const bool direct = m[6] != 0;
    plan.src = (const int32_t *)t[0].data_ptr();
    plan.tile_info = (const int32_t *)t[1].data_ptr();
    plan.n_tiles = (const int32_t *)t[2].data_ptr();
    plan.pos = direct ? nullptr : (const int32_t *)t[3].data_ptr();
    plan.rows = direct ? (const int32_t *)t[3].data_ptr() : nullptr;
    plan.n = m[0];
    plan.m_pad = m[1];
    plan.z_rows = m[2];
    plan.K = (int32_t)m[3];
    plan.groups = (int32_t)m[4];
    plan.mirror = (int32_t)m[5];
    plan.map_id = nboffs.data_ptr();
    live = true;
  }
  const TsClassPlan *get() const { return live ? &plan : nullptr; }
};

class ConvBlock : public torch::autograd::Function<ConvBlock> {
 public:
  static torch::autograd::variable_list forward(torch::autograd::AutogradContext *ctx, const at::Tensor &feats, const at::Tensor &weight,
                            const c10::optional<at::Tensor> &residual, const at::Tensor &bn_weight,
                            const at::Tensor &bn_bias, const at::Tensor &nbmaps, const at::Tensor &nboffs, int64_t total,
                            const at::Tensor &pos_out, const at::Tensor &pos_in, int64_t n_in, int64_t n_out,
                            bool transposed, const c10::optional<at::Tensor> &running_mean,
                            const c10::optional<at::Tensor> &running_var, const c10::optional<at::Tensor> &nbt,
                            double momentum, double eps, bool relu, int64_t comm, bool half, int64_t stream,
                            const c10::optional<at::Tensor> &planes, bool passthrough,
                            const c10::optional<at::Tensor> &grad_dest, int64_t group_id,
                            const c10::optional<at::Tensor> &pf0, const c10::optional<at::Tensor> &pf1,
                            const c10::optional<at::Tensor> &pf2, const c10::optional<at::Tensor> &pf3,
                            const std::vector<int64_t> &plan_f_meta, const c10::optional<at::Tensor> &pd0,
                            const c10::optional<at::Tensor> &pd1, const c10::optional<at::Tensor> &pd2,
                            const c10::optional<at::Tensor> &pd3, const std::vector<int64_t> &plan_d_meta, bool wgrad_side_ok,
                            bool natural) {
    const int64_t t_in = now_ns();
    // a backward pass that died mid-way leaves its join behind: nothing of the second stream outlives the next forward call
    if (wg_join_queued.exchange(false)) {
      wg_worker.drain();
      check(api.stream_join((ts_stream_t)stream, (ts_stream_t)wg_side.raw), "ts_stream_join");
    }
    // (the plans cross the autograd boundary as four optional tensors + their meta each: a fixed number of node inputs)
    std::vector<at::Tensor> plan_f, plan_d;
    if (pf0.has_value() && pf0->defined()) plan_f = {*pf0, *pf1, *pf2, *pf3};
    if (pd0.has_value() && pd0->defined()) plan_d = {*pd0, *pd1, *pd2, *pd3};
    // a 1x1x1 convolution's weight is [C_in, C_out] (conv.py:135-140); it comes with the identity rulebook and natural = true
    const bool flat = weight.dim() == 2;
    TORCH_CHECK(flat == natural, "conv_block: a [C_in, C_out] weight goes with the identity rulebook (natural) and vice versa");
    const int64_t k = flat ? 1 : weight.size(0), c_in = weight.size(flat ? 0 : 1), c_out = weight.size(flat ? 1 : 2);
    const bool split = comm == 0 && group_id >= 0;
    const auto dt = half ? at::kHalf : at::kFloat;
    const int64_t rows = transposed ? n_in : n_out;
    const at::Tensor &table = transposed ? pos_in : pos_out;
    at::Tensor x = feats.contiguous().to(dt);
    at::Tensor w32 = weight.detach().contiguous().to(at::kFloat);
    at::Tensor res;
    if (residual.has_value() && residual->defined()) res = residual->contiguous().to(dt);
    const auto opts = x.options();
    at::Tensor conv_out = at::empty({rows, c_out}, opts), out = at::empty({rows, c_out}, opts);
    at::Tensor stats = at::empty({2, c_out}, opts.dtype(at::kFloat));
    at::Tensor mask, w16, pack;
    if (relu) mask = at::empty({rows * (c_out / (half ? 8 : 4))}, opts.dtype(at::kByte));
    const bool own_weight = w32.data_ptr() == weight.data_ptr();
    const bool have = planes.has_value() && planes->defined() && own_weight;
    // half storage: `planes` = the kept half copy of the weight (planes.half_for): w16 IS that tensor and the call casts nothing
    const bool kept16 = half && have && planes->scalar_type() == at::kHalf && planes->numel() == k * c_in * c_out;
    if (half) w16 = kept16 ? *planes : at::empty({k, c_in, c_out}, opts.dtype(at::kHalf));
    if (comm || split) pack = at::empty({2 * c_out + 1}, opts.dtype(at::kDouble));
    const size_t nb = api.workspace_bytes(total, std::max(n_in, n_out), (int32_t)c_in, (int32_t)c_out, (int32_t)k, half ? 1 : 0);
    at::Tensor ws = workspace(nb, x, stream);
    float *st = stats.data_ptr<float>();
    at::Tensor pl;                       // pre-split planes of the weight (taseg_amd/planes.py); fp32 blocks only
    if (!half && have && planes->scalar_type() == at::kShort) pl = *planes;
    // everything the call may use beyond the rulebook, explicitly (TsConvBlockOpts): the class plans of this block's kernel map
    // (csrc/conv_class.hip), the pre-split planes / the kept half copy of the weight
    const PlanRef pf(plan_f, plan_f_meta, nboffs);
    TsConvBlockOpts bopts = {pf.get(), nullptr, pl.defined() ? pl.data_ptr() : nullptr, kept16 ? 1 : 0, nullptr, nullptr, nullptr, 0, 0, 0,
                             natural ? 1 : 0};
    auto call = [&](void *c) {
      check(api.forward(x.data_ptr(), x.size(0), (int32_t)c_in, w32.data_ptr<float>(), (int32_t)k,
                        (const int32_t *)nbmaps.data_ptr(), (const int32_t *)nboffs.data_ptr(), total, transposed ? 1 : 0,
                        (const int32_t *)table.data_ptr(), rows, (int32_t)c_out, ptr(res), (const float *)bn_weight.data_ptr(),
                        (const float *)bn_bias.data_ptr(), (float *)optr(running_mean), (float *)optr(running_var),
                        (int64_t *)optr(nbt), (float)eps, (float)momentum, relu ? 1 : 0, half ? 1 : 0, c,
                        (double *)ptr(pack), conv_out.data_ptr(), st, st + c_out, out.data_ptr(), (uint8_t *)ptr(mask), ptr(w16),
                        &bopts, ws.data_ptr(), (size_t)ws.numel(), (ts_stream_t)stream),
            "ts_conv_block_forward");
    };
    const int64_t t_api = now_ns();
    if (split) {
      call(COMM_PRE);                    // convolution + this rank's sums (the planes hint is consumed here)
      group_sum(group_id, pack);
      call(COMM_POST);                   // statistics over all ranks + elementwise pass
    } else {
      call((void *)comm);
    }
    host_clock.ns_fwd_api += now_ns() - t_api;
    // (the running statistics were written through raw pointers: whoever caches something derived from them keys on the version)
    if (running_mean.has_value() && running_mean->defined()) running_mean->unsafeGetTensorImpl()->bump_version();
    if (running_var.has_value() && running_var->defined()) running_var->unsafeGetTensorImpl()->bump_version();
    ctx->save_for_backward({x, half ? w16 : w32, conv_out, stats, mask, bn_weight, nbmaps, nboffs, pos_out, pos_in, pack});
    ctx->saved_data["planes"] = pl;      // not a graph tensor: refreshed in place when the optimizer has stepped
    ctx->saved_data["plan_d"] = c10::List<at::Tensor>(plan_d);        // the input gradient's plan (tensors + meta)
    ctx->saved_data["plan_d_meta"] = c10::List<int64_t>(plan_d_meta);
    // the weight gradient may leave for the second stream only if autograd will ADOPT it as p.grad (p.grad undefined now): an
    // accumulation into an existing p.grad reads it on this stream, at once
    ctx->saved_data["wgrad_side_ok"] = wgrad_side_ok;
    // where the weight gradient is wanted (a gradient bucket's view, taseg_amd/parallel.py), if anywhere
    ctx->saved_data["grad_dest"] = (grad_dest.has_value() && grad_dest->defined()) ? *grad_dest : at::Tensor();
    ctx->saved_data["natural"] = natural;
    ctx->saved_data["total"] = total;
    ctx->saved_data["n_in"] = n_in;
    ctx->saved_data["n_out"] = n_out;
    ctx->saved_data["transposed"] = transposed;
    ctx->saved_data["half"] = half;
    ctx->saved_data["comm"] = comm;
    ctx->saved_data["group_id"] = split ? group_id : (int64_t)-1;
    ctx->saved_data["stream"] = stream;
    ctx->saved_data["has_res"] = res.defined();
    ctx->saved_data["in_dtype"] = (int64_t)feats.scalar_type();
    ctx->saved_data["res_dtype"] = (int64_t)(res.defined() ? residual->scalar_type() : at::kFloat);
    host_clock.ns_fwd += now_ns() - t_in;
    host_clock.n_fwd += 1;
    // passthrough: the input leaves the node a second time (autograd aliases it); whatever consumes THAT tensor - the
    // shortcut of a residual block - sends its gradient back into this node, where it joins the input gradient's store
    if (passthrough) return {out, feats};
    return {out};
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx,
                                                 torch::autograd::variable_list grads) {
    const int64_t t_in = now_ns();
    const auto saved = ctx->get_saved_variables();
    const at::Tensor &x = saved[0], &w = saved[1], &conv_out = saved[2], &stats = saved[3], &mask = saved[4],
                     &bn_weight = saved[5], &nbmaps = saved[6], &nboffs = saved[7], &pos_out = saved[8], &pos_in = saved[9],
                     &pack = saved[10];
    const int64_t total = ctx->saved_data["total"].toInt(), n_in = ctx->saved_data["n_in"].toInt(),
                  n_out = ctx->saved_data["n_out"].toInt(), comm = ctx->saved_data["comm"].toInt();
    const bool transposed = ctx->saved_data["transposed"].toBool(), half = ctx->saved_data["half"].toBool(),
               has_res = ctx->saved_data["has_res"].toBool();
    const auto in_dtype = (at::ScalarType)ctx->saved_data["in_dtype"].toInt();
    const auto res_dtype = (at::ScalarType)ctx->saved_data["res_dtype"].toInt();
    const bool natural = ctx->saved_data["natural"].toBool();
    const bool flat = w.dim() == 2;                      // (the fp32 weight of a 1x1x1 convolution; its half copy is [1, C_in, C_out])
    const int64_t k = flat ? 1 : w.size(0), c_in = w.size(flat ? 0 : 1), c_out = w.size(flat ? 1 : 2), rows = conv_out.size(0);
    const int64_t stream = ctx->saved_data["stream"].toInt();   // the engine runs a node on its forward stream
    TORCH_CHECK(grads[0].defined(), "conv_block: the block's output received no gradient");
    at::Tensor g = grads[0].contiguous().to(conv_out.scalar_type());
    const at::Tensor &table = transposed ? pos_out : pos_in;
    const int64_t drows = transposed ? n_out : n_in;
    const auto opts = conv_out.options();
    at::Tensor grad_feat, grad_w, grad_res, sums;
    if (ctx->needs_input_grad(0)) grad_feat = at::empty({drows, c_in}, opts);
    if (ctx->needs_input_grad(1)) {
      const at::Tensor dest = ctx->saved_data["grad_dest"].toTensor();
      const std::vector<int64_t> wshape = natural ? std::vector<int64_t>{c_in, c_out} : std::vector<int64_t>{k, c_in, c_out};
      if (dest.defined() && dest.scalar_type() == at::kFloat && dest.is_contiguous() && dest.sizes().vec() == wshape &&
          dest.device() == conv_out.device())
        grad_w = dest.alias();             // a fresh alias of the bucket slot: autograd adopts it as p.grad, no copy
      else
        grad_w = at::empty(wshape, opts.dtype(at::kFloat));
    }
    if (has_res && ctx->needs_input_grad(2)) grad_res = at::empty_like(conv_out);
    at::Tensor gwb = at::empty({2, c_out}, opts.dtype(at::kFloat));
    const int64_t group_id = ctx->saved_data["group_id"].toInt();
    const bool split = group_id >= 0;
    if (comm || split) sums = at::empty({2, c_out}, opts.dtype(at::kDouble));
    const size_t nb = api.workspace_bytes(total, std::max(n_in, n_out), (int32_t)c_in, (int32_t)c_out, (int32_t)k, half ? 1 : 0);
    at::Tensor ws = workspace(nb, x, stream);
    const float *st = stats.data_ptr<float>();
    float *gw = gwb.data_ptr<float>();
    const at::Tensor pl = ctx->saved_data["planes"].toTensor();
    const std::vector<at::Tensor> plan_d = ctx->saved_data["plan_d"].toTensorVector();
    const std::vector<int64_t> plan_d_meta = ctx->saved_data["plan_d_meta"].toIntVector();
    const PlanRef pd(grad_feat.defined() ? plan_d : std::vector<at::Tensor>(), plan_d_meta, nboffs);
    TsConvBlockOpts bopts = {nullptr, pd.get(), (pl.defined() && !half) ? pl.data_ptr() : nullptr, 0, nullptr, nullptr, nullptr, 0, 0, 0,
                             natural ? 1 : 0};
    std::function<void()> side_job;
    if (wg_side.on && ctx->saved_data["wgrad_side_ok"].toBool() && grad_w.defined() && !comm && !split &&
        x.get_device() == wg_side.device_index && stream != wg_side.raw && (c_in * c_out) % 4 == 0 &&
        (((uintptr_t)grad_w.data_ptr()) & 15) == 0) {
      std::lock_guard<std::mutex> lock(wg_mutex);
      WgSide &sd = wg_side;
      const size_t need = api.wgrad_ws_bytes(total, rows, (int32_t)c_in, (int32_t)c_out, (int32_t)k, half ? 1 : 0);
      const int slot = sd.next;
      sd.next = (sd.next + 1) % WG_SLOTS;
      if (!sd.ring[slot].defined() || (size_t)sd.ring[slot].numel() < need) {
        wg_worker.drain();               // (a queued job may still point into the old buffer)
        sd.ring[slot] = at::empty({(int64_t)(need * 1.25) + 256}, x.options().dtype(at::kByte));
        sd.ring[slot].record_stream(c10::Stream::unpack3(sd.stream_id, (c10::DeviceIndex)sd.device_index, (c10::DeviceType)sd.device_type));
      }
      bopts.wgrad_stream = (ts_stream_t)sd.raw;
      bopts.wgrad_ws = sd.ring[slot].data_ptr();
      bopts.wgrad_ws_bytes = (size_t)sd.ring[slot].numel();
      bopts.wgrad_slot = slot;
      bopts.wgrad_deferred = 1;
      // the second stream reads x and writes grad_w: the allocator must not hand their memory out again before it has
      const c10::Stream side = c10::Stream::unpack3(sd.stream_id, (c10::DeviceIndex)sd.device_index, (c10::DeviceType)sd.device_type);
      x.record_stream(side);
      grad_w.record_stream(side);
      {
        // (grad_w by address only: one more owner and autograd's AccumulateGrad would copy the gradient - before it exists -
        // instead of adopting the tensor as p.grad; the memory lives on as p.grad or as the bucket slot it aliases)
        const at::Tensor xk = x, nbk = nbmaps, nok = nboffs, ringk = sd.ring[slot];
        float *const gw_ptr = (float *)grad_w.data_ptr();
        const ts_stream_t side_raw = bopts.wgrad_stream;
        const int64_t n_feat = x.size(0);
        const int32_t col_a = transposed ? 1 : 0, chunk_order = (grad_feat.defined() && drows > 0 && !natural) ? 1 : 0;
        side_job = [=]() {
          check(api.wgrad_side(xk.data_ptr(), n_feat, (int32_t)c_in, (int32_t)k, (const int32_t *)nbk.data_ptr(),
                               (const int32_t *)nok.data_ptr(), total, col_a, rows, (int32_t)c_out, half ? 1 : 0,
                               gw_ptr, chunk_order, ringk.data_ptr(), (size_t)ringk.numel(), slot, side_raw),
                "ts_conv_block_wgrad_side");
        };
      }
      if (!wg_join_queued.exchange(true)) {
        const ts_stream_t side_raw = bopts.wgrad_stream;
        const int64_t main_raw = stream;
        torch::autograd::Engine::get_default_engine().queue_callback([side_raw, main_raw]() {
          wg_join_queued = false;
          wg_worker.drain();
          check(api.stream_join((ts_stream_t)main_raw, side_raw), "ts_stream_join");
        });
      }
    }
    auto call = [&](void *c) {
      check(api.backward(g.data_ptr(), (const uint8_t *)ptr(mask), conv_out.data_ptr(), st, st + c_out,
                         (const float *)bn_weight.data_ptr(), pack.defined() ? pack.data_ptr<double>() + 2 * c_out : nullptr,
                         c, (double *)ptr(sums), rows, (int32_t)c_out, half ? 1 : 0, x.data_ptr(), x.size(0),
                         (int32_t)c_in, w.data_ptr(), (int32_t)k, (const int32_t *)nbmaps.data_ptr(),
                         (const int32_t *)nboffs.data_ptr(), total, transposed ? 0 : 1, (const int32_t *)table.data_ptr(), drows,
                         transposed ? 1 : 0, ptr(grad_feat), ptr(grad_res), (float *)ptr(grad_w), gw, gw + c_out, &bopts,
                         ws.data_ptr(), (size_t)ws.numel(), (ts_stream_t)stream),
            "ts_conv_block_backward");
    };
    if (split) {
      call(COMM_PRE);                    // this rank's sums of the BatchNorm backward
      group_sum(group_id, sums.view({-1}));
    }
    at::Tensor addend;                   // gradient of the passed-through input, in grad_feat's storage type
    if (grads.size() > 1 && grads[1].defined() && grad_feat.defined()) {
      TORCH_CHECK(!natural, "conv_block: a 1x1x1 block does not pass its input through");
      addend = grads[1].contiguous().to(conv_out.scalar_type());
      TORCH_CHECK(addend.sizes() == grad_feat.sizes(), "conv_block: pass-through gradient has the wrong shape");
      bopts.addend = addend.data_ptr();
    }
    const int64_t t_api = now_ns();
    call(split ? COMM_POST : (void *)comm);
    host_clock.ns_bwd_api += now_ns() - t_api;
    if (side_job) wg_worker.push(std::move(side_job));      // (the call has recorded the slot's ready event)
    if (grad_feat.defined() && grad_feat.scalar_type() != in_dtype) grad_feat = grad_feat.to(in_dtype);
    if (grad_res.defined() && grad_res.scalar_type() != res_dtype) grad_res = grad_res.to(res_dtype);
    at::Tensor none;
    at::Tensor gbw = gwb[0], gbb = gwb[1];
    host_clock.ns_bwd += now_ns() - t_in;
    host_clock.n_bwd += 1;
    return {grad_feat, grad_w, grad_res, gbw, gbb, none, none, none, none, none, none, none, none,
            none, none, none, none, none, none, none, none, none, none, none, none, none, none, none, none, none,
            none, none, none, none, none, none, none, none};
  }
};

#include "stage_program.h"

}  // namespace

void load_backend(const std::string &libpath) {
  if (api.handle) return;
  void *h = dlopen(libpath.c_str(), RTLD_NOW | RTLD_GLOBAL);
  TORCH_CHECK(h, "dlopen(", libpath, "): ", dlerror());
  api.workspace_bytes = (decltype(api.workspace_bytes))dlsym(h, "ts_conv_block_workspace_bytes");
  api.forward = (decltype(api.forward))dlsym(h, "ts_conv_block_forward");
  api.backward = (decltype(api.backward))dlsym(h, "ts_conv_block_backward");
  api.eval = (decltype(api.eval))dlsym(h, "ts_conv_block_eval");
  api.last_error = (decltype(api.last_error))dlsym(h, "ts_last_error");
  TORCH_CHECK(api.workspace_bytes && api.forward && api.backward && api.eval, "libtaseg_hip.so lacks the ts_conv_block_* entry points");
#define TS_BIND(field, sym)                                      \
  api.field = (decltype(api.field))dlsym(h, sym);                \
  TORCH_CHECK(api.field, "libtaseg_hip.so lacks ", sym)
  TS_BIND(wgrad_ws_bytes, "ts_conv_block_wgrad_ws_bytes");
  TS_BIND(stream_join, "ts_stream_join");
  TS_BIND(wgrad_side, "ts_conv_block_wgrad_side");
  TS_BIND(set_device, "ts_set_device");
  TS_BIND(downsample_ws, "ts_downsample_workspace_bytes");
  TS_BIND(downsample, "ts_downsample");
  TS_BIND(build_kmap_ws, "ts_build_kmap_workspace_bytes");
  TS_BIND(build_kmap, "ts_build_kmap");
  TS_BIND(build_kmap_sym, "ts_build_kmap_sym");
  TS_BIND(trilinear_ws, "ts_trilinear_workspace_bytes");
  TS_BIND(trilinear_map, "ts_trilinear_map");
  TS_BIND(devox_order_ws, "ts_devox_order_workspace_bytes");
  TS_BIND(devox_order, "ts_devox_order");
  TS_BIND(cat_cols, "ts_cat_cols");
  TS_BIND(copy_cols, "ts_copy_cols");
  TS_BIND(get_option, "ts_get_option");
#undef TS_BIND
  api.handle = h;
}

std::vector<at::Tensor> conv_block(const at::Tensor &feats, const at::Tensor &weight, const c10::optional<at::Tensor> &residual,
                      const at::Tensor &bn_weight, const at::Tensor &bn_bias, const at::Tensor &nbmaps,
                      const at::Tensor &nboffs, int64_t total, const at::Tensor &pos_out, const at::Tensor &pos_in,
                      int64_t n_in, int64_t n_out, bool transposed, const c10::optional<at::Tensor> &running_mean,
                      const c10::optional<at::Tensor> &running_var, const c10::optional<at::Tensor> &nbt, double momentum,
                      double eps, bool relu, int64_t comm, bool half, int64_t stream,
                      const c10::optional<at::Tensor> &planes, bool passthrough,
                      const c10::optional<at::Tensor> &grad_dest, int64_t group_id,
                      const std::vector<at::Tensor> &plan_f, const std::vector<int64_t> &plan_f_meta,
                      const std::vector<at::Tensor> &plan_d, const std::vector<int64_t> &plan_d_meta, bool wgrad_side_ok, bool natural) {
  TORCH_CHECK(api.handle, "taseg_amd fast path: load_backend() has not been called");
  TORCH_CHECK(!natural || (!passthrough && !transposed && plan_f.empty() && plan_d.empty() && n_in == n_out && total == n_out),
              "conv_block: a natural (1x1x1, identity rulebook) block takes no plans, no pass-through and one pair per row");
  TORCH_CHECK((plan_f.empty() || plan_f.size() == 4) && (plan_d.empty() || plan_d.size() == 4),
              "conv_block: a plan is (src, tile_info, n_tiles, pos | rows)");
  auto at_ = [](const std::vector<at::Tensor> &v, size_t i) { return v.size() == 4 ? c10::optional<at::Tensor>(v[i]) : c10::nullopt; };
  return ConvBlock::apply(feats, weight, residual, bn_weight, bn_bias, nbmaps, nboffs, total, pos_out, pos_in, n_in, n_out,
                          transposed, running_mean, running_var, nbt, momentum, eps, relu, comm, half, stream, planes,
                          passthrough, grad_dest, group_id, at_(plan_f, 0), at_(plan_f, 1), at_(plan_f, 2), at_(plan_f, 3),
                          plan_f_meta, at_(plan_d, 0), at_(plan_d, 1), at_(plan_d, 2), at_(plan_d, 3), plan_d_meta, wgrad_side_ok, natural);
}

// Evaluation form of the block (eval-mode BatchNorm on its running statistics, no graph): ts_conv_block_eval without the
// interpreter around it - the Python wrapper (functional.conv_block_eval) costs ~70 us per block, 63 blocks per pass, which made
// the evaluation forward host-bound (tools/eval_probe.py: 4.9 ms for ~3.5 ms of kernels).
at::Tensor conv_block_eval(const at::Tensor &feats, const at::Tensor &weight, const c10::optional<at::Tensor> &residual,
                           const at::Tensor &bn_weight, const at::Tensor &bn_bias, const at::Tensor &mean,
                           const at::Tensor &invstd, const at::Tensor &nbmaps, const at::Tensor &nboffs, int64_t total,
                           const at::Tensor &pos_out, const at::Tensor &pos_in, int64_t n_in, int64_t n_out, bool transposed,
                           bool relu, bool half, int64_t stream, const c10::optional<at::Tensor> &planes,
                           const std::vector<at::Tensor> &plan_f, const std::vector<int64_t> &plan_f_meta, bool natural) {
  TORCH_CHECK(api.handle, "taseg_amd fast path: load_backend() has not been called");
  at::NoGradGuard nograd;
  const bool flat = weight.dim() == 2;
  TORCH_CHECK(flat == natural, "conv_block_eval: a [C_in, C_out] weight goes with the identity rulebook (natural) and vice versa");
  TORCH_CHECK(!natural || (!transposed && plan_f.empty() && n_in == n_out && total == n_out),
              "conv_block_eval: a natural (1x1x1, identity rulebook) block takes no plan and one pair per row");
  const int64_t k = flat ? 1 : weight.size(0), c_in = weight.size(flat ? 0 : 1), c_out = weight.size(flat ? 1 : 2);
  const auto dt = half ? at::kHalf : at::kFloat;
  const int64_t rows = transposed ? n_in : n_out;
  const at::Tensor &table = transposed ? pos_in : pos_out;
  at::Tensor x = feats.contiguous().to(dt);
  at::Tensor w32 = weight.detach().contiguous().to(at::kFloat);
  at::Tensor res;
  if (residual.has_value() && residual->defined()) res = residual->contiguous().to(dt);
  at::Tensor out = at::empty({rows, c_out}, x.options());
  const bool own_weight = w32.data_ptr() == weight.data_ptr();
  const bool have = planes.has_value() && planes->defined() && own_weight;
  const bool kept16 = half && have && planes->scalar_type() == at::kHalf && planes->numel() == k * c_in * c_out;
  at::Tensor w16, pl;
  if (half) w16 = kept16 ? *planes : at::empty({k, c_in, c_out}, x.options().dtype(at::kHalf));
  if (!half && have && planes->scalar_type() == at::kShort) pl = *planes;
  const size_t nb = api.workspace_bytes(total, std::max(n_in, n_out), (int32_t)c_in, (int32_t)c_out, (int32_t)k, half ? 1 : 0);
  at::Tensor ws = workspace(nb, x, stream);
  const PlanRef pf(plan_f, plan_f_meta, nboffs);
  TsConvBlockOpts bopts = {pf.get(), nullptr, pl.defined() ? pl.data_ptr() : nullptr, kept16 ? 1 : 0, nullptr, nullptr, nullptr, 0, 0, 0,
                           natural ? 1 : 0};
  check(api.eval(x.data_ptr(), x.size(0), (int32_t)c_in, w32.data_ptr<float>(), (int32_t)k, (const int32_t *)nbmaps.data_ptr(),
                 (const int32_t *)nboffs.data_ptr(), total, transposed ? 1 : 0, (const int32_t *)table.data_ptr(), rows,
                 (int32_t)c_out, ptr(res), (const float *)bn_weight.data_ptr(), (const float *)bn_bias.data_ptr(),
                 (const float *)mean.data_ptr(), (const float *)invstd.data_ptr(), relu ? 1 : 0, half ? 1 : 0, out.data_ptr(),
                 ptr(w16), &bopts, ws.data_ptr(), (size_t)ws.numel(), (ts_stream_t)stream),
        "ts_conv_block_eval");
  return out;
}

// torch.distributed process group -> the index conv_block takes as `group_id` (SyncBatchNorm's all-reduce through c10d)
int64_t register_group(const c10::intrusive_ptr<c10d::ProcessGroup> &pg) {
  std::lock_guard<std::mutex> lock(group_mutex);
  for (size_t i = 0; i < groups.size(); ++i)
    if (groups[i].get() == pg.get()) return (int64_t)i;
  groups.push_back(pg);
  return (int64_t)groups.size() - 1;
}

void clear_groups() {
  std::lock_guard<std::mutex> lock(group_mutex);
  groups.clear();
}

// ------------------------------------------------------------------------------------------------ index plan
// Everything of a U-Net pass that depends on coordinates only (MinkUNetBackbone._index_plan): the coordinate set of
// every stride (spdownsample, downsample.py:25-51), the submanifold (kernel 3) and strided (kernel 2, stride 2)
// kernel maps with the reference's rulebook order (conv.py:144-177) and the trilinear point <-> voxel maps of
// voxel_to_point at strides 1, 16 and 4 (minkunet/utils.py:72-82).  Same backend calls as the Python path
// (functional.build_pyramid, backend.trilinear_map, backend.devox_order), issued here WITHOUT the interpreter lock:
// the stage has five host reads (4 coordinate counts, the pair totals) during which a Python thread would hold up
// the training thread; a data-stage thread that calls this function does not.
namespace {

std::mutex off_mutex;
std::unordered_map<int64_t, at::Tensor> off_cache;

// nn/utils/kernel.py:11-32 - per axis arange(-size // 2 + 1, size // 2 + 1) * stride; odd volumes enumerate z outermost /
// x innermost, even volumes x outermost / z innermost
at::Tensor kernel_offsets(int size, int stride, const at::Tensor &like) {
  std::lock_guard<std::mutex> lock(off_mutex);
  const int64_t key = ((int64_t)like.get_device() << 32) | ((int64_t)size << 16) | stride;
  auto it = off_cache.find(key);
  if (it != off_cache.end()) return it->second;
  std::vector<int> ax;
  const int lo = -((size + 1) / 2) + 1;                             // python: -size // 2 + 1  (3 -> -1, 2 -> 0)
  for (int i = 0; i < size; ++i) ax.push_back((lo + i) * stride);   // ... up to size // 2
  std::vector<int> rows;
  if ((size * size * size) % 2 == 1) {
    for (int z : ax) for (int y : ax) for (int x : ax) { rows.push_back(x); rows.push_back(y); rows.push_back(z); }
  } else {
    for (int x : ax) for (int y : ax) for (int z : ax) { rows.push_back(x); rows.push_back(y); rows.push_back(z); }
  }
  at::Tensor t = at::from_blob(rows.data(), {(int64_t)rows.size() / 3, 3}, at::kInt).clone().to(like.device());
  off_cache[key] = t;
  return t;
}

struct Kmap {
  at::Tensor nbr, nbmaps, nbsizes, nboffs, pos_out, pos_in;
  int64_t n_in = 0, n_out = 0;
};

// sym: a submanifold map (in_c IS out_c, odd symmetric offsets) on half the probes (ts_build_kmap_sym); its pair total reads -1 if
// the coordinates hold a duplicate - index_plan then calls again with sym = false
Kmap make_kmap(const at::Tensor &in_c, const at::Tensor &out_c, const at::Tensor &offsets, int64_t stream, bool sym = false) {
  Kmap km;
  km.n_in = in_c.size(0);
  km.n_out = out_c.size(0);
  const int64_t k = offsets.size(0);
  const auto o = in_c.options();
  km.nbr = at::empty({k, km.n_out}, o);
  km.nbmaps = at::empty({std::max<int64_t>(k * km.n_out, 1), 2}, o);
  km.nbsizes = at::empty({k}, o);
  km.nboffs = at::empty({k + 1}, o);
  km.pos_out = at::empty({k, km.n_out}, o);
  km.pos_in = at::empty({k, km.n_in}, o);
  at::Tensor ws = workspace(api.build_kmap_ws(km.n_in, km.n_out, (int32_t)k), in_c, stream);
  if (sym && km.n_out > 0)
    check(api.build_kmap_sym((const int32_t *)in_c.data_ptr(), km.n_in, (const int32_t *)offsets.data_ptr(), (int32_t)k,
                             (int32_t *)km.nbr.data_ptr(), nullptr, (int32_t *)km.nbmaps.data_ptr(), (int32_t *)km.nbsizes.data_ptr(),
                             (int32_t *)km.nboffs.data_ptr(), (int32_t *)km.pos_out.data_ptr(), (int32_t *)km.pos_in.data_ptr(),
                             ws.data_ptr(), (size_t)ws.numel(), (ts_stream_t)stream),
          "ts_build_kmap_sym");
  else
    check(api.build_kmap((const int32_t *)in_c.data_ptr(), km.n_in, (const int32_t *)out_c.data_ptr(), km.n_out,
                         (const int32_t *)offsets.data_ptr(), (int32_t)k, (int32_t *)km.nbr.data_ptr(), nullptr,
                         (int32_t *)km.nbmaps.data_ptr(), (int32_t *)km.nbsizes.data_ptr(), (int32_t *)km.nboffs.data_ptr(),
                         (int32_t *)km.pos_out.data_ptr(), (int32_t *)km.pos_in.data_ptr(), ws.data_ptr(), (size_t)ws.numel(),
                         (ts_stream_t)stream),
          "ts_build_kmap");
  return km;
}

std::vector<at::Tensor> kmap_tensors(const Kmap &km) { return {km.nbr, km.nbmaps, km.nbsizes, km.nboffs, km.pos_out, km.pos_in}; }

}  // namespace

// returns (coords per level [L+1], submanifold kmaps [L+1] x 6 tensors, strided kmaps [L] x 6 tensors, pair totals
// [2L+1] in the order sub0, down0, sub1, down1, ..., trilinear idx [3], weights [3], devox order [1]) for strides
// (1, 16, 4) / (16); coords [N, 4] int32, point_coords [P, 4] float32, both on the device, `stream` the caller's
// current raw stream.
std::tuple<std::vector<at::Tensor>, std::vector<std::vector<at::Tensor>>, std::vector<std::vector<at::Tensor>>,
           std::vector<int64_t>, std::vector<at::Tensor>, std::vector<at::Tensor>, std::vector<at::Tensor>>
index_plan(const at::Tensor &coords_in, const at::Tensor &points_in, int64_t num_levels, int64_t stream) {
  TORCH_CHECK(api.handle, "taseg_amd fast path: load_backend() has not been called");
  TORCH_CHECK(coords_in.is_cuda() && coords_in.scalar_type() == at::kInt && coords_in.dim() == 2 && coords_in.size(1) == 4,
              "index_plan: coords must be a device int32 [N, 4] tensor");
  TORCH_CHECK(points_in.is_cuda() && points_in.scalar_type() == at::kFloat && points_in.dim() == 2 && points_in.size(1) == 4,
              "index_plan: point coordinates must be a device float32 [P, 4] tensor");
  TORCH_CHECK(num_levels == 4, "index_plan: the MinkUNet pyramid has 4 down-sampling levels");
  py::gil_scoped_release nogil;
  at::NoGradGuard nograd;
  at::Tensor coords = coords_in.contiguous(), points = points_in.contiguous();
  const bool sym_probe = api.get_option(TS_OPT_KMAP_FULL_PROBE) == 0;
  std::vector<at::Tensor> cmaps;
  std::vector<Kmap> sub, down;
  cmaps.push_back(coords);
  int stride = 1;
  for (int64_t level = 0; level <= num_levels; ++level) {
    const at::Tensor &cur = cmaps.back();
    sub.push_back(make_kmap(cur, cur, kernel_offsets(3, stride, cur), stream, sym_probe));
    if (level == num_levels) break;
    // spdownsample(kernel 2, stride 2): unique strided coordinates, (b, x, y, z)-sorted; one host read (the count)
    const int64_t n = cur.size(0);
    at::Tensor out = at::empty({std::max<int64_t>(n, 1), 4}, cur.options());
    at::Tensor cnt = at::empty({1}, cur.options());
    at::Tensor ws = workspace(api.downsample_ws(n), cur, stream);
    const int step = stride * 2;
    check(api.downsample((const int32_t *)cur.data_ptr(), n, step, step, step, (int32_t *)out.data_ptr(),
                         (int32_t *)cnt.data_ptr(), ws.data_ptr(), (size_t)ws.numel(), (ts_stream_t)stream),
          "ts_downsample");
    const int m = cnt.item<int>();
    TORCH_CHECK(m >= 0, "downsample: coordinate outside the supported range (0 <= batch < 1024, -2^17 <= x,y,z < 2^17)");
    at::Tensor nxt = out.narrow(0, 0, m);
    down.push_back(make_kmap(cur, nxt, kernel_offsets(2, stride, cur), stream));
    cmaps.push_back(nxt);
    stride = step;
  }
  // pair totals of all maps in ONE device -> host copy
  std::vector<at::Tensor> lasts;
  for (int64_t level = 0; level <= num_levels; ++level) {
    lasts.push_back(sub[level].nboffs.narrow(0, sub[level].nboffs.size(0) - 1, 1));
    if (level < num_levels) lasts.push_back(down[level].nboffs.narrow(0, down[level].nboffs.size(0) - 1, 1));
  }
  at::Tensor tot = at::cat(lasts).cpu();
  std::vector<int64_t> totals(tot.data_ptr<int>(), tot.data_ptr<int>() + tot.numel());
  // a submanifold map whose coordinates hold a duplicate (the symmetric builder says so with a total of -1): the full probe
  for (int64_t level = 0; level <= num_levels; ++level) {
    if (totals[2 * level] >= 0) continue;
    int st = 1 << level;
    sub[level] = make_kmap(cmaps[level], cmaps[level], kernel_offsets(3, st, cmaps[level]), stream, false);
    // (reported as -(pairs + 1): the caller marks the map - no class plans, scatter form of the input gradient)
    totals[2 * level] = -(int64_t)sub[level].nboffs.narrow(0, sub[level].nboffs.size(0) - 1, 1).cpu().item<int>() - 1;
  }
  // trilinear maps at strides 1, 16, 4 (+ the devoxelize-backward walk order of the coarse ones)
  std::vector<at::Tensor> tri_idx, tri_w, orders;
  const int64_t np = points.size(0);
  for (int s : {1, 16, 4}) {
    const int level = s == 1 ? 0 : (s == 4 ? 2 : 4);
    const at::Tensor &vox = cmaps[level];
    at::Tensor idx = at::empty({np, 8}, coords.options());
    at::Tensor w = at::empty({np, 8}, points.options());
    at::Tensor ws = workspace(api.trilinear_ws(vox.size(0)), coords, stream);
    check(api.trilinear_map((const float *)points.data_ptr(), np, (const int32_t *)vox.data_ptr(), vox.size(0), s,
                            (int32_t *)idx.data_ptr(), (float *)w.data_ptr(), ws.data_ptr(), (size_t)ws.numel(),
                            (ts_stream_t)stream),
          "ts_trilinear_map");
    tri_idx.push_back(idx);
    tri_w.push_back(w);
    if (s == 16) {   // strides 1 and 4 walk the inverse map instead (backend.devox_csr, built by the caller)
      at::Tensor order = at::empty({np}, coords.options());
      at::Tensor ws2 = workspace(api.devox_order_ws(np), coords, stream);
      check(api.devox_order((const int32_t *)idx.data_ptr(), np, vox.size(0), (int32_t *)order.data_ptr(), ws2.data_ptr(),
                            (size_t)ws2.numel(), (ts_stream_t)stream),
            "ts_devox_order");
      orders.push_back(order);
    }
  }
  std::vector<std::vector<at::Tensor>> sub_t, down_t;
  for (auto &k : sub) sub_t.push_back(kmap_tensors(k));
  for (auto &k : down) down_t.push_back(kmap_tensors(k));
  return {cmaps, sub_t, down_t, totals, tri_idx, tri_w, orders};
}

// (calls, ns in total, ns inside the backend call) of the node's forward and backward since the last call; resets the counters
std::vector<int64_t> host_times() {
  std::vector<int64_t> v = {host_clock.n_fwd.exchange(0), host_clock.ns_fwd.exchange(0), host_clock.ns_fwd_api.exchange(0),
                            host_clock.n_bwd.exchange(0), host_clock.ns_bwd.exchange(0), host_clock.ns_bwd_api.exchange(0)};
  return v;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  py::class_<stage::Program, std::shared_ptr<stage::Program>>(m, "StageProgram",
                                                               "op list + parameter tensors of one MinkUNet stage (csrc/fastpath/stage_program.h)")
      .def(py::init<int, int, const std::vector<std::tuple<int, int, int, int, int, bool, bool>> &,
                    const std::vector<std::tuple<at::Tensor, at::Tensor, at::Tensor, c10::optional<at::Tensor>, c10::optional<at::Tensor>,
                                                 c10::optional<at::Tensor>, double, double>> &>())
      .def("set_planes", &stage::Program::set_planes)
      .def("set_grad_dests", &stage::Program::set_grad_dests)
      .def("set_deliver", &stage::Program::set_deliver)
      .def_readonly("n_inputs", &stage::Program::n_inputs);
  py::class_<stage::Geometry, std::shared_ptr<stage::Geometry>>(m, "StageGeometry", "kernel maps + class plans per op of a stage for one batch")
      .def(py::init<const std::vector<std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, int64_t, int64_t, int64_t>> &,
                    const std::vector<int> &, const std::vector<std::vector<at::Tensor>> &, const std::vector<std::vector<int64_t>> &,
                    const std::vector<std::vector<at::Tensor>> &, const std::vector<std::vector<int64_t>> &, bool>());
  m.def("stage_run", &stage::run, "a whole stage (block calls + concatenation) as ONE autograd node, training mode");
  m.def("stage_run_eval", &stage::run_eval, "a whole stage on the running statistics, no graph");
  m.def("unet_run", &stage::unet_run, "stage1 .. up4 of the U-Net pass in one call (eight stage programs, the dropouts between them)");
  m.def("host_times", &host_times, "diagnostic: (calls, ns, ns in the backend call) of the block node's forward and backward; resets");
  m.def("index_plan", &index_plan, "coordinate pyramid + kernel maps + trilinear maps of a MinkUNet pass (releases the GIL)");
  m.def("load_backend", &load_backend, "bind libtaseg_hip.so");
  m.def("conv_block", &conv_block, "act(BN(conv(x)) [+ residual]) as one native autograd node");
  m.def("conv_block_eval", &conv_block_eval, "act(BN_eval(conv(x)) [+ residual]) on the running statistics, no graph");
  m.def("join_wgrad_stream", &join_wgrad_stream, "the given stream waits for every weight gradient handed to the second stream so far");
  m.def("set_wgrad_stream", &set_wgrad_stream, "weight gradients of conv_block's backward on a second stream (joined at the end of the pass)");
  m.def("register_group", &register_group, "process group -> id for conv_block's c10d SyncBatchNorm path");
  m.def("clear_groups", &clear_groups, "drop the registered process groups (before destroy_process_group)");
}
