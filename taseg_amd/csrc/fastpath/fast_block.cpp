// Native autograd node for conv -> BatchNorm(train) [+ residual] [-> ReLU] (the unit of the MinkUNet family,
// R/pcseg/model/segmentor/voxel/minkunet/minkunet.py:31-129).  Same backend calls as the Python Function
// taseg_amd.torchsparse.nn.functional._ConvBlock (ts_conv_block_forward / ts_conv_block_backward of libtaseg_hip.so,
// bound with dlopen - this file contains no device code); what it removes is interpreter work: the Python node costs
// ~53 us (forward) + ~85 us (backward) of host time per block, 55 blocks per step, about half of it tensor
// bookkeeping and argument marshalling around ~25 us of launches.  PyTorch supplies memory, streams and the autograd
// graph only.
#include <dlfcn.h>
#include <torch/extension.h>

#include <mutex>
#include <unordered_map>

#include "taseg_hip.h"

namespace {

struct Api {
  void *handle = nullptr;
  decltype(&ts_conv_block_workspace_bytes) workspace_bytes = nullptr;
  decltype(&ts_conv_block_forward) forward = nullptr;
  decltype(&ts_conv_block_backward) backward = nullptr;
  decltype(&ts_last_error) last_error = nullptr;
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

inline void *ptr(const at::Tensor &t) { return t.defined() ? t.data_ptr() : nullptr; }
inline void *optr(const c10::optional<at::Tensor> &t) { return (t.has_value() && t->defined()) ? t->data_ptr() : nullptr; }

class ConvBlock : public torch::autograd::Function<ConvBlock> {
 public:
  static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &feats, const at::Tensor &weight,
                            const c10::optional<at::Tensor> &residual, const at::Tensor &bn_weight,
                            const at::Tensor &bn_bias, const at::Tensor &nbmaps, const at::Tensor &nboffs, int64_t total,
                            const at::Tensor &pos_out, const at::Tensor &pos_in, int64_t n_in, int64_t n_out,
                            bool transposed, const c10::optional<at::Tensor> &running_mean,
                            const c10::optional<at::Tensor> &running_var, const c10::optional<at::Tensor> &nbt,
                            double momentum, double eps, bool relu, int64_t comm, bool half, int64_t stream) {
    const int64_t k = weight.size(0), c_in = weight.size(1), c_out = weight.size(2);
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
    if (half) w16 = at::empty({k, c_in, c_out}, opts.dtype(at::kHalf));
    if (comm) pack = at::empty({2 * c_out + 1}, opts.dtype(at::kDouble));
    const size_t nb = api.workspace_bytes(total, std::max(n_in, n_out), (int32_t)c_in, (int32_t)c_out, (int32_t)k, half ? 1 : 0);
    at::Tensor ws = workspace(nb, x, stream);
    float *st = stats.data_ptr<float>();
    check(api.forward(x.data_ptr(), x.size(0), (int32_t)c_in, w32.data_ptr<float>(), (int32_t)k,
                      (const int32_t *)nbmaps.data_ptr(), (const int32_t *)nboffs.data_ptr(), total, transposed ? 1 : 0,
                      (const int32_t *)table.data_ptr(), rows, (int32_t)c_out, ptr(res), (const float *)bn_weight.data_ptr(),
                      (const float *)bn_bias.data_ptr(), (float *)optr(running_mean), (float *)optr(running_var),
                      (int64_t *)optr(nbt), (float)eps, (float)momentum, relu ? 1 : 0, half ? 1 : 0, (void *)comm,
                      (double *)ptr(pack), conv_out.data_ptr(), st, st + c_out, out.data_ptr(), (uint8_t *)ptr(mask), ptr(w16),
                      ws.data_ptr(), (size_t)ws.numel(), (ts_stream_t)stream),
          "ts_conv_block_forward");
    ctx->save_for_backward({x, half ? w16 : w32, conv_out, stats, mask, bn_weight, nbmaps, nboffs, pos_out, pos_in, pack});
    ctx->saved_data["total"] = total;
    ctx->saved_data["n_in"] = n_in;
    ctx->saved_data["n_out"] = n_out;
    ctx->saved_data["transposed"] = transposed;
    ctx->saved_data["half"] = half;
    ctx->saved_data["comm"] = comm;
    ctx->saved_data["stream"] = stream;
    ctx->saved_data["has_res"] = res.defined();
    ctx->saved_data["in_dtype"] = (int64_t)feats.scalar_type();
    ctx->saved_data["res_dtype"] = (int64_t)(res.defined() ? residual->scalar_type() : at::kFloat);
    return out;
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx,
                                                 torch::autograd::variable_list grads) {
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
    const int64_t k = w.size(0), c_in = w.size(1), c_out = w.size(2), rows = conv_out.size(0);
    const int64_t stream = ctx->saved_data["stream"].toInt();   // the engine runs a node on its forward stream
    at::Tensor g = grads[0].contiguous().to(conv_out.scalar_type());
    const at::Tensor &table = transposed ? pos_out : pos_in;
    const int64_t drows = transposed ? n_out : n_in;
    const auto opts = conv_out.options();
    at::Tensor grad_feat, grad_w, grad_res, sums;
    if (ctx->needs_input_grad(0)) grad_feat = at::empty({drows, c_in}, opts);
    if (ctx->needs_input_grad(1)) grad_w = at::empty({k, c_in, c_out}, opts.dtype(at::kFloat));
    if (has_res && ctx->needs_input_grad(2)) grad_res = at::empty_like(conv_out);
    at::Tensor gwb = at::empty({2, c_out}, opts.dtype(at::kFloat));
    if (comm) sums = at::empty({2, c_out}, opts.dtype(at::kDouble));
    const size_t nb = api.workspace_bytes(total, std::max(n_in, n_out), (int32_t)c_in, (int32_t)c_out, (int32_t)k, half ? 1 : 0);
    at::Tensor ws = workspace(nb, x, stream);
    const float *st = stats.data_ptr<float>();
    float *gw = gwb.data_ptr<float>();
    check(api.backward(g.data_ptr(), (const uint8_t *)ptr(mask), conv_out.data_ptr(), st, st + c_out,
                       (const float *)bn_weight.data_ptr(), pack.defined() ? pack.data_ptr<double>() + 2 * c_out : nullptr,
                       (void *)comm, (double *)ptr(sums), rows, (int32_t)c_out, half ? 1 : 0, x.data_ptr(), x.size(0),
                       (int32_t)c_in, w.data_ptr(), (int32_t)k, (const int32_t *)nbmaps.data_ptr(),
                       (const int32_t *)nboffs.data_ptr(), total, transposed ? 0 : 1, (const int32_t *)table.data_ptr(), drows,
                       transposed ? 1 : 0, ptr(grad_feat), ptr(grad_res), (float *)ptr(grad_w), gw, gw + c_out, ws.data_ptr(),
                       (size_t)ws.numel(), (ts_stream_t)stream),
          "ts_conv_block_backward");
    if (grad_feat.defined() && grad_feat.scalar_type() != in_dtype) grad_feat = grad_feat.to(in_dtype);
    if (grad_res.defined() && grad_res.scalar_type() != res_dtype) grad_res = grad_res.to(res_dtype);
    at::Tensor none;
    return {grad_feat, grad_w, grad_res, gwb[0], gwb[1], none, none, none, none, none, none, none, none,
            none, none, none, none, none, none, none, none, none};
  }
};

}  // namespace

void load_backend(const std::string &libpath) {
  if (api.handle) return;
  void *h = dlopen(libpath.c_str(), RTLD_NOW | RTLD_GLOBAL);
  TORCH_CHECK(h, "dlopen(", libpath, "): ", dlerror());
  api.workspace_bytes = (decltype(api.workspace_bytes))dlsym(h, "ts_conv_block_workspace_bytes");
  api.forward = (decltype(api.forward))dlsym(h, "ts_conv_block_forward");
  api.backward = (decltype(api.backward))dlsym(h, "ts_conv_block_backward");
  api.last_error = (decltype(api.last_error))dlsym(h, "ts_last_error");
  TORCH_CHECK(api.workspace_bytes && api.forward && api.backward, "libtaseg_hip.so lacks the ts_conv_block_* entry points");
  api.handle = h;
}

at::Tensor conv_block(const at::Tensor &feats, const at::Tensor &weight, const c10::optional<at::Tensor> &residual,
                      const at::Tensor &bn_weight, const at::Tensor &bn_bias, const at::Tensor &nbmaps,
                      const at::Tensor &nboffs, int64_t total, const at::Tensor &pos_out, const at::Tensor &pos_in,
                      int64_t n_in, int64_t n_out, bool transposed, const c10::optional<at::Tensor> &running_mean,
                      const c10::optional<at::Tensor> &running_var, const c10::optional<at::Tensor> &nbt, double momentum,
                      double eps, bool relu, int64_t comm, bool half, int64_t stream) {
  TORCH_CHECK(api.handle, "taseg_amd fast path: load_backend() has not been called");
  return ConvBlock::apply(feats, weight, residual, bn_weight, bn_bias, nbmaps, nboffs, total, pos_out, pos_in, n_in, n_out,
                          transposed, running_mean, running_var, nbt, momentum, eps, relu, comm, half, stream);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("load_backend", &load_backend, "bind libtaseg_hip.so");
  m.def("conv_block", &conv_block, "act(BN(conv(x)) [+ residual]) as one native autograd node");
}
