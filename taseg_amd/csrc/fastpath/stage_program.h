// The issue loop of a MinkUNet stage in C++ (included by fast_block.cpp after its helpers: api, check, workspace, the second-stream
// ring, group_sum).  A stage of the reference's backbone - an encoder stage `BasicConvolutionBlock(k2, s2) + n x ResidualBlock`, a
// decoder stage `BasicDeconvolutionBlock + torchsparse.cat + n x ResidualBlock` (R/pcseg/model/segmentor/voxel/minkunet/
// minkunet.py:186-356, forward :393-422) - is a short straight-line program over a few feature matrices: block calls
// (conv -> BatchNorm [+ residual] [-> ReLU], csrc/block.hip) and one concatenation.  The per-block autograd nodes (ConvBlock above,
// functional._ConvBlock) issue exactly these calls, but each behind ~10 allocator calls, a Python -> C++ crossing, an autograd node
// and three AccumulateGrad edges: ~130 us of host time per block against ~30 us of launches, which made the reference's default mode
// (autocast) and the evaluation loop host-bound.  Here ONE autograd node runs the whole stage:
//   * StageProgram  - what depends on the MODEL only: the op list and the parameter / buffer tensors of every layer, resolved once;
//   * StageGeometry - what depends on the BATCH only: kernel maps, class plans and row counts per op, resolved once per batch (on the
//                     staging thread, together with the index plan);
//   * StageRun      - forward: activations, BatchNorm statistics, ReLU masks of all blocks in ONE arena sized from the geometry
//                     (pointer arithmetic instead of tensors); backward: the same op list in reverse, gradients in a second arena.
// Same kernels, same launch order, same bits as the per-block nodes (tests/test_gpu_stage_program.py compares them bit for bit).
#pragma once

namespace stage {

constexpr size_t ALIGN = 256;
inline size_t up(size_t x) { return (x + ALIGN - 1) / ALIGN * ALIGN; }

struct Layer {                            // one convolution + its BatchNorm / SyncBatchNorm
  at::Tensor kernel, bn_w, bn_b, rmean, rvar, nbt;      // the modules' own tensors (running statistics may be undefined)
  int64_t k = 1, c_in = 0, c_out = 0;
  bool natural = false;                   // 1x1x1: [C_in, C_out] weight on the identity rulebook
  double momentum = 0.1, eps = 1e-5;
  at::Tensor planes32, half16;            // pre-split bf16 planes / kept IEEE-half copy of the weight (taseg_amd/planes.py), or undefined
  at::Tensor dest_w, dest_g, dest_b;      // gradient-bucket slots (parallel.GradBucketReducer), or undefined
  int64_t claimed = -1;                   // gradient epoch in which the slots were last handed out
  bool claimed_direct = false;            // ... to a pass that delivers them itself (its parameters are off the autograd graph)
  at::Tensor invstd;                      // evaluation: 1 / sqrt(running_var + eps), in step with the buffer
  uint32_t invstd_version = 0;
  const void *invstd_src = nullptr;
};

enum OpKind : int { BLOCK = 0, CAT = 1 };
struct Op {
  int kind = BLOCK, layer = -1, src = -1, dst = -1, aux = -1;      // aux: residual register (block) / second source (cat), or -1
  bool transposed = false, relu = true;
};

struct Program {
  std::vector<Layer> layers;
  std::vector<Op> ops;
  int n_inputs = 1, n_regs = 0, out_reg = 0;
  std::vector<int> uses;                  // consumers of every register inside the stage
  // Direct delivery (parallel.GradBucketReducer.deliver): with the reducer's slots on every parameter of the stage the node does not
  // hand the gradients to autograd at all - it writes them into the slots and calls this ONCE at the end of its backward pass; the
  // reducer re-points p.grad and counts the bucket down.  183 AccumulateGrad nodes and 183 Python hooks per step (each a trip to
  // the interpreter lock from the engine's thread, beside a staging thread that also wants it) become 8 calls.
  std::function<void()> deliver;
  std::function<void()> deliver_check;    // raises when a bucket the slots belong to is no longer open (before the first write to a slot)

  // ops: (kind, layer, src, dst, aux, transposed, relu); layers: (kernel, bn_w, bn_b, running_mean, running_var, nbt, momentum, eps)
  Program(int n_inputs_, int out_reg_, const std::vector<std::tuple<int, int, int, int, int, bool, bool>> &ops_,
          const std::vector<std::tuple<at::Tensor, at::Tensor, at::Tensor, c10::optional<at::Tensor>, c10::optional<at::Tensor>,
                                       c10::optional<at::Tensor>, double, double>> &layers_)
      : n_inputs(n_inputs_), out_reg(out_reg_) {
    for (const auto &l : layers_) {
      Layer y;
      y.kernel = std::get<0>(l);
      y.bn_w = std::get<1>(l);
      y.bn_b = std::get<2>(l);
      if (std::get<3>(l).has_value()) y.rmean = *std::get<3>(l);
      if (std::get<4>(l).has_value()) y.rvar = *std::get<4>(l);
      if (std::get<5>(l).has_value()) y.nbt = *std::get<5>(l);
      y.momentum = std::get<6>(l);
      y.eps = std::get<7>(l);
      TORCH_CHECK(y.kernel.dim() == 2 || y.kernel.dim() == 3, "stage program: a convolution weight is [K, C_in, C_out] or [C_in, C_out]");
      y.natural = y.kernel.dim() == 2;
      y.k = y.natural ? 1 : y.kernel.size(0);
      y.c_in = y.kernel.size(y.natural ? 0 : 1);
      y.c_out = y.kernel.size(y.natural ? 1 : 2);
      TORCH_CHECK(y.bn_w.numel() == y.c_out && y.bn_b.numel() == y.c_out, "stage program: BatchNorm width does not match the convolution");
      layers.push_back(std::move(y));
    }
    int regs = n_inputs;
    for (const auto &o : ops_) {
      Op p;
      p.kind = std::get<0>(o);
      p.layer = std::get<1>(o);
      p.src = std::get<2>(o);
      p.dst = std::get<3>(o);
      p.aux = std::get<4>(o);
      p.transposed = std::get<5>(o);
      p.relu = std::get<6>(o);
      TORCH_CHECK(p.kind == BLOCK || p.kind == CAT, "stage program: unknown op kind");
      TORCH_CHECK(p.src >= 0 && p.src < regs && p.aux < regs && p.dst == regs, "stage program: registers are written once, in order");
      TORCH_CHECK(p.kind == CAT ? p.aux >= 0 : (p.layer >= 0 && p.layer < (int)layers.size()), "stage program: bad op operands");
      ++regs;
      ops.push_back(p);
    }
    n_regs = regs;
    TORCH_CHECK(out_reg >= n_inputs && out_reg < n_regs, "stage program: the output register must be written by an op");
    uses.assign(n_regs, 0);
    for (const Op &o : ops) {
      ++uses[o.src];
      if (o.aux >= 0) ++uses[o.aux];
    }
    TORCH_CHECK(uses[out_reg] == 0, "stage program: the output register is not read inside the stage");
  }

  std::vector<at::Tensor> parameters() const {
    std::vector<at::Tensor> v;
    for (const Layer &l : layers) {
      v.push_back(l.kernel);
      v.push_back(l.bn_w);
      v.push_back(l.bn_b);
    }
    return v;
  }
  // planes32 / half16 per layer (undefined = none), as taseg_amd.planes hands them out: the tensors are persistent objects that the
  // module refreshes in place when a weight has changed, so they are set once and again only when an entry was re-created
  void set_planes(const std::vector<c10::optional<at::Tensor>> &p32, const std::vector<c10::optional<at::Tensor>> &h16) {
    TORCH_CHECK(p32.size() == layers.size() && h16.size() == layers.size(), "stage program: one planes entry per layer");
    for (size_t i = 0; i < layers.size(); ++i) {
      layers[i].planes32 = p32[i].has_value() ? *p32[i] : at::Tensor();
      layers[i].half16 = h16[i].has_value() ? *h16[i] : at::Tensor();
    }
  }
  static std::function<void()> hold_callable(py::object fn) {
    if (fn.is_none()) return nullptr;
    // (the callable is kept in a shared holder whose deleter takes the interpreter lock)
    std::shared_ptr<py::object> hold(new py::object(std::move(fn)), [](py::object *o) {
      py::gil_scoped_acquire gil;
      delete o;
    });
    return [hold]() {
      py::gil_scoped_acquire gil;
      (*hold)();
    };
  }
  void set_deliver(py::object fn, py::object check) {
    deliver = hold_callable(std::move(fn));
    deliver_check = deliver ? hold_callable(std::move(check)) : nullptr;
  }
  // A pass of this epoch already took its parameters OFF the autograd graph (direct delivery).  A second pass before that one's
  // backward - two views, a consistency loss, recompute with gradients - would put them back ON it: AccumulateGrad then fires after
  // the second pass's contribution alone, the reducer's hooks count the buckets down and launch the all-reduce on partial gradients,
  // and the first pass's node afterwards writes into slots that are in flight.  Refused where it starts.
  void refuse_second_pass(int64_t grad_epoch) const {
    for (const Layer &l : layers)
      TORCH_CHECK(!(l.claimed == grad_epoch && l.claimed_direct),
                  "taseg_amd stage program: a second forward pass with gradients through the same stage before the backward pass of the "
                  "first (two views / consistency losses / recompute).  The first pass delivers its parameter gradients straight into "
                  "the optimizer's buckets (exactly one forward + backward per optimizer step); for such loops set "
                  "taseg_amd.options.options.direct_grads = False before building the model");
  }
  // may this pass deliver its gradients directly?  (every slot present, nothing accumulated, slots not yet handed out this epoch)
  bool direct_ok(int64_t grad_epoch) const {
    if (!deliver) return false;
    for (const Layer &l : layers)
      if (!(l.dest_w.defined() && l.dest_g.defined() && l.dest_b.defined()) || l.claimed == grad_epoch || l.kernel.grad().defined() ||
          l.bn_w.grad().defined() || l.bn_b.grad().defined() || !l.kernel.requires_grad() || !l.bn_w.requires_grad() ||
          !l.bn_b.requires_grad())
        return false;
    return true;
  }
  // gradient-bucket slots of (kernel, bn weight, bn bias) per layer, or none
  void set_grad_dests(const std::vector<c10::optional<at::Tensor>> &d) {
    TORCH_CHECK(d.size() == 3 * layers.size(), "stage program: three gradient slots per layer");
    for (size_t i = 0; i < layers.size(); ++i) {
      layers[i].dest_w = d[3 * i].has_value() ? *d[3 * i] : at::Tensor();
      layers[i].dest_g = d[3 * i + 1].has_value() ? *d[3 * i + 1] : at::Tensor();
      layers[i].dest_b = d[3 * i + 2].has_value() ? *d[3 * i + 2] : at::Tensor();
      layers[i].claimed = -1;
      layers[i].claimed_direct = false;
    }
  }
};

struct MapRef {
  at::Tensor nbmaps, nboffs, pos_out, pos_in;
  int64_t total = 0, n_in = 0, n_out = 0;
};

struct PlanHold {                         // a class plan of one op: its tensors (kept alive) and the struct the backend reads
  std::vector<at::Tensor> t;
  TsClassPlan plan;
  bool live = false;
  void set(const std::vector<at::Tensor> &tensors, const std::vector<int64_t> &m, const at::Tensor &nboffs) {
    if (tensors.size() != 4 || m.size() != 7) return;
    t = tensors;
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

struct Geometry {
  std::vector<MapRef> maps;
  std::vector<int> op_map;                // per op: index into maps (-1 for a concatenation)
  std::vector<PlanHold> plan_f, plan_d;   // per op
  bool half = false;                      // the storage mode the plans were chosen for

  Geometry(const std::vector<std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, int64_t, int64_t, int64_t>> &maps_,
           const std::vector<int> &op_map_, const std::vector<std::vector<at::Tensor>> &pf, const std::vector<std::vector<int64_t>> &pfm,
           const std::vector<std::vector<at::Tensor>> &pd, const std::vector<std::vector<int64_t>> &pdm, bool half_)
      : op_map(op_map_), half(half_) {
    for (const auto &m : maps_) {
      MapRef r;
      r.nbmaps = std::get<0>(m);
      r.nboffs = std::get<1>(m);
      r.pos_out = std::get<2>(m);
      r.pos_in = std::get<3>(m);
      r.total = std::get<4>(m);
      r.n_in = std::get<5>(m);
      r.n_out = std::get<6>(m);
      TORCH_CHECK(r.nbmaps.is_cuda() && r.nbmaps.scalar_type() == at::kInt && r.nboffs.scalar_type() == at::kInt, "stage geometry: int32 device rulebooks");
      maps.push_back(std::move(r));
    }
    const size_t n = op_map.size();
    TORCH_CHECK(pf.size() == n && pfm.size() == n && pd.size() == n && pdm.size() == n, "stage geometry: one plan entry per op");
    plan_f.resize(n);
    plan_d.resize(n);
    for (size_t i = 0; i < n; ++i) {
      TORCH_CHECK(op_map[i] < (int)maps.size(), "stage geometry: unknown kernel map");
      if (op_map[i] < 0) continue;
      plan_f[i].set(pf[i], pfm[i], maps[op_map[i]].nboffs);
      plan_d[i].set(pd[i], pdm[i], maps[op_map[i]].nboffs);
    }
  }
};

// what one run keeps for its backward pass
struct OpRec {
  const void *x = nullptr;                // the block's input rows
  int64_t x_rows = 0, rows = 0;
  void *conv_out = nullptr, *out = nullptr, *w16 = nullptr;
  float *stats = nullptr;
  uint8_t *mask = nullptr;
  double *pack = nullptr;
  bool side_ok = false;                   // the weight gradient may leave for the second stream (p.grad undefined at forward time)
  bool use_dest = false;                  // ... and the three gradients go straight into the bucket slots
};

struct State : torch::CustomClassHolder {
  std::shared_ptr<Program> prog;
  std::shared_ptr<Geometry> geom;
  at::Tensor arena;
  std::vector<at::Tensor> inputs;         // converted (contiguous, storage dtype) inputs: what the first blocks read
  std::vector<OpRec> recs;
  std::vector<int64_t> reg_rows, reg_ch;
  std::vector<at::ScalarType> in_dtypes;
  bool half = false, direct = false;
  int64_t stream = 0, comm = 0, group_id = -1;
};

inline void *cptr(const at::Tensor &t) { return t.defined() ? t.data_ptr() : nullptr; }

// rows / channels of every register for these inputs on this geometry; checks the program against the maps
inline void shapes(const Program &p, const Geometry &g, const std::vector<at::Tensor> &inputs, std::vector<int64_t> &rows,
                   std::vector<int64_t> &ch) {
  TORCH_CHECK((int)inputs.size() == p.n_inputs, "stage run: ", p.n_inputs, " input matrices expected");
  TORCH_CHECK(g.op_map.size() == p.ops.size(), "stage run: the geometry was built for another program");
  rows.assign(p.n_regs, 0);
  ch.assign(p.n_regs, 0);
  for (int i = 0; i < p.n_inputs; ++i) {
    TORCH_CHECK(inputs[i].dim() == 2 && inputs[i].is_cuda(), "stage run: inputs are device matrices [rows, channels]");
    rows[i] = inputs[i].size(0);
    ch[i] = inputs[i].size(1);
  }
  for (size_t i = 0; i < p.ops.size(); ++i) {
    const Op &o = p.ops[i];
    if (o.kind == CAT) {
      TORCH_CHECK(rows[o.src] == rows[o.aux], "stage run: concatenation of ", rows[o.src], " and ", rows[o.aux], " rows");
      rows[o.dst] = rows[o.src];
      ch[o.dst] = ch[o.src] + ch[o.aux];
      continue;
    }
    const Layer &l = p.layers[o.layer];
    const MapRef &m = g.maps[g.op_map[i]];
    TORCH_CHECK(ch[o.src] == l.c_in, "stage run: op ", i, " reads ", ch[o.src], " channels, its weight takes ", l.c_in);
    TORCH_CHECK(rows[o.src] == (o.transposed ? m.n_out : m.n_in), "stage run: op ", i, " reads ", rows[o.src], " rows, its kernel map has ",
                o.transposed ? m.n_out : m.n_in);
    TORCH_CHECK(m.total > 0 && rows[o.src] > 0, "stage run: empty kernel map");
    TORCH_CHECK(!l.natural || (m.n_in == m.n_out && m.total == m.n_out && !o.transposed), "stage run: a 1x1x1 block runs on the identity rulebook");
    rows[o.dst] = o.transposed ? m.n_in : m.n_out;
    ch[o.dst] = l.c_out;
    if (o.aux >= 0) TORCH_CHECK(rows[o.aux] == rows[o.dst] && ch[o.aux] == ch[o.dst], "stage run: residual of another shape");
  }
}

inline size_t block_ws_bytes(const Program &p, const Geometry &g, bool half) {
  size_t nb = 0;
  for (size_t i = 0; i < p.ops.size(); ++i) {
    if (p.ops[i].kind != BLOCK) continue;
    const Layer &l = p.layers[p.ops[i].layer];
    const MapRef &m = g.maps[g.op_map[i]];
    nb = std::max(nb, api.workspace_bytes(m.total, std::max(m.n_in, m.n_out), (int32_t)l.c_in, (int32_t)l.c_out, (int32_t)l.k, half ? 1 : 0));
  }
  return nb;
}

inline at::Tensor view_of(const at::Tensor &arena, const void *p, int64_t rows, int64_t ch, at::ScalarType dt) {
  const int64_t off = (const char *)p - (const char *)arena.data_ptr();
  const int64_t nbytes = rows * ch * (int64_t)c10::elementSize(dt);
  return arena.narrow(0, off, nbytes).view(dt).view({rows, ch});
}

class StageRun : public torch::autograd::Function<StageRun> {
 public:
  // inputs: the stage's feature matrices; params: (kernel, bn weight, bn bias) per layer - or empty when the gradients are delivered
  // to the bucket slots behind autograd's back is NOT done here: parameters always travel through autograd
  static torch::autograd::variable_list forward(torch::autograd::AutogradContext *ctx, at::TensorList inputs_, at::TensorList params,
                                                std::shared_ptr<Program> prog, std::shared_ptr<Geometry> geom, bool half,
                                                int64_t stream, int64_t comm, int64_t group_id, int64_t grad_epoch, bool direct) {
    const int64_t t_in = now_ns();
    if (wg_join_queued.exchange(false)) {
      wg_worker.drain();
      check(api.stream_join((ts_stream_t)stream, (ts_stream_t)wg_side.raw), "ts_stream_join");
    }
    Program &p = *prog;
    const Geometry &g = *geom;
    TORCH_CHECK(params.size() == (direct ? 0 : 3 * p.layers.size()), "stage run: three parameters per layer (none with direct delivery)");
    TORCH_CHECK(g.half == half, "stage run: the geometry was resolved for the other storage mode");
    auto st = c10::make_intrusive<State>();
    st->prog = prog;
    st->geom = geom;
    st->half = half;
    st->direct = direct;
    st->stream = stream;
    st->comm = comm;
    st->group_id = (comm == 0 && group_id >= 0) ? group_id : -1;
    const bool split = st->group_id >= 0;
    const bool sync = comm != 0 || split;
    const auto dt = half ? at::kHalf : at::kFloat;
    const size_t es = half ? 2 : 4;
    for (const at::Tensor &x : inputs_) {
      st->in_dtypes.push_back(x.scalar_type());
      st->inputs.push_back(x.contiguous().to(dt));
    }
    shapes(p, g, st->inputs, st->reg_rows, st->reg_ch);
    const auto &rows = st->reg_rows;
    const auto &ch = st->reg_ch;
    // ---- one arena for everything the stage produces and keeps
    size_t bytes = 0;
    for (size_t i = 0; i < p.ops.size(); ++i) {
      const Op &o = p.ops[i];
      const size_t mat = up((size_t)rows[o.dst] * ch[o.dst] * es);
      if (o.kind == CAT) {
        bytes += mat;
        continue;
      }
      const Layer &l = p.layers[o.layer];
      bytes += 2 * mat + up(2 * l.c_out * 4);
      if (o.relu) bytes += up((size_t)rows[o.dst] * (l.c_out / (half ? 8 : 4)));
      if (sync) bytes += up((2 * l.c_out + 1) * 8);
      if (half && !(l.half16.defined() && l.half16.numel() == l.k * l.c_in * l.c_out)) bytes += up((size_t)l.k * l.c_in * l.c_out * 2);
    }
    const at::Tensor &like = st->inputs[0];
    st->arena = at::empty({(int64_t)bytes}, like.options().dtype(at::kByte));
    char *cur = (char *)st->arena.data_ptr();
    auto take = [&](size_t n) {
      void *r = cur;
      cur += up(n);
      return r;
    };
    at::Tensor ws = workspace(block_ws_bytes(p, g, half), like, stream);
    std::vector<const void *> reg(p.n_regs, nullptr);
    for (int i = 0; i < p.n_inputs; ++i) reg[i] = st->inputs[i].data_ptr();
    st->recs.resize(p.ops.size());
    for (size_t i = 0; i < p.ops.size(); ++i) {
      const Op &o = p.ops[i];
      OpRec &r = st->recs[i];
      if (o.kind == CAT) {
        // torchsparse.cat (operators.py:10-17): [a | b] along the channels, one launch
        void *dst = take((size_t)rows[o.dst] * ch[o.dst] * es);
        check(api.cat_cols(reg[o.src], ch[o.src] * (int64_t)es, reg[o.aux], ch[o.aux] * (int64_t)es, rows[o.dst], dst, (ts_stream_t)stream),
              "ts_cat_cols");
        reg[o.dst] = dst;
        continue;
      }
      Layer &l = p.layers[o.layer];
      const MapRef &m = g.maps[g.op_map[i]];
      const int64_t n_rows = rows[o.dst];
      r.x = reg[o.src];
      r.x_rows = rows[o.src];
      r.rows = n_rows;
      r.conv_out = take((size_t)n_rows * l.c_out * es);
      r.out = take((size_t)n_rows * l.c_out * es);
      r.stats = (float *)take(2 * l.c_out * 4);
      if (o.relu) r.mask = (uint8_t *)take((size_t)n_rows * (l.c_out / (half ? 8 : 4)));
      if (sync) r.pack = (double *)take((2 * l.c_out + 1) * 8);
      const bool kept16 = half && l.half16.defined() && l.half16.numel() == l.k * l.c_in * l.c_out;
      if (half) r.w16 = kept16 ? l.half16.data_ptr() : take((size_t)l.k * l.c_in * l.c_out * 2);
      // the gradients of this layer may go where the reducer wants them / to the second stream only if autograd will ADOPT them as
      // p.grad: no gradient accumulated yet, slots not handed out since the last reducer.finish() / zero_grad()
      const bool fresh = !l.kernel.grad().defined() && !l.bn_w.grad().defined() && !l.bn_b.grad().defined();
      r.side_ok = !l.kernel.grad().defined();
      r.use_dest = fresh && l.dest_w.defined() && l.dest_g.defined() && l.dest_b.defined() && l.claimed != grad_epoch;
      TORCH_CHECK(!direct || r.use_dest, "stage run: direct delivery without the reducer's slots");
      if (r.use_dest) {
        l.claimed = grad_epoch;
        l.claimed_direct = direct;
      }
      const at::Tensor &table = o.transposed ? m.pos_in : m.pos_out;
      const void *pl = (!half && l.planes32.defined()) ? l.planes32.data_ptr() : nullptr;
      TsConvBlockOpts bopts = {g.plan_f[i].get(), nullptr, pl, kept16 ? 1 : 0, nullptr, nullptr, nullptr, 0, 0, 0, l.natural ? 1 : 0};
      auto call = [&](void *c) {
        check(api.forward(r.x, r.x_rows, (int32_t)l.c_in, l.kernel.data_ptr<float>(), (int32_t)l.k, (const int32_t *)m.nbmaps.data_ptr(),
                          (const int32_t *)m.nboffs.data_ptr(), m.total, o.transposed ? 1 : 0, (const int32_t *)table.data_ptr(), n_rows,
                          (int32_t)l.c_out, o.aux >= 0 ? reg[o.aux] : nullptr, (const float *)l.bn_w.data_ptr(),
                          (const float *)l.bn_b.data_ptr(), (float *)cptr(l.rmean), (float *)cptr(l.rvar), (int64_t *)cptr(l.nbt),
                          (float)l.eps, (float)l.momentum, o.relu ? 1 : 0, half ? 1 : 0, c, r.pack, r.conv_out, r.stats,
                          r.stats + l.c_out, r.out, r.mask, r.w16, &bopts, ws.data_ptr(), (size_t)ws.numel(), (ts_stream_t)stream),
              "ts_conv_block_forward");
      };
      const int64_t t_api = now_ns();
      if (split) {
        call(COMM_PRE);
        group_sum(st->group_id, view_of(st->arena, r.pack, 1, 2 * l.c_out + 1, at::kDouble).view({-1}));
        call(COMM_POST);
      } else {
        call((void *)comm);
      }
      host_clock.ns_fwd_api += now_ns() - t_api;
      // (the running statistics were written through raw pointers: whoever caches something derived from them must see a new version)
      if (l.rvar.defined()) l.rvar.unsafeGetTensorImpl()->bump_version();
      if (l.rmean.defined()) l.rmean.unsafeGetTensorImpl()->bump_version();
      reg[o.dst] = r.out;
    }
    ctx->saved_data["state"] = c10::IValue::make_capsule(st);
    at::Tensor result = view_of(st->arena, reg[p.out_reg], rows[p.out_reg], ch[p.out_reg], dt);
    host_clock.ns_fwd += now_ns() - t_in;
    host_clock.n_fwd += (int64_t)p.layers.size();
    return {result};
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx, torch::autograd::variable_list grads) {
    const int64_t t_in = now_ns();
    auto st = c10::static_intrusive_pointer_cast<State>(ctx->saved_data["state"].toCapsule());
    Program &p = *st->prog;
    const Geometry &g = *st->geom;
    const bool half = st->half, split = st->group_id >= 0;
    const int64_t stream = st->stream, comm = st->comm;
    const bool sync = comm != 0 || split;
    const auto dt = half ? at::kHalf : at::kFloat;
    const size_t es = half ? 2 : 4;
    const auto &rows = st->reg_rows;
    const auto &ch = st->reg_ch;
    TORCH_CHECK(grads[0].defined(), "stage run: the stage's output received no gradient");
    if (st->direct && p.deliver_check) p.deliver_check();      // the buckets must still be open BEFORE the first write to a slot
    at::Tensor g_out = grads[0].contiguous().to(dt);
    const int n_in = p.n_inputs;
    // which registers need a gradient: the inputs that require one; every block output is on a differentiable path (parameters)
    std::vector<char> need(p.n_regs, 1);
    for (int i = 0; i < n_in; ++i) need[i] = ctx->needs_input_grad(i) ? 1 : 0;
    // ---- the gradient arena.  Every producer writes a matrix of its own (a block's input gradient, its residual gradient, the left
    // column block of a concatenation): a register with two consumers (the input of a residual block) gets the earlier gradient
    // ADDED in the store of the later one (TsConvBlockOpts.addend), never in place.
    size_t bytes = 0, bn_floats = 0;
    const size_t n_ops = p.ops.size();
    std::vector<size_t> off_gf(n_ops, 0), off_gr(n_ops, 0), off_gwb(n_ops, 0), off_sums(n_ops, 0), off_cat(n_ops, 0);
    // an input that is only concatenated gets its column block of the concatenation's gradient as a strided view (autograd sums it
    // with the gradient of the matrix's other consumer: no copy here)
    auto cat_view = [&](const Op &o, int part) { const int r = part ? o.aux : o.src; return r < n_in && p.uses[r] == 1; };
    for (size_t i = 0; i < n_ops; ++i) {
      const Op &o = p.ops[i];
      if (o.kind == CAT) {
        off_cat[i] = bytes;
        for (int part = 0; part < 2; ++part) {
          const int r = part ? o.aux : o.src;
          if (need[r] && !cat_view(o, part)) bytes += up((size_t)rows[r] * ch[r] * es);
        }
        continue;
      }
      const Layer &l = p.layers[o.layer];
      const OpRec &r = st->recs[i];
      if (need[o.src]) {
        off_gf[i] = bytes;
        bytes += up((size_t)r.x_rows * l.c_in * es);
      }
      if (o.aux >= 0 && need[o.aux]) {
        off_gr[i] = bytes;
        bytes += up((size_t)r.rows * l.c_out * es);
      }
      off_gwb[i] = bn_floats;              // (BatchNorm gradients: a small tensor of their own - what autograd adopts as p.grad must
      bn_floats += 2 * l.c_out;           //  not keep the whole gradient arena alive)
      if (sync) {
        off_sums[i] = bytes;
        bytes += up(2 * l.c_out * 8);
      }
    }
    const at::Tensor &like = st->inputs[0];
    at::Tensor garena = at::empty({(int64_t)std::max<size_t>(bytes, ALIGN)}, like.options().dtype(at::kByte));
    char *gbase = (char *)garena.data_ptr();
    at::Tensor bn_grads = at::empty({(int64_t)std::max<size_t>(bn_floats, 1)}, like.options().dtype(at::kFloat));
    at::Tensor ws = workspace(block_ws_bytes(p, g, half), like, stream);
    std::vector<void *> G(p.n_regs, nullptr);         // gradient of a register once something has produced it (contiguous rows)
    std::vector<at::Tensor> gin(n_in);                // ... or, for an input, the strided view described above
    G[p.out_reg] = g_out.data_ptr();
    std::vector<at::Tensor> gparams(3 * p.layers.size());
    // (not beside SyncBatchNorm: the block call keeps the weight gradient of a statistics-exchanging block on its own stream)
    const bool side_possible = wg_side.on && !comm && !split && like.get_device() == wg_side.device_index && stream != wg_side.raw;
    for (int i = (int)n_ops - 1; i >= 0; --i) {
      const Op &o = p.ops[i];
      if (o.kind == CAT) {
        TORCH_CHECK(G[o.dst], "stage run: concatenation without a gradient");
        const int64_t pitch = ch[o.dst] * (int64_t)es, wa = ch[o.src] * (int64_t)es, wb = ch[o.aux] * (int64_t)es;
        char *slot = gbase + off_cat[i];
        for (int part = 0; part < 2; ++part) {
          const int r = part ? o.aux : o.src;
          if (!need[r]) continue;
          TORCH_CHECK(!G[r], "stage run: a concatenated matrix has an earlier gradient (not a shape this loop serves)");
          if (cat_view(o, part)) {
            gin[r] = view_of(garena, G[o.dst], rows[o.dst], ch[o.dst], dt).narrow(1, part ? ch[o.src] : 0, ch[r]);
            continue;
          }
          check(api.copy_cols(G[o.dst], pitch, part ? wa : 0, part ? wb : wa, rows[o.dst], slot, part ? wb : wa, (ts_stream_t)stream),
                "ts_copy_cols");
          G[r] = slot;
          slot += up((size_t)rows[r] * ch[r] * es);
        }
        continue;
      }
      Layer &l = p.layers[o.layer];
      const OpRec &r = st->recs[i];
      const MapRef &m = g.maps[g.op_map[i]];
      TORCH_CHECK(G[o.dst], "stage run: op ", i, " received no gradient");
      const at::Tensor &table = o.transposed ? m.pos_out : m.pos_in;
      const int64_t drows = r.x_rows;
      void *grad_feat = need[o.src] ? (void *)(gbase + off_gf[i]) : nullptr;
      void *grad_res = (o.aux >= 0 && need[o.aux]) ? (void *)(gbase + off_gr[i]) : nullptr;
      // an earlier-processed consumer of this block's input has left a gradient: added in the store of this block's input gradient
      // (the shortcut of a residual block, minkunet.py:117-129); a natural block takes no addend and adds afterwards
      const void *addend = (grad_feat && G[o.src] && !l.natural) ? G[o.src] : nullptr;
      const bool feat_add = grad_feat && G[o.src] && l.natural;
      const bool res_add = grad_res && G[o.aux];
      float *gwb = bn_grads.data_ptr<float>() + off_gwb[i];
      double *sums = sync ? (double *)(gbase + off_sums[i]) : nullptr;
      // parameter gradients: the bucket slots where the reducer named them and autograd will adopt them, else fresh tensors
      at::Tensor grad_w, gbw, gbb;
      float *gw_ptr = gwb, *gb_ptr = gwb + l.c_out;
      if (r.use_dest) {
        grad_w = l.dest_w.alias();
        gbw = l.dest_g.alias();
        gbb = l.dest_b.alias();
        gw_ptr = (float *)gbw.data_ptr();
        gb_ptr = (float *)gbb.data_ptr();
      } else {
        const std::vector<int64_t> wshape = l.natural ? std::vector<int64_t>{l.c_in, l.c_out} : std::vector<int64_t>{l.k, l.c_in, l.c_out};
        grad_w = at::empty(wshape, like.options().dtype(at::kFloat));
      }
      const void *pl = (!half && l.planes32.defined()) ? l.planes32.data_ptr() : nullptr;
      TsConvBlockOpts bopts = {nullptr, grad_feat ? g.plan_d[i].get() : nullptr, pl, 0, addend, nullptr, nullptr, 0, 0, 0, l.natural ? 1 : 0};
      std::function<void()> side_job;
      if (side_possible && r.side_ok && (l.c_in * l.c_out) % 4 == 0 && (((uintptr_t)grad_w.data_ptr()) & 15) == 0) {
        std::lock_guard<std::mutex> lock(wg_mutex);
        WgSide &sd = wg_side;
        const size_t needb = api.wgrad_ws_bytes(m.total, r.rows, (int32_t)l.c_in, (int32_t)l.c_out, (int32_t)l.k, half ? 1 : 0);
        const int slot = sd.next;
        sd.next = (sd.next + 1) % WG_SLOTS;
        const c10::Stream side = c10::Stream::unpack3(sd.stream_id, (c10::DeviceIndex)sd.device_index, (c10::DeviceType)sd.device_type);
        if (!sd.ring[slot].defined() || (size_t)sd.ring[slot].numel() < needb) {
          wg_worker.drain();              // (a queued job may still point into the old buffer)
          sd.ring[slot] = at::empty({(int64_t)(needb * 1.25) + 256}, like.options().dtype(at::kByte));
          sd.ring[slot].record_stream(side);
        }
        bopts.wgrad_stream = (ts_stream_t)sd.raw;
        bopts.wgrad_ws = sd.ring[slot].data_ptr();
        bopts.wgrad_ws_bytes = (size_t)sd.ring[slot].numel();
        bopts.wgrad_slot = slot;
        bopts.wgrad_deferred = 1;
        // What the second stream reads (block inputs: the forward arena, the stage's inputs) stays alive through `keep` below until
        // the job has been enqueued and through the graph until the pass has ended - and the pass ends with the join of the second
        // stream into this one (the engine callback below): whatever reuses that memory afterwards is ordered behind the reads.  No
        // record_stream on the arena: a block with a pending cross-stream event is withheld from the allocator's pool while the
        // device is behind the host, and a device-bound fp32 step then spent 3 ms per pass in the allocator.  A fresh weight
        // gradient becomes p.grad and may be dropped by the caller at any time: that one is marked.
        if (!r.use_dest) grad_w.record_stream(side);
        {
          // (grad_w by address only: one more owner and AccumulateGrad would copy the gradient instead of adopting the tensor)
          const at::Tensor ringk = sd.ring[slot];
          float *const gwp = (float *)grad_w.data_ptr();
          const ts_stream_t side_raw = bopts.wgrad_stream;
          const void *xk = r.x;
          const int64_t n_feat = r.x_rows, total = m.total, nrows = r.rows;
          const int32_t c_in = (int32_t)l.c_in, c_out = (int32_t)l.c_out, kk = (int32_t)l.k, hf = half ? 1 : 0;
          const int32_t col_a = o.transposed ? 1 : 0, chunk_order = (grad_feat && drows > 0 && !l.natural) ? 1 : 0;
          const int32_t *nbm = (const int32_t *)m.nbmaps.data_ptr(), *nbo = (const int32_t *)m.nboffs.data_ptr();
          auto keep = st;                 // (forward arena, inputs, geometry)
          side_job = [=]() {
            (void)keep;
            check(api.wgrad_side(xk, n_feat, c_in, kk, nbm, nbo, total, col_a, nrows, c_out, hf, gwp, chunk_order, ringk.data_ptr(),
                                 (size_t)ringk.numel(), slot, side_raw),
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
      const void *w = half ? (const void *)r.w16 : (const void *)l.kernel.data_ptr();
      auto call = [&](void *c) {
        check(api.backward(G[o.dst], r.mask, r.conv_out, r.stats, r.stats + l.c_out, (const float *)l.bn_w.data_ptr(),
                           r.pack ? r.pack + 2 * l.c_out : nullptr, c, sums, r.rows, (int32_t)l.c_out, half ? 1 : 0, r.x, r.x_rows,
                           (int32_t)l.c_in, w, (int32_t)l.k, (const int32_t *)m.nbmaps.data_ptr(), (const int32_t *)m.nboffs.data_ptr(),
                           m.total, o.transposed ? 0 : 1, (const int32_t *)table.data_ptr(), drows, o.transposed ? 1 : 0, grad_feat,
                           grad_res, (float *)grad_w.data_ptr(), gw_ptr, gb_ptr, &bopts, ws.data_ptr(), (size_t)ws.numel(),
                           (ts_stream_t)stream),
              "ts_conv_block_backward");
      };
      if (split) {
        call(COMM_PRE);                   // this rank's sums of the BatchNorm backward
        group_sum(st->group_id, view_of(garena, sums, 1, 2 * l.c_out, at::kDouble).view({-1}));
      }
      const int64_t t_api = now_ns();
      call(split ? COMM_POST : (void *)comm);
      host_clock.ns_bwd_api += now_ns() - t_api;
      if (side_job) wg_worker.push(std::move(side_job));      // (the call has recorded the slot's ready event)
      if (feat_add) view_of(garena, grad_feat, drows, l.c_in, dt).add_(view_of(garena, G[o.src], drows, l.c_in, dt));
      if (grad_feat) G[o.src] = grad_feat;
      if (res_add) view_of(garena, grad_res, r.rows, l.c_out, dt).add_(view_of(garena, G[o.aux], r.rows, l.c_out, dt));
      if (grad_res) G[o.aux] = grad_res;
      if (!r.use_dest) {
        gbw = bn_grads.narrow(0, (int64_t)off_gwb[i], l.c_out);
        gbb = bn_grads.narrow(0, (int64_t)off_gwb[i] + l.c_out, l.c_out);
      }
      gparams[3 * o.layer] = std::move(grad_w);
      gparams[3 * o.layer + 1] = std::move(gbw);
      gparams[3 * o.layer + 2] = std::move(gbb);
    }
    torch::autograd::variable_list out;
    for (int i = 0; i < n_in; ++i) {
      at::Tensor gi = gin[i];
      if (!gi.defined() && need[i] && G[i]) gi = view_of(garena, G[i], rows[i], ch[i], dt);
      if (gi.defined() && gi.scalar_type() != st->in_dtypes[i]) gi = gi.to(st->in_dtypes[i]);
      out.push_back(std::move(gi));
    }
    if (!st->direct)
      for (auto &t : gparams) out.push_back(std::move(t));
    for (int i = 0; i < 8; ++i) out.push_back(at::Tensor());      // prog, geom, half, stream, comm, group_id, grad_epoch, direct
    if (st->direct) {
      gparams.clear();                    // (the aliases of the slots: the reducer's own views become p.grad)
      p.deliver();
    }
    host_clock.ns_bwd += now_ns() - t_in;
    host_clock.n_bwd += (int64_t)p.layers.size();
    return out;
  }
};

// (callers hold no interpreter lock: the issue loop touches no Python object)
inline at::Tensor run_inner(const std::vector<at::Tensor> &inputs, std::shared_ptr<Program> prog, std::shared_ptr<Geometry> geom, bool half,
                            int64_t stream, int64_t comm, int64_t group_id, int64_t grad_epoch) {
  // (the node must still be recorded: with its parameters off the graph that takes an input that requires a gradient)
  bool wanted = false;
  for (const at::Tensor &x : inputs) wanted = wanted || x.requires_grad();
  if (at::GradMode::is_enabled() && wanted) prog->refuse_second_pass(grad_epoch);
  const bool direct = at::GradMode::is_enabled() && wanted && prog->direct_ok(grad_epoch);
  const std::vector<at::Tensor> params = direct ? std::vector<at::Tensor>() : prog->parameters();
  return StageRun::apply(at::TensorList(inputs), at::TensorList(params), prog, geom, half, stream, comm, group_id, grad_epoch, direct)[0];
}

inline at::Tensor run(const std::vector<at::Tensor> &inputs, std::shared_ptr<Program> prog, std::shared_ptr<Geometry> geom, bool half,
                      int64_t stream, int64_t comm, int64_t group_id, int64_t grad_epoch) {
  TORCH_CHECK(api.handle, "taseg_amd fast path: load_backend() has not been called");
  // another thread (the data stage of the next batch) may have the interpreter while the stage is issued
  py::gil_scoped_release nogil;
  return run_inner(inputs, std::move(prog), std::move(geom), half, stream, comm, group_id, grad_epoch);
}

// The evaluation form (modules in eval mode, no graph: minkunet.py:435-455, R/train.py:452-540): the same op list on
// ts_conv_block_eval with the running statistics.
inline at::Tensor run_eval_inner(const std::vector<at::Tensor> &inputs_, std::shared_ptr<Program> prog, std::shared_ptr<Geometry> geom,
                                 bool half, int64_t stream) {
  at::NoGradGuard nograd;
  Program &p = *prog;
  const Geometry &g = *geom;
  TORCH_CHECK(g.half == half, "stage run: the geometry was resolved for the other storage mode");
  const auto dt = half ? at::kHalf : at::kFloat;
  const size_t es = half ? 2 : 4;
  std::vector<at::Tensor> inputs;
  for (const at::Tensor &x : inputs_) inputs.push_back(x.contiguous().to(dt));
  std::vector<int64_t> rows, ch;
  shapes(p, g, inputs, rows, ch);
  size_t bytes = 0;
  for (size_t i = 0; i < p.ops.size(); ++i) {
    const Op &o = p.ops[i];
    bytes += up((size_t)rows[o.dst] * ch[o.dst] * es);
    if (o.kind == BLOCK) {
      const Layer &l = p.layers[o.layer];
      if (half && !(l.half16.defined() && l.half16.numel() == l.k * l.c_in * l.c_out)) bytes += up((size_t)l.k * l.c_in * l.c_out * 2);
    }
  }
  const at::Tensor &like = inputs[0];
  at::Tensor arena = at::empty({(int64_t)bytes}, like.options().dtype(at::kByte));
  char *cur = (char *)arena.data_ptr();
  auto take = [&](size_t n) {
    void *r = cur;
    cur += up(n);
    return r;
  };
  at::Tensor ws = workspace(block_ws_bytes(p, g, half), like, stream);
  std::vector<const void *> reg(p.n_regs, nullptr);
  for (int i = 0; i < p.n_inputs; ++i) reg[i] = inputs[i].data_ptr();
  for (size_t i = 0; i < p.ops.size(); ++i) {
    const Op &o = p.ops[i];
    void *dst = take((size_t)rows[o.dst] * ch[o.dst] * es);
    if (o.kind == CAT) {
      check(api.cat_cols(reg[o.src], ch[o.src] * (int64_t)es, reg[o.aux], ch[o.aux] * (int64_t)es, rows[o.dst], dst, (ts_stream_t)stream),
            "ts_cat_cols");
      reg[o.dst] = dst;
      continue;
    }
    Layer &l = p.layers[o.layer];
    const MapRef &m = g.maps[g.op_map[i]];
    TORCH_CHECK(l.rmean.defined() && l.rvar.defined() && l.rmean.scalar_type() == at::kFloat, "stage run: evaluation needs float32 running statistics");
    // 1 / sqrt(running_var + eps), recomputed when the buffer has changed
    const uint32_t ver = l.rvar._version();
    if (!l.invstd.defined() || l.invstd_version != ver || l.invstd_src != l.rvar.data_ptr() || l.invstd.device() != l.rvar.device()) {
      l.invstd = at::rsqrt(l.rvar.to(at::kFloat) + l.eps);
      l.invstd_version = ver;
      l.invstd_src = l.rvar.data_ptr();
    }
    const bool kept16 = half && l.half16.defined() && l.half16.numel() == l.k * l.c_in * l.c_out;
    void *w16 = nullptr;
    if (half) w16 = kept16 ? l.half16.data_ptr() : take((size_t)l.k * l.c_in * l.c_out * 2);
    const at::Tensor &table = o.transposed ? m.pos_in : m.pos_out;
    const void *pl = (!half && l.planes32.defined()) ? l.planes32.data_ptr() : nullptr;
    TsConvBlockOpts bopts = {g.plan_f[i].get(), nullptr, pl, kept16 ? 1 : 0, nullptr, nullptr, nullptr, 0, 0, 0, l.natural ? 1 : 0};
    check(api.eval(reg[o.src], rows[o.src], (int32_t)l.c_in, l.kernel.data_ptr<float>(), (int32_t)l.k, (const int32_t *)m.nbmaps.data_ptr(),
                   (const int32_t *)m.nboffs.data_ptr(), m.total, o.transposed ? 1 : 0, (const int32_t *)table.data_ptr(), rows[o.dst],
                   (int32_t)l.c_out, o.aux >= 0 ? reg[o.aux] : nullptr, (const float *)l.bn_w.data_ptr(), (const float *)l.bn_b.data_ptr(),
                   (const float *)l.rmean.data_ptr(), (const float *)l.invstd.data_ptr(), o.relu ? 1 : 0, half ? 1 : 0, dst, w16, &bopts,
                   ws.data_ptr(), (size_t)ws.numel(), (ts_stream_t)stream),
          "ts_conv_block_eval");
    reg[o.dst] = dst;
  }
  return view_of(arena, reg[p.out_reg], rows[p.out_reg], ch[p.out_reg], dt);
}

inline at::Tensor run_eval(const std::vector<at::Tensor> &inputs, std::shared_ptr<Program> prog, std::shared_ptr<Geometry> geom, bool half,
                           int64_t stream) {
  TORCH_CHECK(api.handle, "taseg_amd fast path: load_backend() has not been called");
  py::gil_scoped_release nogil;
  return run_eval_inner(inputs, std::move(prog), std::move(geom), half, stream);
}

// stage1 .. stage4, up1 .. up4 of the U-Net pass (minkunet.py:393-422) in ONE call: the eight stage programs in order, the two
// dropouts between them (at::dropout = torch.nn.functional.dropout: same generator use), the skip connections handed on as matrices;
// returns the three feature matrices the point head devoxelises (stride-16 encoder output, stride-4 and stride-1 decoder outputs,
// each BEFORE its dropout: minkunet.py:400-412).  The whole backbone is issued without the interpreter lock.
inline std::vector<at::Tensor> unet_run(const at::Tensor &f0, const std::vector<std::shared_ptr<Program>> &progs,
                                        const std::vector<std::shared_ptr<Geometry>> &geoms, bool training, bool half, int64_t stream,
                                        int64_t comm, int64_t group_id, int64_t grad_epoch, double dropout_p) {
  TORCH_CHECK(api.handle, "taseg_amd fast path: load_backend() has not been called");
  TORCH_CHECK(progs.size() == 8 && geoms.size() == 8, "unet_run: eight stage programs (stage1-4, up1-4) and their geometries");
  py::gil_scoped_release nogil;
  auto go = [&](int i, std::vector<at::Tensor> in) {
    return training ? run_inner(in, progs[i], geoms[i], half, stream, comm, group_id, grad_epoch) : run_eval_inner(in, progs[i], geoms[i], half, stream);
  };
  auto drop = [&](const at::Tensor &x) { return at::dropout(x, dropout_p, training); };
  at::Tensor f1 = go(0, {f0}), f2 = go(1, {f1}), f3 = go(2, {f2}), f4 = go(3, {f3});
  at::Tensor y1 = go(4, {drop(f4), f3}), y2 = go(5, {y1, f2}), y3 = go(6, {drop(y2), f1}), y4 = go(7, {y3, f0});
  return {f4, y2, y4};
}

}  // namespace stage
