"""Stage programs: the encoder / decoder stages of the MinkUNet family issued from C++ (csrc/fastpath/stage_program.h).

The reference's backbone (R/pcseg/model/segmentor/voxel/minkunet/minkunet.py:186-356; forward :393-422) is nine stages - stem,
stage1-4 (`BasicConvolutionBlock(k2, s2)` + ResidualBlocks), up1-4 (`BasicDeconvolutionBlock`, `torchsparse.cat` with the skip
connection, ResidualBlocks).  Module by module that is 63 conv -> BatchNorm [-> residual] [-> ReLU] block calls per pass, each behind
a Python call chain, an autograd node and ~10 allocator calls: ~130 us of host time per block against ~30 us of launches.  Here a
stage is compiled ONCE per model into an op list over a few feature matrices (`StageProgram`), its kernel maps / class plans are
resolved ONCE per batch (`StageGeometry`, cached in the index plan) and one call runs the whole stage as ONE autograd node (or, in
evaluation mode, without a graph).  Same backend calls in the same order as `spnn.conv_bn_act` block by block - the per-module
path stays behind `TASEG_STAGE_PROGRAM=0` and serves every model the programs do not (Bottleneck blocks, hooks on the conv
modules, channel counts off the full-tile paths, frozen BatchNorm layers ...).
"""
import os
import weakref

import torch
from torch import nn

from taseg_amd import _fast
from taseg_amd import backend as B
from taseg_amd import parallel as _parallel
from taseg_amd import planes as _planes
from taseg_amd.options import options
from taseg_amd.torchsparse.nn import functional as spF
from taseg_amd.torchsparse.nn import modules as spM
from taseg_amd.torchsparse.utils import make_ntuple

__all__ = ["StagePrograms", "enabled", "programs_of", "forget", "compiled"]

_ON = options.stage_program
_DIRECT_GRADS = options.direct_grads
_ONES = (1, 1, 1)
_BLOCK, _CAT = 0, 1


def enabled() -> bool:
    return _ON and _fast.module() is not None and hasattr(_fast.module(), "stage_run")


class _Unsupported(Exception):
    pass


# model -> StagePrograms | False (this model's structure is not served).  Kept here, not on the module: a module's __dict__ is
# what copy.deepcopy / pickle walk, and a compiled program is neither
_PROGRAMS = weakref.WeakKeyDictionary()


def programs_of(model, compile=True):
    """the compiled programs of `model`, compiling them at the first call; None when its structure is not served"""
    hit = _PROGRAMS.get(model)
    if hit is None and compile:
        try:
            hit = StagePrograms(model)
        except _Unsupported:
            hit = False
        _PROGRAMS[model] = hit
    return hit or None


def forget(model):
    _PROGRAMS.pop(model, None)


def compiled(model) -> bool:
    return bool(_PROGRAMS.get(model))


class _Stage:
    """one compiled stage: the C++ program, its (conv, bn) modules and the kernel-map key of every op"""
    __slots__ = ("program", "layers", "ops", "keys", "in_strides", "planes_state", "dest_state", "bns", "convs", "modules", "entries")


def _tup(v):
    return tuple(make_ntuple(v, ndim=3))


class StagePrograms:
    def __init__(self, model):
        fast = _fast.module()
        self.fast = fast
        self.stale = False
        self.half = False                 # storage mode of the last pass (what prepare() resolves the next batch for)
        self.stages = {}
        from .minkunet import BasicConvolutionBlock, BasicDeconvolutionBlock, ResidualBlock
        self._kinds = (BasicConvolutionBlock, BasicDeconvolutionBlock, ResidualBlock)
        for name, stride in (("stage1", 1), ("stage2", 2), ("stage3", 4), ("stage4", 8)):
            self.stages[name] = self._compile(list(getattr(model, name).children()), (stride,), cat_after_first=False)
        for name, stride in (("up1", 16), ("up2", 8), ("up3", 4), ("up4", 2)):
            up = getattr(model, name)
            self.stages[name] = self._compile([up[0]] + list(up[1].children()), (stride, stride // 2), cat_after_first=True)
        for name, st in self.stages.items():
            # every module the program stands in for: a forward / backward hook on any of them must still fire (module path then)
            st.modules = list(getattr(model, name).modules())
        self._finish_compile()

    # ------------------------------------------------------------------ model side
    def _compile(self, blocks, in_strides, cat_after_first):
        conv_block, deconv_block, res_block = self._kinds
        st = _Stage()
        st.layers, st.ops, st.keys = [], [], []
        stride = {i: s for i, s in enumerate(in_strides)}
        nreg = [len(in_strides)]

        def block(conv, bn, src, aux, relu):
            if not isinstance(conv, spM.Conv3d) or not isinstance(bn, (nn.BatchNorm1d, nn.SyncBatchNorm)):
                raise _Unsupported("not a Conv3d + BatchNorm pair")
            if conv.bias is not None or not bn.affine or bn.momentum is None or bn.weight.dtype != torch.float32 \
                    or conv.kernel.dtype != torch.float32:
                raise _Unsupported("bias / non-affine / cumulative-average BatchNorm / non-fp32 parameters")
            ks, cs, dil = _tup(conv.kernel_size), _tup(conv.stride), _tup(conv.dilation)
            s = stride[src]
            if ks == _ONES:
                if cs != _ONES or dil != _ONES or conv.transposed:
                    raise _Unsupported("strided 1x1x1 convolution")
                key, out_stride = ("identity", s), s
            elif not conv.transposed:
                out_stride = s * cs[0]
                key = ((s,) * 3, ks, cs, dil)
            else:
                out_stride = s // cs[0]
                key = ((out_stride,) * 3, ks, cs, dil)
            if len(set(cs)) != 1 or conv.kernel_volume > 63:
                raise _Unsupported("anisotropic stride / kernel volume")
            dst = nreg[0]
            nreg[0] += 1
            stride[dst] = out_stride
            st.layers.append((conv, bn))
            st.ops.append((_BLOCK, len(st.layers) - 1, src, dst, -1 if aux is None else aux, bool(conv.transposed), bool(relu)))
            st.keys.append(key)
            return dst

        cur = 0
        for i, blk in enumerate(blocks):
            if type(blk) in (conv_block, deconv_block):
                net = blk.net
                cur = block(net[0], net[1], cur, None, True)
            elif type(blk) is res_block:
                net, down = blk.net, blk.downsample
                h = block(net[0], net[1], cur, None, True)
                if isinstance(down, nn.Identity):
                    sc = cur
                else:
                    sc = block(down[0], down[1], cur, None, False)
                cur = block(net[3], net[4], h, sc, True)
            else:
                raise _Unsupported(f"block type {type(blk).__name__}")
            if cat_after_first and i == 0:
                dst = nreg[0]
                nreg[0] += 1
                if stride[cur] != stride[1]:
                    raise _Unsupported("skip connection at another stride")
                stride[dst] = stride[cur]
                st.ops.append((_CAT, -1, cur, dst, 1, False, False))       # torchsparse.cat([up(x), skip])
                st.keys.append(None)
                cur = dst
        st.in_strides = tuple(in_strides)
        st.convs = [c for c, _ in st.layers]
        st.bns = [b for _, b in st.layers]
        layers = []
        for conv, bn in st.layers:
            track = bn.track_running_stats and bn.running_mean is not None
            layers.append((conv.kernel, bn.weight, bn.bias, bn.running_mean if track else None, bn.running_var if track else None,
                           bn.num_batches_tracked if track else None, float(bn.momentum), float(bn.eps)))
        st.entries = {False: None, True: None}      # per storage mode: [(planes entry | None, weight)] as last handed to the program
        st.program = self.fast.StageProgram(len(in_strides), cur, st.ops, layers)
        st.planes_state = {False: None, True: None}
        st.dest_state = None
        return st

    # ------------------------------------------------------------------ batch side
    def geometry(self, name, plan, half):
        """StageGeometry of stage `name` on this index plan for this storage mode (cached in the plan)"""
        cache = plan.get("_stage_geom")
        if cache is None:
            cache = plan["_stage_geom"] = {}
        key = (id(self), name, half)
        geom = cache.get(key)
        if geom is None:
            geom = cache[key] = self._build_geometry(self.stages[name], plan, half)
        return geom

    def prepare(self, plan, half):
        """all stages at once (the data stage calls this with the index plan, off the training thread)"""
        for name in self.stages:
            self.geometry(name, plan, half)

    def _build_geometry(self, st, plan, half):
        kmaps, cmaps = plan["kmaps"], plan["cmaps"]
        if any(km.dup for km in kmaps.values()):
            raise _Unsupported("a coordinate set with a duplicate: the block calls' list-form input gradient does not apply")
        rows = {i: cmaps[(s,) * 3].shape[0] for i, s in enumerate(st.in_strides)}
        dev = plan["coords"].device
        maps, op_map, pf, pfm, pd, pdm = [], [], [], [], [], []
        for op, key in zip(st.ops, st.keys):
            kind, layer, src, dst, aux, transposed, _relu = op
            if kind == _CAT:
                rows[dst] = rows[src]
                op_map.append(-1)
                for lst in (pf, pfm, pd, pdm):
                    lst.append([])
                continue
            conv = st.convs[layer]
            if key[0] == "identity":
                km = spF.identity_map(rows[src], dev)
            else:
                km = kmaps.get(key)
                if km is None:
                    raise _Unsupported(f"the index plan holds no kernel map {key}")
            n_in, n_out = km.sizes
            rows[dst] = n_in if transposed else n_out
            k = conv.kernel
            c_in, c_out = (k.shape[0], k.shape[1]) if k.dim() == 2 else (k.shape[1], k.shape[2])
            plan_f, plan_d = km.plans_for(transposed, c_in, c_out, half)
            maps.append((km.nbmaps_buf, km.nboffs, km.pos_out, km.pos_in, int(km.total), int(n_in), int(n_out)))
            op_map.append(len(maps) - 1)
            a, b = spM._plan_args(plan_f)
            pf.append(a)
            pfm.append(b)
            a, b = spM._plan_args(plan_d)
            pd.append(a)
            pdm.append(b)
        return self.fast.StageGeometry(maps, op_map, pf, pfm, pd, pdm, bool(half))

    # ------------------------------------------------------------------ per call
    def usable(self, feats, training, grad):
        """can the programs serve this pass?  (training with a graph, or evaluation without one; full-tile channel counts).  Runs
        once per pass over ~300 modules: only dictionary truth tests and identity comparisons on objects collected at compile time"""
        if not feats.is_cuda or (grad and not training):
            return False           # (eval-mode BatchNorm with a graph: the module path)
        # (training-mode BatchNorm without a graph - the frozen teacher of MinkUNetMsKd, minkunet_ms_kd.py:533 - runs the training
        # program: its autograd node records nothing under no_grad and the forward arena goes with the call)
        half = spF._amp_half(feats)
        if not half and feats.dtype != torch.float32:
            return False
        if not (self.ok_half if half else self.ok_f32) or (not training and not self.ok_eval):
            return False
        for d in self.hook_dicts:              # a forward / backward hook on any module the programs stand in for must still fire
            if d:
                return False
        for bn in self.all_bns:
            if bn.training != training:
                return False
        # the programs hold these tensor objects: a module whose parameter / buffer objects were replaced since (`.to()` swaps
        # buffers, an assignment swaps a parameter) needs a new program
        for d, key, held in self.held_slots:
            if d.get(key) is not held:
                self.stale = True
                return False
        return True

    def _finish_compile(self):
        """what usable() reads: shape rules decided once, the hook dictionaries and (dictionary, key, tensor) slots of every module"""
        layers = [(c, b) for st in self.stages.values() for c, b in st.layers]
        # run / run_unet take the statistics scope (SyncBatchNorm or not, process group) from the FIRST layer for all of them: a
        # partially converted model, or layers on different process groups, keep the per-module path, which decides per layer
        kinds = {(isinstance(b, nn.SyncBatchNorm), id(getattr(b, "process_group", None))) for _, b in layers}
        if len(kinds) > 1:
            raise _Unsupported("BatchNorm layers of mixed type / process group")
        shapes = [(c.kernel.shape[-2], c.kernel.shape[-1], c.kernel.dim()) for c, _ in layers]
        base = all(co <= 1024 and (dim == 3 or spF._dense_ok(ci, co)) for ci, co, dim in shapes)
        self.ok_f32 = base and all(co % 4 == 0 for _, co, _ in shapes)
        self.ok_half = base and all(ci % 32 == 0 and co % 32 == 0 for ci, co, _ in shapes)
        self.ok_eval = all(b.track_running_stats and b._buffers.get("running_var") is not None
                           and b._buffers["running_mean"].dtype == torch.float32 for _, b in layers)
        self.all_bns = [b for _, b in layers]
        self.hook_dicts = []
        for st in self.stages.values():
            for m in st.modules:
                self.hook_dicts += [m._forward_hooks, m._forward_pre_hooks, m._backward_hooks, m._backward_pre_hooks]
        self.held_slots = []
        for c, b in layers:
            self.held_slots += [(c._parameters, "kernel", c._parameters["kernel"]), (b._parameters, "weight", b._parameters["weight"]),
                                (b._parameters, "bias", b._parameters["bias"]), (b._buffers, "running_mean", b._buffers.get("running_mean")),
                                (b._buffers, "running_var", b._buffers.get("running_var"))]

    def _refresh(self, st, half):
        """planes / kept half copies of the stage's weights in step with the weights (taseg_amd/planes.py refreshes every stale
        weight of the model in one batch of launches at the first stale one it is asked for), gradient-bucket slots of the
        parameters as the reducer names them"""
        stream = B.L.stream()
        epoch = _planes._epoch
        ents = st.entries[half]
        fresh = ents is not None
        if fresh:
            # (taseg_amd/planes._Entry.fresh inlined: 61 of these per pass)
            for e, w in ents:
                if e is not None and (e.ptr != w.data_ptr() or e.version != w._version or e.epoch != epoch or e.stream != stream):
                    fresh = False
                    break
        if not fresh:
            get, table = (_planes.half_for, _planes._half_entries) if half else (_planes.planes_for, _planes._entries)
            cur = [get(c.kernel) for c in st.convs]
            st.entries[half] = [(table.get(id(c.kernel)) if t is not None else None, c.kernel) for c, t in zip(st.convs, cur)]
            ident = tuple(id(t) for t in cur)
            if st.planes_state[half] != ident:
                none = [None] * len(cur)
                st.program.set_planes(none if half else cur, cur if half else none)
                st.planes_state[True], st.planes_state[False] = (ident, None) if half else (None, ident)
        first = getattr(st.convs[0].kernel, "_taseg_grad_dest", None)
        if st.dest_state is not first:
            dests = []
            for conv, bn in st.layers:
                dests += [getattr(conv.kernel, "_taseg_grad_dest", None), getattr(bn.weight, "_taseg_grad_dest", None),
                          getattr(bn.bias, "_taseg_grad_dest", None)]
            st.program.set_grad_dests(dests)
            st.dest_state = first
            # with a reducer behind the slots the stage delivers its gradients itself, once per backward pass (TASEG_DIRECT_GRADS=0:
            # through autograd's AccumulateGrad and the reducer's per-parameter hooks)
            reducer = _parallel.reducer_of(st.convs[0].kernel)
            params = [p for conv, bn in st.layers for p in (conv.kernel, bn.weight, bn.bias)]
            # (direct delivery bypasses AccumulateGrad: tensor hooks a user put on a parameter would never fire - the reducer's own
            # post-accumulate hook is the one expected there)
            hooked = any(p._backward_hooks or len(p._post_accumulate_grad_hooks or ()) > 1 for p in params)
            if reducer is not None and _DIRECT_GRADS and not hooked and all(d is not None for d in dests) \
                    and all(_parallel.reducer_of(p) is reducer for p in params):
                rref = weakref.ref(reducer)

                def _live():
                    red = rref()
                    if red is None:
                        raise RuntimeError("taseg_amd stage program: the GradBucketReducer / FlatSGD that owns this model's gradient slots "
                                           "no longer exists; build a new one before the next backward pass")
                    return red
                st.program.set_deliver(lambda: _live().deliver(params), lambda: _live().check_open(params))
            else:
                st.program.set_deliver(None, None)
        return stream

    ORDER = ("stage1", "stage2", "stage3", "stage4", "up1", "up2", "up3", "up4")

    def run_unet(self, f0, plan, training, dropout_p):
        """stage1 .. up4 in ONE native call (csrc/fastpath/stage_program.h::unet_run): returns the features of the stride-16
        encoder output and the stride-4 / stride-1 decoder outputs, each before its dropout"""
        half = spF._amp_half(f0)
        progs, geoms, stream = [], [], None
        for name in self.ORDER:
            st = self.stages[name]
            geoms.append(self.geometry(name, plan, half))
            stream = self._refresh(st, half)
            progs.append(st.program)
        comm, group_id = 0, -1
        if training:
            group = spM._sync_group(self.stages["stage1"].bns[0])
            if group is not None:
                from taseg_amd.rccl import direct_comm
                spM._require_rows(f0, group)
                c = direct_comm(group)
                if c is not None:
                    comm = c.value or 0
                else:
                    group_id = spM._group_id(self.fast, group)
        return self.fast.unet_run(f0, progs, geoms, bool(training), half, stream, comm, group_id, _parallel.grad_epoch(), float(dropout_p))

    def run(self, name, inputs, plan, training):
        st = self.stages[name]
        half = spF._amp_half(inputs[0])
        geom = self.geometry(name, plan, half)
        stream = self._refresh(st, half)
        if not training:
            return self.fast.stage_run_eval(list(inputs), st.program, geom, half, stream)
        comm, group_id = 0, -1
        group = spM._sync_group(st.bns[0])
        if group is not None:
            from taseg_amd.rccl import direct_comm
            spM._require_rows(inputs[0], group)
            c = direct_comm(group)
            if c is not None:
                comm = c.value or 0
            else:
                group_id = spM._group_id(self.fast, group)
        return self.fast.stage_run(list(inputs), st.program, geom, half, stream, comm, group_id, _parallel.grad_epoch())
