"""MinkUNet segmentor (reference pcseg/model/segmentor/voxel/minkunet/minkunet.py:186-458).

Module tree, attribute names and therefore every state_dict key / shape are identical to
the reference (`stem.0.kernel`, `stage1.0.net.0.kernel`, `stage2.1.downsample.0.kernel`,
`up1.1.0.net.0.kernel`, `classifier.0.weight`, BN `weight/bias/running_*` ...), so reference
checkpoints load here and vice versa.  The sparse ops run on the HIP backend through
`taseg_amd.torchsparse`.
"""
import torch
from torch import nn

from taseg_amd import torchsparse
from taseg_amd.torchsparse import PointTensor, SparseTensor
from taseg_amd.torchsparse import nn as spnn
from taseg_amd.torchsparse.nn import functional as spF
from taseg_amd.pcseg.loss import Losses
from ...base_segmentors import BaseSegmentor
from taseg_amd import backend as B
from taseg_amd import _fast
from taseg_amd.options import options
from .utils import voxel_to_point, voxelize_index
from . import stage_program as _SP

__all__ = ["MinkUNet", "unvoxelise_predictions"]

import contextlib
import os as _os
_DEVOX_ATOMIC = options.devox_atomic
_DEVOX_CELLS = options.devox_cells     # stride-16 devoxelize backward: cell-reduced two-stage sum


def _coarse_devox_plan(idx, w, n_vox):
    """How the stride-16 devoxelize backward walks its map (~700 contributions per voxel, ~4 voxels per point): by default
    the cell-reduced two-stage sum (every gradient row read once, backend.devox_cells); TASEG_DEVOX_CELLS=0: the plain
    gather along the inverse map (every row read once per live corner)."""
    return B.devox_cells(idx, w, n_vox) if _DEVOX_CELLS else B.devox_csr(idx, w, n_vox)


class SyncBatchNorm(spnn.SyncBatchNorm):
    """nn.SyncBatchNorm applied to SparseTensor features (minkunet.py:23-25); training-mode reductions on HIP."""


class BatchNorm(spnn.BatchNorm):
    """nn.BatchNorm1d applied to SparseTensor features (minkunet.py:27-29); training-mode reductions on HIP."""


def _norm(channels: int, if_dist: bool) -> nn.Module:
    return SyncBatchNorm(channels) if if_dist else BatchNorm(channels)


class BasicConvolutionBlock(nn.Module):
    """conv -> BN -> ReLU (minkunet.py:31-54)."""

    def __init__(self, inc, outc, ks=3, stride=1, dilation=1, if_dist=False):
        super().__init__()
        self.net = nn.Sequential(
            spnn.Conv3d(inc, outc, kernel_size=ks, dilation=dilation, stride=stride),
            _norm(outc, if_dist),
            spnn.ReLU(True),
        )

    def forward(self, x):
        # (children straight from the module dictionaries: `self.net[0]` is nn.Module.__getattr__ + nn.Sequential.__getitem__,
        # ~2 us each and five of them per residual block)
        net = self._modules["net"]._modules
        return spnn.conv_bn_act(net["0"], net["1"], x, relu=True)      # conv -> BN -> ReLU, one node


class BasicDeconvolutionBlock(nn.Module):
    """transposed conv -> BN -> ReLU (minkunet.py:57-80)."""

    def __init__(self, inc, outc, ks=3, stride=1, if_dist=False):
        super().__init__()
        self.net = nn.Sequential(
            spnn.Conv3d(inc, outc, kernel_size=ks, stride=stride, transposed=True),
            _norm(outc, if_dist),
            spnn.ReLU(True),
        )

    def forward(self, x):
        net = self._modules["net"]._modules
        return spnn.conv_bn_act(net["0"], net["1"], x, relu=True)


def _shortcut(inc, outc, stride, if_dist):
    if inc == outc and stride == 1:
        return nn.Identity()
    return nn.Sequential(spnn.Conv3d(inc, outc, kernel_size=1, dilation=1, stride=stride), _norm(outc, if_dist))


class ResidualBlock(nn.Module):
    """two 3x3x3 convs + identity / 1x1x1 shortcut (minkunet.py:83-129)."""
    expansion = 1

    def __init__(self, inc, outc, ks=3, stride=1, dilation=1, if_dist=False):
        super().__init__()
        self.net = nn.Sequential(
            spnn.Conv3d(inc, outc, kernel_size=ks, dilation=dilation, stride=stride),
            _norm(outc, if_dist),
            spnn.ReLU(True),
            spnn.Conv3d(outc, outc, kernel_size=ks, dilation=dilation, stride=1),
            _norm(outc, if_dist),
        )
        self.downsample = _shortcut(inc, outc * self.expansion, stride, if_dist)
        self.relu = spnn.ReLU(True)

    def forward(self, x):
        # relu(net(x) + downsample(x)) with the BN / add / ReLU tails fused (same module parameters / buffers)
        # (the first block hands x through: the shortcut's gradient then lands in the store of conv1's input gradient)
        mods = self._modules
        net, down = mods["net"]._modules, mods["downsample"]
        h, x = spnn.conv_bn_act(net["0"], net["1"], x, relu=True, passthrough=True)
        # (the 1x1x1 shortcut + its BatchNorm: one block call on the identity rulebook, spnn.conv_bn_act)
        if isinstance(down, nn.Identity):
            shortcut = x
        else:
            dm = down._modules
            shortcut = spnn.conv_bn_act(dm["0"], dm["1"], x, relu=False)
        return spnn.conv_bn_act(net["3"], net["4"], h, relu=True, residual=shortcut)


class Bottleneck(nn.Module):
    """1x1x1 -> 3x3x3 -> 1x1x1 (x4) with shortcut (minkunet.py:132-183)."""
    expansion = 4

    def __init__(self, inc, outc, ks=3, stride=1, dilation=1, if_dist=False):
        super().__init__()
        wide = outc * self.expansion
        self.net = nn.Sequential(
            spnn.Conv3d(inc, outc, kernel_size=1, bias=False),
            _norm(outc, if_dist),
            spnn.Conv3d(outc, outc, kernel_size=ks, stride=stride, bias=False, dilation=dilation),
            _norm(outc, if_dist),
            spnn.Conv3d(outc, wide, kernel_size=1, bias=False),
            _norm(wide, if_dist),
        )
        self.downsample = _shortcut(inc, wide, stride, if_dist)
        self.relu = spnn.ReLU(True)

    def forward(self, x):
        return self.relu(self.net(x) + self.downsample(x))


class LazyScalar:
    """float-like view of a device scalar: the host sync happens only if somebody formats or
    converts it (the reference calls `loss.item()` inside forward, stalling the launch queue
    before backward; TASeg's trainer only logs the value)."""

    def __init__(self, t):
        self._t = t.detach()
        self._v = None

    def item(self):
        if self._v is None:
            self._v = float(self._t.item())
        return self._v

    __float__ = item

    def __format__(self, spec):
        return format(self.item(), spec)

    def __repr__(self):
        return repr(self.item())

    def _num(self, other):
        return float(other)

    def __add__(self, o):
        return self.item() + self._num(o)

    __radd__ = __add__

    def __mul__(self, o):
        return self.item() * self._num(o)

    __rmul__ = __mul__

    def __truediv__(self, o):
        return self.item() / self._num(o)


class MinkUNetBackbone(BaseSegmentor):
    """Everything MinkUNet and MinkUNetMs share: construction (minkunet.py:187-356 ==
    minkunet_ms.py:187-356) and the encoder / decoder / point-head pass (minkunet.py:393-422)."""

    def __init__(self, model_cfgs, num_class: int):
        super().__init__(model_cfgs, num_class)
        self.in_feature_dim = model_cfgs.IN_FEATURE_DIM
        self.num_layer = model_cfgs.get("NUM_LAYER", [2, 3, 4, 6, 2, 2, 2, 2])
        self.block = {"ResBlock": ResidualBlock, "Bottleneck": Bottleneck}[model_cfgs.get("BLOCK", "Bottleneck")]
        cr = model_cfgs.get("cr", 1.0)
        cs = [int(cr * x) for x in model_cfgs.get("PLANES", [32, 32, 64, 128, 256, 256, 128, 96, 96])]
        self.pres = model_cfgs.get("pres", 0.05)
        self.vres = model_cfgs.get("vres", 0.05)
        if_dist = model_cfgs.IF_DIST
        exp = self.block.expansion

        self.stem = nn.Sequential(
            spnn.Conv3d(self.in_feature_dim, cs[0], kernel_size=3, stride=1), _norm(cs[0], if_dist), spnn.ReLU(True),
            spnn.Conv3d(cs[0], cs[0], kernel_size=3, stride=1), _norm(cs[0], if_dist), spnn.ReLU(True),
        )
        self.in_channels = cs[0]

        def encoder_stage(width, depth):
            return nn.Sequential(
                BasicConvolutionBlock(self.in_channels, self.in_channels, ks=2, stride=2, dilation=1, if_dist=if_dist),
                *self._make_layer(self.block, width, depth, if_dist=if_dist))

        self.stage1 = encoder_stage(cs[1], self.num_layer[0])
        self.stage2 = encoder_stage(cs[2], self.num_layer[1])
        self.stage3 = encoder_stage(cs[3], self.num_layer[2])
        self.stage4 = encoder_stage(cs[4], self.num_layer[3])

        def decoder_stage(width, skip_channels, depth):
            up = BasicDeconvolutionBlock(self.in_channels, width, ks=2, stride=2, if_dist=if_dist)
            self.in_channels = width + skip_channels
            body = nn.Sequential(*self._make_layer(self.block, width, depth, if_dist=if_dist))
            return nn.ModuleList([up, body])

        self.up1 = decoder_stage(cs[5], cs[3] * exp, self.num_layer[4])
        self.up2 = decoder_stage(cs[6], cs[2] * exp, self.num_layer[5])
        self.up3 = decoder_stage(cs[7], cs[1] * exp, self.num_layer[6])
        self.up4 = decoder_stage(cs[8], cs[0], self.num_layer[7])

        self.classifier = nn.Sequential(spnn.PointLinear((cs[4] + cs[6] + cs[8]) * exp, self.num_class))
        self.weight_initialization()
        self.dropout = nn.Dropout(model_cfgs.get("DROPOUT_P", 0.3), True)

        default_loss = {"LOSS_TYPES": ["CELoss", "LovLoss"], "LOSS_WEIGHTS": [1.0, 1.0], "KNN": 10}
        loss_cfg = self.model_cfgs.get("LOSS_CONFIG", default_loss)
        loss_types = loss_cfg.get("LOSS_TYPES", default_loss["LOSS_TYPES"])
        loss_weights = loss_cfg.get("LOSS_WEIGHTS", default_loss["LOSS_WEIGHTS"])
        assert len(loss_types) == len(loss_weights)
        self.criterion_losses = Losses(
            loss_types=loss_types, loss_weights=loss_weights, ignore_index=model_cfgs.IGNORE_LABEL,
            knn=loss_cfg.get("KNN", default_loss["KNN"]), label_smoothing=model_cfgs.get("LABEL_SMOOTHING", 0.0))

    def _make_layer(self, block, out_channels, num_block, stride=1, if_dist=False):
        layers = [block(self.in_channels, out_channels, stride=stride, if_dist=if_dist)]
        self.in_channels = out_channels * block.expansion
        layers += [block(self.in_channels, out_channels, if_dist=if_dist) for _ in range(1, num_block)]
        return layers

    def weight_initialization(self):
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm1d, nn.SyncBatchNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    @staticmethod
    def _index_plan(coords: torch.Tensor, point_coords: torch.Tensor, backward: bool = True, **extra):
        """Everything of a pass that depends on coordinates only - no features, no parameters: the
        coordinate set and kernel map of every stride (same `cmaps` / `kmaps` entries conv3d would create
        lazily, conv.py:144-177) and the trilinear point<->voxel maps `voxel_to_point` caches per stride
        (utils.py:72-82; the U-Net devoxelises at strides 1, 16 and 4).  Built before the first convolution
        so the host reads (voxel counts, pair totals) do not stall the launch stream mid-network; a data stage
        may build it for the NEXT batch on another stream (`taseg_amd.data.stage.DevicePrefetcher`).  backward=False (an
        evaluation pass): without the walk orders of the devoxelize BACKWARD (three plans, one host read, ~25 launches)."""
        fast = _fast.module()
        if fast is not None and coords.is_cuda and coords.dtype == torch.int32 and point_coords.dtype == torch.float32:
            # native, interpreter-lock-free form of the block below (csrc/fastpath: same backend calls, same results)
            pc = point_coords.contiguous()
            cm, sub_t, down_t, totals, t_idx, t_w, orders = fast.index_plan(coords.contiguous(), pc, 4, B.L.stream())
            names = ("nbr", "nbmaps", "nbsizes", "nboffs", "pos_out", "pos_in")
            cmaps, kmaps = {}, {}
            for lvl, c in enumerate(cm):
                s = 1 << lvl
                cmaps[(s, s, s)] = c
            for lvl in range(5):
                s = 1 << lvl
                km = spF.KernelMap(dict(zip(names, sub_t[lvl])), (cm[lvl].shape[0], cm[lvl].shape[0]))
                t = int(totals[2 * lvl])
                # (a negative entry: the level's coordinates hold a duplicate - the map was built by the full probe, -(pairs + 1))
                km._total, km._dup = (-t - 1, True) if t < 0 else (t, False)
                km.build_class_plan(defer=True)       # large maps: plan of the class-sorted implicit GEMM (csrc/conv_class.hip)
                kmaps[((s, s, s), (3, 3, 3), (1, 1, 1), (1, 1, 1))] = km
                if lvl < 4:
                    km = spF.KernelMap(dict(zip(names, down_t[lvl])), (cm[lvl].shape[0], cm[lvl + 1].shape[0]))
                    km._total = int(totals[2 * lvl + 1])
                    km.build_direct_plans()               # 2x2x2 strided map: one-pass plans of its two directions
                    kmaps[((s, s, s), (2, 2, 2), (2, 2, 2), (1, 1, 1))] = km
            spF.accept_class_plans(list(kmaps.values()))       # ONE host read for the class plans of all levels
            keys = ((1, 1, 1), (16, 16, 16), (4, 4, 4))
            tri_idx, tri_w = dict(zip(keys, t_idx)), dict(zip(keys, t_w))
            tri_order = {keys[1]: orders[0]} if (_DEVOX_ATOMIC and backward) else {}
            for key in ((keys if not _DEVOX_ATOMIC else (keys[0], keys[2])) if backward else ()):
                plan = _coarse_devox_plan if key == keys[1] else B.devox_csr
                tri_order[key] = plan(tri_idx[key], tri_w[key], cmaps[key].shape[0])
            return dict(coords=coords, point_coords=pc, cmaps=cmaps, kmaps=kmaps, tri_idx=tri_idx, tri_w=tri_w,
                        tri_order=tri_order, **extra)
        with torch.no_grad():
            probe = SparseTensor(None, coords, 1)
            spF.build_pyramid(probe, num_levels=4)
            for key, km in probe.kmaps.items():
                if key[1] == (3, 3, 3) and key[2] == (1, 1, 1):
                    km.build_class_plan()
                elif key[1] == key[2] == (2, 2, 2):
                    km.build_direct_plans()
            tri_idx, tri_w, tri_order = {}, {}, {}
            pc = point_coords.contiguous()
            for s in (1, 16, 4):
                key = (s, s, s)
                tri_idx[key], tri_w[key] = B.trilinear_map(pc, probe.cmaps[key], s)
                if not backward:
                    continue
                # how the devoxelize backward walks this map: a gather along the inverse map, no atomics, fixed summation
                # order (the whole training step is run-to-run deterministic).  TASEG_DEVOX_ATOMIC=1: stride 16 (~4 live
                # corners, ~60 points per voxel) adds runs of points of one interpolation cell with float atomics instead,
                # every gradient row read once (147 vs 235 us per step, last-bit noise in the stage-4 gradients)
                if s == 16 and _DEVOX_ATOMIC:
                    tri_order[key] = B.devox_order(tri_idx[key], probe.cmaps[key].shape[0])
                elif s == 16:
                    tri_order[key] = _coarse_devox_plan(tri_idx[key], tri_w[key], probe.cmaps[key].shape[0])
                else:
                    tri_order[key] = B.devox_csr(tri_idx[key], tri_w[key], probe.cmaps[key].shape[0])
        return dict(coords=coords, point_coords=pc, cmaps=probe.cmaps, kmaps=probe.kmaps, tri_idx=tri_idx,
                    tri_w=tri_w, tri_order=tri_order, **extra)

    def prepare(self, batch_dict):
        """Build the index plan of `batch_dict` and leave it under batch_dict['_plan'] (forward() does this
        itself when it is absent)."""
        raise NotImplementedError

    def _unet(self, feats: torch.Tensor, point_feats: torch.Tensor, plan) -> torch.Tensor:
        """stem .. classifier on the stride-1 voxel features and their point view; returns logits [N, num_class]."""
        return self.classifier(self._unet_point_features(feats, point_feats, plan, concat=True))

    def _unet_point_features(self, feats: torch.Tensor, point_feats: torch.Tensor, plan, concat: bool = False):
        """The encoder / decoder pass; returns the three per-point feature blocks the classifier concatenates:
        stride-16 encoder output, stride-4 and stride-1 decoder outputs, each devoxelised onto the points.
        concat=True returns them as ONE [N, C1 + C2 + C3] tensor, interpolated straight into its column blocks
        (`spF.spdevoxelize_cat`: no torch.cat, no copies of the gradient slices in the backward pass)."""
        x0 = SparseTensor(feats, plan["coords"], 1)
        x0.cmaps, x0.kmaps = plan["cmaps"], plan["kmaps"]
        z = PointTensor(point_feats, plan["point_coords"], idx_query=plan["tri_idx"], weights=plan["tri_w"])
        z.additional_features["devox_order"] = plan["tri_order"]
        if "vox_idx" in plan:
            z.additional_features["idx_query"][1] = plan["vox_idx"]
            z.additional_features["counts"][1] = plan["vox_counts"]
        x0 = spnn.conv_bn_act(self.stem[0], self.stem[1], x0, relu=True)          # stem = 2 x (conv, BN, ReLU)
        x0 = spnn.conv_bn_act(self.stem[3], self.stem[4], x0, relu=True)
        progs = self._stage_programs(x0.F, plan)
        if progs is not None:
            return self._unet_stages(progs, x0.F, plan, concat)
        z0 = voxel_to_point(x0, z, nearest=False, features=False)      # z0.F is never read (cache carrier only)

        x1 = self.stage1(x0)
        x2 = self.stage2(x1)
        x3 = self.stage3(x2)
        x4 = self.stage4(x3)
        concat = concat and all(k in plan["tri_idx"] for k in (x4.s, x2.s, x0.s))
        sources = []                                   # (voxel features before dropout, stride) of z1, z2, z3
        if concat:
            sources.append((x4.F, x4.s))
        else:
            z1 = voxel_to_point(x4, z0)

        # out of place: `sources` (and the ReLU node that produced x4.F) still hold the features BEFORE dropout - the
        # reference devoxelises z1 / z2 before its in-place dropout (minkunet.py:400-412)
        x4.F = torch.nn.functional.dropout(x4.F, self.dropout.p, self.training, False)
        y1 = self.up1[1](torchsparse.cat([self.up1[0](x4), x3]))
        y2 = self.up2[1](torchsparse.cat([self.up2[0](y1), x2]))
        if concat:
            sources.append((y2.F, y2.s))
        else:
            z2 = voxel_to_point(y2, z1)

        y2.F = torch.nn.functional.dropout(y2.F, self.dropout.p, self.training, False)
        y3 = self.up3[1](torchsparse.cat([self.up3[0](y2), x1]))
        y4 = self.up4[1](torchsparse.cat([self.up4[0](y3), x0]))
        if concat:
            sources.append((y4.F, y4.s))
            orders = plan["tri_order"]
            return spF.spdevoxelize_cat([f for f, _ in sources],
                                        [(plan["tri_idx"][k], plan["tri_w"][k], orders.get(k)) for _, k in sources])
        z3 = voxel_to_point(y4, z2)
        return z1.F, z2.F, z3.F

    def _stage_programs(self, feats, plan):
        """the compiled stage programs of this model (stage_program.StagePrograms) if they can serve this pass on this index plan,
        else None: the module-by-module path then runs (TASEG_STAGE_PROGRAM=0, no native binding, Bottleneck blocks, hooks on
        the conv modules, widths off the full-tile paths, an index plan without the U-Net's kernel maps ...)"""
        if not _SP.enabled():
            return None
        progs = _SP.programs_of(self)
        if progs is None:
            return None
        if not progs.usable(feats, self.training, torch.is_grad_enabled()):
            if progs.stale:                          # parameter / buffer objects were replaced: compile again at the next pass
                _SP.forget(self)
            return None
        if not all(k in plan["tri_idx"] for k in ((16, 16, 16), (4, 4, 4), (1, 1, 1))):
            return None
        half = spF._amp_half(feats)
        progs.half = half                            # (prepare() resolves the NEXT batch's geometry for this storage mode)
        try:
            progs.prepare(plan, half)                # cached in the plan: a no-op when the data stage has done it
        except _SP._Unsupported:
            return None
        return progs

    def _stage_prepare(self, plan):
        """resolve the stage programs' geometry (kernel maps + class plans per op) with the index plan - on the thread that stages
        the batch, not on the one that issues the step; a no-op until the first pass has compiled the programs"""
        progs = _SP.programs_of(self, compile=False)
        if progs is not None:
            try:
                progs.prepare(plan, progs.half)
            except _SP._Unsupported:
                pass
        return plan

    def _unet_stages(self, progs, f0, plan, concat):
        """stage1 .. up4 of `_unet_point_features` on the stage programs: one call (and one autograd node) per stage"""
        # (the eight stages and the two out-of-place dropouts between them - the point head devoxelises the features BEFORE
        # dropout, minkunet.py:400-412 - in one native call, issued without the interpreter lock)
        f4, y2, y4 = progs.run_unet(f0, plan, self.training, self.dropout.p)
        keys = ((16, 16, 16), (4, 4, 4), (1, 1, 1))
        tri_idx, tri_w, orders = plan["tri_idx"], plan["tri_w"], plan["tri_order"]
        maps = [(tri_idx[k], tri_w[k], orders.get(k)) for k in keys]
        if concat:
            return spF.spdevoxelize_cat([f4, y2, y4], maps)
        return tuple(spF.spdevoxelize(f, i, w, o) for f, (i, w, o) in zip((f4, y2, y4), maps))

    def _train_outputs(self, logits, target, coords_xyz, offset):
        loss = self.criterion_losses(logits, target, xyz=coords_xyz, offset=offset)
        lazy = LazyScalar(loss)
        return {"loss": loss}, {"loss": lazy}, {"loss": lazy}


class _PinnedRing:
    """host buffers for the evaluation tail's device -> host copies: page-locked (the copies are asynchronous), grown on demand,
    four generations deep - a deferred result stays valid until three further deferred tails have been issued"""
    DEPTH = 4

    def __init__(self):
        self.bufs, self.at = {}, 0

    def next_generation(self):
        self.at = (self.at + 1) % self.DEPTH

    def take(self, key, like, numel):
        k = (key, self.at)
        b = self.bufs.get(k)
        if b is None or b.dtype != like.dtype or b.numel() < numel:
            b = self.bufs[k] = torch.empty(max(int(numel * 1.25), 1024), dtype=like.dtype, pin_memory=True)
        return b[:numel]


_pinned = _PinnedRing()
# TASEG_EVAL_COPY_STREAM=0: the deferred tail's device -> host copies on the launch stream
_COPY_STREAM = options.eval_copy_stream
_copy_streams = {}


def _copy_stream(device):
    s = _copy_streams.get(device)
    if s is None:
        s = _copy_streams[device] = torch.cuda.Stream(device=device)
    return s


class PendingPredictions:
    """The evaluation tail in flight (`model(batch, defer=True)`): every device -> host copy has been enqueued into page-locked
    buffers behind the forward pass, one event marks their end.  `result()` waits for the event, checks what the reference's
    indexing would have raised on, and slices the dictionary of numpy arrays `model(batch)` returns.  The arrays are views of a
    ring of host buffers: valid until three further deferred tails have been issued (copy them to keep them longer)."""

    def __init__(self, event, meta_h, result_h, mapped_h, labels_h, shapes, names, n_scenes, has_ms, fused=False, fallback=None):
        self.event, self.meta_h, self.result_h, self.mapped_h, self.labels_h = event, meta_h, result_h, mapped_h, labels_h
        self.shapes, self.names, self.n_scenes, self.has_ms = shapes, names, n_scenes, has_ms
        self.fused, self.fallback = fused, fallback
        self._out = None

    def result(self):
        if self._out is not None:
            return self._out
        self.event.synchronize()
        n = self.n_scenes
        meta = self.meta_h.numpy()
        if self.fused:
            # the fused tail (csrc/evaltail.hip): [flags, voxels per scene, points per scene, labels per scene, ...]; flags bit 0:
            # a scene index outside the batch, bit 1: an index array not grouped by scene (the sorted form then serves the batch),
            # bit 2: inverse map outside its scene
            flags = int(meta[0])
            if flags & 2:
                self._out = self.fallback()
                return self._out
            bad = (1 if flags & 4 else 0) | (2 if flags & 1 else 0)
            cnt_p_h = cnt_k_h = meta[1 + n:1 + 2 * n].tolist()
        else:
            bad, cnt_p_h, cnt_k_h = int(meta[0]), meta[1:1 + n].tolist(), meta[1 + n:1 + 2 * n].tolist()
        cnt_l_h = meta[1 + 2 * n:1 + 3 * n].tolist()
        n_cur_h = meta[1 + 3 * n:1 + 4 * n].tolist()
        if bad & 2:
            raise IndexError(f"batch indices beyond the {n} scenes of the batch")
        if bad & 1:
            raise IndexError("inverse_map names a voxel outside its scene")
        if self.has_ms:
            n_ms_h = meta[1 + 4 * n:1 + 5 * n].tolist()
            if n_ms_h != cnt_p_h:
                raise IndexError(f"num_points_ms {n_ms_h} does not match the inverse map's points per scene {cnt_p_h}")
        result_h = self.result_h.numpy().reshape(self.shapes[0])
        mapped_h = None if self.mapped_h is None else self.mapped_h.numpy().reshape(self.shapes[1])
        labels_h = self.labels_h.numpy().reshape(self.shapes[2])
        point_predict, point_labels, point_predict_logits = [], [], []
        at_k = at_l = 0
        for b in range(n):
            n_cur = int(n_cur_h[b])
            seg = slice(at_k, at_k + min(int(cnt_k_h[b]), n_cur))
            point_predict.append(result_h[seg])                 # (views of the batch's arrays: no second host copy)
            if mapped_h is not None:
                point_predict_logits.append(mapped_h[seg])
            point_labels.append(labels_h[at_l: at_l + min(int(cnt_l_h[b]), n_cur)])
            at_k += int(cnt_k_h[b])
            at_l += int(cnt_l_h[b])
        self._out = {"point_predict": point_predict, "point_labels": point_labels, "name": self.names,
                     "point_predict_logits": point_predict_logits}
        return self._out


_FUSED_TAIL = options.fused_eval_tail


def _fused_tail(out, vox_batch, invs, all_labels, num_points, want_probs, num_points_ms, names, defer, n_scenes):
    """unvoxelise_predictions for index arrays grouped by scene (what sparse_collate builds) in two launches of csrc/evaltail.hip -
    rows per scene, then gather + arg-max - instead of three stable sorts and ~45 tensor ops; None where it does not apply.  A batch
    that is not grouped is detected on the device and served by the sorted form when the arrays are collected."""
    bv, bp, bl = vox_batch, invs.C[:, -1], all_labels.C[:, -1]
    if not (out.is_cuda and out.dim() == 2 and out.dtype in (torch.float32, torch.float16) and n_scenes <= 64
            and all(t.dtype == torch.int32 and t.is_cuda and t.dim() == 1 for t in (bv, bp, bl))):
        return None
    L = B.L
    lib = L.load()
    dev = out.device
    inv = invs.F if invs.F.dtype == torch.int64 else invs.F.long()
    inv = inv.contiguous()
    n = n_scenes
    has_ms = num_points_ms is not None
    meta = torch.zeros(1 + (5 if has_ms else 4) * n, dtype=torch.int64, device=dev)      # [flags | counts 3 x n | num_points (| _ms)]
    st = L.stream()
    stride = lambda t: int(t.stride(0)) if t.shape[0] > 1 else 1  # noqa: E731
    L.check(lib.ts_scene_counts(L.ptr(bv), stride(bv), bv.shape[0], L.ptr(bp), stride(bp), bp.shape[0], L.ptr(bl), stride(bl),
                                bl.shape[0], n, meta.data_ptr() + 8, meta.data_ptr(), st), "ts_scene_counts")
    logits = out.contiguous()
    pts, c = inv.shape[0], logits.shape[1]
    mapped = torch.empty((pts, c), dtype=logits.dtype, device=dev)
    pred = None if want_probs else torch.empty(pts, dtype=torch.int64, device=dev)
    L.check(lib.ts_unvoxelise(L.ptr(logits), 1 if logits.dtype == torch.float16 else 0, c, meta.data_ptr() + 8, n, L.ptr(bp), stride(bp),
                              L.ptr(inv), pts, L.ptr(mapped), L.ptr(pred), meta.data_ptr(), st), "ts_unvoxelise")
    meta[1 + 3 * n:1 + 4 * n].copy_(torch.as_tensor(num_points).reshape(-1)[:n], non_blocking=True)
    if has_ms:
        meta[1 + 4 * n:1 + 5 * n].copy_(torch.as_tensor(num_points_ms).reshape(-1)[:n], non_blocking=True)
    result = mapped.softmax(1) if want_probs else pred
    return meta, result, mapped, all_labels.F


def unvoxelise_predictions(out, vox_batch, invs, all_labels, num_points, want_probs, point_mask=None, num_points_ms=None,
                           names=None, defer=False, _fused=True):
    """The evaluation tail of the segmentors (minkunet.py:435-455, minkunet_ms.py:433-458) for the WHOLE batch at once: per scene
    `out[scene][inverse_map of the scene]` (Ms: `[point_mask of the scene]`), trimmed to the scan's own point count; arg-max and
    logits (or the soft-max under return_logit / return_tta) and the mapped labels as numpy arrays, scene by scene - what the
    reference's per-scene loop of boolean masks returns, with three stable sorts, one gather and a handful of device -> host
    copies instead of ~12 launches and 6 host reads per scene.  defer: return a PendingPredictions instead of waiting for the
    copies - the caller issues the next batch's forward pass first and collects `result()` afterwards (pcseg/eval.py, bench.py
    --eval): the host never waits for the device between two batches."""
    dev = out.device
    n_scenes = len(names) if names is not None else int(torch.as_tensor(num_points).numel())
    fused = _fused_tail(out, vox_batch, invs, all_labels, num_points, want_probs, num_points_ms, names, defer, n_scenes) \
        if (_fused and _FUSED_TAIL and point_mask is None) else None
    if fused is not None:
        meta, result, mapped, labels_sorted = fused
        return _tail_to_host(meta, result, mapped, labels_sorted, want_probs, defer, names, n_scenes, num_points_ms is not None, True,
                             lambda: unvoxelise_predictions(out, vox_batch, invs, all_labels, num_points, want_probs, point_mask,
                                                            num_points_ms, names, False, _fused=False))
    b_vox, b_pts, b_lab = vox_batch.long(), invs.C[:, -1].long(), all_labels.C[:, -1].long()
    # stable sorts by scene: 8-bit keys where the batch allows it (one radix pass instead of the eight of a 64-bit key; an index
    # outside 0 .. 254 wraps to a value that is still >= n_scenes or lands in the wrong scene's count - both are reported below)
    narrow = (lambda t: t.clamp(-1, 255).to(torch.uint8)) if n_scenes < 255 else (lambda t: t)
    bv_sorted, order_v = torch.sort(narrow(b_vox), stable=True)   # out[scene] = out[order_v[start : start + count]] (original order kept)
    bp_sorted, order_p = torch.sort(narrow(b_pts), stable=True)
    bl_sorted, order_l = torch.sort(narrow(b_lab), stable=True)
    # rows per scene from the SORTED keys (torch.bincount reads the largest index on the host: with four of them the "deferred" tail
    # waited for the forward pass four times per batch - the host issue time of bench.py --eval was 2.3 ms of bincount)
    edges = torch.arange(n_scenes + 1, device=dev)

    def counts(sorted_keys):
        at = torch.searchsorted(sorted_keys.long(), edges)
        # (second value: rows whose scene index is outside 0 .. n_scenes - 1 - the reference's boolean masks would drop them
        # silently and fail on the shapes later; reported when the arrays are collected)
        return at[1:] - at[:-1], (at[0] != 0) | (at[-1] != sorted_keys.shape[0])
    cnt_v, out_v = counts(bv_sorted)
    cnt_p, out_p = counts(bp_sorted)
    cnt_l, out_l = counts(bl_sorted)
    bp_sorted = bp_sorted.long()
    start_v = torch.cumsum(cnt_v, 0) - cnt_v
    local = invs.F[order_p].long()
    scene_p = bp_sorted.clamp(0, n_scenes - 1)
    bad = ((local < 0) | (local >= cnt_v[scene_p])).any()           # read with the counts (the reference's indexing raises)
    rows = order_v[(start_v[scene_p] + local).clamp_(0, max(int(order_v.shape[0]) - 1, 0))]
    mapped = out[rows]                                      # [points, classes], scene-major, the scene's own point order
    cnt_k = cnt_p
    if point_mask is not None:
        keep = point_mask.to(dev).bool()
        mapped, kept_scene = mapped[keep], bp_sorted[keep]          # (boolean indexing: one host read, Ms models only)
        cnt_k, _ = counts(kept_scene)
    result = mapped.softmax(1) if want_probs else mapped.argmax(1)
    labels_sorted = all_labels.F[order_l]
    # one small tensor with everything the per-scene slicing needs, then the arrays themselves: asynchronous copies into page-locked
    # buffers, one event behind them
    code = bad.long() + 2 * (out_v | out_p | out_l).long()      # bit 0: inverse map outside its scene; bit 1: scene index outside the batch
    parts = [code.reshape(1), cnt_p.long(), cnt_k.long(), cnt_l.long(),
             torch.as_tensor(num_points).reshape(-1)[:n_scenes].to(dev, non_blocking=True).long()]
    if num_points_ms is not None:
        parts.append(torch.as_tensor(num_points_ms).reshape(-1)[:n_scenes].to(dev, non_blocking=True).long())
    meta = torch.cat(parts)
    return _tail_to_host(meta, result, mapped, labels_sorted, want_probs, defer, names, n_scenes, num_points_ms is not None, False, None)


def _tail_to_host(meta, result, mapped, labels_sorted, want_probs, defer, names, n_scenes, has_ms, fused, fallback):
    """one small tensor with everything the per-scene slicing needs, then the arrays themselves: asynchronous copies into page-locked
    buffers, one event behind them"""
    if defer:
        _pinned.next_generation()
    meta, result, labels_sorted = meta.contiguous(), result.contiguous(), labels_sorted.contiguous()
    mapped = mapped if want_probs else mapped.contiguous()

    def to_host(key, t):
        if not defer:
            return t.reshape(-1).cpu()           # fresh arrays, as `model(batch)` has always returned them
        h = _pinned.take(key, t, t.numel())
        h.copy_(t.reshape(-1), non_blocking=True)
        return h

    # deferred: the copies (the logits of every point among them: 19 MB per bench batch) run on a stream of their own behind an
    # event of the launch stream - there they are blit kernels that would hold the next pass's forward up (0.3 ms per pass)
    side = _copy_stream(meta.device) if (defer and _COPY_STREAM) else None
    if side is not None:
        ready = torch.cuda.Event()
        ready.record()
        side.wait_event(ready)
    with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        meta_h = to_host("meta", meta)
        result_h = to_host("result", result)
        mapped_h = None if want_probs else to_host("mapped", mapped)
        labels_h = to_host("labels", labels_sorted)
        event = torch.cuda.Event()
        event.record()
    pending = PendingPredictions(event, meta_h, result_h, mapped_h, labels_h, (tuple(result.shape), tuple(mapped.shape), tuple(labels_sorted.shape)),
                                 names, n_scenes, has_ms, fused, fallback)
    # (the device tensors stay alive until the copies have run: the event's owner keeps them)
    pending._keep = (meta, result, mapped, labels_sorted)
    return pending if defer else pending.result()


class MinkUNet(MinkUNetBackbone):
    """Single-frame model: re-voxelises `batch_dict['lidar']` on device, then the U-Net
    (minkunet.py:385-455)."""

    def prepare(self, batch_dict):
        with torch.no_grad():
            z = PointTensor(None, batch_dict["lidar"].C.float())
            coords, vox_idx, vox_counts = voxelize_index(z, self.pres, self.vres)      # minkunet.py:388-390
        plan = self._stage_prepare(self._index_plan(coords, z.C, backward=self.training, vox_idx=vox_idx, vox_counts=vox_counts))
        batch_dict["_plan"] = plan
        return plan

    def forward(self, batch_dict, return_logit=False, return_tta=False, defer=False):
        x = batch_dict["lidar"]
        x.F = x.F[:, :self.in_feature_dim]
        plan = batch_dict.get("_plan") or self.prepare(batch_dict)
        feats = spF.spvoxelize(x.F, plan["vox_idx"], plan["vox_counts"])      # feature half of initial_voxelize
        out = self._unet(feats, x.F, plan)

        if self.training:
            target = batch_dict["targets"].F.long().cuda(non_blocking=True)
            return self._train_outputs(out, target, batch_dict["lidar"].C[:, :3].float(), batch_dict["offset"])

        return unvoxelise_predictions(out, x.C[:, -1], batch_dict["inverse_map"], batch_dict["targets_mapped"],
                                      batch_dict["num_points"], return_logit or return_tta, names=batch_dict["name"], defer=defer)

    def forward_ensemble(self, batch_dict):
        return self.forward(batch_dict, return_tta=True)
