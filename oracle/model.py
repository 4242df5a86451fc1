"""CPU ORACLE (model level) - test infrastructure, not product code.

A functional restatement of the reference's MinkUNet / MinkUNetMs forward pass
(R/pcseg/model/segmentor/voxel/minkunet/minkunet.py:385-434, minkunet_ms.py:385-432,
minkunet/utils.py:11-107, TS/torchsparse/nn/functional/conv.py:122-205) driven by a plain
state_dict, on the CPU.  Sparse ops come from oracle/ts_oracle.py (numpy) or, when
`backend="ref"`, from oracle/_ref (the reference's own C++ kernels compiled by
oracle/build_ref.py) - the latter is what bench.py times as the CPU baseline.  Dense pieces
(BatchNorm, Linear, cross-entropy, sort) are torch CPU ops, as in the reference.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import numpy as np
import torch
import torch.nn.functional as TF
from torch.autograd import Function

from . import ts_oracle as O


def _load_ref():
    import importlib
    import os
    import sys
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
    if d not in sys.path:
        sys.path.insert(0, d)
    return importlib.import_module("ts_ref_backend")


class _Conv(Function):
    @staticmethod
    def forward(ctx, feats, weight, kmap, transposed, backend):
        nbmaps, nbsizes, sizes = kmap
        if backend == "ref":
            ref = _load_ref()
            n_out = sizes[0] if transposed else sizes[1]
            out = torch.zeros(n_out, weight.shape[-1])
            ref.convolution_forward_cpu(feats.contiguous(), out, weight.contiguous(),
                                        torch.from_numpy(nbmaps.astype(np.int32)),
                                        torch.from_numpy(nbsizes.astype(np.int32)), transposed)
        else:
            out = torch.from_numpy(O.conv_forward(feats.numpy(), weight.numpy(), nbmaps, nbsizes, sizes, transposed))
        ctx.save_for_backward(feats, weight)
        ctx.misc = (kmap, transposed, backend)
        return out

    @staticmethod
    def backward(ctx, gout):
        feats, weight = ctx.saved_tensors
        (nbmaps, nbsizes, sizes), transposed, backend = ctx.misc
        if backend == "ref":
            ref = _load_ref()
            gin, gw = torch.zeros_like(feats), torch.zeros_like(weight)
            ref.convolution_backward_cpu(feats.contiguous(), gin, gout.contiguous(), weight.contiguous(), gw,
                                         torch.from_numpy(nbmaps.astype(np.int32)),
                                         torch.from_numpy(nbsizes.astype(np.int32)), transposed)
        else:
            gin, gw = O.conv_backward(feats.numpy(), weight.numpy(), gout.contiguous().numpy(), nbmaps, nbsizes,
                                      transposed)
            gin, gw = torch.from_numpy(gin), torch.from_numpy(gw)
        return gin, gw, None, None, None


class _Devox(Function):
    @staticmethod
    def forward(ctx, feats, idx, w):
        ctx.misc = (idx, w, feats.shape[0])
        return torch.from_numpy(O.devoxelize_forward(feats.numpy(), idx, w))

    @staticmethod
    def backward(ctx, gout):
        idx, w, m = ctx.misc
        return torch.from_numpy(O.devoxelize_backward(gout.contiguous().numpy(), idx, w, m)), None, None


class _Vox(Function):
    @staticmethod
    def forward(ctx, feats, idx, counts):
        ctx.misc = (idx, counts, feats.shape[0])
        return torch.from_numpy(O.voxelize_forward(feats.numpy(), idx, counts))

    @staticmethod
    def backward(ctx, gout):
        idx, counts, n = ctx.misc
        return torch.from_numpy(O.voxelize_backward(gout.contiguous().numpy(), idx, counts, n)), None, None


class _Sparse:
    """coords [N,4] int32 numpy + feats torch tensor + shared caches (TS/torchsparse/tensor.py:10-72)."""

    def __init__(self, feats, coords, stride, cmaps=None, kmaps=None):
        self.F, self.C, self.s = feats, coords, stride
        self.cmaps = {} if cmaps is None else cmaps
        self.kmaps = {} if kmaps is None else kmaps

    def like(self, feats):
        return _Sparse(feats, self.C, self.s, self.cmaps, self.kmaps)


class OracleMinkUNet:
    """params: dict name -> torch CPU tensor (the reference state_dict layout)."""

    def __init__(self, params, cfg, num_class=20, backend="numpy", training=True, act=None):
        self.p = params
        # the activation of every block: ReLU (minkunet.py:42-51) - a smooth stand-in lets a gradient comparison judge arithmetic
        # alone (no ReLU flips between two roundings of the same pre-activation: tests/test_gpu_bench_backward.py)
        self.act = torch.relu if act is None else act
        self.cfg = cfg
        self.backend = backend
        self.training = training
        self.num_layer = cfg.get("NUM_LAYER", [2, 3, 4, 6, 2, 2, 2, 2])
        self.in_dim = cfg["IN_FEATURE_DIM"]
        self.debug = {}

    # -- layers ----------------------------------------------------------------------
    def conv(self, x, name, ks, stride=1, transposed=False):
        """TS/torchsparse/nn/functional/conv.py:122-205."""
        w = self.p[name + ".kernel"]
        if ks == 1:
            return x.like(x.F.matmul(w))
        if not transposed:
            out_s = x.s * stride
            if out_s in x.cmaps:
                out_c = x.cmaps[out_s]
            elif stride == 1:
                out_c = x.C
            else:
                out_c = O.spdownsample(x.C, stride, ks, x.s)
            key = (x.s, ks, stride)
            if key not in x.kmaps:
                off = O.get_kernel_offsets(ks, x.s, 1)
                _, nbmaps, nbsizes = O.build_kmap(x.C, out_c, off)
                x.kmaps[key] = (nbmaps, nbsizes, (x.C.shape[0], out_c.shape[0]))
            feats = _Conv.apply(x.F, w, x.kmaps[key], False, self.backend)
        else:
            out_s = x.s // stride
            out_c = x.cmaps[out_s]
            feats = _Conv.apply(x.F, w, x.kmaps[(out_s, ks, stride)], True, self.backend)
        out = _Sparse(feats, out_c, out_s, x.cmaps, x.kmaps)
        out.cmaps.setdefault(out_s, out_c)
        return out

    def bn(self, x, name):
        p = self.p
        f = TF.batch_norm(x.F, p[name + ".running_mean"], p[name + ".running_var"], p[name + ".weight"],
                          p[name + ".bias"], self.training, 0.1, 1e-5)
        return x.like(f)

    def conv_bn_relu(self, x, name, ks, stride=1, transposed=False):
        x = self.bn(self.conv(x, name + ".net.0", ks, stride, transposed), name + ".net.1")
        return x.like(self.act(x.F))

    def resblock(self, x, name):
        """R/.../minkunet.py:83-129."""
        y = self.bn(self.conv(x, name + ".net.0", 3), name + ".net.1")
        y = y.like(self.act(y.F))
        y = self.bn(self.conv(y, name + ".net.3", 3), name + ".net.4")
        if (name + ".downsample.0.kernel") in self.p:
            sc = self.bn(self.conv(x, name + ".downsample.0", 1), name + ".downsample.1")
        else:
            sc = x
        return y.like(self.act(y.F + sc.F))

    def stage(self, x, name, depth):
        x = self.conv_bn_relu(x, name + ".0", 2, 2)
        for i in range(depth):
            x = self.resblock(x, f"{name}.{i + 1}")
        return x

    def up(self, x, skip, name, depth):
        y = self.conv_bn_relu(x, name + ".0", 2, 2, transposed=True)
        y = y.like(torch.cat([y.F, skip.F], 1))
        for i in range(depth):
            y = self.resblock(y, f"{name}.1.{i}")
        return y

    def voxel_to_point(self, x, zC, cache):
        """R/.../minkunet/utils.py:69-107."""
        if x.s not in cache:
            cache[x.s] = O.trilinear_map(zC, x.C, x.s)
        idx, w = cache[x.s]
        return _Devox.apply(x.F, idx, w)

    # -- forward ---------------------------------------------------------------------
    def unet(self, x0, zC):
        nl = self.num_layer
        cache = {}
        x0 = self.bn(self.conv(x0, "stem.0", 3), "stem.1")
        x0 = x0.like(self.act(x0.F))
        x0 = self.bn(self.conv(x0, "stem.3", 3), "stem.4")
        x0 = x0.like(self.act(x0.F))
        self.voxel_to_point(x0, zC, cache)  # z0: only the cached maps matter
        x1 = self.stage(x0, "stage1", nl[0])
        x2 = self.stage(x1, "stage2", nl[1])
        x3 = self.stage(x2, "stage3", nl[2])
        x4 = self.stage(x3, "stage4", nl[3])
        z1 = self.voxel_to_point(x4, zC, cache)
        y1 = self.up(x4, x3, "up1", nl[4])
        y2 = self.up(y1, x2, "up2", nl[5])
        z2 = self.voxel_to_point(y2, zC, cache)
        y3 = self.up(y2, x1, "up3", nl[6])
        y4 = self.up(y3, x0, "up4", nl[7])
        z3 = self.voxel_to_point(y4, zC, cache)
        self.debug.update(kmaps=x0.kmaps, cmaps=x0.cmaps, tri=cache)
        feat = torch.cat([z1, z2, z3], 1)
        self.debug["point_features"] = feat
        return TF.linear(feat, self.p["classifier.0.weight"], self.p["classifier.0.bias"])

    def unet3d(self, x0, zC):
        """R/.../unet3d.py:297-316 - the FOV encoder of TIAF: stem + four down stages of one residual block each, point features
        at strides 1 / 4 / 16 -> a linear classifier; returns (logits, x4, x2, x0) like the reference."""
        cache = {}
        x0 = self.bn(self.conv(x0, "stem.0", 3), "stem.1")
        x0 = x0.like(self.act(x0.F))
        x0 = self.bn(self.conv(x0, "stem.3", 3), "stem.4")
        x0 = x0.like(self.act(x0.F))
        z0 = self.voxel_to_point(x0, zC, cache)
        x1 = self.stage(x0, "stage1", 1)
        x2 = self.stage(x1, "stage2", 1)
        z1 = self.voxel_to_point(x2, zC, cache)
        x3 = self.stage(x2, "stage3", 1)
        x4 = self.stage(x3, "stage4", 1)
        z2 = self.voxel_to_point(x4, zC, cache)
        out = TF.linear(torch.cat([z0, z1, z2], 1), self.p["classifier.0.weight"], self.p["classifier.0.bias"])
        return out, x4, x2, x0

    def forward_minkunet(self, coords, feats):
        """R/.../minkunet.py:385-422 (with initial_voxelize)."""
        feats = feats[:, :self.in_dim]
        zC0 = np.asarray(coords, dtype=np.float32)
        scaled, cell, sparse_hash, idx_query, counts = O.initial_voxelize_maps(zC0, self.cfg.get("pres", 0.05),
                                                                                self.cfg.get("vres", 0.05))
        vc = np.round(O.voxelize_forward(cell, idx_query, counts)).astype(np.int32)
        vf = _Vox.apply(feats, idx_query, counts)
        x0 = _Sparse(vf, vc, 1)
        x0.cmaps[1] = vc
        self.debug.update(idx_query=idx_query, counts=counts, vox_coords=vc)
        return self.unet(x0, scaled)

    def forward_minkunet_ms(self, coords, feats):
        """R/.../minkunet_ms.py:385-420 (no re-voxelisation)."""
        feats = feats[:, :self.in_dim]
        c = np.asarray(coords, dtype=np.int32)
        x0 = _Sparse(feats, c, 1)
        return self.unet(x0, c.astype(np.float32))


def lovasz_softmax_ref(probas, labels, ignore=0):
    """R/tools/utils/common/lovasz_losses.py:158-227 restated as the per-class loop (classes='present')."""
    valid = labels != ignore
    probas, labels = probas[valid], labels[valid]
    if probas.numel() == 0:
        return probas.sum() * 0.0
    losses = []
    for c in range(probas.shape[1]):
        fg = (labels == c).to(probas.dtype)
        if fg.sum() == 0:
            continue
        err = (fg - probas[:, c]).abs()
        err_sorted, perm = torch.sort(err, 0, descending=True)
        fg_sorted = fg[perm]
        gts = fg_sorted.sum()
        inter = gts - fg_sorted.cumsum(0)
        union = gts + (1 - fg_sorted).cumsum(0)
        jac = 1.0 - inter / union
        if len(jac) > 1:
            jac = torch.cat([jac[:1], jac[1:] - jac[:-1]])
        losses.append(torch.dot(err_sorted, jac))
    return sum(losses) / len(losses)


def loss_ce_lovasz(logits, target, ignore=0, label_smoothing=0.1):
    """R/pcseg/loss/__init__.py:52-56,106-115: CE(ignore, label smoothing) + Lovasz-softmax, weights 1."""
    ce = TF.cross_entropy(logits, target, ignore_index=ignore, label_smoothing=label_smoothing)
    return ce + lovasz_softmax_ref(logits.softmax(1), target, ignore)


class _Renamed:
    """a state_dict seen through a renaming of its first name component: `prefix` in front (the sub-module `lidar_backbone.`),
    `part_suffix` behind it (the teacher's `stem_gt`, `stage1_gt`, ... of MinkUNetMsKd, minkunet_ms_kd.py:231-380)"""

    def __init__(self, params, prefix="", part_suffix=""):
        self.params, self.prefix, self.suffix = params, prefix, part_suffix

    def _k(self, name):
        head, dot, rest = name.partition(".")
        return f"{self.prefix}{head}{self.suffix}{dot}{rest}"

    def __getitem__(self, name):
        return self.params[self._k(name)]

    def __contains__(self, name):
        return self._k(name) in self.params


def forward_minkunet_ms_mm(params, cfg, coords, feats, fov_coords, fov_feats, image_features_fov, image_logits_fov=None,
                           training=True, backend="numpy", act=None):
    """MinkUNetMsMm (TIAF) below the image branch - R/.../minkunet_ms_mm.py:456-516: the FOV encoder (UNet3D) on
    [LiDAR attributes | gathered image features (| gathered image logits)], the MinkUNet of MinkUNetMs on the fused cloud, the FOV
    encoder's stride-16 / 4 / 1 features trilinearly interpolated onto EVERY point of the fused cloud (`voxel_to_point_fov`,
    utils.py:149-170: a point with no FOV voxel around it gets a zero row), and `classifier_fusion` (Linear, BatchNorm1d, ReLU,
    Linear) on the rows whose stride-16 FOV feature is not all zero.  The dense 2-D branch stays outside: `image_features_fov`
    [n_fov, 224] (and `image_logits_fov`) are INPUTS - tensors, so that gradients with respect to them can be compared too.
    Dropouts are identities (the tests switch them off).  Returns dict(logits, fusion_logits, fov_logits, overlap)."""
    in_dim = cfg["IN_FEATURE_DIM"]
    use = cfg.get("INPUT_FEAT_LIDAR")
    fov_feats = torch.as_tensor(fov_feats)
    cols = []
    if "lidar" in use:
        cols.append(fov_feats[:, :in_dim - 1])
    if "image" in use:
        cols.append(image_features_fov)
    if "logit" in use:
        cols.append(image_logits_fov)
    fc = np.asarray(fov_coords, dtype=np.int32)
    enc = OracleMinkUNet(_Renamed(params, prefix="lidar_backbone."), cfg, backend=backend, training=training, act=act)
    xf = _Sparse(torch.cat(cols, 1), fc, 1)
    xf.cmaps[1] = fc
    fov_logits, x4f, x2f, x0f = enc.unet3d(xf, fc.astype(np.float32))

    main = OracleMinkUNet(params, cfg, backend=backend, training=training, act=act)
    logits = main.forward_minkunet_ms(coords, torch.as_tensor(feats))
    point_feats = main.debug["point_features"]
    zC = np.asarray(coords, dtype=np.float32)
    fov_point = []
    for x in (x4f, x2f, x0f):
        idx, w = O.trilinear_map(zC, x.C, x.s)
        fov_point.append(_Devox.apply(x.F, idx, w))
    overlap = fov_point[0].detach().sum(-1) != 0
    fusion = torch.cat([point_feats] + fov_point, 1)[overlap]
    h = TF.linear(fusion, params["classifier_fusion.0.weight"], params["classifier_fusion.0.bias"])
    h = TF.batch_norm(h, params["classifier_fusion.1.running_mean"], params["classifier_fusion.1.running_var"],
                      params["classifier_fusion.1.weight"], params["classifier_fusion.1.bias"], training, 0.1, 1e-5)
    fusion_logits = TF.linear(torch.relu(h), params["classifier_fusion.3.weight"], params["classifier_fusion.3.bias"])
    return dict(logits=logits, fusion_logits=fusion_logits, fov_logits=fov_logits, overlap=overlap)


def loss_minkunet_ms_mm(out, labels, fov_targets, image_logits_fov, image_logits_dense, image_targets_dense, weights,
                        ignore=0, label_smoothing=0.0):
    """the five weighted losses of R/.../minkunet_ms_mm.py:518-528 (each CE + Lovasz) -> (loss, parts[5])"""
    labels = torch.as_tensor(labels).long()
    fov_targets = torch.as_tensor(fov_targets).long()
    crit = lambda lg, t: loss_ce_lovasz(lg, t, ignore, label_smoothing)      # noqa: E731
    parts = [crit(out["logits"], labels) * weights[0],
             crit(out["fusion_logits"], labels[out["overlap"]]) * weights[1],
             crit(image_logits_fov, fov_targets) * weights[2],
             crit(image_logits_dense, torch.as_tensor(image_targets_dense).long()) * weights[3],
             crit(out["fov_logits"], fov_targets) * weights[4]]
    return sum(parts), parts


def forward_minkunet_ms_kd(params, cfg, coords, feats, coords_gt, feats_gt, labels, training=True, backend="numpy",
                           feat_kd_weight=10.0, label_smoothing=0.0, ignore=0):
    """MinkUNetMsKd's training forward - R/.../minkunet_ms_kd.py:532-640: the frozen teacher (the `*_gt` parts, no graph; its
    BatchNorm layers follow the module's mode) on the cloud fused with ground-truth masks, the student on the cloud fused with
    pseudo-label masks, CE + Lovasz on the student's logits, and per sample the MSE between the student's and the teacher's point
    features on the voxels present in both clouds (coordinate hash query), weighted FEAT_KD_WEIGHT / batch size - without the random
    sub-sampling (MAX_VOXEL above every sample's common voxels: the deterministic case the reference's fixture uses).
    Returns dict(loss, loss_seg, loss_feat_kd, logits, teacher_logits)."""
    teacher = OracleMinkUNet(_Renamed(params, part_suffix="_gt"), cfg, backend=backend, training=training)
    with torch.no_grad():
        teacher_logits = teacher.forward_minkunet_ms(coords_gt, torch.as_tensor(feats_gt))
        feat_t = teacher.debug["point_features"]
    student = OracleMinkUNet(params, cfg, backend=backend, training=training)
    logits = student.forward_minkunet_ms(coords, torch.as_tensor(feats))
    feat_s = student.debug["point_features"]
    loss_seg = loss_ce_lovasz(logits, torch.as_tensor(labels).long(), ignore, label_smoothing)
    c, cg = np.asarray(coords, dtype=np.int32), np.asarray(coords_gt, dtype=np.int32)
    s2t = O.sphashquery(O.sphash(c), O.sphash(cg))
    batch_size = int(c[:, 3].max()) + 1
    loss_kd = 0
    for b in range(batch_size):
        pick = np.nonzero((s2t >= 0) & (c[:, 3] == b))[0]
        loss_kd = loss_kd + TF.mse_loss(feat_s[pick], feat_t[s2t[pick]]) * feat_kd_weight / batch_size
    return dict(loss=loss_seg + loss_kd, loss_seg=loss_seg, loss_feat_kd=loss_kd, logits=logits, teacher_logits=teacher_logits)
