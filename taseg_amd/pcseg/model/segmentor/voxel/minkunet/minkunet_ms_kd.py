"""MinkUNetMsKd - TASeg's mask-distillation segmentor (reference
pcseg/model/segmentor/voxel/minkunet/minkunet_ms_kd.py:199-721).

A frozen teacher MinkUNet (modules `stem_gt ... classifier_gt`, loaded from a trained MinkUNetMs checkpoint) runs on
the cloud aggregated with ground-truth labels (`lidar_ms_gt`); the student (`stem ... classifier`) runs on the cloud
aggregated with pseudo labels (`lidar_ms`).  Besides the segmentation loss the student's 480-dim per-voxel features
are pulled towards the teacher's with an MSE over the voxels present in BOTH clouds, matched by coordinate hash
(`sphash` / `sphashquery`, :613-615) and randomly sub-sampled to at most MAX_VOXEL per sample (:621-633).
Same state_dict keys as the reference (teacher keys carry the `_gt` suffix), so its checkpoints and its
checkpoint-renaming loader work.
"""
import os

import torch
from torch import nn

from taseg_amd.torchsparse.nn import functional as spF
from taseg_amd.options import options
from .minkunet import LazyScalar, MinkUNetBackbone, unvoxelise_predictions

__all__ = ["MinkUNetMsKd"]

# TASEG_KD_LOSS_ON_DEVICE=0: the reference's literal per-sample loop (three host reads per sample) instead of _kd_loss_on_device
_KD_ON_DEVICE = options.kd_loss_on_device

_PARTS = ("stem", "stage1", "stage2", "stage3", "stage4", "up1", "up2", "up3", "up4", "classifier", "dropout")


class MinkUNetMsKd(MinkUNetBackbone):
    def __init__(self, model_cfgs, num_class: int):
        super().__init__(model_cfgs, num_class)
        self.sampling_type = model_cfgs.get("SAMPLING_TYPE", "uncertain")
        self.max_voxel = model_cfgs.get("MAX_VOXEL", 3000)
        self.feat_kd = model_cfgs.get("FEAT_KD", "mse")
        self.feat_kd_weight = model_cfgs.get("FEAT_KD_WEIGHT", 1.0)
        if self.feat_kd != "mse":
            raise NotImplementedError(f"FEAT_KD '{self.feat_kd}' (the reference implements 'mse' only)")
        # the teacher is a second backbone whose parts are registered under the reference's `*_gt` names; the backbone
        # object itself stays unregistered (its parts must not appear twice in the state_dict)
        teacher = MinkUNetBackbone(model_cfgs, num_class)
        for part in _PARTS:
            setattr(self, part + "_gt", getattr(teacher, part))
        object.__setattr__(self, "_teacher", teacher)

    def prepare(self, batch_dict):
        x_ms = batch_dict["lidar_ms"]
        plan = self._index_plan(x_ms.C, x_ms.C.float())
        x_gt = batch_dict["lidar_ms_gt"]
        plan["teacher"] = self._index_plan(x_gt.C, x_gt.C.float())
        with torch.no_grad():          # student voxel -> teacher voxel (or -1), by coordinate hash (:613-615)
            plan["s2t"] = spF.sphashquery(spF.sphash(x_ms.C.int()), spF.sphash(x_gt.C.int()))
            # (:618 - read here, where the batch is staged, so that the step itself reads nothing back)
            plan["batch_size"] = int(x_ms.C[:, -1].max()) + 1
        batch_dict["_plan"] = plan
        return plan

    def _kd_loss_on_device(self, feat_s, feat_t, s2t, batch_col, batch_size):
        """The feature-distillation term (minkunet_ms_kd.py:617-633: per sample, the voxels present in both clouds, a uniformly
        random subset of at most MAX_VOXEL of them, MSE against the teacher's features, weighted mean over the samples) without the
        reference's per-sample host reads (`max()`, `sum() >`, `nonzero()`): one random key per voxel, one sort by (sample, key) -
        a voxel is picked iff it is among its sample's first MAX_VOXEL candidates in that order - and the <= B * MAX_VOXEL picked rows
        gathered through a fixed-size index list.  Same distribution of picks, same value when nothing is sub-sampled (up to
        summation order); a sample without common voxels makes the term NaN like `mse_loss` of an empty selection does."""
        n, dev = s2t.shape[0], s2t.device
        b = batch_col.long()
        cand = s2t >= 0
        if batch_size <= 64:                # 31-bit keys: [sample | not a candidate | 24 random bits] - four radix passes
            key = torch.where(cand, torch.randint(0, 1 << 24, (n,), device=dev, dtype=torch.int32),
                              torch.full((), 1 << 24, device=dev, dtype=torch.int32))
            order = torch.argsort(batch_col.int() * (1 << 25) + key)
        else:
            key = torch.where(cand, torch.rand(n, device=dev, dtype=torch.float64), torch.full((), 2.0, device=dev, dtype=torch.float64))
            order = torch.argsort(b.double() * 4.0 + key)                       # sample-major, candidates first, random among them
        # (rows per sample without torch.bincount: it reads the largest value back to size its result - a host wait for the whole
        # forward pass of both networks in the middle of the step, 45 ms at bs 6)
        rows = torch.zeros(batch_size, dtype=torch.int64, device=dev).index_add_(0, b.clamp(max=batch_size - 1), torch.ones_like(b))
        start = torch.cumsum(rows, 0) - rows
        rank = torch.empty(n, dtype=torch.int64, device=dev)
        rank[order] = torch.arange(n, device=dev) - start[b[order]]
        pick = cand & (rank < self.max_voxel)
        n_b = torch.zeros(batch_size, dtype=torch.float32, device=dev).index_add_(0, b, pick.float())
        cap = min(n, batch_size * int(self.max_voxel))
        sel = torch.argsort((~pick).to(torch.uint8), stable=True)[:cap]   # the picked rows first (ascending row), fixed length
        valid = pick[sel].float()
        d = feat_s[sel] - feat_t[s2t[sel].clamp(min=0)].detach()
        per_row = (d * d).sum(1) * valid / (n_b[b[sel]].clamp(min=1.0) * feat_s.shape[1])
        empty = torch.where(n_b == 0, torch.full((), float("nan"), device=dev), torch.zeros((), device=dev)).sum()
        return (per_row.sum() + empty) * (self.feat_kd_weight / batch_size)

    def forward(self, batch_dict, return_logit=False, return_tta=False):
        plan = batch_dict.get("_plan") or self.prepare(batch_dict)
        x_gt = batch_dict["lidar_ms_gt"]
        x_gt.F = x_gt.F[:, :self.in_feature_dim]
        with torch.no_grad():
            # (concat=True: the three feature blocks interpolated straight into the columns of one matrix, as MinkUNet's own pass)
            feat_t = self._teacher._unet_point_features(x_gt.F, x_gt.F, plan["teacher"], concat=True)
            batch_dict["teacher_logits"] = self.classifier_gt(feat_t)     # the reference computes them too (:573), unused
        x_ms = batch_dict["lidar_ms"]
        x_ms.F = x_ms.F[:, :self.in_feature_dim]
        feat_s = self._unet_point_features(x_ms.F, x_ms.F, plan, concat=True)
        out_ms = self.classifier(feat_s)

        if self.training:
            target = batch_dict["targets_ms"].F.long().cuda(non_blocking=True)
            loss_seg = self.criterion_losses(out_ms, target, xyz=x_ms.C[:, :3].float(), offset=batch_dict["offset_ms"])
            if self.sampling_type != "random":
                raise NotImplementedError("SAMPLING_TYPE must be 'random' (the only branch the reference implements, :621)")
            s2t = plan["s2t"]
            batch_size = plan["batch_size"]
            if _KD_ON_DEVICE:
                loss_kd = self._kd_loss_on_device(feat_s, feat_t, s2t, x_ms.C[:, -1], batch_size)
            else:
                loss_kd = out_ms.new_zeros(())
                for b in range(batch_size):
                    pick = ((s2t >= 0) & (x_ms.C[:, -1] == b)).nonzero().reshape(-1)
                    if pick.numel() > self.max_voxel:
                        pick = pick[torch.randperm(pick.numel(), device=pick.device)[:self.max_voxel]]
                    mse = torch.nn.functional.mse_loss(feat_s[pick], feat_t[s2t[pick]].detach())
                    loss_kd = loss_kd + mse * self.feat_kd_weight / batch_size
            loss = loss_seg + loss_kd
            disp = {"loss": LazyScalar(loss), "loss_seg": LazyScalar(loss_seg), "loss_feat_kd": LazyScalar(loss_kd)}
            return {"loss": loss}, disp, dict(disp)

        # the evaluation tail of MinkUNetMs for the whole batch at once (minkunet.unvoxelise_predictions)
        return unvoxelise_predictions(out_ms, x_ms.C[:, -1], batch_dict["inverse_map_ms"], batch_dict["targets_mapped"],
                                      batch_dict["num_points"], return_logit or return_tta, point_mask=batch_dict["point_mask"],
                                      num_points_ms=batch_dict["num_points_ms"], names=batch_dict["name"])

    def forward_ensemble(self, batch_dict):
        return self.forward(batch_dict, return_tta=True)

    def load_params_from_file(self, filename, logger, to_cpu=False):
        """A MinkUNetMs checkpoint initialises BOTH networks: every key is loaded as is (student) and once more under
        its `_gt` name (teacher) (minkunet_ms_kd.py:680-717)."""
        if not os.path.isfile(filename):
            raise FileNotFoundError
        logger.info("==> Loading parameters from checkpoint %s to %s" % (filename, "CPU" if to_cpu else "GPU"))
        state = torch.load(filename, map_location=torch.device("cpu") if to_cpu else None)
        state = state.get("model_state", state)
        logger.info(f"==> Done {self.load_params(self.with_teacher_keys(state))}")

    @staticmethod
    def with_teacher_keys(state):
        out = dict(state)
        for k, v in state.items():
            k = k[len("module."):] if k.startswith("module.") else k        # DDP prefix
            head, _, rest = k.partition(".")
            if head in _PARTS:
                out[f"{head}_gt.{rest}" if rest else f"{head}_gt"] = v
        return out

    def fix_part_param(self, keywards="gt"):
        for name, p in self.named_parameters():
            if keywards in name:
                p.requires_grad = False
