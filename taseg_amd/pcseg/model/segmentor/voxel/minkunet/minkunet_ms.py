"""MinkUNetMs - TASeg's multi-scan (FSA / SMSA) segmentor
(reference pcseg/model/segmentor/voxel/minkunet/minkunet_ms.py:186-461).

The temporally aggregated cloud `batch_dict['lidar_ms']` (current scan + class-filtered,
pose-aligned history scans + time flag, built by the data stage) is already voxel-unique, so
unlike `MinkUNet` there is no device re-voxelisation: the stem consumes it directly
(minkunet_ms.py:386-392).  Same parameters / state_dict as `MinkUNet`.
"""
from .minkunet import MinkUNetBackbone, unvoxelise_predictions

__all__ = ["MinkUNetMs"]


class MinkUNetMs(MinkUNetBackbone):
    def prepare(self, batch_dict):
        x_ms = batch_dict["lidar_ms"]
        plan = self._stage_prepare(self._index_plan(x_ms.C, x_ms.C.float(), backward=self.training))
        batch_dict["_plan"] = plan
        return plan

    def forward(self, batch_dict, return_logit=False, return_tta=False, defer=False):
        x_ms = batch_dict["lidar_ms"]
        x_ms.F = x_ms.F[:, :self.in_feature_dim]
        plan = batch_dict.get("_plan") or self.prepare(batch_dict)
        out_ms = self._unet(x_ms.F, x_ms.F, plan)

        if self.training:
            target_ms = batch_dict["targets_ms"].F.long().cuda(non_blocking=True)
            return self._train_outputs(out_ms, target_ms, batch_dict["lidar_ms"].C[:, :3].float(),
                                       batch_dict["offset_ms"])

        # evaluation: un-voxelise onto every point of the fused cloud, keep the current-frame
        # points (`point_mask`), trim to the scan's own point count (minkunet_ms.py:433-458)
        return unvoxelise_predictions(out_ms, x_ms.C[:, -1], batch_dict["inverse_map_ms"], batch_dict["targets_mapped"],
                                      batch_dict["num_points"], return_logit or return_tta, point_mask=batch_dict["point_mask"],
                                      num_points_ms=batch_dict["num_points_ms"], names=batch_dict["name"], defer=defer)

    def forward_ensemble(self, batch_dict):
        return self.forward(batch_dict, return_tta=True)
