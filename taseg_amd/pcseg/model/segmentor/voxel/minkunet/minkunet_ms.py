"""MinkUNetMs - TASeg's multi-scan (FSA / SMSA) segmentor
(reference pcseg/model/segmentor/voxel/minkunet/minkunet_ms.py:186-461).

The temporally aggregated cloud `batch_dict['lidar_ms']` (current scan + class-filtered,
pose-aligned history scans + time flag, built by the data stage) is already voxel-unique, so
unlike `MinkUNet` there is no device re-voxelisation: the stem consumes it directly
(minkunet_ms.py:386-392).  Same parameters / state_dict as `MinkUNet`.
"""
from .minkunet import MinkUNetBackbone

__all__ = ["MinkUNetMs"]


class MinkUNetMs(MinkUNetBackbone):
    def prepare(self, batch_dict):
        x_ms = batch_dict["lidar_ms"]
        plan = self._index_plan(x_ms.C, x_ms.C.float())
        batch_dict["_plan"] = plan
        return plan

    def forward(self, batch_dict, return_logit=False, return_tta=False):
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
        invs_ms = batch_dict["inverse_map_ms"]
        all_labels = batch_dict["targets_mapped"]
        point_mask = batch_dict["point_mask"]
        num_points_ms = batch_dict["num_points_ms"]
        point_predict, point_labels, point_predict_logits = [], [], []
        cursor = 0
        for idx in range(int(invs_ms.C[:, -1].max()) + 1):
            scene = x_ms.C[:, -1] == idx
            cur_inv = invs_ms.F[invs_ms.C[:, -1] == idx]
            n_ms = int(num_points_ms[idx])
            keep = point_mask[cursor: cursor + n_ms]
            mapped = out_ms[scene][cur_inv][keep]
            n_cur = int(batch_dict["num_points"][idx])
            if return_logit or return_tta:
                point_predict.append(mapped.softmax(1)[:n_cur].cpu().numpy())
            else:
                point_predict.append(mapped.argmax(1)[:n_cur].cpu().numpy())
                point_predict_logits.append(mapped[:n_cur].cpu().numpy())
            point_labels.append(all_labels.F[all_labels.C[:, -1] == idx][:n_cur].cpu().numpy())
            cursor += n_ms
        return {"point_predict": point_predict, "point_labels": point_labels, "name": batch_dict["name"],
                "point_predict_logits": point_predict_logits}

    def forward_ensemble(self, batch_dict):
        return self.forward(batch_dict, return_tta=True)
