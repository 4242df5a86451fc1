"""Golden vectors added in round 3 (build container only; reads /root/reference, inert on the GPU box).

    python tests/golden/make_golden_r3.py

Same arrangement as make_golden_r2.py (the REAL reference's Python, import environment of _ref_env.py, no reference code in
this file):

  tiaf_nus.npz
      nuScenes TIAF dataset stage: NuscenesMsMmDataset.__getitem__ (nuscenes_ms_mm.py:142-194: the FSA fuse of
      nuscenes_ms.py plus multiscan_fuse_fov :196-327 - image keyframes by driven distance incl. the random fill-up,
      per keyframe the cloud with its interval predecessors, paint radius, per camera view get_fov_points :329-401 through
      calibrated-sensor / ego-pose records, half-resolution pixels, transform into the current frame) ->
      NuscVoxelMsMmDataset.get_single_sample / collate_batch (nuscenes_voxel_ms_mm.py:77-262), on two synthetic scenes in
      the reference's own bookkeeping.  Third-party pieces the reference imports and this image lacks are served under
      their import names from oracle/ts_oracle.py, which restates their published algorithms: pyquaternion's
      Quaternion.rotation_matrix (as for multiscan_nus.npz) and nuscenes-devkit's geometry_utils.view_points; PIL (present)
      does the image resize as in the reference.  Recorded in the fixture (`third_party`).
"""
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden_r2 as G2  # noqa: E402  (sets up the reference import environment; generates nothing on import)
from make_golden_r2 import _ref_env, BACKEND_DESC, FLEX_NUS, nus_scene  # noqa: E402

VIEWS = ['CAM_FRONT', 'CAM_FRONT_RIGHT', 'CAM_BACK_RIGHT', 'CAM_BACK', 'CAM_BACK_LEFT', 'CAM_FRONT_LEFT']
VIEW_YAW = [0.0, -55.0, -110.0, 180.0, 110.0, 55.0]            # degrees, ego frame (x forward, y left)
IMG_H, IMG_W = 68, 96                                           # full-size camera image; half size minus 2 rows = 32 x 48 (UNet2D: multiples of 16)


def _rot_to_quat(m):
    """unit quaternion (w, x, y, z) of a rotation matrix (trace method)"""
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        return [0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s]
    i = int(np.argmax(np.diag(m)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(1.0 + m[i, i] - m[j, j] - m[k, k]) * 2
    q = [0.0] * 4
    q[0] = (m[k, j] - m[j, k]) / s
    q[1 + i] = 0.25 * s
    q[1 + j] = (m[j, i] + m[i, j]) / s
    q[1 + k] = (m[k, i] + m[i, k]) / s
    return [float(v) for v in q]


def _cam_rotation(yaw_deg):
    """camera -> ego rotation of a camera looking along the ego-frame direction `yaw` (camera: z forward, x right, y down)"""
    a = np.deg2rad(yaw_deg)
    fwd = np.array([np.cos(a), np.sin(a), 0.0])
    right = np.array([np.sin(a), -np.cos(a), 0.0])
    down = np.array([0.0, 0.0, -1.0])
    return np.stack([right, down, fwd], axis=1)


def _install_devkit_stub():
    from oracle import ts_oracle as O
    for name in ("nuscenes", "nuscenes.utils", "nuscenes.utils.geometry_utils"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["nuscenes.utils.geometry_utils"].view_points = O.view_points
    sys.modules["nuscenes"].utils = sys.modules["nuscenes.utils"]
    sys.modules["nuscenes.utils"].geometry_utils = sys.modules["nuscenes.utils.geometry_utils"]


def camera_tables(sc, seed):
    """devkit records of the cameras and of the lidar key frames of a nus_scene(): sample_data / calibrated_sensor / ego_pose
    tables, image and semantic-map files (in memory)."""
    rs = np.random.RandomState(500 + seed)
    tabs = {"sample_data": {}, "calibrated_sensor": {}, "ego_pose": {}}
    tabs["calibrated_sensor"]["cs_lidar"] = {"rotation": sc["infos"][0]["lidar2ego_rotation"],
                                             "translation": sc["infos"][0]["lidar2ego_translation"]}
    for v, name in enumerate(VIEWS):
        tabs["calibrated_sensor"]["cs_" + name] = {"rotation": _rot_to_quat(_cam_rotation(VIEW_YAW[v])),
                                                   "translation": [1.5 - 0.2 * v, 0.1 * (v - 2), 1.5],
                                                   "camera_intrinsic": [[70.0 + v, 0.0, IMG_W / 2 - 1.3 * v],
                                                                        [0.0, 70.0 + v, IMG_H / 2 + 0.7 * v], [0.0, 0.0, 1.0]]}
    files = {}
    for i, info in enumerate(sc["infos"]):
        sd = sc["sample_tab"][info["token"]]["data"]["LIDAR_TOP"]
        tabs["ego_pose"]["pose_" + sd] = {"rotation": info["ego2global_rotation"], "translation": info["ego2global_translation"]}
        tabs["sample_data"][sd] = {"calibrated_sensor_token": "cs_lidar", "ego_pose_token": "pose_" + sd}
        for v, name in enumerate(VIEWS):
            tok = f"{name}_{i:02d}"
            sc["sample_tab"][info["token"]]["data"][name] = tok
            # the camera fires a few ms away from the lidar sweep: its own ego pose
            t = np.array(info["ego2global_translation"]) + np.array([0.011 * (v + 1), -0.004 * v, 0.0])
            q = np.array(info["ego2global_rotation"]) + np.array([0.0, 0.0, 0.0, 2e-4 * (v + 1)])
            tabs["ego_pose"]["pose_" + tok] = {"rotation": (q / np.linalg.norm(q)).tolist(), "translation": t.tolist()}
            fname = f"samples/{name}/{tok}.jpg"
            tabs["sample_data"][tok] = {"calibrated_sensor_token": "cs_" + name, "ego_pose_token": "pose_" + tok, "filename": fname}
            files["/nus/" + fname] = rs.randint(0, 256, size=(IMG_H, IMG_W, 3)).astype(np.uint8)
            view = '' if name == 'CAM_FRONT' else name.replace('CAM', '')
            files["/nus/" + fname.replace(name, 'SEMANTIC_MAP_SAM%s' % view).replace('.jpg', '.npy')] = \
                rs.randint(0, 17, size=(IMG_H // 2, IMG_W // 2, 1)).astype(np.float32)
    return tabs, files


def gen_tiaf_nus(fname="tiaf_nus.npz", multiscan=4, step=1.0):
    import yaml
    from PIL import Image
    G2._install_pyquaternion()
    _install_devkit_stub()
    _ref_env._pkg("pcseg.data.dataset.nuscenes", os.path.join(_ref_env.REF, "pcseg", "data", "dataset", "nuscenes"))
    _ref_env._pkg("tools", os.path.join(_ref_env.REF, "tools"))
    _ref_env._pkg("tools.utils", os.path.join(_ref_env.REF, "tools", "utils"))
    _ref_env._pkg("tools.utils.common", os.path.join(_ref_env.REF, "tools", "utils", "common"))
    for alias, typ in (("int", int), ("bool", bool), ("float", float)):
        if not hasattr(np, alias):
            setattr(np, alias, typ)                              # aliases numpy >= 1.24 dropped
    from pcseg.data.dataset.nuscenes.nuscenes_ms_mm import NuscenesMsMmDataset
    from pcseg.data.dataset.nuscenes.nuscenes_voxel_ms_mm import NuscVoxelMsMmDataset
    with open(os.path.join(_ref_env.REF, "pcseg", "data", "dataset", "nuscenes", "nuscenes.yaml")) as f:
        learning_map = yaml.safe_load(f)["learning_map"]
    lm = np.zeros(256, dtype=np.int64)
    for k, v in learning_map.items():
        lm[k] = v
    height, width = IMG_H // 2 - 2, IMG_W // 2
    out = {"backend": np.array(BACKEND_DESC), "multiscan": np.array(multiscan), "step": np.array(step),
           "steps": np.array(FLEX_NUS), "learning_map": lm, "height": np.array(height), "width": np.array(width),
           "views": np.array(VIEWS),
           "third_party": np.array("pyquaternion and nuscenes-devkit absent: Quaternion.rotation_matrix and geometry_utils."
                                   "view_points restated in oracle/ts_oracle.py; PIL %s present" % Image.__version__)}
    # sample 0: a keyframe picked per 2 m, nothing passed over; sample 1: 6 m steps on a short scene - two picked, one drawn
    # from the passed-over keyframes by random.sample
    cases = [dict(seed=71, back=0, multiscan_image=2, step_image=2.0, interval=1, used_view=[0, 3], paint_dist=25.0, rng=1234),
             dict(seed=72, back=0, multiscan_image=3, step_image=6.0, interval=2, used_view=[5, 0, 1], paint_dist=-1, rng=99)]
    samples = []
    for b, case in enumerate(cases):
        sc = nus_scene(case["seed"])
        tabs, cam_files = camera_tables(sc, case["seed"])
        sc["files"].update(cam_files)

        class FakeNusc:
            dataroot = "/nus"

            def get(self, table, token):
                if table == "sample":
                    return sc["sample_tab"][token]
                if table == "lidarseg":
                    return sc["lidarseg_tab"][token]
                return tabs[table][token]

            def get_sample_data(self, token):
                rec = tabs["sample_data"][token]
                return os.path.join(self.dataroot, rec["filename"]), [], np.array(tabs["calibrated_sensor"][rec["calibrated_sensor_token"]]["camera_intrinsic"])

        ds = object.__new__(NuscenesMsMmDataset)
        ds.root_path, ds.data_path_ceph, ds.split, ds.seq, ds.augment, ds.tta = "/nus", None, "val", -1, "none", False
        ds.nusc, ds.nusc_infos, ds.nusc_infos_sweep = FakeNusc(), sc["infos"], sc["sweeps"]
        ds.global_indexes, ds.local_indexes, ds.scene_tokens = sc["global_indexes"], sc["local_indexes"], sc["scene_tokens"]
        ds.token2samplelist, ds.token2samplelist_fov, ds.learning_map = {}, {}, learning_map
        ds.multiscan, ds.step, ds.flexible_steps, ds.pseudo_mask = multiscan, step, FLEX_NUS, "mink_sweep_notta"
        ds.img_view, ds.used_view, ds.resize = list(VIEWS), case["used_view"], 0.5
        ds.multiscan_image, ds.step_image, ds.multiscan_interval = case["multiscan_image"], case["step_image"], case["interval"]
        ds.height, ds.width, ds.image_jitter, ds.image_flip, ds.paint_dist = height, width, False, False, case["paint_dist"]
        ds.get_path_infos_cam_lidar()
        real_fromfile, real_load, real_open, real_array = np.fromfile, np.load, Image.open, np.array
        np.fromfile = lambda path, dtype=None, count=-1, **kw: sc["files"][path].copy()
        np.load = lambda path, *a, **kw: sc["files"][path].copy()
        Image.open = lambda path, *a, **kw: Image.fromarray(sc["files"][path])
        # the reference is written against numpy 1.x, where np.array(..., copy=False) means "copy only if needed" (:384)
        np.array = lambda obj, *a, copy=True, **kw: real_array(obj, *a, copy=(None if copy is False else copy), **kw)
        index = len(sc["infos"]) - 1 - case["back"]
        try:
            random.seed(case["rng"])
            pc_data = ds[index]
            lidar_sd = sc["sample_tab"][sc["infos"][index]["token"]]["data"]["LIDAR_TOP"]
            frames = list(ds.token2samplelist_fov[lidar_sd])
            sweeps = list(ds.token2samplelist[lidar_sd])
        finally:
            np.fromfile, np.load, Image.open, np.array = real_fromfile, real_load, real_open, real_array
        p = f"b{b}_"
        for k, v in case.items():
            out[p + "cfg_" + k] = np.array(v)
        out[p + "index"] = np.array(index)
        out[p + "image_keyframes"] = np.array(frames)
        out[p + "sample_list"] = np.array(sweeps)
        for k in ("xyzret", "xyzret_ms", "xyzret_fov_ms", "image_ms", "semantic_map_ms"):
            out[p + k] = np.asarray(pc_data[k], dtype=np.float32)
        for k in ("depth_map_ms", "lidar_map_ms"):
            out[p + k + "_shape"] = np.array(pc_data[k].shape)
            assert not np.any(pc_data[k])
        for k in ("labels", "labels_ms", "labels_fov_ms"):
            out[p + k] = pc_data[k].reshape(-1).astype(np.int64)
        # the scene: bookkeeping arrays (as multiscan_nus.npz), every keyframe's cloud / labels, the frames of the FSA list
        g0 = sc["global_indexes"][index]
        out[p + "scene_tokens"] = np.array(sc["scene_tokens"])
        out[p + "local_indexes"] = np.array(sc["local_indexes"])
        out[p + "global_indexes"] = np.array(sc["global_indexes"])
        out[p + "is_key"] = np.array(["lidar_path" in s for s in sc["sweeps"]])
        out[p + "timestamps"] = np.array([s["timestamp"] for s in sc["sweeps"]], dtype=np.int64)
        key_of = {id(info): i for i, info in enumerate(sc["infos"])}
        out[p + "key_index"] = np.array([key_of.get(id(s), -1) for s in sc["sweeps"]])
        for name, field in (("key_l2e_q", "lidar2ego_rotation"), ("key_l2e_t", "lidar2ego_translation"),
                            ("key_e2g_q", "ego2global_rotation"), ("key_e2g_t", "ego2global_translation")):
            out[p + name] = np.array([i[field] for i in sc["infos"]], dtype=np.float64)
        s2l_r, s2l_t = np.zeros((len(sc["sweeps"]), 3, 3)), np.zeros((len(sc["sweeps"]), 3))
        for g, s in enumerate(sc["sweeps"]):
            if "data_path" in s:
                s2l_r[g], s2l_t[g] = s["sensor2lidar_rotation"], s["sensor2lidar_translation"]
        out[p + "s2l_r"], out[p + "s2l_t"] = s2l_r, s2l_t
        for d in sorted(set(sweeps)):
            s = sc["sweeps"][g0 + d]
            path = "/nus/" + (s["lidar_path"] if "lidar_path" in s else s["data_path"])[16:]
            tok = sc["sample_tab"][s["token"]]["data"]["LIDAR_TOP"] if "lidar_path" in s else s["sample_data_token"]
            out[f"{p}points_d{-d}"] = sc["files"][path]
            out[f"{p}pseudo_d{-d}"] = sc["files"]["/YourHome/PCSeg/logs/voxel/nuscenes/minkunet_mk34_cr10/default/results/"
                                                 "lidarseg/trainval_sweep_notta/" + tok + "_lidarseg.bin"]
            if "lidar_path" in s:
                out[f"{p}rawlabels_d{-d}"] = sc["files"]["/nus/" + sc["lidarseg_tab"][tok]["filename"]]
        for i, info in enumerate(sc["infos"]):
            sd = sc["sample_tab"][info["token"]]["data"]["LIDAR_TOP"]
            out[f"{p}key{i}_points"] = sc["files"]["/nus/" + info["lidar_path"][16:]]
            out[f"{p}key{i}_rawlabels"] = sc["files"]["/nus/" + sc["lidarseg_tab"][sd]["filename"]]
        # cameras: per view the calibrated sensor, per (keyframe, view) the ego pose, image, semantic map
        out[p + "lidar_cs_q"] = np.array(tabs["calibrated_sensor"]["cs_lidar"]["rotation"], dtype=np.float64)
        out[p + "lidar_cs_t"] = np.array(tabs["calibrated_sensor"]["cs_lidar"]["translation"], dtype=np.float64)
        for v in case["used_view"]:
            cs = tabs["calibrated_sensor"]["cs_" + VIEWS[v]]
            out[f"{p}view{v}_cs_q"], out[f"{p}view{v}_cs_t"] = np.array(cs["rotation"], dtype=np.float64), np.array(cs["translation"], dtype=np.float64)
            out[f"{p}view{v}_intrinsic"] = np.array(cs["camera_intrinsic"], dtype=np.float64)
            for d in frames:
                i = index + d
                tok = f"{VIEWS[v]}_{i:02d}"
                pose = tabs["ego_pose"]["pose_" + tok]
                out[f"{p}key{i}_view{v}_pose_q"], out[f"{p}key{i}_view{v}_pose_t"] = np.array(pose["rotation"]), np.array(pose["translation"])
                out[f"{p}key{i}_view{v}_image"] = sc["files"]["/nus/" + tabs["sample_data"][tok]["filename"]]
                view = '' if VIEWS[v] == 'CAM_FRONT' else VIEWS[v].replace('CAM', '')
                out[f"{p}key{i}_view{v}_semantic"] = sc["files"]["/nus/" + tabs["sample_data"][tok]["filename"].replace(
                    VIEWS[v], 'SEMANTIC_MAP_SAM%s' % view).replace('.jpg', '.npy')]
        vox = object.__new__(NuscVoxelMsMmDataset)
        vox.point_cloud_dataset = [pc_data]
        vox.in_feature_dim, vox.training, vox.if_tta, vox.voxel_size, vox.num_points = 4, False, False, 0.1, 1000000
        samples.append(vox.get_single_sample(0))
    batch = NuscVoxelMsMmDataset.collate_batch(samples)
    for key in G2.BATCH_SPARSE + ("lidar_fov_ms", "targets_fov_ms"):
        out[f"batch_{key}_C"] = batch[key].C.numpy()
        out[f"batch_{key}_F"] = batch[key].F.numpy()
    for key in G2.BATCH_DENSE + ("offset_img", "image_ms", "semantic_map_ms"):
        out[f"batch_{key}"] = batch[key].numpy()
    for key in ("depth_map_ms", "lidar_map_ms"):
        out[f"batch_{key}_shape"] = np.array(batch[key].shape)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, os.path.getsize(os.path.join(HERE, fname)) // 1024, "KiB; image keyframes",
          [out[f"b{b}_image_keyframes"].tolist() for b in range(2)], "fov points", [out[f"b{b}_xyzret_fov_ms"].shape for b in range(2)],
          "fov voxels", out["batch_lidar_fov_ms_C"].shape, "images", out["batch_image_ms"].shape)


if __name__ == "__main__":
    print("reference backend:", BACKEND_DESC)
    gen_tiaf_nus()
