"""nuScenes info reader (taseg_amd/data/nuscenes_reader.py): a nuScenes tree - info pickles, devkit JSON tables, lidar /
lidarseg / pseudo-label .bin files - is written from the scenes of tests/golden/multiscan_nus.npz, read back, and must lead
to what the REAL reference's NuscenesMsDataset produced from the same scenes: the bookkeeping arrays and the sweep selection
on the CPU, the fused cloud and labels through the device stage on the GPU box (nuscenes_ms.py:19-131, 226-330)."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from taseg_amd.data import nuscenes as N
from taseg_amd.data.nuscenes_reader import NuscInfoReader, NuscTables

PSEUDO = "pseudo/trainval_sweep_notta"


def _write_tree(root, g, b):
    """scene of sample `b` of the fixture in the reference's on-disk bookkeeping; returns the keyframe index to load"""
    p = f"b{b}_"
    is_key, key_index = g[p + "is_key"], g[p + "key_index"]
    stamps, scenes = g[p + "timestamps"], g[p + "scene_tokens"].tolist()
    n_key = len(g[p + "global_indexes"])
    infos = []
    for i in range(n_key):
        gi = int(g[p + "global_indexes"][i])
        infos.append({"lidar_path": f"./data/nuscenes/samples/LIDAR_TOP/{gi:03d}.bin", "token": f"sample{gi:03d}",
                      "timestamp": int(stamps[gi]), "lidar2ego_rotation": g[p + "key_l2e_q"][i].tolist(),
                      "lidar2ego_translation": g[p + "key_l2e_t"][i].tolist(), "ego2global_rotation": g[p + "key_e2g_q"][i].tolist(),
                      "ego2global_translation": g[p + "key_e2g_t"][i].tolist()})
    sweeps = []
    for f in range(len(is_key)):
        if is_key[f]:
            sweeps.append(infos[int(key_index[f])])
        else:
            sweeps.append({"data_path": f"./data/nuscenes/sweeps/LIDAR_TOP/{f:03d}.bin", "sample_data_token": f"sweep{f:03d}",
                           "timestamp": int(stamps[f]), "sensor2lidar_rotation": g[p + "s2l_r"][f], "sensor2lidar_translation": g[p + "s2l_t"][f]})
    os.makedirs(os.path.join(root, "v1.0-trainval"), exist_ok=True)
    with open(os.path.join(root, "nuscenes_infos_val.pkl"), "wb") as fh:
        pickle.dump({"infos": infos, "metadata": {"version": "v1.0-trainval"}}, fh)
    with open(os.path.join(root, "nuscenes_infos_val_sweep.pkl"), "wb") as fh:
        pickle.dump({"infos_sweep": sweeps, "global_indexes": g[p + "global_indexes"].tolist(),
                     "local_indexes": g[p + "local_indexes"].tolist(), "scene_tokens": scenes}, fh)
    # devkit tables: two sensors, a camera key-frame record and non-key-frame lidar records beside the ones that count
    tables = {"sensor": [{"token": "s_lidar", "channel": "LIDAR_TOP"}, {"token": "s_cam", "channel": "CAM_FRONT"}],
              "calibrated_sensor": [{"token": "cs_lidar", "sensor_token": "s_lidar"}, {"token": "cs_cam", "sensor_token": "s_cam"}],
              "sample": [], "sample_data": [], "lidarseg": []}
    for f in range(len(is_key)):
        father = infos[int(g[p + "local_indexes"][f])]["token"]
        if is_key[f]:
            tok = f"sample{f:03d}"
            tables["sample"].append({"token": tok, "scene_token": scenes[f]})
            tables["sample_data"].append({"token": f"cam{f:03d}", "sample_token": tok, "calibrated_sensor_token": "cs_cam", "is_key_frame": True})
            tables["sample_data"].append({"token": f"sd{f:03d}", "sample_token": tok, "calibrated_sensor_token": "cs_lidar", "is_key_frame": True})
            tables["lidarseg"].append({"sample_data_token": f"sd{f:03d}", "filename": f"lidarseg/v1.0-trainval/sd{f:03d}_lidarseg.bin"})
        else:
            tables["sample_data"].append({"token": f"sweep{f:03d}", "sample_token": father, "calibrated_sensor_token": "cs_lidar", "is_key_frame": False})
    for name, rows in tables.items():
        with open(os.path.join(root, "v1.0-trainval", name + ".json"), "w") as fh:
            json.dump(rows, fh)
    # the files the sample reads: the current keyframe and the frames the reference selected
    index = int(g[p + "index"])
    g0 = int(g[p + "global_indexes"][index])

    def put(rel, arr, dtype):
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        np.asarray(arr, dtype=dtype).tofile(path)

    put(infos[index]["lidar_path"][16:], g[p + "points_cur"], np.float32)
    put(f"lidarseg/v1.0-trainval/sd{g0:03d}_lidarseg.bin", g[p + "rawlabels_cur"], np.uint8)
    for d in g[p + "sample_list"].tolist():
        f = g0 + d
        fr = sweeps[f]
        put((fr["lidar_path"] if is_key[f] else fr["data_path"])[16:], g[f"{p}points_d{-d}"], np.float32)
        tok = f"sd{f:03d}" if is_key[f] else f"sweep{f:03d}"
        put(f"{PSEUDO}/{tok}_lidarseg.bin", g[f"{p}pseudo_d{-d}"], np.uint8)
        if f"{p}rawlabels_d{-d}" in g:
            put(f"lidarseg/v1.0-trainval/sd{f:03d}_lidarseg.bin", g[f"{p}rawlabels_d{-d}"], np.uint8)
    return index


def _reader(root, g):
    lm = {i: int(v) for i, v in enumerate(g["learning_map"].tolist())}
    return NuscInfoReader(root, split="val", learning_map=lm, pseudo_dir=os.path.join(root, PSEUDO))


@pytest.mark.parametrize("b", [0, 1])
def test_info_files_give_the_reference_bookkeeping(tmp_path, g_multiscan_nus, b):
    g = g_multiscan_nus
    index = _write_tree(str(tmp_path), g, b)
    rd = _reader(str(tmp_path), g)
    p = f"b{b}_"
    seq = rd.sequence
    for name, key in (("is_key", "is_key"), ("key_index", "key_index"), ("timestamps", "timestamps"), ("local_indexes", "local_indexes"),
                      ("global_indexes", "global_indexes"), ("s2l_r", "s2l_r"), ("s2l_t", "s2l_t"), ("l2e_q", "key_l2e_q"),
                      ("l2e_t", "key_l2e_t"), ("e2g_q", "key_e2g_q"), ("e2g_t", "key_e2g_t")):
        assert np.array_equal(np.asarray(getattr(seq, name)), g[p + key]), name
    assert list(seq.scene_tokens) == g[p + "scene_tokens"].tolist()
    assert len(rd) == len(g[p + "global_indexes"])
    # devkit look-ups: LIDAR_TOP key-frame record only (not the camera's, not a sweep's), scene tokens, lidarseg files
    g0 = int(g[p + "global_indexes"][index])
    assert rd.tables.lidar_of[f"sample{g0:03d}"] == f"sd{g0:03d}"
    assert rd.tables.scene_of[f"sample{g0:03d}"] == g[p + "scene_tokens"][g0]
    assert rd.tables.lidarseg_of[f"sd{g0:03d}"].endswith(f"sd{g0:03d}_lidarseg.bin")
    assert rd.has_history(index)
    first_of_scene_b = next(i for i, gi in enumerate(g[p + "global_indexes"]) if g[p + "scene_tokens"][gi] == g[p + "scene_tokens"][g0])
    assert not rd.has_history(first_of_scene_b)          # its predecessor in the list belongs to the other scene
    assert rd.sample_list(index, int(g["multiscan"]), float(g["step"])) == g[p + "sample_list"].tolist()
    s = rd.sample(index, int(g["multiscan"]), float(g["step"]), device="cpu")
    assert np.array_equal(s["points"].numpy(), g[p + "points_cur"])
    assert np.array_equal(s["labels"].numpy(), g["learning_map"][g[p + "rawlabels_cur"]])
    for d, pts, lab, ps in zip(s["offsets"], s["hist_points"], s["hist_labels"], s["hist_pseudo"]):
        assert np.array_equal(pts.numpy(), g[f"{p}points_d{-d}"])
        assert np.array_equal(ps.numpy(), g[f"{p}pseudo_d{-d}"].astype(np.int64))
        want = g["learning_map"][g[f"{p}rawlabels_d{-d}"]] if f"{p}rawlabels_d{-d}" in g else np.zeros(len(pts), np.int64)
        assert np.array_equal(lab.numpy(), want)
    assert tuple(s["params"].shape) == (len(s["offsets"]), 28)


def test_reader_refuses_files_that_do_not_belong_together(tmp_path, g_multiscan_nus):
    _write_tree(str(tmp_path), g_multiscan_nus, 0)
    with open(os.path.join(str(tmp_path), "nuscenes_infos_val_sweep.pkl"), "rb") as fh:
        data = pickle.load(fh)
    data["scene_tokens"] = data["scene_tokens"][:-1]
    with open(os.path.join(str(tmp_path), "nuscenes_infos_val_sweep.pkl"), "wb") as fh:
        pickle.dump(data, fh)
    with pytest.raises(ValueError):
        _reader(str(tmp_path), g_multiscan_nus)
    with pytest.raises(FileNotFoundError):
        NuscTables.from_json(str(tmp_path), "v1.0-test")


@pytest.mark.gpu
@pytest.mark.parametrize("b", [0, 1])
def test_files_to_device_stage_matches_reference_golden(tmp_path, g_multiscan_nus, b):
    """info files -> reader -> ts_fuse_sweeps + class-step mask: the reference's xyzret_ms / labels_ms bit for bit"""
    g = g_multiscan_nus
    index = _write_tree(str(tmp_path), g, b)
    rd = _reader(str(tmp_path), g)
    s = rd.sample(index, int(g["multiscan"]), float(g["step"]), device="cuda")
    raw, lab, keep = N.fuse_sweeps(s["points"], s["labels"], s["hist_points"], s["hist_labels"], s["hist_pseudo"], s["params"],
                                   g["steps"].tolist())
    assert np.array_equal(raw[keep].cpu().numpy(), g[f"b{b}_xyzret_ms"])
    assert np.array_equal(lab[keep].cpu().numpy(), g[f"b{b}_labels_ms"])
    batch = N.build_nuscenes_batch([s], 0.1, g["steps"].tolist())
    assert int(batch["num_points_ms"].view(-1)[0]) <= len(g[f"b{b}_xyzret_ms"]) and batch["lidar_ms"].C.shape[1] == 4
