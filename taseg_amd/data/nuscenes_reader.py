"""nuScenes info reader for the device data stage (the nuScenes half): the file side of
R/pcseg/data/dataset/nuscenes/nuscenes_ms.py:19-100 (construction: info pickles, sweep tables, devkit look-ups) and
:103-131, 278-330 (which files a sample reads) - feeding `taseg_amd.data.nuscenes`, which selects the sweeps on the host
and runs ego-box filter, transforms, class-step mask and voxelisation on the GPU.

Inputs, as the reference's dataset class finds them on disk:

    <root>/nuscenes_infos_<split>.pkl          {"infos": [ {lidar_path, token, timestamp, lidar2ego_rotation / _translation,
                                                ego2global_rotation / _translation, ...} per keyframe ]}        (:56-66)
    <root>/nuscenes_infos_<split>_sweep.pkl    {"infos_sweep": [ keyframe info | {data_path, sample_data_token, timestamp,
                                                sensor2lidar_rotation / _translation} per frame in time order ],
                                                "global_indexes", "local_indexes", "scene_tokens"}               (:72-78)
    <root>/<version>/{sample, sample_data, calibrated_sensor, sensor, lidarseg}.json     the devkit's tables; the
                                                reference asks them three questions only, which `NuscTables` answers from
                                                the JSON files directly (no nuscenes-devkit import):
                                                  nusc.get('sample', token)['scene_token']                       (:119)
                                                  nusc.get('sample', token)['data']['LIDAR_TOP']                  (:113)
                                                  nusc.get('lidarseg', sd_token)['filename']                      (:115)
    <root>/samples|sweeps/LIDAR_TOP/*.bin      float32 [n, 5] x, y, z, intensity, ring                           (:106, 299)
    <root>/lidarseg/...                        uint8 [n] raw class ids -> learning_map                           (:116-117)
    <pseudo_dir>/<sample_data token>_lidarseg.bin   uint8 [n] pseudo classes of every frame                      (:320-321)

    reader = NuscInfoReader(root, split="train", learning_map=LEARNING_MAP, pseudo_dir=...)
    sample = reader.sample(index, multiscan=15, step=1.0, device="cuda")
    batch = build_nuscenes_batch([sample], 0.1, FLEXIBLE_STEPS_NUSC)

Host work per sample = reading ~25 files and a few dozen 3x3 products (cached per keyframe like the reference's
`token2samplelist`); everything per point runs on the device.  Augmentation (LaserMix / PolarMix, :132-213) and the Ceph
client are outside the scope contract.
"""
import json
import os
import pickle
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .nuscenes import NuscSequence, select_sweeps, sweep_params

__all__ = ["NuscTables", "NuscInfoReader", "sequence_from_infos"]

_PREFIX = 16        # the info files hold './data/nuscenes/...' paths; the reference cuts the first 16 characters (:105)


class NuscTables:
    """The three devkit look-ups of nuscenes_ms.py answered from the JSON tables: sample token -> scene token, sample token
    -> token of its LIDAR_TOP keyframe sample_data (the devkit builds sample['data'][channel] from the key-frame
    sample_data records, channel = sensor of the record's calibrated sensor), sample_data token -> lidarseg file."""

    def __init__(self, scene_of: Dict[str, str], lidar_of: Dict[str, str], lidarseg_of: Dict[str, str], dataroot: str):
        self.scene_of, self.lidar_of, self.lidarseg_of, self.dataroot = scene_of, lidar_of, lidarseg_of, dataroot

    @classmethod
    def from_json(cls, dataroot: str, version: str = "v1.0-trainval"):
        def table(name, required=True):
            path = os.path.join(dataroot, version, name + ".json")
            if not os.path.exists(path):
                if required:
                    raise FileNotFoundError(path)
                return []
            with open(path) as f:
                return json.load(f)

        channel_of_sensor = {r["token"]: r["channel"] for r in table("sensor")}
        sensor_of_calib = {r["token"]: r["sensor_token"] for r in table("calibrated_sensor")}
        scene_of = {r["token"]: r["scene_token"] for r in table("sample")}
        lidar_of = {}
        for r in table("sample_data"):
            if r.get("is_key_frame") and channel_of_sensor.get(sensor_of_calib.get(r["calibrated_sensor_token"])) == "LIDAR_TOP":
                lidar_of[r["sample_token"]] = r["token"]
        lidarseg_of = {r["sample_data_token"]: r["filename"] for r in table("lidarseg", required=False)}   # absent on the test split
        return cls(scene_of, lidar_of, lidarseg_of, dataroot)


def sequence_from_infos(infos: Sequence[dict], sweeps: Sequence[dict], global_indexes, local_indexes,
                        scene_tokens) -> NuscSequence:
    """The arrays `taseg_amd.data.nuscenes` works on, read out of the reference's info dicts (nuscenes_ms.py:226-276,
    348-360 say which fields matter).  A frame of `sweeps` is a keyframe iff it carries 'lidar_path' (:271, :281)."""
    key_of_token = {info["token"]: i for i, info in enumerate(infos)}
    n = len(sweeps)
    is_key = np.array(["lidar_path" in s for s in sweeps], dtype=bool)
    key_index = np.array([key_of_token[s["token"]] if k else -1 for s, k in zip(sweeps, is_key)], dtype=np.int64)
    s2l_r, s2l_t = np.zeros((n, 3, 3)), np.zeros((n, 3))
    for g, s in enumerate(sweeps):
        if "data_path" in s:
            s2l_r[g], s2l_t[g] = np.asarray(s["sensor2lidar_rotation"], dtype=np.float64), s["sensor2lidar_translation"]
    return NuscSequence(
        is_key=is_key, key_index=key_index, timestamps=np.array([s["timestamp"] for s in sweeps], dtype=np.int64),
        scene_tokens=list(scene_tokens), local_indexes=np.asarray(local_indexes, dtype=np.int64), s2l_r=s2l_r, s2l_t=s2l_t,
        global_indexes=np.asarray(global_indexes, dtype=np.int64),
        l2e_q=np.array([i["lidar2ego_rotation"] for i in infos], dtype=np.float64),
        l2e_t=np.array([i["lidar2ego_translation"] for i in infos], dtype=np.float64),
        e2g_q=np.array([i["ego2global_rotation"] for i in infos], dtype=np.float64),
        e2g_t=np.array([i["ego2global_translation"] for i in infos], dtype=np.float64))


class NuscInfoReader:
    """NuscenesMsDataset's construction (nuscenes_ms.py:19-100) and per-sample file reads (:103-131, :278-330) without the
    devkit object: info pickles + JSON tables -> resident tensors for `build_nuscenes_batch`."""

    def __init__(self, root_path: str, split: str = "train", learning_map: Optional[Dict[int, int]] = None,
                 pseudo_dir: Optional[str] = None, version: str = "v1.0-trainval", tables: Optional[NuscTables] = None,
                 info_path: Optional[str] = None, sweep_info_path: Optional[str] = None):
        self.root_path, self.split, self.pseudo_dir = root_path, split, pseudo_dir
        with open(info_path or os.path.join(root_path, f"nuscenes_infos_{split}.pkl"), "rb") as f:
            self.infos = pickle.load(f)["infos"]
        with open(sweep_info_path or os.path.join(root_path, f"nuscenes_infos_{split}_sweep.pkl"), "rb") as f:
            data = pickle.load(f)
        self.sweeps = data["infos_sweep"]
        self.global_indexes, self.local_indexes, self.scene_tokens = data["global_indexes"], data["local_indexes"], data["scene_tokens"]
        if not (len(self.global_indexes) == len(self.infos) and len(self.local_indexes) == len(self.sweeps) == len(self.scene_tokens)):
            raise ValueError("nuScenes info files do not belong together: %d keyframes / %d global indexes, %d frames / %d local "
                             "indexes / %d scene tokens" % (len(self.infos), len(self.global_indexes), len(self.sweeps),
                                                           len(self.local_indexes), len(self.scene_tokens)))
        self.tables = tables or NuscTables.from_json(root_path, version)
        self.sequence = sequence_from_infos(self.infos, self.sweeps, self.global_indexes, self.local_indexes, self.scene_tokens)
        lut = np.zeros(256, dtype=np.int64)
        for k, v in (learning_map or {}).items():
            lut[int(k)] = int(v)
        self.lut = lut
        self._lists = {}            # keyframe -> selected frame offsets (the reference's token2samplelist, :233-276)

    def __len__(self):
        return len(self.infos)

    # ---- files ----------------------------------------------------------------------------------------------------
    def _points(self, path: str) -> np.ndarray:
        return np.fromfile(os.path.join(self.root_path, path[_PREFIX:]), dtype=np.float32, count=-1).reshape([-1, 5])

    def _labels(self, sd_token: str, n: int) -> np.ndarray:
        """mapped annotation of a keyframe; zeros where the split has none (:121-122, :300-303 `except` branch)"""
        name = self.tables.lidarseg_of.get(sd_token)
        path = None if name is None else os.path.join(self.tables.dataroot, name)
        if path is None or not os.path.exists(path):
            return np.zeros(n, dtype=np.int64)
        return self.lut[np.fromfile(path, dtype=np.uint8)]

    def _pseudo(self, sd_token: str) -> np.ndarray:
        if self.pseudo_dir is None:
            raise ValueError("the multi-scan stage filters history points by pseudo labels (PSEUDO_MASK, nuscenes_ms.py:"
                             "317-321): pass pseudo_dir")
        return np.fromfile(os.path.join(self.pseudo_dir, sd_token + "_lidarseg.bin"), dtype=np.uint8).astype(np.int64)

    # ---- samples --------------------------------------------------------------------------------------------------
    def has_history(self, index: int) -> bool:
        """the reference aggregates only when the PREVIOUS keyframe of the list lies in the same scene (:119-120; index 0
        compares with the last keyframe, as `self.nusc_infos[index-1]` does)"""
        return self.tables.scene_of[self.infos[index - 1]["token"]] == self.tables.scene_of[self.infos[index]["token"]]

    def sample_list(self, index: int, multiscan: int, step: float) -> List[int]:
        key = (index, multiscan, step)
        if key not in self._lists:
            self._lists[key] = select_sweeps(self.sequence, index, multiscan, step)
        return self._lists[key]

    def sample(self, index: int, multiscan: int, step: float, device="cuda") -> Dict:
        """Resident inputs of keyframe `index` for `taseg_amd.data.nuscenes.build_nuscenes_batch`: the current cloud and
        its labels, the selected history frames oldest first (raw points, mapped labels - zeros for sweeps, :318 -, pseudo
        classes) and their transform parameters."""
        info = self.infos[index]
        cur = self._points(info["lidar_path"])
        sd = self.tables.lidar_of[info["token"]]
        cur_lab = self._labels(sd, len(cur))
        if len(cur_lab) != len(cur):
            raise ValueError(f"{info['lidar_path']}: {len(cur)} points but {len(cur_lab)} labels")
        offsets = self.sample_list(index, multiscan, step) if self.has_history(index) else []
        g0 = int(self.global_indexes[index])
        hp, hl, hs = [], [], []
        for d in offsets:
            fr = self.sweeps[g0 + d]
            if "lidar_path" in fr:
                pts = self._points(fr["lidar_path"])
                tok = self.tables.lidar_of[fr["token"]]
                lab = self._labels(tok, len(pts))
            else:
                pts = self._points(fr["data_path"])
                tok = fr["sample_data_token"]
                lab = np.zeros(len(pts), dtype=np.int64)
            ps = self._pseudo(tok)
            if not (len(ps) == len(pts) == len(lab)):
                raise ValueError(f"frame {g0 + d}: {len(pts)} points, {len(lab)} labels, {len(ps)} pseudo labels")
            hp.append(torch.from_numpy(pts).to(device))
            hl.append(torch.from_numpy(lab).to(device))
            hs.append(torch.from_numpy(ps).to(device))
        params = torch.from_numpy(sweep_params(self.sequence, index, offsets)).to(device)
        return dict(points=torch.from_numpy(cur).to(device), labels=torch.from_numpy(cur_lab).to(device), hist_points=hp,
                    hist_labels=hl, hist_pseudo=hs, params=params, name=info["lidar_path"], offsets=list(offsets))
