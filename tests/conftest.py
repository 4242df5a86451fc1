import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. `pytest tests` in the build container."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no ROCm device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def _load(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def g_ops():
    return _load("ops.npz")


@pytest.fixture(scope="session")
def g_minkunet():
    return _load("model_minkunet.npz")


@pytest.fixture(scope="session")
def g_miou():
    return _load("miou_minkunet.npz")


@pytest.fixture(scope="session")
def g_minkunet_ms():
    return _load("model_minkunet_ms.npz")


@pytest.fixture(scope="session")
def g_minkunet_ms_mm():
    return _load("model_minkunet_ms_mm.npz")


@pytest.fixture(scope="session")
def g_minkunet_ms_kd():
    return _load("model_minkunet_ms_kd.npz")


@pytest.fixture(scope="session")
def g_multiscan():
    return _load("multiscan.npz")


@pytest.fixture(scope="session")
def g_multiscan_nus():
    return _load("multiscan_nus.npz")


@pytest.fixture(scope="session")
def g_eval_ms():
    return _load("eval_ms.npz")


def nus_sample(g, b):
    """Sample `b` of tests/golden/multiscan_nus.npz as (oracle sequence dict, taseg_amd NuscSequence, keyframe index,
    per-offset dicts of points / pseudo labels / mapped labels)."""
    from taseg_amd.data.nuscenes import NuscSequence
    p = f"b{b}_"
    keys = [dict(lidar2ego_rotation=g[p + "key_l2e_q"][i], lidar2ego_translation=g[p + "key_l2e_t"][i],
                 ego2global_rotation=g[p + "key_e2g_q"][i], ego2global_translation=g[p + "key_e2g_t"][i])
            for i in range(len(g[p + "key_l2e_q"]))]
    oseq = dict(is_key=g[p + "is_key"], key_index=g[p + "key_index"], scene_tokens=g[p + "scene_tokens"],
                local_indexes=g[p + "local_indexes"], global_indexes=g[p + "global_indexes"], s2l_r=g[p + "s2l_r"],
                s2l_t=g[p + "s2l_t"], timestamps=g[p + "timestamps"], keys=keys)
    seq = NuscSequence(is_key=g[p + "is_key"], key_index=g[p + "key_index"], timestamps=g[p + "timestamps"],
                       scene_tokens=g[p + "scene_tokens"].tolist(), local_indexes=g[p + "local_indexes"], s2l_r=g[p + "s2l_r"],
                       s2l_t=g[p + "s2l_t"], global_indexes=g[p + "global_indexes"], l2e_q=g[p + "key_l2e_q"],
                       l2e_t=g[p + "key_l2e_t"], e2g_q=g[p + "key_e2g_q"], e2g_t=g[p + "key_e2g_t"])
    offsets = g[p + "sample_list"].tolist()
    lm = g["learning_map"]
    pts = {d: g[f"{p}points_d{-d}"] for d in offsets}
    pseudo = {d: g[f"{p}pseudo_d{-d}"] for d in offsets}
    labels = {d: (lm[g[f"{p}rawlabels_d{-d}"]] if f"{p}rawlabels_d{-d}" in g else np.zeros(len(pts[d]), dtype=np.int64))
              for d in offsets}
    return oseq, seq, int(g[p + "index"]), pts, pseudo, labels


def dense_ops_inputs(n):
    """Seeded inputs of tests/golden/ops_dense.npz (the fixture stores only the reference's outputs): submanifold k3
    convolutions 32 -> 64 ("full") and 4 -> 20 ("ragged"), strided k2 32 -> 32 + transposed mirror ("t"), on `n` voxels."""
    import torch
    g = torch.Generator().manual_seed(9)
    out = {}
    for tag, (ci, co) in (("full", (32, 64)), ("ragged", (4, 20))):
        out[f"{tag}_x"] = torch.randn(n, ci, generator=g)
        out[f"{tag}_w"] = torch.randn(27, ci, co, generator=g) * 0.2
        out[f"{tag}_gy"] = torch.randn(n, co, generator=g)
    out["t_x"] = torch.randn(n, 32, generator=g)
    out["t_wd"] = torch.randn(8, 32, 32, generator=g) * 0.2
    out["t_wu"] = torch.randn(8, 32, 32, generator=g) * 0.2
    out["t_gy"] = torch.randn(n, 32, generator=g)
    return out
