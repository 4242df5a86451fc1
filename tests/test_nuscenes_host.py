"""nuScenes FSA stage, the parts that need no GPU: the oracle's restatement of nuscenes_ms.py:226-373 against the golden
vectors the REAL reference produced (tests/golden/multiscan_nus.npz, make_golden_r2.py --nus), and the product's host
logic (sweep selection, relative transforms) against the oracle."""
import numpy as np

from conftest import nus_sample
from oracle import ts_oracle as O
from taseg_amd.data import nuscenes as N


def test_quaternion_rotation_matrix_is_a_rotation():
    rs = np.random.RandomState(0)
    for _ in range(20):
        q = rs.randn(4)
        r = O.quaternion_rotation_matrix(q)
        assert np.allclose(r @ r.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(r) - 1) < 1e-12
        assert np.array_equal(r, N.rotation_matrix(q))
    # known answers: identity, 90 degrees about z
    assert np.array_equal(O.quaternion_rotation_matrix([1, 0, 0, 0]), np.eye(3))
    rz = O.quaternion_rotation_matrix([np.sqrt(0.5), 0, 0, np.sqrt(0.5)])
    assert np.allclose(rz, [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-15)


def test_oracle_nuscenes_fuse_matches_reference_golden(g_multiscan_nus):
    g = g_multiscan_nus
    for b in range(2):
        oseq, _, index, pts, pseudo, labels = nus_sample(g, b)
        offsets = O.nus_select_sweeps(oseq, index, int(g["multiscan"]), float(g["step"]))
        assert offsets == g[f"b{b}_sample_list"].tolist()
        raw, ann, pse, mask = O.nus_multiscan_fuse(oseq, index, offsets, pts, pseudo, labels, g["steps"].tolist())
        assert np.array_equal(raw, g[f"b{b}_fused_all"])                 # float32 bit for bit
        assert np.array_equal(ann, g[f"b{b}_labels_all"]) and np.array_equal(pse, g[f"b{b}_pseudo_all"])
        assert np.array_equal(mask, g[f"b{b}_mask"])
        cur = g[f"b{b}_points_cur"].copy()
        cur[:, 4] = 0                                                     # nuscenes_ms.py:109
        assert np.array_equal(np.concatenate([cur, raw[mask]]), g[f"b{b}_xyzret_ms"])
        assert np.array_equal(g["learning_map"][g[f"b{b}_rawlabels_cur"]], g[f"b{b}_labels"])
        # the ego-box filter removed something and the class-step rule dropped classes with step 0 / kept step-1 ones
        assert len(raw) < sum(len(pts[d]) for d in offsets) and 0 < mask.sum() < len(mask)


def test_host_logic_matches_oracle(g_multiscan_nus):
    g = g_multiscan_nus
    for b in range(2):
        oseq, seq, index, pts, pseudo, labels = nus_sample(g, b)
        for multiscan, step in ((int(g["multiscan"]), float(g["step"])), (2, 1.5), (6, 0.5), (15, 1.0)):
            assert N.select_sweeps(seq, index, multiscan, step) == O.nus_select_sweeps(oseq, index, multiscan, step)
        offsets = N.select_sweeps(seq, index, int(g["multiscan"]), float(g["step"]))
        params = N.sweep_params(seq, index, offsets)
        assert params.shape == (len(offsets), 28)
        probe = np.concatenate([np.zeros((1, 3)), np.eye(3)]).astype(np.float32)
        g0 = int(seq.global_indexes[index])
        for row, d in zip(params, offsets):
            father = int(seq.key_index[g0 + d]) if seq.is_key[g0 + d] else int(seq.local_indexes[g0 + d])
            assert row[12] == (0.0 if seq.is_key[g0 + d] else 1.0) and row[25] == (1.0 if father != index else 0.0)
            if row[25]:
                want = O.nus_transform_point(np.concatenate([probe, np.zeros((4, 2), np.float32)], 1).astype(np.float64),
                                             oseq["keys"][index], oseq["keys"][father])[:, :3]
                rot, trans = row[13:22].reshape(3, 3), row[22:25]
                assert np.allclose(probe @ rot + trans, want, rtol=0, atol=1e-12)
            assert row[26] == seq.timestamps[g0] / 1e6 - seq.timestamps[g0 + d] / 1e6
