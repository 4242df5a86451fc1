"""Host-side logic that needs no GPU: containers, kernel offsets, numpy quantize / collate against the
oracle, state_dict layout against the reference's, the vectorised Lovasz loss against the per-class loop."""
import os

import numpy as np
import pytest
import torch

from oracle import model as OM
from oracle import ts_oracle as O
from taseg_amd.data.synthetic import AttrDict, fill_parameters, make_model_cfg, synth_pose, synth_scan
from taseg_amd.pcseg.loss import Losses, lovasz_softmax
from taseg_amd.pcseg.model import build_network
from taseg_amd.torchsparse import PointTensor, SparseTensor, cat
from taseg_amd.torchsparse.nn.utils import get_kernel_offsets
from taseg_amd.torchsparse.utils import make_ntuple, sparse_collate_fn, sparse_quantize


def test_kernel_offsets_match_oracle():
    for size, stride in ((3, 1), (3, 4), (2, 1), (2, 8), ((3, 1, 3), 2)):
        assert np.array_equal(get_kernel_offsets(size, stride).numpy(), O.get_kernel_offsets(size, stride))
    k3 = get_kernel_offsets(3, 1).tolist()
    assert k3[13] == [0, 0, 0] and k3[0] == [-1, -1, -1] and k3[1] == [0, -1, -1]   # x innermost, centre = 13


def test_sparse_quantize_numpy_matches_oracle():
    pts, _ = synth_scan(3, n_points=5000, n_beams=16, n_az=500)
    pc = np.round(pts[:, :3] / 0.05).astype(np.int32)
    pc -= pc.min(0)
    coords, idx, inv = sparse_quantize(pc, return_index=True, return_inverse=True)
    widx, winv = O.sparse_quantize(pc)
    assert np.array_equal(idx, widx) and np.array_equal(inv, winv)
    assert np.array_equal(coords, pc[widx])
    assert np.array_equal(coords[inv], pc)                       # inverse map reconstructs every point's voxel
    assert np.all(np.diff(O.ravel_hash(coords).astype(np.int64)) > 0)   # strictly ascending ravel key


def test_collate_appends_batch_column():
    a = {"lidar": SparseTensor(np.ones((3, 4), np.float32), np.zeros((3, 3), np.int32)), "name": "a",
         "num_points": np.array([3])}
    b = {"lidar": SparseTensor(np.ones((2, 4), np.float32), np.ones((2, 3), np.int32)), "name": "b",
         "num_points": np.array([2])}
    out = sparse_collate_fn([a, b])
    assert out["lidar"].C.shape == (5, 4) and out["lidar"].C[:, 3].tolist() == [0, 0, 0, 1, 1]
    assert out["name"] == ["a", "b"] and out["num_points"].shape == (2, 1)


def test_containers_share_caches():
    x = SparseTensor(torch.ones(4, 2), torch.zeros(4, 4, dtype=torch.int32), 2)
    assert x.s == (2, 2, 2) and x.F is x.feats and x.C is x.coords
    y = x + x
    z = cat([x, y])
    assert y.kmaps is x.kmaps and z.cmaps is x.cmaps and z.F.shape == (4, 4) and float(y.F.sum()) == 16
    x.F = x.F[:, :1]
    assert x.feats.shape == (4, 1)
    p = PointTensor(torch.ones(4, 2), torch.zeros(4, 4))
    q = p + p
    assert q.idx_query is p.idx_query and q.additional_features is p.additional_features
    assert make_ntuple(3, 3) == (3, 3, 3) and make_ntuple([1, 2, 3], 3) == (1, 2, 3)


def test_state_dict_layout_matches_reference(g_minkunet, g_minkunet_ms):
    for name, in_dim, g in (("MinkUNet", 4, g_minkunet), ("MinkUNetMs", 5, g_minkunet_ms)):
        cfg = make_model_cfg(name, in_dim=in_dim, cr=0.5, num_layer=[1] * 8)
        sd = build_network(cfg, 20).state_dict()
        assert list(sd.keys()) == g["state_keys"].tolist()
        assert [",".join(map(str, v.shape)) for v in sd.values()] == g["state_shapes"].tolist()


def test_mk34_parameter_count():
    m = build_network(make_model_cfg("MinkUNet", in_dim=4, cr=1.0), 20)
    assert sum(p.numel() for p in m.parameters()) == 37882900 and len(m.state_dict()) == 380
    m5 = build_network(make_model_cfg("MinkUNetMs", in_dim=5, cr=1.0), 20)
    assert sum(p.numel() for p in m5.parameters()) == 37883764


def test_fill_parameters_is_order_independent(g_minkunet):
    import zlib
    cfg = make_model_cfg("MinkUNet", in_dim=4, cr=0.5, num_layer=[1] * 8)
    sd = fill_parameters(build_network(cfg, 20), seed=3).state_dict()
    crc = np.array([zlib.crc32(v.numpy().tobytes()) for v in sd.values()], dtype=np.int64)
    assert np.array_equal(crc, g_minkunet["param_crc"])       # same parameters as the reference model got


def test_registry_errors():
    with pytest.raises(NotImplementedError):
        build_network(AttrDict(NAME="SPVCNN"), 20)
    with pytest.raises(NameError):
        build_network(AttrDict(NAME="Nope"), 20)
    with pytest.raises(KeyError):
        build_network(make_model_cfg("MinkUNet", BLOCK="Wrong"), 20)


def test_lovasz_vectorised_equals_per_class_loop():
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(4000, 20, generator=g, requires_grad=True)
    labels = torch.randint(0, 12, (4000,), generator=g)          # classes 12..19 absent, 0 ignored
    a = lovasz_softmax(logits.softmax(1), labels, ignore=0)
    b = OM.lovasz_softmax_ref(logits.softmax(1), labels, ignore=0)
    assert abs(float(a) - float(b)) < 1e-6
    ga, = torch.autograd.grad(a, logits, retain_graph=True)
    gb, = torch.autograd.grad(b, logits)
    assert float((ga - gb).abs().max()) < 1e-7
    crit = Losses(["CELoss", "LovLoss"], [1.0, 1.0], ignore_index=0, label_smoothing=0.1)
    want = OM.loss_ce_lovasz(logits, labels)
    assert abs(float(crit(logits, labels)) - float(want)) < 1e-6
    all_ignored = lovasz_softmax(logits.softmax(1), torch.zeros(4000, dtype=torch.long), ignore=0)
    assert float(all_ignored.sum()) == 0.0
    with pytest.raises(NotImplementedError):
        Losses(["FocalLoss"], [1.0])


def test_synthetic_scan_is_deterministic_and_shaped():
    a, la = synth_scan(1000, n_points=120000)
    b, lb = synth_scan(1000, n_points=120000)
    assert a.shape == (120000, 4) and a.dtype == np.float32 and np.array_equal(a, b) and np.array_equal(la, lb)
    assert set(np.unique(la)) <= set(range(20)) and len(np.unique(la)) >= 6
    h, _ = synth_scan(1001, n_points=4000, n_beams=16, n_az=400, pose=synth_pose(2), scene_seed=1000)
    assert h.shape[1] == 4 and synth_pose(0).tolist() == np.eye(4).tolist()


def test_reducer_refuses_a_second_backward():
    """GradBucketReducer's contract is one backward per finish(): a second one would add into buffers whose
    all-reduce is already in flight - it must raise, not corrupt."""
    from taseg_amd.parallel import GradBucketReducer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.ReLU(), torch.nn.Linear(8, 2))
    red = GradBucketReducer(net)
    x = torch.randn(5, 4)
    net(x).sum().backward()
    with pytest.raises(RuntimeError, match="one backward"):
        net(x).sum().backward()
    red.finish()
    for p in net.parameters():
        p.grad = None
    net(x).sum().backward()          # re-armed after finish()
    red.finish()
    # a parameter that got no gradient is reported, its bucket slice is zero
    net2 = torch.nn.ModuleDict({"a": torch.nn.Linear(4, 4), "b": torch.nn.Linear(4, 4)})
    red2 = GradBucketReducer(net2)
    net2["a"](x).sum().backward()
    red2.finish()
    unused = [red2.buckets[0]["params"][i] for i in red2.buckets[0]["unused"]]
    assert {id(p) for p in unused} == {id(p) for p in net2["b"].parameters()}
    assert all(float(p.grad.abs().sum()) == 0 for p in net2["b"].parameters())


def test_install_as_dropin_binds_reference_import_names():
    """INTEGRATION.md level 1: unmodified OpenPCSeg imports resolve to this package (R/train.py:195, model/__init__.py)."""
    import subprocess
    import sys
    import os
    code = (
        "import taseg_amd; names = taseg_amd.install_as_dropin()\n"
        "import torchsparse, torchsparse.nn as spnn, torchsparse.nn.functional as F\n"
        "from torchsparse import SparseTensor, PointTensor\n"
        "from torchsparse.utils.quantize import sparse_quantize\n"
        "from torchsparse.utils.collate import sparse_collate_fn\n"
        "from torchsparse.nn.utils import get_kernel_offsets\n"
        "from pcseg.model import build_network, load_data_to_gpu\n"
        "from pcseg.loss import Losses\n"
        "import torchsparse.backend as B\n"
        "assert all(hasattr(B, n) for n in ('hash_cuda', 'kernel_hash_cuda', 'hash_query_cuda', 'count_cuda',"
        " 'voxelize_forward_cuda', 'voxelize_backward_cuda', 'devoxelize_forward_cuda', 'devoxelize_backward_cuda',"
        " 'convolution_forward_cuda', 'convolution_backward_cuda'))\n"
        "assert torchsparse is taseg_amd.torchsparse and spnn.Conv3d is taseg_amd.torchsparse.nn.Conv3d\n"
        "from taseg_amd.data.synthetic import make_model_cfg\n"
        "m = build_network(make_model_cfg('MinkUNetMs', in_dim=5, cr=0.5, num_layer=[1] * 8), 20)\n"
        "assert type(m).__name__ == 'MinkUNetMs' and hasattr(m, 'load_params_from_file')\n"
        "print('DROPIN_OK')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert "DROPIN_OK" in out.stdout, out.stderr[-2000:]


def test_eval_helpers(tmp_path, g_eval_ms):
    """taseg_amd/pcseg/eval.py against the reference's outputs stored in eval_ms.npz (vote sum / payload, R/train.py:474-503)
    and against direct restatements of fast_hist_crop / per_class_iu / tta_remap's lookup"""
    from taseg_amd.pcseg import eval as E
    g = g_eval_ms
    votes = int(g["votes"])
    ret = {"point_predict_logits": [g[f"minkunet_ms_tta_point_predict_logits_{v}"] for v in range(votes)],
           "point_predict": [g[f"minkunet_ms_tta_point_predict_{v}"] for v in range(votes)],
           "point_labels": [g[f"minkunet_ms_tta_point_labels_{v}"] for v in range(votes)], "name": g["minkunet_ms_tta_name"].tolist()}
    acc = E.accumulate_votes(ret, votes)
    assert np.array_equal(acc, g["minkunet_ms_tta_sum"])                        # same fp32 additions in the same order
    payload = E.vote_payload(acc, "semantickitti")
    assert np.array_equal(payload, g["minkunet_ms_tta_label"]) and payload.dtype == np.uint32
    with pytest.raises(ValueError):
        E.accumulate_votes(ret, votes + 1)
    path = E.write_prediction(str(tmp_path / "sequences" / "08" / "predictions" / "000000.label"), payload)
    back = np.fromfile(path, dtype=np.uint32)
    assert np.array_equal(back, payload.reshape(-1))
    # tta_remap.py:103-154: class ids -> raw SemanticKITTI ids through learning_map_inv, instance bits kept
    from taseg_amd.data.semantickitti import LEARNING_MAP_INV
    lut = E.remap_lut(LEARNING_MAP_INV)
    withinst = back | (np.arange(len(back), dtype=np.uint32) << 16)
    out = E.remap_labels(withinst, lut)
    assert np.array_equal(out & 0xFFFF, np.array([LEARNING_MAP_INV[int(c)] for c in back], dtype=np.uint32))
    assert np.array_equal(out >> 16, np.arange(len(back), dtype=np.uint32) & 0xFFFF)
    # validation metric: R/train.py:35-52
    pred, lab = g["minkunet_ms_point_predict_0"], g["minkunet_ms_point_labels_0"]
    uniq = np.arange(19)
    h = E.fast_hist_crop(pred, lab, uniq)
    full = O.fast_hist(pred, lab, 21)
    assert np.array_equal(h, full[1:20, 1:20]) and h.shape == (19, 19)
    assert np.allclose(E.per_class_iu(h), O.per_class_iu(h))
    with pytest.raises(ValueError):
        E.vote_payload(np.eye(4)[[0, 1]], "nuscenes")                          # class 0 in a nuScenes submission


def test_bench_roofline_groups_by_family_and_prices_the_weight_gradient_without_a_scatter_term():
    """bench.py's `roofline`: instantiations grouped by kernel family, the family with the largest time share reported
    with BOTH fractions; weight-gradient bytes = P (Cin + Cout) s + 8 P + K Cin Cout 4 (no read-modify-write term)."""
    import bench

    class Ms:
        def __init__(self, ms):
            self.ms = ms

        def elapsed_time(self, _):
            return self.ms

    p, k = 1_000_000, 27
    recs = []
    for name, ms in (("pair_gemm_s_kernel<128,96,2,false,true>", 0.20), ("pair_gemm_d_kernel<128,true>", 0.05)):
        recs.append(("pair_gemm", Ms(ms), None, dict(name=name, pairs=p, c_red=96, c_out=96, k=k, esize=4, n_rows=150000)))
    recs.append(("gather_sum", Ms(0.10), None, dict(name="gather_list_kernel<8>", pairs=p, c_red=0, c_out=96, k=k, n_rows=150000,
                                                    esize=4, side_bytes=0.0)))
    recs.append(("conv_wgrad", Ms(0.12), None, dict(name="wgrad_s_kernel<96,96>", pairs=p, c_red=96, c_out=96, k=k, esize=4,
                                                    n_rows=150000, n_rows_b=150000)))
    prof = bench.summarise_profile(recs, 1)
    wg = [r for r in prof if r["kernel"].startswith("wgrad")][0]
    assert wg["bytes_per_launch"] == p * (96 * 4 + 96 * 4 + 8) + k * 96 * 96 * 4
    roof = bench.build_roofline(prof, False, 4.0)
    assert roof["kernel"] == "pair_gemm" and len(roof["kernels"]) == 2 and roof["launches_per_step"] == 2
    assert abs(roof["ms_per_step"] - 0.25) < 1e-12
    flops = 2 * 2.0 * p * 96 * 96
    assert abs(roof["mfma_frac"] - flops / 0.25e-3 / 1e12 / bench.MFMA_SPLIT_PEAK_TF) < 1e-9
    byts = 2 * (p * (96 * 4 + 96 * 4 + 8) + k * 96 * 96 * 4)
    assert abs(roof["hbm_frac"] - byts / 0.25e-3 / 1e9 / bench.HBM_PEAK_GBS) < 1e-9
    assert roof["bound"] in ("mfma", "hbm") and roof["frac"] == (roof["mfma_frac"] if roof["bound"] == "mfma" else roof["hbm_frac"])
    fams = {f["family"]: f for f in roof["families"]}
    assert set(fams) == {"pair_gemm", "gather", "wgrad"} and fams["gather"]["mfma_frac"] is None
    lb = roof["whole_step_lower_bound_terms_ms"]
    assert roof["whole_step_lower_bound_ms"] == max(lb.values()) > 0


def test_bench_watchdog_ends_a_stuck_rank_with_exit_code_3():
    """bench.Watchdog: a rank without progress for TASEG_BENCH_WATCHDOG_S seconds says where it was and exits 3 (the
    launcher then ends its peers); a rank that keeps beating is left alone"""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench; wd = bench.Watchdog(5); wd.beat('all-reduce of bucket 2');\n"
            "t0 = time.time()\n"
            "while time.time() - t0 < float(sys.argv[1]): time.sleep(0.1); (wd.beat('step') if sys.argv[2] == 'beat' else None)\n"
            "wd.stop(); print('finished')" % ROOT)
    env = dict(os.environ, TASEG_BENCH_WATCHDOG_S="1.0")
    stuck = subprocess.run([sys.executable, "-c", code, "30", "stuck"], env=env, capture_output=True, text=True, timeout=120)
    assert stuck.returncode == 3 and "rank 5" in stuck.stderr and "all-reduce of bucket 2" in stuck.stderr
    alive = subprocess.run([sys.executable, "-c", code, "3", "beat"], env=env, capture_output=True, text=True, timeout=120)
    assert alive.returncode == 0 and "finished" in alive.stdout


def test_traffic_parser_recovers_names_rocprofv3_leaves_mangled():
    """profiles/parse_traffic.py: rocprofv3's demangler does not know _Float16 (DF16_), so the half kernels arrive mangled; the
    table keys must still be `name<template arguments>` - what bench.py's family roofline looks its members up by."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("parse_traffic", os.path.join(root, "profiles", "parse_traffic.py"))
    pt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pt)
    assert pt.short("_Z20gather_list_h_kernelILi8EEvPKDF16_iPKiilPDF16_13TsWgradReduceS1_i") == "gather_list_h_kernel<8>"
    assert pt.short("_Z18pair_gemm_h_kernelILi128ELi2ELb1EEvPKDF16_iS1_iPK15HIP_vector_typeIiLj2EEPKiiiPDF16_") == \
        "pair_gemm_h_kernel<128,2,true>"
    assert pt.short("_Z21devoxelize_fwd_kernelILi4EDF16_EvPKT0_PKiPKfliPS0_l") == "devoxelize_fwd_kernel<4,_Float16>"
    assert pt.short("_Z24bn_act_bwd_coef_h_kernelPKDv8_DF16_PKhS1_PKfS5_S5_S5_liPS_S6_") == "bn_act_bwd_coef_h_kernel"
    assert pt.short("void wgrad_s_kernel<96, 96>(float const*, int)") == "wgrad_s_kernel<96,96>"
    import bench
    table = {"pair_gemm_h_kernel<128,2,true>": {"hbm_bytes_per_launch": 80e6, "launches_sampled": 100},
             "pair_gemm_h_kernel<128,2,false>": {"hbm_bytes_per_launch": 60e6, "launches_sampled": 300}}
    prof = [{"kernel": "pair_gemm_h_kernel<128>", "ms_per_step": 1.0, "launches_per_step": 10.0, "flops_per_launch": 1e9,
             "bytes_per_step": 1e9, "ideal_fused_bytes_per_step": 1e8}]
    fam = bench.summarise_families(prof, True, table)[0]
    assert abs(fam["traffic_bytes_per_launch"] - 65e6) < 1.0          # launch-weighted over the two instantiations


class _NoEvent:
    """stands in for torch.cuda.Event where the tail runs on CPU tensors"""

    def record(self, *a):
        pass

    def synchronize(self):
        pass


def _tail_case(rs, n_vox, n_pts, classes=5):
    from taseg_amd.torchsparse import SparseTensor
    vox_batch = torch.from_numpy(np.concatenate([np.full(n, b) for b, n in enumerate(n_vox)])).int()
    vox_batch = vox_batch[torch.from_numpy(rs.permutation(len(vox_batch)))]          # scenes interleaved
    out = torch.from_numpy(rs.randn(len(vox_batch), classes).astype(np.float32))
    inv = np.concatenate([rs.randint(0, n_vox[b], n_pts[b]) for b in range(len(n_vox))])
    bat = np.concatenate([np.full(n_pts[b], b) for b in range(len(n_vox))])
    perm = rs.permutation(len(inv))
    inv, bat = inv[perm], bat[perm]
    coords = torch.zeros(len(inv), 4, dtype=torch.int32)
    coords[:, 3] = torch.from_numpy(bat).int()
    labels = torch.from_numpy(rs.randint(0, classes, len(inv))).long()
    return out, vox_batch, SparseTensor(torch.from_numpy(inv).long(), coords), SparseTensor(labels, coords)


def test_evaluation_tail_for_the_whole_batch_equals_the_per_scene_loop(monkeypatch):
    """unvoxelise_predictions (the eval branch's tail, minkunet.py:435-455, for all scenes at once and without host reads before the
    arrays are collected) against the reference's per-scene boolean-mask loop; scene indices outside the batch and inverse maps
    outside their scene are reported when the arrays are collected."""
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet.minkunet import unvoxelise_predictions
    from taseg_amd.torchsparse import SparseTensor
    monkeypatch.setattr(torch.cuda, "Event", _NoEvent)
    rs = np.random.RandomState(3)
    n_vox, n_pts = [50, 70, 40], [80, 90, 60]
    out, vox_batch, invs, labs = _tail_case(rs, n_vox, n_pts)
    names = ["a", "b", "c"]
    for want_probs in (False, True):
        r = unvoxelise_predictions(out, vox_batch, invs, labs, torch.tensor(n_pts), want_probs, names=names)
        for b in range(3):
            ob = out[vox_batch == b]
            sel = invs.C[:, 3] == b
            ref = ob[invs.F[sel]]
            want = ref.softmax(1) if want_probs else ref.argmax(1)
            assert np.allclose(r["point_predict"][b], want.numpy(), atol=1e-6)
            if not want_probs:
                assert np.array_equal(r["point_predict_logits"][b], ref.numpy())
            assert np.array_equal(r["point_labels"][b], labs.F[sel].numpy())
    # a scan shorter than its inverse map is trimmed like the reference trims it ([:num_points])
    r = unvoxelise_predictions(out, vox_batch, invs, labs, torch.tensor([80, 50, 60]), False, names=names)
    assert [len(a) for a in r["point_predict"]] == [80, 50, 60]
    for bad_index in (7, -1):
        c2 = invs.C.clone()
        c2[0, 3] = bad_index
        with pytest.raises(IndexError, match="beyond the 3 scenes"):
            unvoxelise_predictions(out, vox_batch, SparseTensor(invs.F, c2), labs, torch.tensor(n_pts), False, names=names)
    f2 = invs.F.clone()
    f2[3] = 1000
    with pytest.raises(IndexError, match="outside its scene"):
        unvoxelise_predictions(out, vox_batch, SparseTensor(f2, invs.C), labs, torch.tensor(n_pts), False, names=names)


def test_wgrad_stream_tuner_keeps_the_second_stream_only_on_a_clear_win(monkeypatch):
    """_fast.tune_wgrad_stream: all rounds but one must win by > 1 % and the best of rounds by > 2 %; an environment pin and a
    world size above one skip the timing."""
    import time
    from taseg_amd import _fast
    state = {"on": False}
    monkeypatch.setattr(_fast, "module", lambda: object())
    monkeypatch.setattr(_fast, "wgrad_stream", lambda on: state.__setitem__("on", bool(on)) or bool(on))
    from taseg_amd.options import options
    monkeypatch.setattr(options, "wgrad_stream", "auto")

    clock = {"t": 0.0}
    monkeypatch.setattr(time, "perf_counter", lambda: clock["t"])      # a clock that only the steps advance: no timing noise

    def run(ms_off, ms_on):
        calls = {"n": 0}

        def step():
            seq = ms_on if state["on"] else ms_off
            clock["t"] += seq[min(calls["n"] // 10, len(seq) - 1)] / 1e3       # 10 steps per round (2 settings x (1 + 4))
            calls["n"] += 1
        return _fast.tune_wgrad_stream(step, lambda: None, rounds=3, steps=4)[0]

    assert run([4.0], [3.6]) is True                      # 10 % faster in every round
    assert run([4.0], [3.98]) is False                    # inside the noise
    assert run([4.0], [4.4]) is False                     # slower
    assert run([4.0, 4.0, 4.0], [3.6, 4.1, 3.6]) is True  # one lost round of three is allowed
    assert run([4.0, 4.0, 4.0], [3.6, 4.1, 4.1]) is False
    monkeypatch.setattr(options, "wgrad_stream", "1")
    assert _fast.tune_wgrad_stream(lambda: None, lambda: None) == (True, None, None)
    monkeypatch.setattr(options, "wgrad_stream", "0")
    assert _fast.tune_wgrad_stream(lambda: None, lambda: None) == (False, None, None)


def test_options_object_is_typed_and_the_environment_only_overrides():
    """taseg_amd.options: one typed object; TASEG_<FIELD> replaces a default when the module is imported (diagnostics), legacy
    spellings are mapped, unknown fields / values are refused, overrides are scoped"""
    from taseg_amd.options import Options
    o = Options()
    assert o.rccl_direct == "c10d" and o.class_gemm is True and o.class_min_rows_96 == 48000       # the defaults
    taken = o.load_environment({"TASEG_RCCL_DIRECT": "1", "TASEG_CLASS_GEMM": "0", "TASEG_CLASS_MIN_ROWS_96": "123", "PATH": "x",
                                "TASEG_BENCH_WATCHDOG_S": "5"})
    assert set(taken) == {"rccl_direct", "class_gemm", "class_min_rows_96"}
    assert o.rccl_direct == "create" and o.class_gemm is False and o.class_min_rows_96 == 123
    with o.override(class_gemm=True, rccl_direct="borrow"):
        assert o.class_gemm is True and o.rccl_direct == "borrow"
    assert o.class_gemm is False and o.rccl_direct == "create"
    with pytest.raises(AttributeError):
        o.no_such_option = 1
    with pytest.raises(ValueError):
        o.rccl_direct = "carrier-pigeon"
    with pytest.raises(TypeError):
        o.class_min_rows_96 = "many" if False else 1.5
    with pytest.warns(UserWarning, match="TASEG_CLAS_GEMM"):
        Options().load_environment({"TASEG_CLAS_GEMM": "0"})                      # a typo is reported, not ignored
    # nothing else under taseg_amd/ reads the environment for configuration
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "taseg_amd")
    bad = []
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")) and f != "options.py":
                text = open(os.path.join(d, f)).read()
                if re.search(r"environ[.\w]*[\[(]\s*[\"']TASEG_|getenv\(\"TASEG_", text):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_a_model_with_a_reducer_still_pickles_and_finds_its_reducer():
    """the parameter -> reducer mapping lives outside the tensors (a weak reference in Parameter.__dict__ made torch.save(model) fail)"""
    import io
    import pickle
    from taseg_amd import parallel
    lin = torch.nn.Linear(8, 4)
    red = parallel.GradBucketReducer(lin)
    assert parallel.reducer_of(lin.weight) is red and parallel.reducer_of(lin.bias) is red
    pickle.dumps(lin)
    torch.save(lin, io.BytesIO())
    other = torch.nn.Linear(8, 4)
    assert parallel.reducer_of(other.weight) is None
    red.check_open([lin.weight])
    red.buckets[0]["launched"] = True
    with pytest.raises(RuntimeError, match="already launched"):
        red.check_open([lin.weight])
