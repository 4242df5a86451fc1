"""Parity AT THE BENCHMARKED SIZES (-m gpu).  The golden fixtures stop at 44k voxels; bench.py's three workloads run at
178k (2 x 120k-point scans), 210k (4-scan TFA) and 391k (nuScenes FSA, bs 4) voxels per step.  Here the very batches
bench.py builds (same generator functions, same seeds) go through the index stage and the convolution kernels and are
compared with the numpy oracle, which restates the reference (oracle/ts_oracle.py, pinned by the golden fixtures):

* data stage (KITTI multi-scan fuse + class-step filter + double voxelisation; nuScenes voxelisation of the fused cloud),
  stride-1 voxel set, coordinate pyramid, all 9 rulebooks (`results` / `nbmaps` / `nbsizes`), position tables and the three
  trilinear maps: bit for bit (conv.py:156-176, downsample.py:25-51, minkunet/utils.py:11-27,72-82);
* one stride-1 96 -> 96 and one stride-2 (kernel 2) 96 -> 96 convolution on the bench rulebooks, forward and both
  gradients against the float64 oracle (1e-5 of the tensor's scale), two-pass kernels against the neighbour-table
  kernel (convolution_cuda.cu:101-278);
* configs[4] as a model: MinkUNetMs, 17 classes, IN_FEATURE_DIM 4, voxel 0.1 m, bs 4 on the nuScenes stage's output
  under torch.autocast against its own fp32 step (nuscenes/minkunet_mk34_cr10_fsa.yaml:13-36).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bench  # noqa: E402  (the workload generators of the measured runs)
from oracle import ts_oracle as O  # noqa: E402
from taseg_amd.data.synthetic import (FLEXIBLE_STEPS_KITTI, FLEXIBLE_STEPS_NUSC, fill_parameters,  # noqa: E402
                                      make_model_cfg)

DEV = "cuda"


def _np(t):
    return t.detach().cpu().numpy()


def _same(got, want, what):
    got = _np(got) if isinstance(got, torch.Tensor) else np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.array_equal(got, want), what


def _tiny_model(name, in_dim, num_class=20):
    """prepare() is parameter-free: a narrow, shallow instance builds the same index plan as mk34 cr 1.0"""
    from taseg_amd.pcseg.model import build_network
    return build_network(make_model_cfg(name, in_dim=in_dim, cr=0.125, num_layer=[1] * 8), num_class).cuda().train()


def _oracle_pyramid(vox, levels=4):
    """coordinate sets and rulebooks of strides 1 .. 16 the way conv3d builds them lazily (conv.py:144-177)"""
    cmaps, kmaps = {1: vox}, {}
    cur, s = vox, 1
    for lvl in range(levels + 1):
        kmaps[(s, 3, 1)] = O.build_kmap(cur, cur, O.get_kernel_offsets(3, s, 1)) + ((len(cur), len(cur)),)
        if lvl < levels:
            down = O.spdownsample(cur, 2, 2, s)
            kmaps[(s, 2, 2)] = O.build_kmap(cur, down, O.get_kernel_offsets(2, s, 1)) + ((len(cur), len(down)),)
            cmaps[2 * s] = down
            cur, s = down, 2 * s
    return cmaps, kmaps


def _check_plan(plan, vox, point_coords):
    """plan: MinkUNet*.prepare() output; vox: oracle stride-1 voxels; point_coords: oracle float [N, 4]"""
    cmaps, kmaps = _oracle_pyramid(vox)
    assert len(plan["kmaps"]) == len(kmaps) == 9
    for s, c in cmaps.items():
        _same(plan["cmaps"][(s, s, s)], c, f"coordinates at stride {s}")
    pairs = {}
    for (s, ks, st), (results, nbmaps, nbsizes, sizes) in kmaps.items():
        km = plan["kmaps"][((s,) * 3, (ks,) * 3, (st,) * 3, (1, 1, 1))]
        tag = f"stride {s} kernel {ks}"
        assert km.sizes == sizes and km.total == len(nbmaps), tag
        _same(km.nbr, results.astype(np.int32), tag + " results")
        _same(km.nbmaps, nbmaps.astype(np.int32), tag + " nbmaps")
        _same(km.nbsizes, nbsizes.astype(np.int32), tag + " nbsizes")
        # position tables: row of nbmaps holding the pair of (k, out voxel) / (k, in voxel), -1 where there is none
        po, pi, nbm = _np(km.pos_out), _np(km.pos_in), nbmaps
        kk, jj = np.nonzero(results >= 0)
        assert np.array_equal(po >= 0, results >= 0), tag
        assert np.array_equal(nbm[po[kk, jj], 1], jj) and np.array_equal(nbm[po[kk, jj], 0], results[kk, jj]), tag
        ki, ii = np.nonzero(pi >= 0)
        assert len(ki) == len(nbm) and np.array_equal(nbm[pi[ki, ii], 0], ii), tag
        offs = np.concatenate([[0], np.cumsum(nbsizes)])
        assert np.array_equal(np.searchsorted(offs, pi[ki, ii], side="right") - 1, ki), tag
        pairs[(s, ks)] = len(nbmaps)
    for s in (1, 16, 4):
        idx, w = O.trilinear_map(point_coords, cmaps[s], s)
        _same(plan["tri_idx"][(s, s, s)], idx.astype(_np(plan["tri_idx"][(s, s, s)]).dtype), f"trilinear indices, stride {s}")
        got_w = _np(plan["tri_w"][(s, s, s)])
        assert np.abs(got_w - w).max() <= 1e-6, f"trilinear weights, stride {s}"
    return cmaps, kmaps, pairs


def test_single_frame_index_stage_at_bench_size():
    """configs[1]: 2 x 120 000 points -> ~178k voxels, 1.16 M pairs at stride 1"""
    from taseg_amd.torchsparse import SparseTensor
    coords, feats, labels, npts = bench.make_scans(0, 2, 120000, "minkunet")
    model = _tiny_model("MinkUNet", 4)
    bd = {"lidar": SparseTensor(feats, coords), "targets": SparseTensor(labels, coords),
          "offset": torch.tensor([len(coords)], device=DEV, dtype=torch.int32)}
    plan = model.prepare(bd)
    zc = _np(coords).astype(np.float32)
    scaled, cell, sparse_hash, idx_query, counts = O.initial_voxelize_maps(zc, model.pres, model.vres)
    vox = np.round(O.voxelize_forward(cell, idx_query, counts)).astype(np.int32)
    _same(plan["coords"], vox, "stride-1 voxels (ascending hash order)")
    _same(plan["vox_idx"], idx_query, "point -> voxel map")
    _same(plan["vox_counts"], counts, "points per voxel")
    _, _, pairs = _check_plan(plan, vox, scaled)
    print(f"single-frame bench batch: {npts} points -> {len(vox)} voxels, pairs per stride "
          f"{ {k: v for k, v in sorted(pairs.items())} }")
    assert len(vox) > 150000 and pairs[(1, 3)] > 5 * len(vox)


def _oracle_kitti_ms_sample(scan, voxel, steps):
    """semantickitti_ms.py:140-149,253-320 + semantickitti_voxel_ms.py:121-170 with the oracle's functions"""
    pts = [_np(p) for p in scan["points"]]
    lab = [_np(l) for l in scan["labels"]]
    poses = [_np(p) for p in scan["poses"]]
    t = len(pts) - 1
    raw, labs = [pts[t][:, :4]], [lab[t]]
    for i in range(t):
        delta = i - t
        keep = np.array([bool(steps[c]) and abs(delta) % steps[c] == 0 for c in range(len(steps))])[lab[i]]
        raw.append(O.fuse_scan(pts[i][:, :4], poses[t], poses[i])[keep])
        labs.append(lab[i][keep])
    raw_ms = O.append_time_flag(len(pts[t]), np.concatenate(raw))
    labels_ms = np.concatenate(labs)
    clamp = (raw_ms[:, :3] >= pts[t][:, :3].min(0)).all(1)
    raw_ms, labels_ms = raw_ms[clamp], labels_ms[clamp]
    pc_ms = O.voxel_coords(raw_ms, voxel)
    mins = pc_ms.min(0)
    pc_ms = pc_ms - mins
    pc = O.voxel_coords(pts[t], voxel) - mins
    idx_ms, inv_ms = O.sparse_quantize(pc_ms)
    idx, inv = O.sparse_quantize(pc)
    return dict(lidar_ms_C=pc_ms[idx_ms], lidar_ms_F=raw_ms[idx_ms], targets_ms_F=labels_ms[idx_ms], inverse_ms=inv_ms,
                lidar_C=pc[idx], lidar_F=pts[t][idx], inverse=inv, n_ms=len(raw_ms))


def test_tfa_stage_and_index_stage_at_bench_size():
    """configs[2]: 4-scan TFA, 2 x 5 x 120 000 points -> ~210k voxels per step, data stage included"""
    from taseg_amd.data.stage import build_multiscan_batch
    scans, npts = bench.make_multiscans(0, 2, 120000)
    bd = build_multiscan_batch(scans, 0.05, FLEXIBLE_STEPS_KITTI)
    want = [_oracle_kitti_ms_sample(s, 0.05, FLEXIBLE_STEPS_KITTI) for s in scans]

    def stacked(key):
        return np.concatenate([np.concatenate([w[key], np.full((len(w[key]), 1), b, np.int32)], 1) for b, w in enumerate(want)])

    _same(bd["lidar_ms"].C, stacked("lidar_ms_C"), "fused voxel coordinates")
    _same(bd["lidar_ms"].F, np.concatenate([w["lidar_ms_F"] for w in want]), "fused voxel features (float32 bits)")
    _same(bd["targets_ms"].F, np.concatenate([w["targets_ms_F"] for w in want]), "fused voxel labels")
    _same(bd["inverse_map_ms"].F, np.concatenate([w["inverse_ms"] for w in want]), "fused inverse map")
    _same(bd["lidar"].C, stacked("lidar_C"), "current-frame voxel coordinates")
    _same(bd["lidar"].F, np.concatenate([w["lidar_F"] for w in want]), "current-frame voxel features")
    _same(bd["inverse_map"].F, np.concatenate([w["inverse"] for w in want]), "current-frame inverse map")
    assert [int(n) for n in bd["num_points_ms"].view(-1)] == [w["n_ms"] for w in want]
    model = _tiny_model("MinkUNetMs", 5)
    plan = model.prepare(bd)
    vox = _np(bd["lidar_ms"].C)
    _same(plan["coords"], vox, "MinkUNetMs convolves the dataset's voxels as they are")
    _, _, pairs = _check_plan(plan, vox, vox.astype(np.float32))
    print(f"4-scan TFA bench batch: {npts} raw points -> {len(vox)} voxels, pairs per stride "
          f"{ {k: v for k, v in sorted(pairs.items())} }")
    assert len(vox) > 180000


def _nusc_batch():
    from taseg_amd.data.nuscenes import build_nuscenes_batch
    samples, npts, n_sweeps = bench.make_nusc_samples(0, 4, 34700)
    return samples, build_nuscenes_batch(samples, 0.1, FLEXIBLE_STEPS_NUSC), npts, n_sweeps


def test_nuscenes_index_stage_at_bench_size():
    """configs[4] shape: bs 4, 34.7k-point sweeps, voxel 0.1 m -> ~390k voxels per step.  The sweep fuse is pinned bit
    for bit by tests/golden/multiscan_nus.npz; here the voxelisation of the device-fused cloud and the whole index stage"""
    from taseg_amd.data import nuscenes as N
    samples, bd, npts, n_sweeps = _nusc_batch()
    c_want, inv_want = [], []
    for b, s in enumerate(samples):
        raw, lab, keep = N.fuse_sweeps(s["points"], s["labels"], s["hist_points"], s["hist_labels"], s["hist_pseudo"],
                                       s["params"], FLEXIBLE_STEPS_NUSC)
        cur = _np(s["points"])
        raw = _np(raw)[_np(keep)][:, :4]
        raw = raw[(raw[:, :3] >= cur[:, :3].min(0)).all(1)]
        pc = O.voxel_coords(raw, 0.1)
        pc = pc - pc.min(0)
        idx, inv = O.sparse_quantize(pc)
        c_want.append(np.concatenate([pc[idx], np.full((len(idx), 1), b, np.int32)], 1))
        inv_want.append(inv)
    _same(bd["lidar_ms"].C, np.concatenate(c_want), "fused voxel coordinates")
    _same(bd["inverse_map_ms"].F, np.concatenate(inv_want), "fused inverse map")
    model = _tiny_model("MinkUNetMs", 4, 17)
    plan = model.prepare(bd)
    vox = _np(bd["lidar_ms"].C)
    _, _, pairs = _check_plan(plan, vox, vox.astype(np.float32))
    print(f"nuScenes bench batch: {n_sweeps} sweeps per sample, {npts} raw points -> {len(vox)} voxels, pairs per stride "
          f"{ {k: v for k, v in sorted(pairs.items())} }")
    assert len(vox) > 300000


@pytest.fixture(scope="module")
def bench_maps():
    """stride-1 k3 and stride-1 -> 2 k2 rulebooks of the single-frame bench batch, device + oracle side"""
    from taseg_amd import backend as B
    from taseg_amd.torchsparse.nn.utils import get_kernel_offsets
    coords, _, _, _ = bench.make_scans(0, 2, 120000, "minkunet")
    vox = _np(coords)              # any 178k-voxel set of the bench density serves the convolution check
    down = O.spdownsample(vox, 2, 2, 1)
    out = {}
    for tag, (oc, ks) in {"s1k3": (vox, 3), "s1k2": (down, 2)}.items():
        res, nbmaps, nbsizes = O.build_kmap(vox, oc, O.get_kernel_offsets(ks, 1, 1))
        km = B.build_kmap(coords, torch.from_numpy(oc).cuda(), get_kernel_offsets(ks, 1, 1, device=DEV), want_inverse=True)
        out[tag] = dict(nbmaps=nbmaps, nbsizes=nbsizes, sizes=(len(vox), len(oc)), km=km)
    return out


def _rel(got, want):
    want = np.asarray(want, dtype=np.float64)
    return float(np.abs(_np(got).astype(np.float64) - want).max() / max(1.0, float(np.abs(want).max())))


@pytest.mark.parametrize("tag,ci,co", [("s1k3", 96, 96), ("s1k2", 96, 96), ("s1k3", 128, 96)])
def test_convolution_at_bench_size_vs_float64_oracle(bench_maps, tag, ci, co):
    """up4 / up3-shaped layers on the bench rulebook (1.16 M pairs at stride 1): forward, input gradient and weight
    gradient of the two-pass split-bf16 kernels against the float64 evaluation of convolution_cuda.cu:101-278"""
    from taseg_amd import backend as B
    m = bench_maps[tag]
    km, (n_in, n_out) = m["km"], m["sizes"]
    k = len(m["nbsizes"])
    rs = np.random.RandomState(ci + co + k)
    x = rs.randn(n_in, ci).astype(np.float32)
    w = (rs.randn(k, ci, co) / np.sqrt(k * ci / 4)).astype(np.float32)
    gy = rs.randn(n_out, co).astype(np.float32)
    want_y = O.conv_forward(x.astype(np.float64), w, m["nbmaps"], m["nbsizes"], m["sizes"])
    want_gx, want_gw = O.conv_backward(x.astype(np.float64), w, gy, m["nbmaps"], m["nbsizes"])
    total = int(km["nboffs"][-1])
    assert total == len(m["nbmaps"])
    xd, wd, gyd = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(), torch.from_numpy(gy).cuda()
    z = B.conv_pair_gemm(xd, wd, km["nbmaps"], km["nboffs"], total, gather_col=0)
    y = B.conv_gather_sum(z, km["pos_out"], n_out)
    zg = B.conv_pair_gemm(gyd, wd, km["nbmaps"], km["nboffs"], total, gather_col=1, weight_transposed=True)
    gx = B.conv_gather_sum(zg, km["pos_in"], n_in)
    gw = B.conv_wgrad(xd, gyd, km["nbmaps"], km["nboffs"], k, 0, total)
    errs = (_rel(y, want_y), _rel(gx, want_gx), _rel(gw, want_gw))
    print(f"{tag} {ci}->{co}: {total} pairs, rel. error vs float64: y {errs[0]:.2e}, grad_x {errs[1]:.2e}, grad_w {errs[2]:.2e}")
    assert max(errs) <= 1e-5, errs
    # the two passes against the single-launch neighbour-table kernel (the reference-form entry points' kernel)
    y_nbr = B.conv_nbr(xd, wd, km["nbr"], n_out)
    assert _rel(y_nbr, want_y) <= 1e-5
    # run-to-run identical
    z2 = B.conv_pair_gemm(xd, wd, km["nbmaps"], km["nboffs"], total, gather_col=0)
    assert torch.equal(B.conv_gather_sum(z2, km["pos_out"], n_out), y)
    if tag == "s1k3":
        # the class-sorted implicit GEMM the training step takes on a map of this size (csrc/conv_class.hip): same bar against
        # float64, every pair of the rulebook on the plan, Z' under a third of Z, results independent of the tile order
        plan = B.conv_class_plan(km["nbr"])
        yc = B.conv_gather_sum(B.conv_class_gemm(xd, wd, plan), plan["pos"], n_out)
        gxc = B.conv_gather_sum(B.conv_class_gemm(gyd, wd, plan, weight_transposed=True), plan["pos"], n_in)
        ec = (_rel(yc, want_y), _rel(gxc, want_gx))
        z_rows = 128 * int(plan["n_tiles"][0])
        print(f"   class-sorted: y {ec[0]:.2e}, grad_x {ec[1]:.2e}; Z' rows {z_rows} = {z_rows / n_out:.2f} N against {total / n_out:.2f} N pairs")
        assert max(ec) <= 1e-5 and z_rows * 3 <= total
        assert int((plan["src"] >= 0).sum()) == total
        plan2 = B.conv_class_plan(km["nbr"])
        assert torch.equal(B.conv_gather_sum(B.conv_class_gemm(xd, wd, plan2), plan2["pos"], n_out), yc)


def test_nuscenes_config_autocast_vs_fp32_at_bench_size():
    """BASELINE configs[4] as a model: MinkUNetMs mk34 cr 1.0, 17 classes, IN_FEATURE_DIM 4, voxel 0.1 m, bs 4, on the
    nuScenes stage's output; one training step under torch.autocast (half storage, fp32 accumulation) against the fp32
    step of the same weights.  SURVEY 8(d) fp16 gate: deviation reported, arg-max agreement >= 99 %."""
    from taseg_amd.pcseg.model import build_network
    _, bd, npts, _ = _nusc_batch()
    cfg = make_model_cfg("MinkUNetMs", in_dim=4, cr=1.0)
    model = fill_parameters(build_network(cfg, 17), seed=11).cuda().train()
    res = {}
    for mode in ("fp32", "amp"):
        grabbed = {}
        h = model.classifier.register_forward_hook(lambda m, i, o: grabbed.__setitem__("logits", o.detach().float()))
        h2 = model.stage2[1].register_forward_hook(lambda m, i, o: grabbed.__setitem__("feat_dtype", o.F.dtype))
        model.zero_grad(set_to_none=True)
        batch = dict(bd)
        batch.pop("_plan", None)
        with torch.autocast("cuda", dtype=torch.float16, enabled=(mode == "amp")):
            ret, _, _ = model(batch)
        ret["loss"].float().backward()
        h.remove()
        h2.remove()
        res[mode] = (grabbed["logits"], float(ret["loss"]), grabbed["feat_dtype"],
                     {k: p.grad.detach().float().clone() for k, p in model.named_parameters()})
        for m in model.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.reset_running_stats()
    (l32, loss32, d32, g32), (l16, loss16, d16, g16) = res["fp32"], res["amp"]
    assert d32 == torch.float32 and d16 == torch.float16
    assert l32.shape == (bd["lidar_ms"].C.shape[0], 17)
    dev = (l16 - l32).abs()
    agree = float((l16.argmax(1) == l32.argmax(1)).float().mean())
    print(f"configs[4] model, AMP vs fp32 on {l32.shape[0]} voxels ({npts} raw points): max |dlogit| {float(dev.max()):.4f}, "
          f"mean {float(dev.mean()):.5f}, argmax agreement {100 * agree:.2f} %, loss {loss16:.5f} vs {loss32:.5f}")
    assert np.isfinite(loss16) and np.isfinite(loss32)
    assert agree >= 0.99 and float(dev.mean()) < 0.02 and abs(loss16 - loss32) < 2e-2
    for k in ("stem.0.kernel", "stage2.1.net.0.kernel", "stage4.1.net.3.kernel", "up4.1.1.net.3.kernel", "classifier.0.weight"):
        cos = float(torch.nn.functional.cosine_similarity(g16[k].flatten(), g32[k].flatten(), dim=0))
        assert cos > 0.99, (k, cos)
