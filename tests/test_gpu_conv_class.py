"""Class-sorted implicit GEMM (csrc/conv_class.hip): plan properties, and the convolution / input gradient it computes against
the float64 oracle and the two-pass kernels (same product, another summation order: 1e-6-close, not bit-identical)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ts_oracle as O  # noqa: E402

DEV = "cuda"


def _cloud(seed, n, extent, batch=2):
    rs = np.random.RandomState(seed)
    c = np.unique(np.concatenate([rs.randint(0, extent, (n, 3)), rs.randint(0, batch, (n, 1))], 1), axis=0)
    return c[rs.permutation(len(c))].astype(np.int32)


def _T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("n,extent", [(1, 4), (100, 3), (127, 40), (129, 6), (5000, 14), (40000, 40)])
def test_class_plan_lists_every_pair_once(n, extent):
    """every (output row, offset) pair of the rulebook appears exactly once in src, at the row pos names; dead rows and the
    padding carry -1; the tile list holds every tile with a live row, longest first, with the union of its rows' masks"""
    from taseg_amd import backend as B
    c = _cloud(n, n, extent)
    offs = O.get_kernel_offsets(3, 1, 1)
    km = B.build_kmap(_T(c), _T(c), _T(offs))
    plan = B.conv_class_plan(km["nbr"])
    nbr = km["nbr"].cpu().numpy()
    nn = nbr.shape[1]
    src, pos = plan["src"].cpu().numpy(), plan["pos"].cpu().numpy()
    m_pad = plan["m_pad"]
    assert m_pad == 3 * ((nn + 127) // 128 * 128) and src.shape == (9, m_pad)
    got = np.full_like(nbr, -1)
    for g in range(3):
        live = np.nonzero(pos[g] >= 0)[0]
        assert len(np.unique(pos[g][live])) == len(live)                      # one Z' row per (row, group)
        assert np.all((pos[g][live] >= g * (m_pad // 3)) & (pos[g][live] < (g + 1) * (m_pad // 3)))
        got[9 * g:9 * g + 9][:, live] = src[:, pos[g][live]]
        dead = np.nonzero(pos[g] < 0)[0]
        assert np.all(nbr[9 * g:9 * g + 9][:, dead] < 0)                        # rows without a neighbour in the group
    assert np.array_equal(got, nbr)
    used = np.zeros(m_pad, bool)
    used[pos[pos >= 0]] = True
    assert np.all(src[:, ~used] == -1)
    nt = int(plan["n_tiles"][0])
    info = plan["tile_info"].cpu().numpy()[:nt]
    tiles = info[:, 0] >> 2
    assert len(np.unique(tiles)) == nt
    live_tiles = np.unique(np.nonzero(used)[0] // 128)
    assert np.array_equal(np.sort(tiles), live_tiles)
    masks = ((src >= 0).reshape(9, -1, 128).any(2) * (1 << np.arange(9))[:, None]).sum(0)
    assert np.array_equal(info[:, 1], masks[tiles]) and np.array_equal(info[:, 0] & 3, tiles // (m_pad // 3 // 128))
    pop = np.array([bin(int(v)).count("1") for v in info[:, 1]])
    assert np.all(pop[:-1] >= pop[1:])                                           # longest first
    assert int(plan["n_tiles"][1]) == int(pop.sum())                             # the plan's (tile, offset) steps


@pytest.mark.parametrize("ci,co", [(32, 32), (32, 64), (64, 64), (96, 96), (128, 96), (96, 128), (256, 128)])
def test_class_gemm_is_the_convolution(ci, co):
    from taseg_amd import backend as B
    c = _cloud(7, 20000, 30)
    offs = O.get_kernel_offsets(3, 1, 1)
    km = B.build_kmap(_T(c), _T(c), _T(offs))
    plan = B.conv_class_plan(km["nbr"])
    _, nbmaps, nbsizes = O.build_kmap(c, c, offs)
    total = len(nbmaps)
    rs = np.random.RandomState(1)
    x = rs.randn(len(c), ci).astype(np.float32)
    x[::9] *= 1e-3
    w = (rs.randn(27, ci, co) / np.sqrt(ci)).astype(np.float32)
    gy = rs.randn(len(c), co).astype(np.float32)
    y64 = O.conv_forward(x.astype(np.float64), w.astype(np.float64), nbmaps, nbsizes, (len(c), len(c)))
    gx64, _ = O.conv_backward(x.astype(np.float64), w.astype(np.float64), gy.astype(np.float64), nbmaps, nbsizes)
    xt, wt, gt = _T(x), _T(w), _T(gy)
    y = B.conv_gather_sum(B.conv_class_gemm(xt, wt, plan), plan["pos"], len(c))
    gx = B.conv_gather_sum(B.conv_class_gemm(gt, wt, plan, weight_transposed=True), plan["pos"], len(c))
    y2 = B.conv_gather_sum(B.conv_pair_gemm(xt, wt, km["nbmaps"], km["nboffs"], total, 0), km["pos_out"], len(c))
    gx2 = B.conv_gather_sum(B.conv_pair_gemm(gt, wt, km["nbmaps"], km["nboffs"], total, 1, weight_transposed=True),
                            km["pos_in"], len(c))

    def rel(a, b):
        b = np.asarray(b, dtype=np.float64)
        return float(np.abs(a.double().cpu().numpy() - b).max()) / float(np.abs(b).max())

    assert rel(y, y64) <= 1e-5 and rel(gx, gx64) <= 1e-5
    assert rel(y, y2.double().cpu().numpy()) <= 2e-6 and rel(gx, gx2.double().cpu().numpy()) <= 2e-6
    # deterministic: the same launch twice gives the same bits (tile order varies, results do not)
    plan_b = B.conv_class_plan(km["nbr"])
    y_b = B.conv_gather_sum(B.conv_class_gemm(xt, wt, plan_b), plan_b["pos"], len(c))
    assert torch.equal(y, y_b)


@pytest.mark.parametrize("ci,co,n,extent", [(32, 32, 20000, 30), (64, 96, 20000, 30), (96, 96, 20000, 30), (128, 96, 20000, 30),
                                            (96, 128, 20000, 30), (192, 96, 20000, 30), (256, 128, 9000, 24), (384, 256, 3000, 16),
                                            (160, 64, 6000, 20), (64, 64, 100, 6), (96, 96, 120000, 70), (32, 64, 120000, 70)])
def test_class_gemm_half_storage(ci, co, n, extent):
    """IEEE-half rows: the class path rounds a Z' row once per (row, z-plane), the two-pass form once per pair - both within
    half precision of the fp32 evaluation.  Row chunks of 128 / 96 / 64 / 32 columns (one to five per row), one and two column
    tiles, tile lists shorter and longer than the persistent grid (120 000 voxels: ~2 800 tiles on <= 1 024 workgroups)"""
    from taseg_amd import backend as B
    c = _cloud(11, n, extent)
    offs = O.get_kernel_offsets(3, 1, 1)
    km = B.build_kmap(_T(c), _T(c), _T(offs))
    plan = B.conv_class_plan(km["nbr"])
    total = int(km["nboffs"][-1])
    rs = np.random.RandomState(2)
    x = _T(rs.randn(len(c), ci).astype(np.float32))
    gy = _T(rs.randn(len(c), co).astype(np.float32))
    w = _T((rs.randn(27, ci, co) / np.sqrt(ci)).astype(np.float32))
    xh, gh, wh = x.half(), gy.half(), w.half()
    y32 = B.conv_gather_sum(B.conv_class_gemm(xh.float(), wh.float(), plan), plan["pos"], len(c))
    g32 = B.conv_gather_sum(B.conv_class_gemm(gh.float(), wh.float(), plan, weight_transposed=True), plan["pos"], len(c))
    y = B.conv_gather_sum_f16(B.conv_class_gemm_f16(xh, wh, plan), plan["pos"], len(c))
    g = B.conv_gather_sum_f16(B.conv_class_gemm_f16(gh, wh, plan, weight_transposed=True), plan["pos"], len(c))
    y2 = B.conv_gather_sum_f16(B.conv_pair_gemm_f16(xh, wh, km["nbmaps"], km["nboffs"], total, 0, natural=True), km["pos_out"], len(c))
    g2 = B.conv_gather_sum_f16(B.conv_pair_gemm_f16(gh, wh, km["nbmaps"], km["nboffs"], total, 1), km["pos_in"], len(c))
    assert y.dtype == torch.float16 and g.dtype == torch.float16

    def err(a, b):
        return float((a.float() - b).abs().max()) / float(b.abs().max())

    for got, two, ref in ((y, y2, y32), (g, g2, g32)):
        assert err(got, ref) <= 2e-3                                   # a few half ulps of the tensor's scale
        assert err(got, ref) <= 1.5 * err(two, ref) + 2e-4       # same error budget: 3 roundings of group sums vs 6.5 of pair products


@pytest.mark.parametrize("ci,co,n,extent", [(32, 32, 3000, 16), (96, 96, 20000, 30), (128, 96, 20000, 30), (64, 128, 9000, 24),
                                            (256, 128, 3000, 16), (96, 96, 120000, 70)])
@pytest.mark.parametrize("half", [False, True])
def test_class_conv_finishes_in_the_product_with_the_bits_of_pass_2(ci, co, n, extent, half):
    """ts_conv_class_conv: the centre group's tiles add the outer groups' Z' rows, the addend, and store the result rows - the same
    additions in the same order as ts_conv_class_gemm + ts_conv_gather_sum (+ addend): bit-identical, forward and transposed product,
    fp32 and half storage"""
    from taseg_amd import backend as B
    c = _cloud(5, n, extent)
    offs = O.get_kernel_offsets(3, 1, 1)
    km = B.build_kmap(_T(c), _T(c), _T(offs))
    plan = B.conv_class_plan(km["nbr"])
    nv = len(c)
    tiles, steps = plan["n_tiles"].tolist()
    info = plan["tile_info"][:tiles].cpu().numpy()
    assert int(((info[:, 0] & 3) == 1).sum()) == (nv + 127) // 128            # every row is in one centre-group tile
    rs = np.random.RandomState(3)
    x = _T(rs.randn(nv, ci).astype(np.float32))
    gy = _T(rs.randn(nv, co).astype(np.float32))
    w = _T((rs.randn(27, ci, co) / np.sqrt(ci)).astype(np.float32))
    add_x = _T(rs.randn(nv, ci).astype(np.float32))
    if half:
        x, gy, w, add_x = x.half(), gy.half(), w.half(), add_x.half()
        gemm, conv, gsum = B.conv_class_gemm_f16, B.conv_class_conv_f16, B.conv_gather_sum_f16
    else:
        gemm, conv, gsum = B.conv_class_gemm, B.conv_class_conv, B.conv_gather_sum
    y_ref = gsum(gemm(x, w, plan), plan["pos"], nv)
    g_ref = gsum(gemm(gy, w, plan, weight_transposed=True), plan["pos"], nv)
    assert torch.equal(conv(x, w, plan), y_ref)
    assert torch.equal(conv(gy, w, plan, weight_transposed=True), g_ref)
    # the addend lands in the same store: one more add after the sum over the offsets, as in ts_conv_gather_sum_ex
    got = conv(gy, w, plan, weight_transposed=True, addend=add_x)
    zt = gemm(gy, w, plan, weight_transposed=True).float()
    acc = torch.zeros(nv, ci, device=DEV)
    for grp_i in range(3):                                      # fp32 sums in group order, the addend last, ONE rounding
        p = plan["pos"][grp_i].long()
        acc[p >= 0] += zt[p[p >= 0]]
    assert torch.equal(acc.to(g_ref.dtype), g_ref)
    assert torch.equal(got, (acc + add_x.float()).to(g_ref.dtype))
    with pytest.raises(ValueError):
        conv(x, w, B.conv_class_plan(km["nbr"][:8].contiguous(), direct=True))


def test_class_plan_is_kept_only_while_its_work_stays_near_the_pairs(monkeypatch):
    """KernelMap.build_class_plan reads the plan's (tile, offset) steps and keeps the plan only while 128 * steps stays under
    _CLASS_MAX_WORK x the rulebook's pairs.  Sorting by mask makes that easy to meet - a LiDAR-like surface gives 1.03, even 35 %
    of a dense box chosen at random 1.46 - so the guard is exercised here with a tighter bound: the surface map keeps its plan,
    the random one goes back to the two passes."""
    from taseg_amd import backend as B
    from taseg_amd.torchsparse.nn import functional as F
    rs = np.random.RandomState(0)
    u, v = np.meshgrid(np.arange(150), np.arange(150), indexing="ij")
    plane = np.stack([u.ravel(), v.ravel(), (u.ravel() // 7) % 3], 1)
    wall = np.stack([u.ravel(), np.full(u.size, 60), v.ravel() % 40], 1)
    surf = np.unique(np.concatenate([plane, wall]), axis=0)
    surf = np.concatenate([surf, np.zeros((len(surf), 1), np.int64)], 1).astype(np.int32)
    box = np.stack(np.meshgrid(np.arange(40), np.arange(40), np.arange(40), indexing="ij"), -1).reshape(-1, 3)
    noise = box[rs.rand(len(box)) < 0.35]
    noise = np.concatenate([noise, np.zeros((len(noise), 1), np.int64)], 1).astype(np.int32)
    monkeypatch.setattr(F, "_CLASS_MAX_WORK", 1.2)
    got = {}
    for name, c in (("surface", surf), ("noise", noise)):
        assert len(c) >= 16384
        km = F.build_kernel_map(_T(c), _T(c), 3, 1)
        plan = km.build_class_plan()
        tiles, steps = B.conv_class_plan(km.nbr)["n_tiles"].tolist()
        got[name] = (plan is not None, 128 * steps / km.total)
        if plan is not None:
            assert plan["z_rows"] == 128 * tiles and km.class_rows() == 128 * tiles
    assert got["surface"][0] and got["surface"][1] < 1.2, got
    assert not got["noise"][0] and 1.2 < got["noise"][1] < 1.6, got


# ---------------------------------------------------------------------------------------------- direct plans (2x2x2 maps)
def _strided_map(seed, n, extent, stride=1):
    from taseg_amd import backend as B
    fine = _cloud(seed, n, extent) * np.array([stride, stride, stride, 1], np.int32)
    coarse = O.spdownsample(fine, 2, 2, stride)
    offs = O.get_kernel_offsets(2, stride, 1)
    km = B.build_kmap(_T(fine), _T(coarse), _T(offs))
    _, nbmaps, nbsizes = O.build_kmap(fine, coarse, offs)
    return fine, coarse, km, nbmaps, nbsizes


@pytest.mark.parametrize("n,extent", [(1, 4), (300, 5), (129, 40), (6000, 24)])
def test_direct_plans_hold_every_pair_of_a_strided_map(n, extent):
    """the two direct plans of a 2x2x2 / stride-2 map (conv.py:184-192, SURVEY App. A: every fine voxel has exactly one
    parent): destination = coarse rows ("down") lists each coarse row once with all its children, destination = fine rows
    ("up") lists each fine row once with its one parent; padding slots carry -1"""
    from taseg_amd import backend as B
    fine, coarse, km, nbmaps, nbsizes = _strided_map(n, n, extent)
    nbr = km["nbr"].cpu().numpy()                                                # [8, n_coarse]: fine row per (offset, coarse row)
    down = B.conv_class_plan(km["nbr"], direct=True)
    nbr_t = B.conv_nbr_transposed(km["pos_in"], km["nbmaps"], 8)
    up = B.conv_class_plan(nbr_t, direct=True)
    want_t = np.full((8, len(fine)), -1, np.int64)
    kk = np.repeat(np.arange(8), nbsizes)
    want_t[kk, nbmaps[:, 0]] = nbmaps[:, 1]
    assert np.array_equal(nbr_t.cpu().numpy(), want_t) and np.all((want_t >= 0).sum(0) == 1)
    for plan, table, n_dest in ((down, nbr, len(coarse)), (up, want_t, len(fine))):
        src, rows = plan["src"].cpu().numpy(), plan["rows"].cpu().numpy()
        assert plan["pos"] is None and plan["groups"] == 1 and plan["mirror"] == 0
        assert plan["m_pad"] == (n_dest + 127) // 128 * 128 and src.shape == (8, plan["m_pad"])
        live = rows >= 0
        assert np.array_equal(np.sort(rows[live]), np.arange(n_dest))             # every destination row exactly once
        assert np.array_equal(src[:, live], table[:, rows[live]]) and np.all(src[:, ~live] == -1)
        nt = int(plan["n_tiles"][0])
        info = plan["tile_info"].cpu().numpy()[:nt]
        assert nt == plan["m_pad"] // 128 and np.array_equal(np.sort(info[:, 0] >> 2), np.arange(nt))
        masks = ((src >= 0).reshape(8, -1, 128).any(2) * (1 << np.arange(8))[:, None]).sum(0)
        assert np.array_equal(info[:, 1], masks[info[:, 0] >> 2])


@pytest.mark.parametrize("ci,co", [(32, 32), (64, 64), (128, 96), (96, 96), (256, 128), (128, 128)])
@pytest.mark.parametrize("half", [False, True])
def test_direct_plans_are_the_strided_and_transposed_convolution(ci, co, half):
    """ONE pass, no Z: strided forward / its input gradient and the transposed convolution's forward / input gradient on the
    direct plans against the float64 oracle (convolution_cuda.cu:101-278 with transpose = 0 / 1) and against pair GEMM +
    pass 2 - the same products added in the same order (k ascending), so the same bits in fp32"""
    from taseg_amd import backend as B
    fine, coarse, km, nbmaps, nbsizes = _strided_map(5, 20000, 36)
    nf, nc, total = len(fine), len(coarse), len(nbmaps)
    down = B.conv_class_plan(km["nbr"], direct=True)
    up = B.conv_class_plan(B.conv_nbr_transposed(km["pos_in"], km["nbmaps"], 8), direct=True)
    rs = np.random.RandomState(3)
    xf = rs.randn(nf, ci).astype(np.float32)           # features on the fine rows
    xc = rs.randn(nc, ci).astype(np.float32)           # ... on the coarse rows (transposed convolution input)
    gc = rs.randn(nc, co).astype(np.float32)
    gf = rs.randn(nf, co).astype(np.float32)
    w = (rs.randn(8, ci, co) / np.sqrt(ci)).astype(np.float32)
    f64 = lambda a: a.astype(np.float64)  # noqa: E731
    want = {
        "strided forward": O.conv_forward(f64(xf), f64(w), nbmaps, nbsizes, (nf, nc)),
        "strided input gradient": O.conv_backward(f64(xf), f64(w), f64(gc), nbmaps, nbsizes)[0],
        "transposed forward": O.conv_forward(f64(xc), f64(w), nbmaps, nbsizes, (nf, nc), transposed=True),
        "transposed input gradient": O.conv_backward(f64(xc), f64(w), f64(gf), nbmaps, nbsizes, transposed=True)[0],
    }
    T = (lambda a: _T(a).half()) if half else _T
    wt = T(w)
    cg = B.conv_class_gemm_f16 if half else B.conv_class_gemm
    pg = B.conv_pair_gemm_f16 if half else B.conv_pair_gemm
    gs = B.conv_gather_sum_f16 if half else B.conv_gather_sum
    nat = dict(natural=True) if half else {}
    wtr = {} if half else dict(weight_transposed=True)
    got = {
        "strided forward": cg(T(xf), wt, down),
        "strided input gradient": cg(T(gc), wt, up, weight_transposed=True),
        "transposed forward": cg(T(xc), wt, up),
        "transposed input gradient": cg(T(gf), wt, down, weight_transposed=True),
    }
    two = {
        "strided forward": gs(pg(T(xf), wt, km["nbmaps"], km["nboffs"], total, 0, **nat), km["pos_out"], nc),
        "strided input gradient": gs(pg(T(gc), wt, km["nbmaps"], km["nboffs"], total, 1, **wtr), km["pos_in"], nf),
        "transposed forward": gs(pg(T(xc), wt, km["nbmaps"], km["nboffs"], total, 1, **nat), km["pos_in"], nf),
        "transposed input gradient": gs(pg(T(gf), wt, km["nbmaps"], km["nboffs"], total, 0, **wtr), km["pos_out"], nc),
    }
    for name, y in got.items():
        ref = want[name]
        assert tuple(y.shape) == ref.shape, name
        err = float(np.abs(y.double().cpu().numpy() - ref).max()) / float(np.abs(ref).max())
        err2 = float(np.abs(two[name].double().cpu().numpy() - ref).max()) / float(np.abs(ref).max())
        same = bool(torch.equal(y, two[name]))
        print(f"{'half' if half else 'fp32'} {ci}->{co} {name}: rel err vs fp64 {err:.2e} (two passes {err2:.2e}), bit-identical to the two passes: {same}")
        if half:
            assert err <= 2e-3 and err <= 1.5 * err2 + 2e-4, name
        else:
            assert err <= 1e-5, name
            if "fine" in name or name in ("strided input gradient", "transposed forward"):
                assert same, name              # one pair per destination row: nothing is added, the product is the result
    y_again = cg(T(xf), wt, B.conv_class_plan(km["nbr"], direct=True))
    assert torch.equal(y_again, got["strided forward"])                       # deterministic across plan builds
    # the sort-free "up" plan (slot = rulebook pair: ts_conv_class_plan_pairs) computes the same rows, bit for bit
    up2 = B.conv_class_plan_pairs(km["nbmaps"], km["nboffs"], 8, total)
    assert total == nf and up2["m_pad"] == up["m_pad"]
    r2 = up2["rows"].cpu().numpy()
    assert np.array_equal(np.sort(r2[r2 >= 0]), np.arange(nf)) and np.array_equal(r2[:total], nbmaps[:, 0])
    assert torch.equal(cg(T(gc), wt, up2, weight_transposed=True), got["strided input gradient"])
    assert torch.equal(cg(T(xc), wt, up2), got["transposed forward"])


def test_direct_plan_writes_zero_rows_without_neighbours():
    """a destination row no pair reaches must come out as zeros (the direct store is the only writer of the result)"""
    from taseg_amd import backend as B
    rs = np.random.RandomState(0)
    n, k = 1000, 8
    nbr = np.full((k, n), -1, np.int32)
    livecols = rs.permutation(n)[:600]
    for j in livecols:
        ks = rs.permutation(k)[:rs.randint(1, 4)]
        nbr[ks, j] = rs.randint(0, 700, size=len(ks))
    plan = B.conv_class_plan(_T(nbr), direct=True)
    x = _T(rs.randn(700, 32).astype(np.float32))
    w = _T(rs.randn(k, 32, 64).astype(np.float32))
    y = B.conv_class_gemm(x, w, plan)
    want = np.zeros((n, 64))
    xn, wn = x.double().cpu().numpy(), w.double().cpu().numpy()
    for kk in range(k):
        j = np.nonzero(nbr[kk] >= 0)[0]
        want[j] += xn[nbr[kk, j]] @ wn[kk]
    assert float(np.abs(y.double().cpu().numpy() - want).max()) <= 1e-4
    dead = np.setdiff1d(np.arange(n), livecols)
    assert len(dead) > 300 and bool((y[_T(dead).long()] == 0).all())
