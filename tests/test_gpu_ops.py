"""Parity of every HIP entry point (through the C ABI, via taseg_amd.backend) against the golden
vectors of the real reference and against the CPU oracle on seeded inputs.

Bars: integer / index results bit-exact; fp32 results <= 1e-5 relative to the tensor's scale
(MFMA f32 accumulates in a different order than the reference's BLAS GEMM + scatter-add)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ts_oracle as O  # noqa: E402

DEV = "cuda"


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def close(a, b, tol=1e-5):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())
    assert err <= tol * scale, (err, scale)


def same(a, b):
    a = a.cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    assert a.shape == np.asarray(b).shape, (a.shape, np.asarray(b).shape)
    assert np.array_equal(a, b)


@pytest.fixture(scope="module")
def B():
    from taseg_amd import backend
    return backend


@pytest.fixture(scope="module")
def F():
    from taseg_amd.torchsparse.nn import functional
    return functional


# --------------------------------------------------------------------------- hashing / query / count
def test_hash_golden(B, g_ops):
    same(B.hash_cuda(T(g_ops["coords"])), g_ops["hash"])
    same(B.hash_cuda(T(g_ops["coords_neg"])), g_ops["hash_neg"])
    same(B.kernel_hash_cuda(T(g_ops["coords"]), T(g_ops["offsets_k3s1"])), g_ops["khash_k3s1"])


def test_hash_known_answers(B):
    c = torch.tensor([[0, 0, 0, 0], [1, 0, 0, 0], [0, 0, 0, 1], [1, 0, 0, 1]], dtype=torch.int32, device=DEV)
    assert B.hash_cuda(c).tolist() == [947293587111810033, 948793285165995886, 947292487600181830,
                                       948794384677624093]


def test_hash_empty_and_large(B):
    assert B.hash_cuda(torch.zeros((0, 4), dtype=torch.int32, device=DEV)).numel() == 0
    rs = np.random.RandomState(0)
    c = rs.randint(-2000, 2000, size=(300001, 4)).astype(np.int32)
    same(B.hash_cuda(T(c)), O.sphash(c))


def test_hash_query_semantics(B, F):
    rs = np.random.RandomState(1)
    refs = np.unique(rs.randint(0, 1 << 40, size=5000).astype(np.int64))
    q = np.concatenate([refs[::3], rs.randint(0, 1 << 40, size=2000).astype(np.int64)])
    want = O.sphashquery(q, refs)
    got = F.sphashquery(T(q), T(refs))
    same(got, want)
    # explicit idx + duplicate reference keys keep the first (query_cpu.cpp:20-24)
    refs2 = np.array([5, 9, 5, 7], dtype=np.int64)
    idx2 = np.array([10, 11, 12, 13], dtype=np.int64)
    got2 = B.hash_query_cuda(T(np.array([5, 7, 8, 9], dtype=np.int64)), T(refs2), T(idx2))
    assert got2.tolist() == [11, 14, 0, 12]
    # 2-D query keeps its shape; empty reference set -> all misses
    q2 = T(q[:12].reshape(3, 4))
    assert F.sphashquery(q2, T(refs)).shape == (3, 4)
    assert (F.sphashquery(q2, torch.zeros(0, dtype=torch.int64, device=DEV)) == -1).all()


def test_count(B):
    rs = np.random.RandomState(2)
    idx = rs.randint(-1, 50, size=10000).astype(np.int32)
    same(B.count_cuda(T(idx), 50), O.spcount(idx, 50))


# --------------------------------------------------------------------------- rulebook
def test_downsample_and_kmaps_golden(B, F, g_ops):
    from taseg_amd.torchsparse.nn.utils import get_kernel_offsets
    cur, ts = T(g_ops["coords"]), 1
    for _ in range(3):
        km = B.build_kmap(cur, cur, get_kernel_offsets(3, ts, 1, device=DEV))
        total = int(km["nboffs"][-1])
        same(km["nbr"], g_ops[f"k3_s{ts}_results"])
        same(km["nbmaps"][:total], g_ops[f"k3_s{ts}_nbmaps"])
        same(km["nbsizes"], g_ops[f"k3_s{ts}_nbsizes"])
        down = F.spdownsample(cur, 2, 2, ts)
        same(down, g_ops[f"down_s{ts}"])
        km2 = B.build_kmap(cur, down, get_kernel_offsets(2, ts, 1, device=DEV), want_inverse=True)
        total2 = int(km2["nboffs"][-1])
        assert total2 == cur.shape[0]
        same(km2["nbr"], g_ops[f"k2_s{ts}_results"])
        same(km2["nbmaps"][:total2], g_ops[f"k2_s{ts}_nbmaps"])
        same(km2["nbsizes"], g_ops[f"k2_s{ts}_nbsizes"])
        # inverse table is the exact inverse of the neighbour table
        nbr, nbr_t = km2["nbr"].cpu().numpy(), km2["nbr_t"].cpu().numpy()
        kk, jj = np.nonzero(nbr >= 0)
        assert np.array_equal(nbr_t[kk, nbr[kk, jj]], jj)
        assert (nbr_t >= 0).sum() == (nbr >= 0).sum()
        # position tables point at the rulebook rows of their pairs
        nbm = km2["nbmaps"][:total2].cpu().numpy()
        po, pi = km2["pos_out"].cpu().numpy(), km2["pos_in"].cpu().numpy()
        assert np.array_equal(po >= 0, nbr >= 0) and np.array_equal(pi >= 0, nbr_t >= 0)
        assert np.array_equal(nbm[po[kk, jj], 1], jj) and np.array_equal(nbm[po[kk, jj], 0], nbr[kk, jj])
        ki, ii = np.nonzero(pi >= 0)
        assert np.array_equal(nbm[pi[ki, ii], 0], ii) and np.array_equal(nbm[pi[ki, ii], 1], nbr_t[ki, ii])
        cur, ts = down, ts * 2


def _blob(seed, n=20000, extent=48, batch=2):
    """dense random blob: many neighbours per voxel, unique rows, shuffled (unsorted) order"""
    rs = np.random.RandomState(seed)
    c = np.unique(np.concatenate([rs.randint(0, extent, size=(n, 3)), rs.randint(0, batch, size=(n, 1))], 1), axis=0)
    return c[rs.permutation(len(c))].astype(np.int32)


def test_kmap_dense_blob_vs_oracle(B, F):
    from taseg_amd.torchsparse.nn.utils import get_kernel_offsets
    c = _blob(3)
    res, nbmaps, nbsizes = O.build_kmap(c, c, O.get_kernel_offsets(3, 1, 1))
    km = B.build_kmap(T(c), T(c), get_kernel_offsets(3, 1, 1, device=DEV), want_inverse=True)
    same(km["nbr"], res.astype(np.int32))
    same(km["nbmaps"][:int(km["nboffs"][-1])], nbmaps.astype(np.int32))
    same(km["nbsizes"], nbsizes.astype(np.int32))
    # symmetry of a submanifold k3 map: pair (i, o, k) <-> (o, i, 26 - k); centre = identity
    nbr = km["nbr"].cpu().numpy()
    assert np.array_equal(nbr[13], np.arange(len(c)))
    assert np.array_equal(km["nbr_t"].cpu().numpy(), nbr[::-1])
    down = O.spdownsample(c, 2, 2, 1)
    same(F.spdownsample(T(c), 2, 2, 1), down)


def test_kmap_tables_are_cleared_by_the_build(B, F):
    """ts_build_kmap clears its hash table, the inverse tables (nbr_t, pos_in: -1 where an input row has no partner) and its
    block counter with one fill launch (ts_fill_segments): recycled, garbage-filled outputs of odd sizes - 1, 3, 5, 257
    rows, so that the fill's unaligned head / tail paths run - must come back exactly as the oracle's tables; an empty
    output set gives an empty rulebook and all-(-1) inverse tables."""
    offs = O.get_kernel_offsets(3, 1, 1)
    rs = np.random.RandomState(3)
    for n in (1, 3, 5, 257, 1000):
        c = np.unique(rs.randint(0, 7 if n < 300 else 12, (4 * n, 3)), axis=0)[:n].astype(np.int32)
        c = np.concatenate([c, np.zeros((len(c), 1), np.int32)], 1)
        out = c[rs.permutation(len(c))[: max(1, len(c) // 2)]]            # a subset in another order: some inputs have no partner
        junk = torch.full((64 * 1024,), 0x5A5A5A5A, dtype=torch.int32, device=DEV)     # dirty the caching allocator's blocks
        del junk
        km = B.build_kmap(T(c), T(out), T(offs.astype(np.int32)), want_inverse=True)
        res, ref_maps, ref_sizes = O.build_kmap(c, out, offs)
        sizes = km["nbsizes"].cpu().numpy()
        assert (sizes == ref_sizes).all()
        total = int(sizes.sum())
        assert (km["nbr"].cpu().numpy() == res).all()
        assert (km["nbmaps"][:total].cpu().numpy() == ref_maps).all()
        nbr_t = np.full((len(offs), len(c)), -1, np.int32)
        pos_in = np.full((len(offs), len(c)), -1, np.int32)
        o = np.concatenate([[0], np.cumsum(ref_sizes)])
        for k in range(len(offs)):
            seg = ref_maps[o[k]:o[k + 1]]
            nbr_t[k, seg[:, 0]] = seg[:, 1]
            pos_in[k, seg[:, 0]] = np.arange(o[k], o[k + 1])
        assert (km["nbr_t"].cpu().numpy() == nbr_t).all()
        assert (km["pos_in"].cpu().numpy() == pos_in).all()
    km = B.build_kmap(T(c), T(np.zeros((0, 4), np.int32)), T(offs.astype(np.int32)), want_inverse=True)
    assert int(km["nbsizes"].sum()) == 0 and int(km["nboffs"].abs().sum()) == 0
    assert (km["nbr_t"] == -1).all() and (km["pos_in"] == -1).all()


def test_downsample_negative_and_range(B, F):
    c = _blob(4, n=3000, extent=30)
    c[:, :3] -= 15                       # negative coordinates: trunc toward zero, like the reference
    same(F.spdownsample(T(c), 2, 2, 1), O.spdownsample(c, 2, 2, 1))
    bad = c.copy()
    bad[0, 0] = 1 << 20
    with pytest.raises(ValueError):
        F.spdownsample(T(bad), 2, 2, 1)


def test_unique_i64(B):
    rs = np.random.RandomState(5)
    keys = rs.randint(0, 1 << 59, size=4000).astype(np.int64)
    keys = np.concatenate([keys, keys[:1500]])
    u, inv = B.unique_i64(T(keys))
    wu, winv = np.unique(keys, return_inverse=True)
    same(u, wu)
    same(inv, winv.astype(np.int32))


# --------------------------------------------------------------------------- voxelize / devoxelize / trilinear
def test_initial_voxelize_pieces_golden(B, F, g_ops):
    pc = T(g_ops["iv_points_c"])
    h = F.sphash(torch.floor(pc).int())
    same(h, g_ops["iv_hash"])
    u, inv = B.unique_i64(h)
    same(u, g_ops["iv_sparse_hash"])
    same(inv, g_ops["iv_idx_query"])
    counts = F.spcount(inv, len(u))
    same(counts, g_ops["iv_counts"])
    vc = torch.round(F.spvoxelize(torch.floor(pc), inv, counts)).int()
    same(vc, g_ops["iv_vox_c"])
    pf = T(g_ops["iv_points_f"]).requires_grad_()
    vf = F.spvoxelize(pf, inv, counts)
    close(vf, g_ops["iv_vox_f"], 1e-6)
    vf.backward(T(g_ops["iv_gv"]))
    close(pf.grad, g_ops["iv_gf"], 1e-6)


@pytest.mark.parametrize("s", [1, 4])
def test_trilinear_and_devoxelize_golden(B, F, g_ops, s):
    idx, w = B.trilinear_map(T(g_ops["tri_points"]), T(g_ops[f"tri_s{s}_vox"]), s)
    same(idx, g_ops[f"tri_s{s}_idx"])
    close(w, g_ops[f"tri_s{s}_w"], 1e-6)
    feat = T(g_ops[f"tri_s{s}_feat"]).requires_grad_()
    out = F.spdevoxelize(feat, idx, w)
    close(out, g_ops[f"tri_s{s}_out"], 1e-6)
    out.backward(T(g_ops[f"tri_s{s}_gout"]))
    close(feat.grad, g_ops[f"tri_s{s}_gfeat"], 1e-5)
    # API-parity helper agrees with the fused kernel
    w2 = F.calc_ti_weights(T(g_ops["tri_points"]), idx.t().contiguous(), scale=s).t()
    close(w2, w, 1e-6)


def test_devoxelize_adjoint_property(B, F):
    """<A x, y> == <x, A^T y> for the voxel->point operator (SURVEY.md section 4 property test)."""
    rs = np.random.RandomState(6)
    m, n, c = 700, 1500, 20
    idx = rs.randint(-1, m, size=(n, 8)).astype(np.int32)
    w = rs.rand(n, 8).astype(np.float32)
    x = rs.randn(m, c).astype(np.float32)
    y = rs.randn(n, c).astype(np.float32)
    ax = B.devoxelize_forward_cuda(T(x), T(idx), T(w))
    aty = B.devoxelize_backward_cuda(T(y), T(idx), T(w), m)
    lhs = float((ax.double() * T(y).double()).sum())
    rhs = float((T(x).double() * aty.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))
    close(ax, O.devoxelize_forward(x, idx, w), 1e-5)


@pytest.mark.parametrize("c", [32, 96, 256])
@pytest.mark.parametrize("stride", [1, 4, 16])
def test_devoxelize_backward_runs_equals_atomic_kernel(B, F, c, stride):
    """run-grouped backward (with and without the cell order) == the per-point atomic kernel == the oracle adjoint,
    on real trilinear maps: many points per cell at coarse strides, points on voxel centres at stride 1"""
    from taseg_amd.data.synthetic import synth_scan
    pts, _ = synth_scan(3, n_points=30000)
    pc = np.unique(np.round(pts[:, :3] / 0.05).astype(np.int32), axis=0)
    pc -= pc.min(0, keepdims=True)
    coords = T(np.concatenate([pc, np.zeros((len(pc), 1), np.int32)], 1))
    vox = F.spdownsample(coords, stride, stride, 1) if stride > 1 else coords
    points = coords.float() + (0.0 if stride == 1 else 0.25)
    idx, w = B.trilinear_map(points.contiguous(), vox, stride)
    order = B.devox_order(idx, vox.shape[0])
    assert sorted(order.tolist()) == list(range(len(pc)))                 # a permutation
    first = (idx >= 0).int().argmax(1)                                   # first present corner identifies the cell
    key = torch.where((idx >= 0).any(1), idx.gather(1, first[:, None])[:, 0].long() * 8 + first, torch.tensor(1 << 40, device=DEV))
    assert bool((key[order.long()][1:] >= key[order.long()][:-1]).all())  # equal tuples are adjacent (sorted by cell key)
    g = T(np.random.RandomState(c + stride).randn(len(pc), c).astype(np.float32))
    want = B.devoxelize_backward_cuda(g, idx, w, vox.shape[0])
    close(B.devoxelize_backward_runs(g, idx, w, vox.shape[0], order), want, 2e-5)
    close(B.devoxelize_backward_runs(g, idx, w, vox.shape[0], None), want, 2e-5)
    if c == 32:
        close(want, O.devoxelize_backward(g.cpu().numpy(), idx.cpu().numpy(), w.cpu().numpy(), vox.shape[0]), 2e-5)


# --------------------------------------------------------------------------- convolution
@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("tag", ["a", "b"])
def test_conv_golden(B, F, g_ops, tag, impl):
    from taseg_amd.torchsparse import SparseTensor
    B.set_conv_impl(impl)
    try:
        x = T(g_ops[f"conv_{tag}_x"]).requires_grad_()
        w = T(g_ops[f"conv_{tag}_w"]).requires_grad_()
        y = F.conv3d(SparseTensor(x, T(g_ops["coords"]), 1), w, 3)
        close(y.F, g_ops[f"conv_{tag}_y"])
        y.F.backward(T(g_ops[f"conv_{tag}_gy"]))
        close(x.grad, g_ops[f"conv_{tag}_gx"])
        close(w.grad, g_ops[f"conv_{tag}_gw"])
    finally:
        B.set_conv_impl(0)


def test_conv_strided_transposed_golden(B, F, g_ops):
    from taseg_amd.torchsparse import SparseTensor
    x = T(g_ops["convt_x"]).requires_grad_()
    wd = T(g_ops["convt_wd"]).requires_grad_()
    wu = T(g_ops["convt_wu"]).requires_grad_()
    st = SparseTensor(x, T(g_ops["coords"]), 1)
    st.cmaps[st.stride] = st.coords
    yd = F.conv3d(st, wd, 2, stride=2)
    same(yd.C, g_ops["convt_coords_d"])
    close(yd.F, g_ops["convt_yd"])
    yu = F.conv3d(yd, wu, 2, stride=2, transposed=True)
    close(yu.F, g_ops["convt_yu"])
    yu.F.backward(T(g_ops["convt_gy"]))
    close(x.grad, g_ops["convt_gx"])
    close(wd.grad, g_ops["convt_gwd"])
    close(wu.grad, g_ops["convt_gwu"])


@pytest.mark.parametrize("ci,co", [(4, 32), (32, 32), (64, 64), (96, 128), (128, 256), (384, 256), (48, 24)])
def test_conv_dense_blob_vs_oracle(B, F, ci, co):
    """every kernel template bucket (C_out <= 32 / 64 / 128 / 256, C_in 4 .. 384), many neighbours per voxel"""
    from taseg_amd.torchsparse import SparseTensor
    rs = np.random.RandomState(ci * 1000 + co)
    c = _blob(7, n=3000, extent=20)
    _, nbmaps, nbsizes = O.build_kmap(c, c, O.get_kernel_offsets(3, 1, 1))
    xn = rs.randn(len(c), ci).astype(np.float32)
    wn = (rs.randn(27, ci, co) / np.sqrt(27 * ci)).astype(np.float32)
    gyn = rs.randn(len(c), co).astype(np.float32)
    want_y = O.conv_forward(xn, wn, nbmaps, nbsizes, (len(c), len(c)))
    want_gx, want_gw = O.conv_backward(xn, wn, gyn, nbmaps, nbsizes)
    x, w = T(xn).requires_grad_(), T(wn).requires_grad_()
    y = F.conv3d(SparseTensor(x, T(c), 1), w, 3)
    close(y.F, want_y, 2e-5)
    y.F.backward(T(gyn))
    close(x.grad, want_gx, 2e-5)
    close(w.grad, want_gw, 2e-5)


def test_conv_two_pass_equals_neighbour_table_kernel(B, F):
    """pair GEMM + gather-sum == the output-stationary ts_conv_nbr kernel == the scalar cross-check"""
    from taseg_amd.torchsparse.nn.utils import get_kernel_offsets
    c = _blob(9, n=6000, extent=26)
    rs = np.random.RandomState(9)
    for ci, co in ((32, 96), (96, 32), (256, 128)):
        x, w = T(rs.randn(len(c), ci).astype(np.float32)), T((rs.randn(27, ci, co) / 30).astype(np.float32))
        km = B.build_kmap(T(c), T(c), get_kernel_offsets(3, 1, 1, device=DEV))
        total = int(km["nboffs"][-1])
        z = B.conv_pair_gemm(x, w, km["nbmaps"], km["nboffs"], total, gather_col=0)
        y2 = B.conv_gather_sum(z, km["pos_out"], len(c))
        y1 = B.conv_nbr(x, w, km["nbr"], len(c))
        B.set_conv_impl(1)
        try:
            y0 = B.conv_nbr(x, w, km["nbr"], len(c))
        finally:
            B.set_conv_impl(0)
        close(y2, y0, 2e-5)
        close(y1, y0, 2e-5)
        # deterministic: two runs are bitwise identical
        z_b = B.conv_pair_gemm(x, w, km["nbmaps"], km["nboffs"], total, gather_col=0)
        assert torch.equal(B.conv_gather_sum(z_b, km["pos_out"], len(c)), y2)


@pytest.mark.parametrize("ci,co", [(32, 32), (64, 96), (96, 64), (128, 128), (384, 256), (256, 192)])
def test_conv_fast_path_equals_generic_kernels(B, F, ci, co):
    """the unguarded full-tile kernels (C_in % 32 == 0, C_out % tile == 0) against the guarded generic ones
    (ts_set_conv_impl(2)): pair GEMM bit-identical in both orientations, weight gradient to atomics-order noise"""
    from taseg_amd.torchsparse.nn.utils import get_kernel_offsets
    c = _blob(11, n=5000, extent=24)
    rs = np.random.RandomState(ci + co)
    x = T(rs.randn(len(c), ci).astype(np.float32))
    gy = T(rs.randn(len(c), co).astype(np.float32))
    w = T((rs.randn(27, ci, co) / 30).astype(np.float32))
    km = B.build_kmap(T(c), T(c), get_kernel_offsets(3, 1, 1, device=DEV))
    total = int(km["nboffs"][-1])

    def run():
        z = B.conv_pair_gemm(x, w, km["nbmaps"], km["nboffs"], total, gather_col=0)
        zt = B.conv_pair_gemm(gy, w, km["nbmaps"], km["nboffs"], total, gather_col=1, weight_transposed=True)
        gw = B.conv_wgrad(x, gy, km["nbmaps"], km["nboffs"], 27, col_a=0, max_pairs=total)
        return z, zt, gw

    B.set_conv_impl(5)              # f32 MFMA full-tile kernels (the default, impl 0, runs the split-bf16 ones)
    try:
        fast = run()
    finally:
        B.set_conv_impl(0)
    variants = []
    for impl in (2, 3, 4):          # generic guarded kernels / one workgroup per tile / persistent workgroups
        B.set_conv_impl(impl)
        try:
            variants.append(run())
        finally:
            B.set_conv_impl(0)
    for other in variants:
        assert torch.equal(fast[0], other[0]) and torch.equal(fast[1], other[1])
        close(fast[2], other[2], 1e-5)


@pytest.mark.parametrize("ci,co", [(32, 32), (64, 96), (96, 64), (128, 128), (384, 256), (256, 192), (32, 128)])
@pytest.mark.parametrize("scale", [1.0, 1e-6])
def test_conv_split_bf16_kernels_are_fp32_grade(B, F, ci, co, scale):
    """the default full-tile kernels evaluate fp32 products as six bf16 x bf16 MFMAs on the exact three-way split of
    both operands (csrc/conv_pairs_s.hip).  Against a float64 evaluation of the same pairs their error must be of the
    size of the f32-MFMA kernels' own (ts_set_conv_impl(5)) - fp32 grade, nowhere near bf16 (4e-3) - also for tiny
    operands (gradients late in training: 1e-6)."""
    from taseg_amd.torchsparse.nn.utils import get_kernel_offsets
    c = _blob(13, n=4000, extent=22)
    rs = np.random.RandomState(ci * 7 + co)
    xn = (rs.randn(len(c), ci) * scale).astype(np.float32)
    gyn = (rs.randn(len(c), co) * scale).astype(np.float32)
    wn = (rs.randn(27, ci, co) / 30).astype(np.float32)
    x, gy, w = T(xn), T(gyn), T(wn)
    km = B.build_kmap(T(c), T(c), get_kernel_offsets(3, 1, 1, device=DEV))
    total = int(km["nboffs"][-1])
    nb = km["nbmaps"][:total].cpu().numpy()
    offs = km["nboffs"].cpu().numpy()
    w64, x64, gy64 = wn.astype(np.float64), xn.astype(np.float64), gyn.astype(np.float64)
    z_ref, zt_ref, gw_ref = np.zeros((total, co)), np.zeros((total, ci)), np.zeros((27, ci, co))
    for k in range(27):
        sl = slice(offs[k], offs[k + 1])
        z_ref[sl] = x64[nb[sl, 0]] @ w64[k]
        zt_ref[sl] = gy64[nb[sl, 1]] @ w64[k].T
        gw_ref[k] = x64[nb[sl, 0]].T @ gy64[nb[sl, 1]]

    def run():
        z = B.conv_pair_gemm(x, w, km["nbmaps"], km["nboffs"], total, gather_col=0)
        zt = B.conv_pair_gemm(gy, w, km["nbmaps"], km["nboffs"], total, gather_col=1, weight_transposed=True)
        gw = B.conv_wgrad(x, gy, km["nbmaps"], km["nboffs"], 27, col_a=0, max_pairs=total)
        return [t.cpu().numpy().astype(np.float64) for t in (z, zt, gw)]

    split = run()
    B.set_conv_impl(5)
    try:
        exact = run()
    finally:
        B.set_conv_impl(0)
    for name, got, ref, want in zip(("z", "zt", "gw"), split, exact, (z_ref, zt_ref, gw_ref)):
        s = np.abs(ref).max()
        assert np.abs(got - ref).max() <= 2e-6 * s, (name, np.abs(got - ref).max() / s)     # ~10 ulp of the scale
        if True:
            e_split, e_mfma = np.abs(got - want).max() / s, np.abs(ref - want).max() / s
            print(f"{name} ci={ci} co={co} scale={scale:g}: max err / scale  split {e_split:.2e}  f32-mfma {e_mfma:.2e}")
            assert e_split <= max(4 * e_mfma, 5e-7), (name, e_split, e_mfma)


@pytest.mark.parametrize("ci,co", [(32, 32), (64, 96), (96, 64), (128, 128), (384, 256), (256, 192), (32, 128)])
def test_conv_f16_path_vs_fp32_on_rounded_operands(B, F, ci, co):
    """half-storage pair GEMM + gather-sum (forward and input gradient) against the fp32 kernels run on the SAME
    half-rounded operands: fp32 accumulation everywhere, so only the half rounding of Z and of the output remains
    (2^-11 relative each, summed over <= 27 terms)"""
    from taseg_amd.torchsparse.nn.utils import get_kernel_offsets
    c = _blob(13, n=5000, extent=24)
    rs = np.random.RandomState(ci * 7 + co)
    x = T(rs.randn(len(c), ci).astype(np.float32)).half()
    gy = T(rs.randn(len(c), co).astype(np.float32)).half()
    w = T((rs.randn(27, ci, co) / np.sqrt(27 * ci)).astype(np.float32))
    km = B.build_kmap(T(c), T(c), get_kernel_offsets(3, 1, 1, device=DEV))
    total = int(km["nboffs"][-1])
    w16, w16t = B.cast_weights_f16(w)
    assert torch.equal(w16, w.half()) and torch.equal(w16t, w.half().transpose(1, 2).contiguous())
    # forward
    z = B.conv_pair_gemm_f16(x, w16t, km["nbmaps"], km["nboffs"], total, gather_col=0)
    y = B.conv_gather_sum_f16(z, km["pos_out"], len(c))
    z32 = B.conv_pair_gemm(x.float(), w16.float(), km["nbmaps"], km["nboffs"], total, gather_col=0)
    y32 = B.conv_gather_sum(z32, km["pos_out"], len(c))
    assert y.dtype == torch.float16
    close(z.float(), z32, 1e-3)
    close(y.float(), y32, 3e-3)
    # input gradient: gather output rows, W_k^T
    zt = B.conv_pair_gemm_f16(gy, w16, km["nbmaps"], km["nboffs"], total, gather_col=1)
    gx = B.conv_gather_sum_f16(zt, km["pos_in"], len(c))
    zt32 = B.conv_pair_gemm(gy.float(), w16.float(), km["nbmaps"], km["nboffs"], total, gather_col=1, weight_transposed=True)
    gx32 = B.conv_gather_sum(zt32, km["pos_in"], len(c))
    close(zt.float(), zt32, 1e-3)
    close(gx.float(), gx32, 3e-3)
    # deterministic
    assert torch.equal(B.conv_pair_gemm_f16(x, w16t, km["nbmaps"], km["nboffs"], total, gather_col=0), z)
    # weight gradient: half rows in, fp32 out (transposing LDS reads feed the MFMA)
    gw = B.conv_wgrad_f16(x, gy, km["nbmaps"], km["nboffs"], 27, col_a=0, max_pairs=total)
    gw32 = B.conv_wgrad(x.float(), gy.float(), km["nbmaps"], km["nboffs"], 27, col_a=0, max_pairs=total)
    assert gw.dtype == torch.float32
    close(gw, gw32, 2e-5)            # identical products, fp32 accumulation in a different order


def test_conv_reference_form_entry_points(B, g_ops):
    """the ten-function lower boundary: explicit nbmaps + host nbsizes, plain and transposed"""
    c = g_ops["coords"]
    x, w, gy = g_ops["conv_b_x"], g_ops["conv_b_w"], g_ops["conv_b_gy"]
    nbmaps, nbsizes = g_ops["k3_s1_nbmaps"], g_ops["k3_s1_nbsizes"]
    out = torch.zeros(len(c), w.shape[2], device=DEV)
    B.convolution_forward_cuda(T(x), out, T(w), T(nbmaps), torch.from_numpy(nbsizes), False)
    close(out, g_ops["conv_b_y"])
    gin, gw = torch.zeros(x.shape, device=DEV), torch.zeros(w.shape, device=DEV)
    B.convolution_backward_cuda(T(x), gin, T(gy), T(w), gw, T(nbmaps), torch.from_numpy(nbsizes), False)
    close(gin, g_ops["conv_b_gx"])
    close(gw, g_ops["conv_b_gw"])
    # transposed through the strided map
    nb2, ns2 = g_ops["k2_s1_nbmaps"], g_ops["k2_s1_nbsizes"]
    yu = torch.zeros(len(c), g_ops["convt_wu"].shape[2], device=DEV)
    B.convolution_forward_cuda(T(g_ops["convt_yd"]), yu, T(g_ops["convt_wu"]), T(nb2), torch.from_numpy(ns2), True)
    close(yu, g_ops["convt_yu"])
    with pytest.raises(ValueError):
        B.convolution_forward_cuda(T(x[:, :3].copy()), out, T(w), T(nbmaps), torch.from_numpy(nbsizes), False)


def test_conv_empty_and_ragged(B, F):
    from taseg_amd.torchsparse import SparseTensor
    # a single voxel, and a row count that is not a multiple of any tile size
    for n in (1, 129, 257):
        c = _blob(8, n=4 * n, extent=12)[:n]
        _, nbmaps, nbsizes = O.build_kmap(c, c, O.get_kernel_offsets(3, 1, 1))
        rs = np.random.RandomState(n)
        xn, wn = rs.randn(n, 16).astype(np.float32), rs.randn(27, 16, 48).astype(np.float32)
        y = F.conv3d(SparseTensor(T(xn), T(c), 1), T(wn), 3)
        close(y.F, O.conv_forward(xn, wn, nbmaps, nbsizes, (n, n)), 2e-5)
    empty = B.conv_nbr(torch.zeros((0, 16), device=DEV), torch.zeros((27, 16, 32), device=DEV),
                       torch.zeros((27, 0), dtype=torch.int32, device=DEV), 0)
    assert empty.shape == (0, 32)


def test_cpu_tensors_are_rejected(B):
    with pytest.raises(RuntimeError):
        B.hash_cuda(torch.zeros((4, 4), dtype=torch.int32))


# --------------------------------------------------------------------------- data stage kernels
def test_fuse_voxelise_quantize_golden(B, g_multiscan):
    g = g_multiscan
    Tn = int(g["T"])
    for b in range(2):
        pose0 = T(g[f"b{b}_pose_t{Tn}"])
        fused = [B.fuse_scan(T(g[f"b{b}_points_t{t}"]), pose0, T(g[f"b{b}_pose_t{t}"])) for t in range(Tn)]
        same(torch.cat(fused), g[f"b{b}_fused_all"])                       # bit-exact float32
    # voxel coords + quantize on the fused cloud of scan 0 against the oracle
    ms = g["b0_raw_data_ms"]
    want_c = O.voxel_coords(ms, 0.05)
    want_c = want_c - want_c.min(0)
    c4, mins = B.voxel_coords(T(ms), 0.05)
    same(c4[:, :3], want_c)
    widx, winv = O.sparse_quantize(want_c)
    idx, inv = B.sparse_quantize(c4)
    same(idx, widx.astype(np.int32))
    same(inv, winv.astype(np.int32))


def test_device_multiscan_stage_golden(g_multiscan):
    """fuse + class-step filter + time flag + double voxelisation + collate on device == the reference's
    numpy dataset code (semantickitti_ms.py / semantickitti_voxel_ms.py), bit for bit, bs = 2"""
    from taseg_amd.data.stage import build_multiscan_batch
    g = g_multiscan
    Tn = int(g["T"])
    lm = g["learning_map"]
    scans = []
    for b in range(2):
        scans.append({"points": [T(g[f"b{b}_points_t{t}"]) for t in range(Tn + 1)],
                      "labels": [T(lm[g[f"b{b}_rawlabels_t{t}"]]) for t in range(Tn + 1)],
                      "poses": [T(g[f"b{b}_pose_t{t}"]) for t in range(Tn + 1)], "name": str(b)})
    batch = build_multiscan_batch(scans, 0.05, g["steps"].tolist())
    for key in ("lidar", "lidar_ms", "inverse_map", "inverse_map_ms", "targets", "targets_ms"):
        same(batch[key].C, g[f"batch_{key}_C"])
        same(batch[key].F.to(torch.from_numpy(g[f"batch_{key}_F"]).dtype), g[f"batch_{key}_F"])
    for key in ("num_points", "num_points_ms", "offset", "offset_ms", "point_mask"):
        same(batch[key].to(torch.from_numpy(g[f"batch_{key}"]).dtype), g[f"batch_{key}"])


def _same_batches(a, b):
    from taseg_amd.torchsparse import SparseTensor
    assert list(a) == list(b) or set(a) == set(b)
    for key, v in a.items():
        w = b[key]
        if isinstance(v, SparseTensor):
            assert v.C.dtype == w.C.dtype and v.F.dtype == w.F.dtype and torch.equal(v.C, w.C) and torch.equal(v.F, w.F), key
        elif isinstance(v, torch.Tensor):
            assert v.dtype == w.dtype and v.shape == w.shape and v.device == w.device and torch.equal(v, w), key
        else:
            assert v == w, key


def test_batched_multiscan_stage_equals_the_per_sample_stage(g_multiscan):
    """stage.build_multiscan_batch (ONE chain of launches for the batch: one pose fuse, one compaction, one batch-keyed
    voxelisation per cloud kind) against the per-sample form it replaced (semantickitti_voxel_ms.py:121-212 sample by sample +
    collate): every tensor of the batch_dict bit for bit, dtypes and devices included - on the golden scans, on a batch whose
    second sample has NO history, with unequal history lengths, and with pseudo classes that are never aggregated (-1)"""
    from taseg_amd.data import stage as S
    g = g_multiscan
    Tn = int(g["T"])
    lm = g["learning_map"]
    steps = g["steps"].tolist()

    def scan(b, keep_hist=None, cut=None, pseudo=False):
        ts = list(range(Tn + 1)) if keep_hist is None else keep_hist + [Tn]
        pts = [T(g[f"b{b}_points_t{t}"]) for t in ts]
        lab = [T(lm[g[f"b{b}_rawlabels_t{t}"]]) for t in ts]
        if cut:
            pts[0], lab[0] = pts[0][:cut].contiguous(), lab[0][:cut].contiguous()
        d = {"points": pts, "labels": lab, "poses": [T(g[f"b{b}_pose_t{t}"]) for t in ts], "name": f"s{b}",
             "deltas": [t - Tn for t in ts[:-1]]}
        if pseudo:
            ps = [l.long().clone() for l in lab[:-1]]
            for p in ps:
                p[::7] = -1
            d["pseudo"] = ps
        return d

    cases = [[scan(0), scan(1)], [scan(0), scan(1, keep_hist=[])], [scan(1, keep_hist=[1, 3], cut=1000), scan(0), scan(0, keep_hist=[2])],
             [scan(0, pseudo=True), scan(1)]]
    for scans in cases:
        _same_batches(S.build_multiscan_batch(scans, 0.05, steps), S.build_multiscan_batch_per_sample(scans, 0.05, steps))


@pytest.mark.parametrize("n,c", [(5000, 32), (20011, 96), (777, 256), (3, 16), (12001, 128), (13000, 256), (30011, 64)])
def test_batchnorm_train_matches_torch(n, c):
    """our reductions + torch's elementwise halves == nn.BatchNorm1d (training): output, grads, running stats"""
    from taseg_amd.torchsparse import SparseTensor
    from taseg_amd.torchsparse import nn as spnn
    torch.manual_seed(n + c)
    x = (torch.randn(n, c, device=DEV) * 2 + 0.5)
    ref = torch.nn.BatchNorm1d(c).to(DEV).train()
    ours = spnn.BatchNorm(c).to(DEV).train()
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5)
        ref.bias.uniform_(-0.5, 0.5)
        ours.weight.copy_(ref.weight)
        ours.bias.copy_(ref.bias)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya = ref(xa)
    yb = ours(SparseTensor(xb, torch.zeros(n, 4, dtype=torch.int32, device=DEV))).F
    close(yb, ya, 2e-5)
    gy = torch.randn_like(ya)
    ya.backward(gy)
    yb.backward(gy)
    close(xb.grad, xa.grad, 5e-5)
    close(ours.weight.grad, ref.weight.grad, 5e-5)
    close(ours.bias.grad, ref.bias.grad, 5e-5)
    close(ours.running_mean, ref.running_mean, 1e-5)
    close(ours.running_var, ref.running_var, 1e-5)
    assert int(ours.num_batches_tracked) == 1
    # eval mode goes through the stock module
    ours.eval(), ref.eval()
    close(ours(SparseTensor(x, torch.zeros(n, 4, dtype=torch.int32, device=DEV))).F, ref(x), 1e-5)


@pytest.mark.parametrize("with_res,relu", [(False, True), (True, True), (True, False)])
def test_fused_bn_act_matches_torch(with_res, relu):
    """relu(BN(x) + residual) fused (2 passes fwd, 2 passes bwd) == the chained torch modules"""
    from taseg_amd.torchsparse import SparseTensor
    from taseg_amd.torchsparse import nn as spnn
    torch.manual_seed(3)
    n, c = 9001, 64
    co = torch.zeros(n, 4, dtype=torch.int32, device=DEV)
    x = torch.randn(n, c, device=DEV) * 1.5 + 0.3
    r = torch.randn(n, c, device=DEV)
    ref = torch.nn.BatchNorm1d(c).to(DEV).train()
    ours = spnn.BatchNorm(c).to(DEV).train()
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5)
        ref.bias.uniform_(-0.5, 0.5)
        ours.weight.copy_(ref.weight)
        ours.bias.copy_(ref.bias)
    xa, xb, ra, rb = (t.clone().requires_grad_() for t in (x, x, r, r))
    ya = ref(xa)
    if with_res:
        ya = ya + ra
    if relu:
        ya = torch.relu(ya)
    yb = spnn.bn_act(ours, SparseTensor(xb, co), relu=relu, residual=SparseTensor(rb, co) if with_res else None).F
    close(yb, ya, 2e-5)
    gy = torch.randn_like(ya)
    ya.backward(gy)
    yb.backward(gy)
    close(xb.grad, xa.grad, 5e-5)
    if with_res:
        close(rb.grad, ra.grad, 1e-6)
    close(ours.weight.grad, ref.weight.grad, 5e-5)
    close(ours.bias.grad, ref.bias.grad, 5e-5)
    close(ours.running_var, ref.running_var, 1e-5)


def test_new_entry_points_accept_empty_inputs(B):
    """zero points / pairs / voxels must return empty (or zero) results, not launch over bad grids"""
    dev = DEV
    e4 = torch.empty((0, 4), dtype=torch.float32, device=dev)
    assert B.fuse_scans(e4, torch.empty(0, dtype=torch.int32, device=dev), torch.eye(4, device=dev),
                        torch.eye(4, device=dev)[None]).shape == (0, 4)
    idx = torch.empty((0, 8), dtype=torch.int32, device=dev)
    w = torch.empty((0, 8), dtype=torch.float32, device=dev)
    assert B.devox_order(idx, 10).shape == (0,)
    g = B.devoxelize_backward_runs(torch.empty((0, 32), device=dev), idx, w, 10, None)
    assert g.shape == (10, 32) and float(g.abs().sum()) == 0.0
    feat = torch.randn(2, 8, 4, 6, device=dev)
    plan = B.image_plan(torch.empty((0, 2), device=dev), torch.empty(0, dtype=torch.int32, device=dev),
                        torch.tensor([2], dtype=torch.int32, device=dev), 2, 4, 6)
    out = B.image_gather_forward(feat, plan)
    assert out.shape == (0, 8) and int(plan["err"]) == 0
    gf = B.image_gather_backward(torch.empty((0, 8), device=dev), plan, 8)
    assert gf.shape == (2, 8, 4, 6) and float(gf.abs().sum()) == 0.0
    # no pairs at all: an isolated voxel convolved with a kernel whose centre is the only hit still works; a rulebook
    # with zero pairs yields zeros
    nb = torch.empty((1, 2), dtype=torch.int32, device=dev)
    offs = torch.zeros(28, dtype=torch.int32, device=dev)
    z = B.conv_pair_gemm_f16(torch.randn(5, 32, device=dev).half(), torch.randn(27, 32, 32, device=dev).half(), nb, offs, 0, 0)
    assert z.shape == (0, 32)
    gw = B.conv_wgrad_f16(torch.randn(5, 32, device=dev).half(), torch.randn(5, 32, device=dev).half(), nb, offs, 27, 0, 0)
    assert gw.shape == (27, 32, 32) and float(gw.abs().sum()) == 0.0
    pos = torch.full((27, 5), -1, dtype=torch.int32, device=dev)
    y = B.conv_gather_sum_f16(z, pos, 5)
    assert y.shape == (5, 32) and float(y.float().abs().sum()) == 0.0
    # round-2 entry points: deterministic fp32 weight gradient with no pairs, nuScenes sweep fuse / TIAF projection with no points
    gw32 = B.conv_wgrad(torch.randn(5, 32, device=dev), torch.randn(5, 32, device=dev), nb, offs, 27, 0, 0)
    assert gw32.shape == (27, 32, 32) and float(gw32.abs().sum()) == 0.0
    fused, keep = B.fuse_sweeps(torch.empty((0, 5), device=dev), torch.empty(0, dtype=torch.int32, device=dev),
                                torch.zeros((1, 28), dtype=torch.float64, device=dev))
    assert fused.shape == (0, 5) and keep.shape == (0,)
    pix, keep = B.project_fov(e4, torch.eye(4, dtype=torch.float64, device=dev)[:3].contiguous(), (10, 10), (8, 8))
    assert pix.shape == (0, 2) and keep.shape == (0,)
    # cell-reduced devoxelize backward and the weight planes with nothing to do
    plan = B.devox_cells(idx, w, 10)
    assert plan[0] == "cells" and plan[2].tolist() == [0]
    g = B.devoxelize_backward_from(torch.empty((0, 32), device=dev), 0, 32, idx, w, 10, plan)
    assert g.shape == (10, 32) and float(g.abs().sum()) == 0.0
    from taseg_amd import _lib as L
    L.check(L.load().ts_conv_split_planes_batch(None, 0, L.stream()), "ts_conv_split_planes_batch")


# --------------------------------------------------------------------------- dense products on the pair-GEMM kernels
@pytest.mark.parametrize("ci,co", [(128, 96), (384, 256), (64, 64), (48, 24)])
@pytest.mark.parametrize("half", [False, True])
def test_pointwise_conv_on_pair_gemm_kernels(B, F, ci, co, half):
    """1x1x1 convolution (conv.py:135-140, `feats.matmul(weight)`) through the identity rulebook: forward, input and
    weight gradient against torch in fp64; (48, 24) takes the library fallback for forward / input gradient."""
    from taseg_amd.torchsparse import SparseTensor
    rs = np.random.RandomState(ci + co)
    n = 20011
    c = T(np.concatenate([rs.randint(0, 200, size=(n, 3)), np.zeros((n, 1))], 1).astype(np.int32))
    xn, wn, gn = rs.randn(n, ci), rs.randn(ci, co) / np.sqrt(ci), rs.randn(n, co)
    dt = torch.float16 if half else torch.float32
    x = T(xn.astype(np.float32)).to(dt).requires_grad_()
    w = T(wn.astype(np.float32)).requires_grad_()
    y = F.conv3d(SparseTensor(x, c, 1), w, 1).F
    assert y.dtype == dt
    y.backward(T(gn.astype(np.float32)).to(dt))
    x64, w64, g64 = x.detach().double().cpu().numpy(), wn, T(gn.astype(np.float32)).to(dt).double().cpu().numpy()
    tol = 4e-3 if half else 2e-5
    close(y.float(), x64 @ w64, tol)
    close(x.grad.float(), g64 @ w64.T, tol)
    close(w.grad, x64.T @ g64, tol)


@pytest.mark.parametrize("half", [False, True])
def test_point_linear_head_on_pair_gemm_kernels(B, F, half):
    """the 480 -> 20 class head (minkunet.py:334-336) padded to 32 output columns on the pair GEMM / weight-gradient
    kernels: y, dx, dW, db against fp64; under autocast the head runs in half storage like nn.Linear would"""
    rs = np.random.RandomState(5)
    n, c, o = 30011, 480, 20
    xn, wn, bn_, gn = rs.randn(n, c), rs.randn(o, c) / np.sqrt(c), rs.randn(o), rs.randn(n, o)
    x = T(xn.astype(np.float32)).requires_grad_()
    w, b = T(wn.astype(np.float32)).requires_grad_(), T(bn_.astype(np.float32)).requires_grad_()
    with torch.autocast("cuda", dtype=torch.float16, enabled=half):
        y = F.point_linear(x, w, b)
    assert y.dtype == (torch.float16 if half else torch.float32) and y.shape == (n, o) and y.is_contiguous()
    g_used = T(gn.astype(np.float32)).to(y.dtype)
    y.backward(g_used)
    g64 = g_used.double().cpu().numpy()
    tol = 4e-3 if half else 2e-5
    close(y.float(), xn @ wn.T + bn_, tol)
    close(x.grad, g64 @ wn, tol)
    close(w.grad, g64.T @ xn, tol)
    close(b.grad, g64.sum(0), 1e-4)


@pytest.mark.parametrize("ci,co", [(32, 32), (64, 96), (96, 64), (128, 128), (384, 256), (32, 128)])
def test_conv_f16_natural_weight_layout_equals_row_layout(B, F, ci, co):
    """the half pair GEMM reading the weight in place ([K, C_in, C_out], fragments through the transposing LDS load)
    against the same kernel on the transposed copy [K, C_out, C_in]: same MFMA stream, bit-identical Z"""
    from taseg_amd.torchsparse.nn.utils import get_kernel_offsets
    c = _blob(17, n=4000, extent=22)
    rs = np.random.RandomState(ci * 3 + co)
    x = T(rs.randn(len(c), ci).astype(np.float32)).half()
    w = T((rs.randn(27, ci, co) / 30).astype(np.float32))
    km = B.build_kmap(T(c), T(c), get_kernel_offsets(3, 1, 1, device=DEV))
    total = int(km["nboffs"][-1])
    w16, w16t = B.cast_weights_f16(w)
    z_rows = B.conv_pair_gemm_f16(x, w16t, km["nbmaps"], km["nboffs"], total, gather_col=0)
    z_nat = B.conv_pair_gemm_f16(x, w16, km["nbmaps"], km["nboffs"], total, gather_col=0, natural=True)
    assert torch.equal(z_rows, z_nat)


@pytest.mark.parametrize("stride", [1, 4, 16])
@pytest.mark.parametrize("c", [32, 96, 256])
def test_devoxelize_backward_along_inverse_map(B, F, c, stride):
    """ts_devox_csr + ts_devoxelize_backward_csr (gather per voxel, no atomics) against the atomic kernel and the oracle
    on real trilinear maps; the inverse map lists exactly the live (point, corner) slots, sorted, and two runs are
    bitwise identical"""
    from taseg_amd.data.synthetic import synth_scan
    pts, _ = synth_scan(5, n_points=30000)
    pc = np.unique(np.round(pts[:, :3] / 0.05).astype(np.int32), axis=0)
    pc -= pc.min(0, keepdims=True)
    coords = T(np.concatenate([pc, np.zeros((len(pc), 1), np.int32)], 1))
    vox = F.spdownsample(coords, stride, stride, 1) if stride > 1 else coords
    points = coords.float() + (0.0 if stride == 1 else 0.3)
    idx, w = B.trilinear_map(points.contiguous(), vox, stride)
    off, ent = B.devox_csr(idx, w, vox.shape[0])
    live = ((idx >= 0) & (w != 0)).cpu().numpy()
    assert int(off[-1]) == int(live.sum()) and int(off[0]) == 0
    offn, entn = off.cpu().numpy(), ent.cpu().numpy()[:int(off[-1])]
    assert np.all(np.diff(offn) >= 0)
    vox_of_slot = idx.cpu().numpy().reshape(-1)[entn]
    assert np.array_equal(vox_of_slot, np.repeat(np.arange(vox.shape[0]), np.diff(offn)))      # grouped by voxel
    assert np.all(live.reshape(-1)[entn])
    within = np.diff(entn) > 0
    assert np.all(within | (np.diff(vox_of_slot) > 0))                                         # ascending slots per voxel
    g = T(np.random.RandomState(c + stride).randn(len(pc), c).astype(np.float32))
    want = B.devoxelize_backward_cuda(g, idx, w, vox.shape[0])
    got = B.devoxelize_backward_csr(g, w, (off, ent), vox.shape[0])
    close(got, want, 2e-5)
    assert torch.equal(got, B.devoxelize_backward_csr(g, w, (off, ent), vox.shape[0]))
    if c == 32:
        close(got, O.devoxelize_backward(g.cpu().numpy(), idx.cpu().numpy(), w.cpu().numpy(), vox.shape[0]), 2e-5)


def test_devoxelize_cat_equals_cat_of_devoxelize(B, F):
    """spdevoxelize_cat (three interpolations into column blocks of one matrix, gradient blocks read in place) against
    torch.cat of three spdevoxelize calls: same values, same feature gradients (walk order / inverse map / plain)"""
    from taseg_amd.data.synthetic import synth_scan
    pts, _ = synth_scan(9, n_points=20000)
    pc = np.unique(np.round(pts[:, :3] / 0.05).astype(np.int32), axis=0)
    pc -= pc.min(0, keepdims=True)
    coords = T(np.concatenate([pc, np.zeros((len(pc), 1), np.int32)], 1))
    rs = np.random.RandomState(3)
    maps, feats = [], []
    for stride, c, kind in ((16, 64, "order"), (4, 32, "csr"), (1, 96, None)):
        vox = F.spdownsample(coords, stride, stride, 1) if stride > 1 else coords
        points = coords.float() + (0.0 if stride == 1 else 0.3)
        idx, w = B.trilinear_map(points.contiguous(), vox, stride)
        order = B.devox_order(idx, vox.shape[0]) if kind == "order" else B.devox_csr(idx, w, vox.shape[0]) if kind == "csr" else None
        maps.append((idx, w, order))
        feats.append(T(rs.randn(vox.shape[0], c).astype(np.float32)))
    gy = T(rs.randn(len(pc), 64 + 32 + 96).astype(np.float32))
    a = [f.clone().requires_grad_() for f in feats]
    ya = F.spdevoxelize_cat(a, maps)
    ya.backward(gy)
    b = [f.clone().requires_grad_() for f in feats]
    yb = torch.cat([F.spdevoxelize(f, i, w, o) for f, (i, w, o) in zip(b, maps)], dim=1)
    yb.backward(gy)
    assert torch.equal(ya, yb)
    for fa, fb in zip(a, b):
        close(fa.grad, fb.grad, 2e-5)


def test_weight_gradient_is_deterministic_and_matches_atomic_form():
    """ts_conv_wgrad_det (partial tiles + ordered sum; what every training path uses) gives identical bits run to run,
    agrees with the float64 product and with the atomic form ts_conv_wgrad to fp32 rounding - fp32 and half operands,
    full-tile and ragged channel counts, an empty offset in the rulebook"""
    from taseg_amd import _lib as L
    from taseg_amd import backend as B
    from taseg_amd.data.synthetic import synth_scan
    from taseg_amd.torchsparse import SparseTensor
    from taseg_amd.torchsparse.nn import functional as spF
    pts, _ = synth_scan(9, n_points=30000, n_beams=32, n_az=1000)
    pc = np.round(pts[:, :3] / 0.05).astype(np.int32)
    pc -= pc.min(0)
    idx, _ = O.sparse_quantize(pc)
    coords = torch.from_numpy(np.concatenate([pc[idx], np.zeros((len(idx), 1), np.int32)], 1)).cuda()
    x = SparseTensor(None, coords, 1)
    spF.build_pyramid(x, 2)
    lib = L.load()
    g = torch.Generator().manual_seed(0)
    for stride, ca, cb in ((1, 96, 96), (2, 64, 128), (1, 20, 36), (4, 256, 256)):
        km = x.kmaps[((stride,) * 3, (3, 3, 3), (1, 1, 1), (1, 1, 1))]
        n, P = km.sizes[0], km.total
        a = torch.randn(n, ca, generator=g).cuda()
        b = torch.randn(n, cb, generator=g).cuda()
        runs = [B.conv_wgrad(a, b, km.nbmaps_buf, km.nboffs, 27, col_a=0, max_pairs=P) for _ in range(3)]
        assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
        atomic = torch.empty_like(runs[0])
        L.check(lib.ts_conv_wgrad(L.ptr(a), ca, L.ptr(b), cb, L.ptr(km.nbmaps_buf), L.ptr(km.nboffs), 27, 0, P,
                                  L.ptr(atomic), L.stream()), "ts_conv_wgrad")
        nb = km.nbmaps[:, :].cpu().numpy()
        sizes = km.nbsizes.cpu().numpy()
        want = np.zeros((27, ca, cb))
        a64, b64 = a.double().cpu().numpy(), b.double().cpu().numpy()
        p0 = 0
        for k in range(27):
            sl = nb[p0:p0 + sizes[k]]
            want[k] = a64[sl[:, 0]].T @ b64[sl[:, 1]]
            p0 += sizes[k]
        scale = np.abs(want).max()
        assert np.abs(runs[0].cpu().numpy() - want).max() <= 2e-5 * scale
        assert np.abs(atomic.cpu().numpy() - want).max() <= 2e-5 * scale
        if ca % 32 == 0 and cb % 32 == 0:
            ah, bh = a.half(), b.half()
            hr = [B.conv_wgrad_f16(ah, bh, km.nbmaps_buf, km.nboffs, 27, col_a=0, max_pairs=P) for _ in range(2)]
            assert torch.equal(hr[0], hr[1])
            wanth = np.zeros_like(want)
            ah64, bh64 = ah.double().cpu().numpy(), bh.double().cpu().numpy()
            p0 = 0
            for k in range(27):
                sl = nb[p0:p0 + sizes[k]]
                wanth[k] = ah64[sl[:, 0]].T @ bh64[sl[:, 1]]
                p0 += sizes[k]
            assert np.abs(hr[0].cpu().numpy() - wanth).max() <= 2e-5 * scale


def test_conv_dense_rulebook_reference_golden(B, F):
    """the reference's own convolution results on a DENSE rulebook (6.3 pairs per voxel: tests/golden/ops_dense.npz, a 30
    degree sector of a full-resolution scan): rulebook bit-exact; submanifold k3 with full-tile channels (split-bf16 MFMA
    kernels) and ragged ones, strided k2 + transposed mirror - forward, input gradient, weight gradient"""
    import os
    from conftest import GOLDEN, dense_ops_inputs
    from taseg_amd.torchsparse import SparseTensor
    g = dict(np.load(os.path.join(GOLDEN, "ops_dense.npz"), allow_pickle=False))
    coords = T(g["coords"])
    n = coords.shape[0]
    inp = dense_ops_inputs(n)
    assert len(g["k3_nbmaps"]) / n > 5.0
    for tag in ("full", "ragged"):
        x = inp[f"{tag}_x"].to(DEV).requires_grad_()
        w = inp[f"{tag}_w"].to(DEV).requires_grad_()
        y = F.conv3d(SparseTensor(x, coords, 1), w, 3)
        km = y.kmaps[((1, 1, 1), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
        same(km.nbmaps.int(), g["k3_nbmaps"])
        same(km.nbsizes.int(), g["k3_nbsizes"])
        close(y.F[::4], g[f"{tag}_y"])
        y.F.backward(inp[f"{tag}_gy"].to(DEV))
        close(x.grad[::4], g[f"{tag}_gx"])
        close(w.grad, g[f"{tag}_gw"])
    x = inp["t_x"].to(DEV).requires_grad_()
    wd, wu = inp["t_wd"].to(DEV).requires_grad_(), inp["t_wu"].to(DEV).requires_grad_()
    st = SparseTensor(x, coords, 1)
    st.cmaps[st.stride] = st.coords
    yd = F.conv3d(st, wd, 2, stride=2)
    same(yd.C, g["t_coords_d"])
    close(yd.F[::2], g["t_yd"])
    yu = F.conv3d(yd, wu, 2, stride=2, transposed=True)
    close(yu.F[::4], g["t_yu"])
    yu.F.backward(inp["t_gy"].to(DEV))
    close(x.grad[::4], g["t_gx"])
    close(wd.grad, g["t_gwd"])
    close(wu.grad, g["t_gwu"])


@pytest.mark.parametrize("ci,co,impl", [(128, 128, 0), (256, 128, 0), (128, 256, 0), (64, 128, 0), (128, 64, 0),
                                         (96, 96, 11), (192, 96, 11), (384, 256, 0)])
def test_presplit_planes_give_the_same_bits(B, F, ci, co, impl):
    """ts_conv_split_planes + ts_conv_planes_hint: the direct-rows pair GEMM on pre-split weight planes (forward product and
    input gradient on the same planes) returns bit for bit what the kernels that split the weight slice
    themselves return; the hint is one-shot and a hint for another weight or other shapes is ignored."""
    from taseg_amd import _lib as L
    from taseg_amd.torchsparse.nn.utils import get_kernel_offsets
    lib = L.load()
    rs = np.random.RandomState(ci + 7 * co)
    c = _blob(11, n=9000, extent=24)
    km = B.build_kmap(T(c), T(c), get_kernel_offsets(3, 1, 1, device=DEV), want_inverse=True)
    total = int(km["nboffs"][-1])
    x, gy = T(rs.randn(len(c), ci).astype(np.float32)), T(rs.randn(len(c), co).astype(np.float32))
    w = T((rs.randn(27, ci, co) / np.sqrt(27 * ci)).astype(np.float32))
    planes = torch.empty(3 * w.numel(), dtype=torch.int16, device=DEV)
    L.check(lib.ts_conv_split_planes(L.ptr(w), 27, ci, co, L.ptr(planes), L.stream()), "ts_conv_split_planes")
    # the planes themselves: h + m + l == w exactly
    pl = planes.view(torch.bfloat16).view(3, -1).float()
    assert torch.equal((pl[0] + pl[1] + pl[2]).view(27, ci, co), w)
    fwd = lambda: B.conv_pair_gemm(x, w, km["nbmaps"], km["nboffs"], total, 0)
    dgr = lambda: B.conv_pair_gemm(gy, w, km["nbmaps"], km["nboffs"], total, 1, weight_transposed=True)
    want_f, want_d = fwd(), dgr()
    B.set_conv_impl(impl)
    try:
        lib.ts_conv_planes_hint(L.ptr(w), L.ptr(planes), 27, ci, co)
        got_f = fwd()
        lib.ts_conv_planes_hint(L.ptr(w), L.ptr(planes), 27, ci, co)
        got_d = dgr()
        assert torch.equal(got_f, want_f) and torch.equal(got_d, want_d)
        # one-shot: poison the planes, the next call without a hint must not read them
        junk = torch.zeros_like(planes)
        lib.ts_conv_planes_hint(L.ptr(w), L.ptr(junk), 27, ci, co)
        if co % 128 == 0 and co % 96 != 0 or impl == 11:
            assert not torch.equal(fwd(), want_f), "the hinted call did not take the planes"
        else:
            fwd()                                                 # consumed (and ignored) all the same
        assert torch.equal(fwd(), want_f), "a hint outlived its call"
        # hints that do not fit this call are ignored: other weight pointer, other shapes
        w2 = w.clone()
        lib.ts_conv_planes_hint(L.ptr(w2), L.ptr(junk), 27, ci, co)
        assert torch.equal(fwd(), want_f)
        lib.ts_conv_planes_hint(L.ptr(w), L.ptr(junk), 27, co, ci + 32)
        assert torch.equal(fwd(), want_f)
    finally:
        B.set_conv_impl(0)
        lib.ts_conv_planes_hint(None, None, 0, 0, 0)


@pytest.mark.parametrize("stride,c", [(16, 256), (16, 64), (4, 32)])
def test_devoxelize_backward_cell_reduced(B, F, stride, c):
    """backend.devox_cells + ts_devoxelize_backward_cells_ld: the two-stage sum over interpolation cells equals the
    gather along the inverse map (and the oracle's scatter-add) up to summation order, twice gives the same bits, and
    reads a column block of a wider gradient matrix in place."""
    from taseg_amd.data.synthetic import synth_scan
    pts, _ = synth_scan(3, n_points=60000, n_beams=64, n_az=1000)
    pc = np.round(pts[:, :3] / 0.05).astype(np.int32)
    pc -= pc.min(0)
    coords = np.unique(np.concatenate([pc // stride * stride, np.zeros((len(pc), 1), np.int32)], 1), axis=0)
    pcf = T(np.concatenate([pc.astype(np.float32), np.zeros((len(pc), 1), np.float32)], 1))
    idx, w = B.trilinear_map(pcf, T(coords), stride)
    m = len(coords)
    rs = np.random.RandomState(stride + c)
    g = T(rs.randn(len(pc), c + 32).astype(np.float32))
    plan = B.devox_cells(idx, w, m)
    assert plan[0] == "cells" and int(plan[2][-1]) == len(pc)
    got = B.devoxelize_backward_from(g, 32, c, idx, w, m, plan)
    again = B.devoxelize_backward_from(g, 32, c, idx, w, m, plan)
    assert torch.equal(got, again)
    ref = B.devoxelize_backward_from(g, 32, c, idx, w, m, B.devox_csr(idx, w, m))
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-5 * scale
    want = O.spdevoxelize_backward(g[:, 32:].cpu().numpy(), idx.cpu().numpy(), w.cpu().numpy(), m) \
        if hasattr(O, "spdevoxelize_backward") else None
    if want is not None:
        close(got, want, 2e-5)
    # the one-matrix form used by _Devoxelize.backward
    close(B.devoxelize_backward_csr(g[:, 32:].contiguous(), w, plan, m), ref, 2e-5)


@pytest.mark.parametrize("p,c,ignore", [(70001, 20, 0), (4099, 17, 0), (1000, 20, None), (50000, 5, 3), (1200003, 20, 0)])
def test_lovasz_kernels_match_the_tensor_form_and_the_oracle(p, c, ignore):
    """csrc/loss.hip (ts_lovasz_errors -> sort -> ts_lovasz_grad) against taseg_amd.pcseg.loss.lovasz.lovasz_softmax_flat
    (torch ops) and the oracle's per-class loop (the reference's lovasz_losses.py restated): value and gradient, classes that
    do not occur, ignored rows, a tile boundary inside the rows; 1.2 M rows: the byte label table of the large calls (the dense image
    loss of TIAF)."""
    from oracle import model as OM
    from taseg_amd.pcseg.loss import lovasz as LV
    rs = np.random.RandomState(p + c)
    logits = torch.from_numpy(rs.randn(p, c).astype(np.float32) * 2).cuda()
    labels = torch.from_numpy(rs.randint(0, c, size=p)).cuda()
    labels[labels == c - 1] = 1                                   # the last class never occurs
    labels[:17] = 0 if ignore is None else ignore
    a = logits.clone().requires_grad_()
    got = LV.lovasz_softmax(a.softmax(1), labels, ignore=ignore)
    got.backward()
    b = logits.clone().requires_grad_()
    valid = (labels != ignore) if ignore is not None else None
    want = LV.lovasz_softmax_flat(b.softmax(1), labels, valid=valid)
    want.backward()
    assert abs(float(got) - float(want)) <= 2e-6 * max(1.0, abs(float(want)))
    assert float((a.grad - b.grad).abs().max()) <= 1e-7 + 1e-4 * float(b.grad.abs().max())
    ref = OM.lovasz_softmax_ref(logits.cpu().softmax(1), labels.cpu(), ignore=-1 if ignore is None else ignore)
    assert abs(float(got) - float(ref)) <= 5e-6 * max(1.0, abs(float(ref)))
    # run-to-run identical
    a2 = logits.clone().requires_grad_()
    again = LV.lovasz_softmax(a2.softmax(1), labels, ignore=ignore)
    again.backward()
    assert float(again) == float(got) and torch.equal(a2.grad, a.grad)
    # nothing but ignored rows: zero loss, zero gradient
    if ignore is not None:
        z = logits.clone().requires_grad_()
        zero = LV.lovasz_softmax(z.softmax(1), torch.full_like(labels, ignore), ignore=ignore)
        zero.backward()
        assert float(zero) == 0.0 and float(z.grad.abs().max()) == 0.0


@pytest.mark.parametrize("p,c,smoothing,half", [(70001, 20, 0.0, False), (4099, 17, 0.1, False), (30000, 20, 0.05, True), (1100003, 20, 0.0, False)])
def test_fused_ce_lovasz_loss_matches_the_tensor_form_and_the_oracle(monkeypatch, p, c, smoothing, half):
    """pcseg.loss.Losses on the csrc/loss.hip kernels (softmax + CE + Lovasz errors in one pass, one backward kernel) against
    the same module on tensor ops and against the oracle's CE + Lovasz: value and gradient w.r.t. the logits."""
    from oracle import model as OM
    import taseg_amd.pcseg.loss as LS
    rs = np.random.RandomState(p)
    logits = torch.from_numpy(rs.randn(p, c).astype(np.float32) * 2).cuda()
    if half:
        logits = logits.half()
    labels = torch.from_numpy(rs.randint(0, c, size=p)).cuda()
    labels[labels == c - 1] = 1
    crit = LS.Losses(["CELoss", "LovLoss"], [1.0, 0.7], ignore_index=0, label_smoothing=smoothing)
    a = logits.clone().requires_grad_()
    got = crit(a, labels)
    (got * 3.0).backward()
    monkeypatch.setattr(LS, "_FUSED", False)
    b = logits.float().clone().requires_grad_()      # half logits: the kernels compute in fp32 like torch does under autocast
    want = crit(b, labels)
    (want * 3.0).backward()
    tol = 2e-3 if half else 3e-6
    assert abs(float(got) - float(want)) <= tol * max(1.0, abs(float(want)))
    # (a million rows: gradients of 1e-6, and equal errors whose order a stable and an unstable sort settle differently - either order is
    # a valid subgradient, the elements of a tie trade their values)
    floor = 1e-9 if p < 1000000 else 5e-9
    assert float((a.grad.float() - b.grad.float()).abs().max()) <= (2e-3 if half else 1e-4) * float(b.grad.float().abs().max()) + floor
    if not half:
        ref = OM.loss_ce_lovasz(logits.cpu(), labels.cpu(), label_smoothing=smoothing)      # weights 1 / 1
        monkeypatch.setattr(LS, "_FUSED", True)
        one = LS.Losses(["CELoss", "LovLoss"], [1.0, 1.0], ignore_index=0, label_smoothing=smoothing)(logits, labels)
        assert abs(float(one) - float(ref)) <= 5e-6 * max(1.0, abs(float(ref)))


def test_fused_loss_fails_loudly_on_a_label_outside_the_classes():
    """a label that is neither the ignore index nor in [0, C) - e.g. an unmapped raw id 255 - makes torch's CrossEntropyLoss
    (the reference's) assert on the device; the fused kernels answer with a NaN loss instead of a silently wrong term"""
    import taseg_amd.pcseg.loss as LS
    rs = np.random.RandomState(1)
    logits = torch.from_numpy(rs.randn(5000, 20).astype(np.float32)).cuda()
    labels = torch.from_numpy(rs.randint(0, 20, size=5000)).cuda()
    crit = LS.Losses(["CELoss", "LovLoss"], [1.0, 1.0], ignore_index=0, label_smoothing=0.0)
    assert np.isfinite(float(crit(logits, labels)))
    labels[1234] = 255
    a = logits.clone().requires_grad_()
    loss = crit(a, labels)
    assert np.isnan(float(loss))
    loss.backward()                  # ... and the gradient of that row is poisoned too: no optimizer step on a wrong gradient
    assert bool(torch.isnan(a.grad[1234]).all())


@pytest.mark.parametrize("k,n,c", [(27, 10007, 96), (27, 4099, 128), (27, 257, 256), (8, 5003, 64), (27, 3001, 32), (27, 1500, 20),
                                   (27, 700, 16), (1, 2000, 96), (27, 9, 1024)])
def test_gather_sum_list_form_gives_the_bits_of_the_register_form(B, k, n, c):
    """Pass 2 of the convolution: gather_list_kernel (live positions compacted in LDS, the default) against
    gather_sum_kernel (K position registers per lane; what ts_set_conv_impl(1) keeps) - the same additions in the same
    order, so the same bits; rows without any position, positions at both ends of Z, a ragged last workgroup; fp32 rows and
    (C >= 64) half rows."""
    rs = np.random.RandomState(k * 1000 + c)
    p = int(0.3 * k * n) + 1
    pos = np.full((k, n), -1, dtype=np.int32)
    flat = rs.permutation(k * n)[:p]
    pos.reshape(-1)[flat] = rs.permutation(p).astype(np.int32)          # every Z row used exactly once
    pos[:, n // 2] = -1                                                   # a row with no neighbour at all
    z = T(rs.randn(p, c).astype(np.float32))
    pos_t = T(pos)
    try:
        B.set_conv_impl(1)
        want = B.conv_gather_sum(z, pos_t, n)
        want_h = B.conv_gather_sum_f16(z.half(), pos_t, n) if c % 8 == 0 else None
    finally:
        B.set_conv_impl(0)
    got = B.conv_gather_sum(z, pos_t, n)
    assert torch.equal(got, want)
    assert float(got[n // 2].abs().max()) == 0.0 or bool((pos[:, n // 2] >= 0).any())
    ref = np.zeros((n, c), np.float32)
    zz = z.cpu().numpy()
    for kk in range(k):
        live = pos[kk] >= 0
        ref[live] += zz[pos[kk][live]]
    close(got, ref, 1e-5)
    if want_h is not None:
        assert torch.equal(B.conv_gather_sum_f16(z.half(), pos_t, n), want_h)


@pytest.mark.parametrize("dup", [False, True])
def test_symmetric_submanifold_kernel_maps_equal_the_full_probe(dup):
    """ts_build_kmap_sym (half the probes + mirrored stores, what the index plan builds its five submanifold maps with) against
    ts_build_kmap on the same coordinates: every table bit for bit (nn/functional/conv.py:156-176 order kept); coordinates with a
    duplicate are detected on the device and the index plan falls back to the full probe"""
    from taseg_amd import _fast
    from taseg_amd import backend as B
    from taseg_amd.data.synthetic import synth_scan
    from taseg_amd.torchsparse.nn import functional as spF
    from taseg_amd.torchsparse.nn.utils import get_kernel_offsets
    fast = _fast.module()
    if fast is None:
        pytest.skip("native fast path not built")
    pts, _ = synth_scan(3, n_points=40000, n_beams=32, n_az=1400)
    pc = np.unique(np.round(pts[:, :3] / 0.05).astype(np.int32), axis=0)
    pc -= pc.min(0)
    coords = np.concatenate([pc, np.zeros((len(pc), 1), np.int32)], 1)
    if dup:
        coords = np.concatenate([coords, coords[100:103]], 0)          # three voxels twice
    c = torch.from_numpy(coords).cuda()
    cm, sub_t, down_t, totals, _, _, _ = fast.index_plan(c, c.float(), 4, B.L.stream())
    names = ("nbr", "nbmaps", "nbsizes", "nboffs", "pos_out", "pos_in")
    for lvl in range(5):
        s = 1 << lvl
        offs = get_kernel_offsets(3, stride=s, dilation=1, device=c.device)
        full = B.build_kmap(cm[lvl], cm[lvl], offs)
        total = int(full["nboffs"][-1])
        # (a level whose coordinates hold a duplicate is reported as -(pairs + 1): the caller marks its map, functional.KernelMap.dup;
        # every coarser level is unique again - spdownsample deduplicates)
        assert int(totals[2 * lvl]) == (-(total + 1) if (dup and lvl == 0) else total)
        got = dict(zip(names, sub_t[lvl]))
        for k in names:
            if dup and lvl == 0 and k == "pos_in":
                continue      # two pairs per (offset, input row) there: the table holds whichever was written last
            a, b = got[k], full[k]
            if k == "nbmaps":
                a, b = a[:total], b[:total]
            assert torch.equal(a, b), (lvl, k)


@pytest.mark.parametrize("half", [False, True])
def test_convolution_over_duplicated_coordinates_sums_every_pair(half):
    """a coordinate held twice (the reference's API does not forbid it): the rulebook then has two pairs per (offset, input row),
    which the list-form input gradient cannot hold - the map is marked (`KernelMap.dup`) and takes the scatter form, like the
    reference's `convolution_backward` (convolution_cuda.cu:153-161, 236-263).  Output, input gradient and weight gradient against
    the plain sum over the rulebook in float64"""
    from taseg_amd.torchsparse import SparseTensor
    from taseg_amd.torchsparse import nn as spnn
    g = torch.Generator().manual_seed(3)
    base = torch.unique(torch.randint(0, 12, (600, 3), generator=g), dim=0)
    coords = torch.cat([base, base[5:9]], 0)                       # four voxels twice
    coords = torch.cat([coords, torch.zeros(len(coords), 1, dtype=coords.dtype)], 1).int().cuda()
    c_in, c_out = 32, 64
    feats = torch.randn(len(coords), c_in, generator=g).cuda().requires_grad_(True)
    conv = spnn.Conv3d(c_in, c_out, 3).cuda()
    x = SparseTensor(feats.half() if half else feats, coords)
    with torch.autocast("cuda", dtype=torch.float16, enabled=half):
        y = conv(x)
    km = x.kmaps[((1, 1, 1), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
    assert km.dup
    go = torch.randn(y.F.shape, generator=g).cuda()
    (y.F.float() * go).sum().backward()
    # the rulebook sum in float64
    nbmaps, nbsizes = km.nbmaps.cpu(), km.nbsizes.cpu()
    f64 = feats.detach().double().cpu().requires_grad_(True)
    w64 = conv.kernel.detach().double().cpu().requires_grad_(True)
    out = torch.zeros(len(coords), c_out, dtype=torch.float64)
    at = 0
    for k, n in enumerate(nbsizes.tolist()):
        pr = nbmaps[at:at + n]
        out = out.index_add(0, pr[:, 1], f64[pr[:, 0]] @ w64[k])
        at += n
    (out * go.double().cpu()).sum().backward()
    tol = 2e-2 if half else 1e-5
    rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())  # noqa: E731
    assert rel(y.F.detach(), out.detach()) <= tol
    assert rel(feats.grad, f64.grad) <= tol
    assert rel(conv.kernel.grad, w64.grad) <= tol
    # the same coordinates without the repeats: an ordinary map
    x1 = SparseTensor(feats.detach()[:len(base)], coords[:len(base)].contiguous())
    conv(x1)
    assert not x1.kmaps[((1, 1, 1), (3, 3, 3), (1, 1, 1), (1, 1, 1))].dup
