"""Property tests (hypothesis) of the rulebook / point-voxel / convolution semantics - SURVEY.md §4 item 2: the reference
ships no tests, so the properties its algorithms must have are checked on random inputs, once on the CPU oracle
(`-m "not gpu"`) and once through the HIP entry points (`-m gpu`, against the same oracle):

  * submanifold k3 map: pair (i, o, k) <=> (o, i, 26 - k), centre offset = identity (conv.py:160-176, kernel.py:11-32)
  * k2 / s2 down-sampling map: every fine voxel is in exactly one pair, coarse coordinates sorted by (b, x, y, z)
    (downsample.py:25-51)
  * voxelize / devoxelize are adjoint to their backward passes: <A x, y> = <x, A^T y> (voxelize_cuda.cu, devoxelize_cuda.cu)
  * the convolution's gradients are the derivatives of its forward pass (central differences in float64;
    convolution_cuda.cu:101-278)
"""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import ts_oracle as O

COMMON = dict(deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)


@st.composite
def clouds(draw, max_n=400, max_extent=14, max_batch=3):
    """unique voxel coordinates [N, 4] (x, y, z, b) in a random order; small extents make dense neighbourhoods"""
    n = draw(st.integers(1, max_n))
    ext = draw(st.integers(2, max_extent))
    nb = draw(st.integers(1, max_batch))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rs = np.random.RandomState(seed)
    c = np.concatenate([rs.randint(0, ext, (n, 3)), rs.randint(0, nb, (n, 1))], 1)
    c = np.unique(c, axis=0)
    return c[rs.permutation(len(c))].astype(np.int32), seed


def _check_submanifold(c, res, nbmaps, nbsizes):
    n = len(c)
    assert res.shape == (27, n)
    assert np.array_equal(res[13], np.arange(n))                    # centre offset: every voxel is its own neighbour
    assert np.array_equal(res[::-1], _inverse(res, n))               # (i, o, k) <=> (o, i, 26 - k)
    assert int(nbsizes.sum()) == len(nbmaps) == int((res >= 0).sum())
    assert np.array_equal(nbsizes, nbsizes[::-1])


def _inverse(res, n):
    inv = np.full_like(res, -1)
    kk, oo = np.nonzero(res >= 0)
    inv[kk, res[kk, oo]] = oo
    return inv


def _check_downsample(c, coarse, res, nbsizes):
    assert int(nbsizes.sum()) == len(c)                              # every fine voxel has exactly one coarse parent
    hit = (res >= 0)
    assert np.array_equal(np.sort(res[hit]), np.arange(len(c)))
    key = coarse[:, [3, 0, 1, 2]].astype(np.int64)
    assert all(tuple(key[i]) < tuple(key[i + 1]) for i in range(len(key) - 1))     # sorted, unique
    assert np.all(coarse[:, :3] % 2 == 0)


@settings(max_examples=40, **COMMON)
@given(clouds())
def test_oracle_submanifold_map_is_symmetric(cs):
    c, _ = cs
    res, nbmaps, nbsizes = O.build_kmap(c, c, O.get_kernel_offsets(3, 1, 1))
    _check_submanifold(c, res, nbmaps, nbsizes)


@settings(max_examples=40, **COMMON)
@given(clouds())
def test_oracle_downsample_map_partitions_the_fine_voxels(cs):
    c, _ = cs
    coarse = O.spdownsample(c, 2, 2, 1)
    res, nbmaps, nbsizes = O.build_kmap(c, coarse, O.get_kernel_offsets(2, 1, 1))
    _check_downsample(c, coarse, res, nbsizes)


@settings(max_examples=25, **COMMON)
@given(clouds(max_n=200), st.integers(1, 6))
def test_oracle_point_voxel_ops_are_adjoint(cs, ch):
    c, seed = cs
    rs = np.random.RandomState(seed)
    m, n = len(c), 3 * len(c)
    idx = rs.randint(-1, m, n)                                       # -1: a point outside every voxel
    counts = O.spcount(idx, m)
    x, y = rs.randn(n, ch), rs.randn(m, ch)
    lhs = (O.voxelize_forward(x, idx, counts) * y).sum()
    rhs = (x * O.voxelize_backward(y, idx, counts, n)).sum()
    assert abs(lhs - rhs) <= 1e-9 * (1 + abs(lhs))
    idx8 = rs.randint(-1, m, (n, 8))
    w = rs.rand(n, 8).astype(np.float32)
    f, g = rs.randn(m, ch), rs.randn(n, ch)
    lhs = (O.devoxelize_forward(f, idx8, w) * g).sum()
    rhs = (f * O.devoxelize_backward(g, idx8, w, m)).sum()
    assert abs(lhs - rhs) <= 1e-9 * (1 + abs(lhs))


@settings(max_examples=12, **COMMON)
@given(clouds(max_n=60, max_extent=5), st.integers(1, 4), st.integers(1, 4), st.booleans())
def test_oracle_conv_gradients_match_central_differences(cs, ci, co, strided):
    c, seed = cs
    rs = np.random.RandomState(seed)
    if strided:
        out = O.spdownsample(c, 2, 2, 1)
        offs = O.get_kernel_offsets(2, 1, 1)
    else:
        out, offs = c, O.get_kernel_offsets(3, 1, 1)
    _, nbmaps, nbsizes = O.build_kmap(c, out, offs)
    sizes = (len(c), len(out))
    x, w = rs.randn(len(c), ci), rs.randn(len(offs), ci, co)
    gy = rs.randn(len(out), co)
    gx, gw = O.conv_backward(x, w, gy, nbmaps, nbsizes)
    f = lambda xx, ww: float((O.conv_forward(xx, ww, nbmaps, nbsizes, sizes) * gy).sum())  # noqa: E731
    eps = 1e-6
    for _ in range(6):
        i, j = rs.randint(len(c)), rs.randint(ci)
        d = np.zeros_like(x)
        d[i, j] = eps
        assert abs((f(x + d, w) - f(x - d, w)) / (2 * eps) - gx[i, j]) <= 1e-6 * (1 + abs(gx[i, j]))
        k, a, b = rs.randint(len(offs)), rs.randint(ci), rs.randint(co)
        d = np.zeros_like(w)
        d[k, a, b] = eps
        assert abs((f(x, w + d) - f(x, w - d)) / (2 * eps) - gw[k, a, b]) <= 1e-6 * (1 + abs(gw[k, a, b]))


# ------------------------------------------------------------------------------------------------ the HIP path
def _T(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@settings(max_examples=25, **COMMON)
@given(clouds(max_n=3000, max_extent=24))
def test_hip_rulebooks_equal_the_oracle_and_keep_its_properties(cs):
    from taseg_amd import backend as B
    from taseg_amd.torchsparse.nn import functional as F
    c, _ = cs
    offs = O.get_kernel_offsets(3, 1, 1)
    km = B.build_kmap(_T(c), _T(c), _T(offs), want_inverse=True)
    res = km["nbr"].cpu().numpy()
    sizes = km["nbsizes"].cpu().numpy()
    total = int(sizes.sum())
    ref_res, ref_maps, ref_sizes = O.build_kmap(c, c, offs)
    assert np.array_equal(res, ref_res) and np.array_equal(sizes, ref_sizes)
    assert np.array_equal(km["nbmaps"][:total].cpu().numpy(), ref_maps)
    _check_submanifold(c, res, ref_maps, sizes)
    assert np.array_equal(km["nbr_t"].cpu().numpy(), res[::-1])
    coarse = F.spdownsample(_T(c), 2, 2, 1).cpu().numpy()
    assert np.array_equal(coarse, O.spdownsample(c, 2, 2, 1))
    offs2 = O.get_kernel_offsets(2, 1, 1)
    km2 = B.build_kmap(_T(c), _T(coarse), _T(offs2))
    _check_downsample(c, coarse, km2["nbr"].cpu().numpy(), km2["nbsizes"].cpu().numpy())


@pytest.mark.gpu
@settings(max_examples=15, **COMMON)
@given(clouds(max_n=1500, max_extent=16), st.sampled_from([4, 16, 32, 48, 96]), st.sampled_from([4, 16, 32, 64, 96]),
       st.booleans())
def test_hip_conv_is_the_oracle_conv_and_its_gradients_are_adjoint(cs, ci, co, strided):
    """forward and both gradients against the float64 oracle on random clouds and channel counts (full-tile and ragged
    kernels), plus <conv(x), gy> = <x, dgrad(gy)> = <W, wgrad(x, gy)> through the HIP kernels themselves"""
    import torch
    from taseg_amd import backend as B
    c, seed = cs
    rs = np.random.RandomState(seed)
    if strided:
        out = O.spdownsample(c, 2, 2, 1)
        offs = O.get_kernel_offsets(2, 1, 1)
    else:
        out, offs = c, O.get_kernel_offsets(3, 1, 1)
    km = B.build_kmap(_T(c), _T(out), _T(offs))
    _, nbmaps, nbsizes = O.build_kmap(c, out, offs)
    total = len(nbmaps)
    x = rs.randn(len(c), ci).astype(np.float32)
    w = (rs.randn(len(offs), ci, co) / np.sqrt(ci)).astype(np.float32)
    gy = rs.randn(len(out), co).astype(np.float32)
    y64 = O.conv_forward(x.astype(np.float64), w.astype(np.float64), nbmaps, nbsizes, (len(c), len(out)))
    gx64, gw64 = O.conv_backward(x.astype(np.float64), w.astype(np.float64), gy.astype(np.float64), nbmaps, nbsizes)
    xt, wt, gt = _T(x), _T(w), _T(gy)
    y = B.conv_gather_sum(B.conv_pair_gemm(xt, wt, km["nbmaps"], km["nboffs"], total, 0), km["pos_out"], len(out))
    gx = B.conv_gather_sum(B.conv_pair_gemm(gt, wt, km["nbmaps"], km["nboffs"], total, 1, weight_transposed=True),
                           km["pos_in"], len(c))
    gw = B.conv_wgrad(xt, gt, km["nbmaps"], km["nboffs"], len(offs), 0, total)

    def rel(a, b):
        return float(np.abs(a.double().cpu().numpy() - b).max()) / max(1.0, float(np.abs(b).max()))

    assert rel(y, y64) <= 1e-5 and rel(gx, gx64) <= 1e-5 and rel(gw, gw64) <= 1e-5
    s1 = float((y.double() * gt.double()).sum())
    s2 = float((xt.double() * gx.double()).sum())
    s3 = float((wt.double() * gw.double()).sum())
    scale = 1.0 + abs(s1)
    assert abs(s1 - s2) <= 2e-5 * scale and abs(s1 - s3) <= 2e-5 * scale


@pytest.mark.gpu
@settings(max_examples=15, **COMMON)
@given(clouds(max_n=800), st.sampled_from([4, 32, 96]))
def test_hip_point_voxel_ops_are_adjoint(cs, ch):
    import torch
    from taseg_amd import backend as B
    c, seed = cs
    rs = np.random.RandomState(seed)
    m, n = len(c), 3 * len(c)
    idx = rs.randint(-1, m, n).astype(np.int32)
    counts = O.spcount(idx, m)
    x, y = rs.randn(n, ch).astype(np.float32), rs.randn(m, ch).astype(np.float32)
    vx = B.voxelize_forward_cuda(_T(x), _T(idx), _T(counts))
    assert np.abs(vx.cpu().numpy() - O.voxelize_forward(x, idx, counts)).max() <= 1e-5
    gx = B.voxelize_backward_cuda(_T(y), _T(idx), _T(counts), n)
    lhs, rhs = float((vx.double() * _T(y).double()).sum()), float((_T(x).double() * gx.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * (1 + abs(lhs))
    idx8 = rs.randint(-1, m, (n, 8)).astype(np.int32)
    w8 = rs.rand(n, 8).astype(np.float32)
    f, g = rs.randn(m, ch).astype(np.float32), rs.randn(n, ch).astype(np.float32)
    df = B.devoxelize_forward_cuda(_T(f), _T(idx8), _T(w8))
    assert np.abs(df.cpu().numpy() - O.devoxelize_forward(f, idx8, w8)).max() <= 1e-5 * 8
    gb = B.devoxelize_backward_cuda(_T(g), _T(idx8), _T(w8), m)
    lhs, rhs = float((df.double() * _T(g).double()).sum()), float((_T(f).double() * gb.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * (1 + abs(lhs))
