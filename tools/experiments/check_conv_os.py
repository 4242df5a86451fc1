"""Output-stationary (Z-free) convolution kernel (csrc/conv_os.hip) against the two-pass kernels it replaces on the wide
shallow layers - bit for bit - and against the float64 oracle (convolution_cuda.cu:101-258)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ts_oracle as O  # noqa: E402

DEV = "cuda"


def _blob(seed, n=9000, extent=30, batch=2):
    rs = np.random.RandomState(seed)
    c = np.unique(np.concatenate([rs.randint(0, extent, size=(n, 3)), rs.randint(0, batch, size=(n, 1))], 1), axis=0)
    return c[rs.permutation(len(c))].astype(np.int32)


def _planes(w):
    from taseg_amd import _lib as L
    pl = torch.empty(3 * w.numel(), dtype=torch.int16, device=w.device)
    L.check(L.load().ts_conv_split_planes(w.data_ptr(), w.shape[0], w.shape[1], w.shape[2], pl.data_ptr(), L.stream()),
            "ts_conv_split_planes")
    return pl


def _case(coords, ci, co, seed):
    from taseg_amd import backend as B
    from taseg_amd.torchsparse.nn.utils import get_kernel_offsets
    c = torch.from_numpy(coords).to(DEV)
    km = B.build_kmap(c, c, get_kernel_offsets(3, 1, 1, device=DEV))
    g = torch.Generator().manual_seed(seed)
    n = len(coords)
    x = torch.randn(n, ci, generator=g).to(DEV)
    gy = torch.randn(n, co, generator=g).to(DEV)
    w = (torch.randn(27, ci, co, generator=g) / np.sqrt(6 * ci)).to(DEV)
    res = torch.randn(n, ci, generator=g).to(DEV)
    return km, x, gy, w, res


@pytest.mark.parametrize("ci,co", [(96, 96), (128, 96)])
@pytest.mark.parametrize("n", [9000, 700, 193])
def test_output_stationary_kernel_gives_the_bits_of_the_two_passes(ci, co, n):
    """forward and input gradient (with and without the residual addend), ragged last tile, tiles with empty offsets"""
    from taseg_amd import backend as B
    coords = _blob(ci + n, n=n, extent=30 if n > 1000 else 12)
    km, x, gy, w, res = _case(coords, ci, co, seed=n + co)
    total = int(km["nboffs"][-1])
    pl = _planes(w)
    nrows = len(coords)
    y2 = B.conv_gather_sum(B.conv_pair_gemm(x, w, km["nbmaps"], km["nboffs"], total, 0), km["pos_out"], nrows)
    y1 = B.conv_os(x, pl, w.shape, km["nbr"])
    assert torch.equal(y1, y2)
    z = B.conv_pair_gemm(gy, w, km["nbmaps"], km["nboffs"], total, 1, weight_transposed=True)
    gx2 = B.conv_gather_sum(z, km["pos_in"], nrows)
    gx1 = B.conv_os(gy, pl, w.shape, km["nbr"], weight_transposed=True)
    assert torch.equal(gx1, gx2)
    gx1r = B.conv_os(gy, pl, w.shape, km["nbr"], weight_transposed=True, addend=res)
    assert torch.equal(gx1r, gx2 + res)
    # run-to-run identical, and against the float64 oracle
    assert torch.equal(B.conv_os(x, pl, w.shape, km["nbr"]), y1)
    _, nbmaps, nbsizes = O.build_kmap(coords, coords, O.get_kernel_offsets(3, 1, 1))
    want = O.conv_forward(x.cpu().numpy().astype(np.float64), w.cpu().numpy(), nbmaps, nbsizes, (nrows, nrows))
    want_gx, _ = O.conv_backward(x.cpu().numpy().astype(np.float64), w.cpu().numpy(), gy.cpu().numpy(), nbmaps, nbsizes)
    for got, ref in ((y1, want), (gx1, want_gx)):
        err = np.abs(got.cpu().numpy().astype(np.float64) - ref).max() / max(1.0, np.abs(ref).max())
        assert err <= 1e-5, err


def test_output_stationary_kernel_at_bench_size():
    """up4's 96 -> 96 layer on the rulebook of the benchmarked batch (178k voxels, 1.16 M pairs): bits of the two passes"""
    import bench
    from taseg_amd import backend as B
    coords, _, _, _ = bench.make_scans(0, 2, 120000, "minkunet")
    km, x, gy, w, res = _case(coords.cpu().numpy(), 96, 96, seed=1)
    total = int(km["nboffs"][-1])
    pl = _planes(w)
    n = coords.shape[0]
    y2 = B.conv_gather_sum(B.conv_pair_gemm(x, w, km["nbmaps"], km["nboffs"], total, 0), km["pos_out"], n)
    assert torch.equal(B.conv_os(x, pl, w.shape, km["nbr"]), y2)
    gx2 = B.conv_gather_sum(B.conv_pair_gemm(gy, w, km["nbmaps"], km["nboffs"], total, 1, weight_transposed=True), km["pos_in"], n)
    assert torch.equal(B.conv_os(gy, pl, w.shape, km["nbr"], weight_transposed=True), gx2)


def test_output_stationary_kernel_walks_a_dense_tile_in_batches():
    """a near-full voxel grid (~23 pairs per voxel): the lists of a 192-row tile exceed the LDS capacity and the kernel takes
    the offsets in several batches - same bits as the two passes"""
    from taseg_amd import backend as B
    rs = np.random.RandomState(5)
    grid = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(12), indexing="ij"), -1).reshape(-1, 3)
    keep = rs.rand(len(grid)) < 0.9
    c = np.concatenate([grid[keep], np.zeros((int(keep.sum()), 1), np.int64)], 1).astype(np.int32)
    c = c[rs.permutation(len(c))]
    km, x, gy, w, _ = _case(c, 96, 96, seed=3)
    total = int(km["nboffs"][-1])
    assert total > 20 * len(c)
    pl = _planes(w)
    y2 = B.conv_gather_sum(B.conv_pair_gemm(x, w, km["nbmaps"], km["nboffs"], total, 0), km["pos_out"], len(c))
    assert torch.equal(B.conv_os(x, pl, w.shape, km["nbr"]), y2)
    gx2 = B.conv_gather_sum(B.conv_pair_gemm(gy, w, km["nbmaps"], km["nboffs"], total, 1, weight_transposed=True), km["pos_in"], len(c))
    assert torch.equal(B.conv_os(gy, pl, w.shape, km["nbr"], weight_transposed=True), gx2)
