"""Torch-facing wrappers over the C ABI (include/taseg_hip.h).

Section 1 mirrors the ten functions of the reference's pybind module
`torchsparse.backend` (TS/torchsparse/backend/pybind_cuda.cpp:18-39) name for name and
argument for argument, so `torchsparse.nn.functional` code written against the reference
binds unchanged.  Section 2 exposes the fused rulebook / convolution entry points the
MI355X design adds.  Tensors must be contiguous ROCm tensors; outputs are allocated with
torch (memory + stream plumbing only).
"""
import torch

from . import _lib as L
from ._lib import BackendError, check
from .options import options

__all__ = [
    "hash_cuda", "kernel_hash_cuda", "hash_query_cuda", "count_cuda",
    "voxelize_forward_cuda", "voxelize_backward_cuda",
    "devoxelize_forward_cuda", "devoxelize_backward_cuda", "devox_order", "devoxelize_backward_runs", "devox_csr",
    "devoxelize_backward_csr",
    "convolution_forward_cuda", "convolution_backward_cuda",
    "downsample", "unique_i64", "build_kmap", "trilinear_map", "conv_nbr", "conv_wgrad", "conv_class_plan", "conv_class_gemm", "conv_class_gemm_f16", "conv_class_conv", "conv_class_conv_f16", "class_finish_pays",
    "fuse_scan", "fuse_scans", "fuse_sweeps", "project_fov", "voxel_coords", "sparse_quantize", "set_conv_impl", "image_gather_forward", "image_gather_backward", "image_gather_rows_forward", "image_gather_rows_backward", "avgpool3s2_rows_forward", "avgpool3s2_rows_backward", "conv3x3c32_pack", "conv3x3c32_rows", "conv3x3c32_wgrad", "conv3x3_rows_takes", "conv3x3_rows_pack", "conv3x3_rows", "conv3x3_wgrad", "conv1x1c32_pack", "conv1x1c32_rows", "conv1x1c32_wgrad", "shuffle_cat_rows_takes", "shuffle_cat_rows_forward", "shuffle_cat_rows_backward",
]


def _f32(t, name):
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (got {t.dtype}); the HIP path computes in f32")
    return t.contiguous()


def _i32(t, name):
    if t.dtype != torch.int32:
        raise TypeError(f"{name} must be int32 (got {t.dtype})")
    return t.contiguous()


# --------------------------------------------------------------------------- section 1
def hash_cuda(idx):
    """hash_cuda(idx[N,4] int32) -> int64[N]   (hash_cuda.cu:67-73)."""
    L.require_device(idx)
    idx = _i32(idx, "coords")
    assert idx.ndim == 2 and idx.shape[1] == 4, idx.shape
    out = torch.empty(idx.shape[0], dtype=torch.int64, device=idx.device)
    L.check(L.load().ts_hash(L.ptr(idx), idx.shape[0], L.ptr(out), L.stream()), "ts_hash")
    return out


def kernel_hash_cuda(idx, kernel_offset):
    """kernel_hash_cuda(idx[N,4], offsets[K,3]) -> int64[K,N]   (hash_cuda.cu:75-84)."""
    L.require_device(idx, kernel_offset)
    idx = _i32(idx, "coords")
    kernel_offset = _i32(kernel_offset, "offsets")
    n, k = idx.shape[0], kernel_offset.shape[0]
    out = torch.empty((k, n), dtype=torch.int64, device=idx.device)
    L.check(L.load().ts_kernel_hash(L.ptr(idx), n, L.ptr(kernel_offset), k, L.ptr(out), L.stream()),
            "ts_kernel_hash")
    return out


def hash_query_cuda(hash_query, hash_target, idx_target):
    """hash_query_cuda(queries, refs, ref_idx) -> int64, 0 = miss else idx+1 (query_cuda.cu:9-56)."""
    L.require_device(hash_query, hash_target, idx_target)
    q = hash_query.contiguous()
    r = hash_target.contiguous()
    if q.dtype != torch.int64 or r.dtype != torch.int64:
        raise TypeError("hash_query: hashes must be int64")
    it = None if idx_target is None else idx_target.contiguous()
    if it is not None and it.dtype != torch.int64:
        raise TypeError("hash_query: idx_target must be int64")
    lib = L.load()
    nb = lib.ts_hash_query_workspace_bytes(r.numel())
    ws = L.workspace(nb, q.device)
    out = torch.empty(q.numel(), dtype=torch.int64, device=q.device)
    L.check(lib.ts_hash_query(L.ptr(q), q.numel(), L.ptr(r), L.ptr(it), r.numel(), L.ptr(out), L.ptr(ws),
                              ws.numel(), L.stream()), "ts_hash_query")
    return out


def count_cuda(idx, s):
    """count_cuda(idx int32[N], s) -> int32[s] histogram of idx >= 0 (count_cuda.cu:24-31)."""
    L.require_device(idx)
    idx = _i32(idx, "idx")
    out = torch.empty(int(s), dtype=torch.int32, device=idx.device)
    L.check(L.load().ts_count(L.ptr(idx), idx.numel(), L.ptr(out), int(s), L.stream()), "ts_count")
    return out


def voxelize_forward_cuda(inputs, idx, counts):
    L.require_device(inputs, idx, counts)
    inputs, idx, counts = _f32(inputs, "inputs"), _i32(idx, "idx"), _i32(counts, "counts")
    n, c = inputs.shape
    m = counts.shape[0]
    out = torch.empty((m, c), dtype=torch.float32, device=inputs.device)
    L.check(L.load().ts_voxelize_forward(L.ptr(inputs), L.ptr(idx), L.ptr(counts), n, c, m, L.ptr(out), L.stream()),
            "ts_voxelize_forward")
    return out


def voxelize_backward_cuda(top_grad, idx, counts, n):
    L.require_device(top_grad, idx, counts)
    top_grad, idx, counts = _f32(top_grad, "top_grad"), _i32(idx, "idx"), _i32(counts, "counts")
    m, c = top_grad.shape
    out = torch.empty((int(n), c), dtype=torch.float32, device=top_grad.device)
    L.check(L.load().ts_voxelize_backward(L.ptr(top_grad), L.ptr(idx), L.ptr(counts), int(n), c, m, L.ptr(out),
                                          L.stream()), "ts_voxelize_backward")
    return out


def devoxelize_forward_cuda(feat, indices, weight):
    L.require_device(feat, indices, weight)
    feat, indices, weight = _f32(feat, "feat"), _i32(indices, "indices"), _f32(weight, "weight")
    m, c = feat.shape
    n = indices.shape[0]
    assert indices.shape == (n, 8) and weight.shape == (n, 8), (indices.shape, weight.shape)
    out = torch.empty((n, c), dtype=torch.float32, device=feat.device)
    L.check(L.load().ts_devoxelize_forward(L.ptr(feat), L.ptr(indices), L.ptr(weight), n, c, m, L.ptr(out),
                                           L.stream()), "ts_devoxelize_forward")
    return out


def devoxelize_backward_cuda(top_grad, indices, weight, n):
    L.require_device(top_grad, indices, weight)
    top_grad, indices, weight = _f32(top_grad, "top_grad"), _i32(indices, "indices"), _f32(weight, "weight")
    npts, c = top_grad.shape
    out = torch.empty((int(n), c), dtype=torch.float32, device=top_grad.device)
    L.check(L.load().ts_devoxelize_backward(L.ptr(top_grad), L.ptr(indices), L.ptr(weight), npts, c, int(n),
                                            L.ptr(out), L.stream()), "ts_devoxelize_backward")
    return out


def devox_order(indices, n_vox):
    """int32 [n] walk order that groups points with identical 8-corner tuples (one interpolation cell) for
    `devoxelize_backward_runs`."""
    L.require_device(indices)
    indices = _i32(indices, "indices")
    n = indices.shape[0]
    lib = L.load()
    ws = L.workspace(lib.ts_devox_order_workspace_bytes(n), indices.device)
    order = torch.empty(n, dtype=torch.int32, device=indices.device)
    L.check(lib.ts_devox_order(L.ptr(indices), n, int(n_vox), L.ptr(order), L.ptr(ws), ws.numel(), L.stream()),
            "ts_devox_order")
    return order


def devoxelize_forward_into(feat, indices, weight, out, col):
    """devoxelize_forward_cuda writing out[:, col : col + C] of a wider point matrix in place (no later torch.cat).
    feat and out both float32, or both float16 (half storage, float32 sums)."""
    L.require_device(feat, indices, weight, out)
    half = feat.dtype == torch.float16
    feat = _f16(feat, "feat") if half else _f32(feat, "feat")
    indices, weight = _i32(indices, "indices"), _f32(weight, "weight")
    m, c = feat.shape
    n = indices.shape[0]
    if out.dtype != feat.dtype or not out.is_contiguous() or out.shape[0] != n or col + c > out.shape[1] or col % 4 \
            or (half and (c % 4 or out.shape[1] % 4)):
        raise ValueError("devoxelize_forward_into: out must be a contiguous [n, ld] matrix of feat's dtype with room at `col`")
    if half:
        L.check(L.load().ts_devoxelize_forward_f16_ld(L.ptr(feat), L.ptr(indices), L.ptr(weight), n, c, m,
                                                      out.data_ptr() + 2 * col, out.shape[1], L.stream()),
                "ts_devoxelize_forward_f16_ld")
        return
    L.check(L.load().ts_devoxelize_forward_ld(L.ptr(feat), L.ptr(indices), L.ptr(weight), n, c, m, out.data_ptr() + 4 * col,
                                              out.shape[1], L.stream()), "ts_devoxelize_forward_ld")


def devoxelize_backward_from(grad, col, c, indices, weight, n_vox, order=None):
    """Adjoint of devoxelize_forward_into for the column block grad[:, col : col + c]; `order` = a walk order
    (devox_order), an inverse map (devox_csr), a cell plan (devox_cells) or None.  A float16 `grad` (inverse map or cell
    plan only) gives a float16 result: float32 sums, one rounding."""
    L.require_device(grad, indices, weight)
    indices, weight = _i32(indices, "indices"), _f32(weight, "weight")
    half = grad.dtype == torch.float16
    if grad.dtype not in (torch.float32, torch.float16) or not grad.is_contiguous() or col + c > grad.shape[1] or col % 4 \
            or grad.shape[1] % 4:
        raise ValueError("devoxelize_backward_from: grad must be a contiguous float32 / float16 [n, ld] matrix, ld and col % 4 == 0")
    if half and not isinstance(order, tuple):
        raise TypeError("devoxelize_backward_from: float16 gradients need an inverse map or a cell plan")
    n, ld = grad.shape
    out = torch.empty((int(n_vox), c), dtype=grad.dtype, device=grad.device)
    lib = L.load()
    base = grad.data_ptr() + grad.element_size() * col
    if isinstance(order, tuple) and isinstance(order[0], str):        # ("cells", ...): backend.devox_cells
        _, walk, seg_start, off, ent = order
        n_seg = seg_start.shape[0] - 1
        part = torch.empty((max(8 * n_seg, 1), c), dtype=torch.float32, device=grad.device)
        fn, name = (lib.ts_devoxelize_backward_cells_f16_ld, "ts_devoxelize_backward_cells_f16_ld") if half else \
            (lib.ts_devoxelize_backward_cells_ld, "ts_devoxelize_backward_cells_ld")
        L.check(fn(base, ld, L.ptr(weight), L.ptr(walk), L.ptr(seg_start), n_seg, L.ptr(off), L.ptr(ent), n, c, int(n_vox),
                   L.ptr(part), L.ptr(out), L.stream()), name)
    elif isinstance(order, tuple):
        off, ent = order
        fn, name = (lib.ts_devoxelize_backward_csr_f16_ld, "ts_devoxelize_backward_csr_f16_ld") if half else \
            (lib.ts_devoxelize_backward_csr_ld, "ts_devoxelize_backward_csr_ld")
        L.check(fn(base, ld, L.ptr(weight), L.ptr(off), L.ptr(ent), n, c, int(n_vox), L.ptr(out), L.stream()), name)
    else:
        L.check(lib.ts_devoxelize_backward_runs_ld(base, ld, L.ptr(indices), L.ptr(weight),
                                                   L.ptr(order), n, c, int(n_vox), L.ptr(out), L.stream()),
                "ts_devoxelize_backward_runs_ld")
    return out


def devox_csr(indices, weight, n_vox):
    """Inverse of a trilinear map for `devoxelize_backward_csr`: (offsets [n_vox + 1], entries [8 n]) int32."""
    L.require_device(indices, weight)
    indices, weight = _i32(indices, "indices"), _f32(weight, "weight")
    n = indices.shape[0]
    lib = L.load()
    ws = L.workspace(lib.ts_devox_csr_workspace_bytes(n), indices.device)
    off = torch.empty(int(n_vox) + 1, dtype=torch.int32, device=indices.device)
    ent = torch.empty(max(8 * n, 1), dtype=torch.int32, device=indices.device)
    L.check(lib.ts_devox_csr(L.ptr(indices), L.ptr(weight), n, int(n_vox), L.ptr(off), L.ptr(ent), L.ptr(ws), ws.numel(),
                             L.stream()), "ts_devox_csr")
    return off, ent


def devox_cells(indices, weight, n_vox, max_len=64):
    """Plan of the cell-reduced devoxelize backward (include/taseg_hip.h, ts_devoxelize_backward_cells_ld) for a trilinear
    map (indices, weight) [n, 8]: ("cells", walk order [n], seg_start [n_seg + 1], offsets [n_vox + 1], entries) - for
    coarse strides, where many points share an interpolation cell.  Coordinates only; one host read (the segment
    count) on the calling stream."""
    L.require_device(indices, weight)
    indices, weight = _i32(indices, "indices"), _f32(weight, "weight")
    n = indices.shape[0]
    walk = devox_order(indices, n_vox)
    flags = torch.empty(max(n, 1), dtype=torch.int32, device=indices.device)
    L.check(L.load().ts_devox_segments(L.ptr(indices), L.ptr(walk), n, int(max_len), L.ptr(flags), L.stream()),
            "ts_devox_segments")
    starts = torch.nonzero(flags[:n]).flatten().int()
    seg_start = torch.cat([starts, torch.tensor([n], dtype=torch.int32, device=indices.device)])
    first = walk[starts.long()].long()
    tuples = indices[first].contiguous()                                   # [n_seg, 8]: the corner tuple of every segment
    # a corner takes part if ANY point of the segment weighs on it; weight 1 on every present corner is a superset
    # (rows of zeros add nothing) and keeps the plan free of a second pass over the weights
    off, ent = devox_csr(tuples, (tuples >= 0).float(), n_vox)
    return ("cells", walk, seg_start, off, ent)


def devoxelize_backward_csr(top_grad, weight, csr, n):
    """devoxelize_backward_cuda as a gather along the inverse map `csr` = devox_csr(...) (or the cell-reduced plan of
    devox_cells): no atomics, deterministic."""
    if isinstance(csr[0], str):
        top_grad = _f32(top_grad, "top_grad")
        return devoxelize_backward_from(top_grad, 0, top_grad.shape[1], csr[1], weight, n, csr)   # (indices unused by this plan)
    off, ent = csr
    L.require_device(top_grad, weight, off, ent)
    top_grad, weight = _f32(top_grad, "top_grad"), _f32(weight, "weight")
    npts, c = top_grad.shape
    if off.shape[0] != int(n) + 1:
        raise ValueError("inverse map was built for another voxel count")
    out = torch.empty((int(n), c), dtype=torch.float32, device=top_grad.device)
    L.check(L.load().ts_devoxelize_backward_csr(L.ptr(top_grad), L.ptr(weight), L.ptr(off), L.ptr(ent), npts, c, int(n),
                                                L.ptr(out), L.stream()), "ts_devoxelize_backward_csr")
    return out


def devoxelize_backward_runs(top_grad, indices, weight, n, order=None):
    """devoxelize_backward_cuda with the atomics issued once per run of points sharing their corner tuple."""
    L.require_device(top_grad, indices, weight, order)
    top_grad, indices, weight = _f32(top_grad, "top_grad"), _i32(indices, "indices"), _f32(weight, "weight")
    if order is not None:
        order = _i32(order, "order")
        if order.shape[0] != top_grad.shape[0]:
            raise ValueError("order must hold one entry per point")
    npts, c = top_grad.shape
    out = torch.empty((int(n), c), dtype=torch.float32, device=top_grad.device)
    L.check(L.load().ts_devoxelize_backward_runs(L.ptr(top_grad), L.ptr(indices), L.ptr(weight), L.ptr(order), npts, c,
                                                 int(n), L.ptr(out), L.stream()), "ts_devoxelize_backward_runs")
    return out


def _host_sizes(neighbor_offset):
    if neighbor_offset.is_cuda:
        raise RuntimeError("neighbor_offset (nbsizes) must be a host tensor, as in the reference "
                           "(nn/functional/conv.py:56 passes nbsizes.cpu())")
    return neighbor_offset.to(torch.int32).contiguous()


def convolution_forward_cuda(in_feat, out_feat, kernel, neighbor_map, neighbor_offset, transpose):
    """Reference-form forward (convolution_cuda.cu:53-165): writes out_feat in place."""
    L.require_device(in_feat, out_feat, kernel, neighbor_map)
    if in_feat.shape[1] != kernel.shape[1]:
        raise ValueError("Input feature size and kernel size mismatch")  # convolution_cuda.cu:57-59
    in_feat, kernel = _f32(in_feat, "in_feat"), _f32(kernel, "kernel")
    nbmap = _i32(neighbor_map, "neighbor_map")
    sizes = _host_sizes(neighbor_offset)
    assert out_feat.is_contiguous() and out_feat.dtype == torch.float32
    lib = L.load()
    k, ci, co = kernel.shape
    nb = lib.ts_convolution_workspace_bytes(in_feat.shape[0], out_feat.shape[0], ci, co, k)
    ws = L.workspace(nb, in_feat.device)
    L.check(lib.ts_convolution_forward(L.ptr(in_feat), in_feat.shape[0], ci, L.ptr(out_feat), out_feat.shape[0], co,
                                       L.ptr(kernel), k, L.ptr(nbmap), sizes.data_ptr(), int(bool(transpose)),
                                       L.ptr(ws), ws.numel(), L.stream()), "ts_convolution_forward")
    return out_feat


def convolution_backward_cuda(in_feat, grad_in_feat, grad_out_feat, kernel, grad_kernel, neighbor_map,
                              neighbor_offset, transpose):
    """Reference-form backward (convolution_cuda.cu:167-278): fills grad_in_feat / grad_kernel."""
    L.require_device(in_feat, grad_in_feat, grad_out_feat, kernel, grad_kernel, neighbor_map)
    in_feat, kernel = _f32(in_feat, "in_feat"), _f32(kernel, "kernel")
    grad_out_feat = _f32(grad_out_feat, "grad_out_feat")
    nbmap = _i32(neighbor_map, "neighbor_map")
    sizes = _host_sizes(neighbor_offset)
    assert grad_in_feat.is_contiguous() and grad_kernel.is_contiguous()
    lib = L.load()
    k, ci, co = kernel.shape
    nb = lib.ts_convolution_workspace_bytes(in_feat.shape[0], grad_out_feat.shape[0], ci, co, k)
    ws = L.workspace(nb, in_feat.device)
    L.check(lib.ts_convolution_backward(L.ptr(in_feat), in_feat.shape[0], ci, L.ptr(grad_in_feat),
                                        L.ptr(grad_out_feat), grad_out_feat.shape[0], co, L.ptr(kernel),
                                        L.ptr(grad_kernel), k, L.ptr(nbmap), sizes.data_ptr(),
                                        int(bool(transpose)), L.ptr(ws), ws.numel(), L.stream()),
            "ts_convolution_backward")


# --------------------------------------------------------------------------- section 2
def _count_to_host(cnt, what):
    n = int(cnt.item())  # the one host sync of a rulebook level: the output shape
    if n < 0:
        raise ValueError(f"{what}: coordinate outside the supported range "
                         "(0 <= batch < 1024, -2^17 <= x,y,z < 2^17)")
    return n


def downsample(coords, stride):
    """spdownsample for stride in {1, kernel_size} (downsample.py:25-51): unique strided coords, (b,x,y,z)-sorted."""
    L.require_device(coords)
    coords = _i32(coords, "coords")
    n = coords.shape[0]
    lib = L.load()
    ws = L.workspace(lib.ts_downsample_workspace_bytes(n), coords.device)
    out = torch.empty((max(n, 1), 4), dtype=torch.int32, device=coords.device)
    cnt = torch.empty(1, dtype=torch.int32, device=coords.device)
    sx, sy, sz = (int(s) for s in stride)
    L.check(lib.ts_downsample(L.ptr(coords), n, sx, sy, sz, L.ptr(out), L.ptr(cnt), L.ptr(ws), ws.numel(),
                              L.stream()), "ts_downsample")
    m = _count_to_host(cnt, "downsample")
    return out[:m]


def unique_i64(keys, return_inverse=True):
    """torch.unique(int64) (+ inverse positions) as initial_voxelize uses it (minkunet/utils.py:16-18)."""
    L.require_device(keys)
    keys = keys.contiguous()
    assert keys.dtype == torch.int64 and keys.ndim == 1
    n = keys.numel()
    lib = L.load()
    ws = L.workspace(lib.ts_unique_workspace_bytes(n), keys.device)
    uniq = torch.empty(max(n, 1), dtype=torch.int64, device=keys.device)
    inv = torch.empty(max(n, 1), dtype=torch.int32, device=keys.device) if return_inverse else None
    cnt = torch.empty(1, dtype=torch.int32, device=keys.device)
    L.check(lib.ts_unique_i64(L.ptr(keys), n, L.ptr(uniq), L.ptr(inv), L.ptr(cnt), L.ptr(ws), ws.numel(),
                              L.stream()), "ts_unique_i64")
    m = int(cnt.item())
    if m < 0:
        raise ValueError("unique_i64: keys must be in [0, 2^62)")
    return (uniq[:m], inv[:n]) if return_inverse else uniq[:m]


def build_kmap(in_coords, out_coords, offsets, want_pairs=True, want_inverse=False, want_pos=True):
    """Neighbour table + reference-order rulebook (conv.py:156-176) in one call, no host sync.

    Returns dict(nbr [K,n_out], nbr_t [K,n_in] | None, nbmaps [K*n_out,2] capacity | None,
                 nbsizes [K], nboffs [K+1], pos_out [K,n_out] | None, pos_in [K,n_in] | None).
    """
    L.require_device(in_coords, out_coords, offsets)
    in_coords, out_coords, offsets = _i32(in_coords, "in_coords"), _i32(out_coords, "out_coords"), _i32(offsets, "offsets")
    n_in, n_out, k = in_coords.shape[0], out_coords.shape[0], offsets.shape[0]
    dev = in_coords.device
    lib = L.load()
    ws = L.workspace(lib.ts_build_kmap_workspace_bytes(n_in, n_out, k), dev)
    nbr = torch.empty((k, n_out), dtype=torch.int32, device=dev)
    nbr_t = torch.empty((k, n_in), dtype=torch.int32, device=dev) if want_inverse else None
    nbmaps = torch.empty((max(k * n_out, 1), 2), dtype=torch.int32, device=dev) if want_pairs else None
    nbsizes = torch.empty(k, dtype=torch.int32, device=dev)
    nboffs = torch.empty(k + 1, dtype=torch.int32, device=dev)
    pos_out = torch.empty((k, n_out), dtype=torch.int32, device=dev) if want_pos else None
    pos_in = torch.empty((k, n_in), dtype=torch.int32, device=dev) if want_pos else None
    L.check(lib.ts_build_kmap(L.ptr(in_coords), n_in, L.ptr(out_coords), n_out, L.ptr(offsets), k, L.ptr(nbr),
                              L.ptr(nbr_t), L.ptr(nbmaps), L.ptr(nbsizes), L.ptr(nboffs), L.ptr(pos_out),
                              L.ptr(pos_in), L.ptr(ws), ws.numel(), L.stream()), "ts_build_kmap")
    return dict(nbr=nbr, nbr_t=nbr_t, nbmaps=nbmaps, nbsizes=nbsizes, nboffs=nboffs, pos_out=pos_out, pos_in=pos_in)


def gather_sum_kernel_name(c, k, half=False):
    """The kernel ts_conv_gather_sum[_f16] launches for rows of c channels and K = k offsets (csrc/conv_pairs{,_h}.hip)."""
    import os
    lists = not options.gather_positions
    kt = k if k in (8, 27) else 0
    if half:
        return "gather_list_h_kernel<8>" if lists and k <= 32 and 64 <= c <= 2048 else f"gather_sum_h_kernel<{kt}>"
    if c % 4:
        return "gather_sum_kernel<1,0>"
    return "gather_list_kernel<8>" if lists and k <= 32 and 16 <= c <= 1024 else f"gather_sum_kernel<4,{kt}>"


planes_in_use = False        # set by bench.py when the run's convolutions go through taseg_amd.planes (kernel naming only)


def pair_gemm_kernel_name(c_out, weight_transposed=False, c_red=None):
    bn, wr = (32, 4) if c_out <= 32 else (64, 2) if c_out <= 64 else (96, 2) if c_out % 96 == 0 else (128, 2)
    fast = c_red is not None and c_red % 32 == 0 and c_out % bn == 0
    if fast and _conv_impl == 0 and bn == 128 and planes_in_use:
        # the block calls of a model run with pre-split weight planes (taseg_amd/planes.py): direct-rows kernel
        return f"pair_gemm_d_kernel<128,{'true' if weight_transposed else 'false'}>"
    if fast and _conv_impl == 0:
        return f"pair_gemm_s_kernel<128,{bn},{wr},{'true' if weight_transposed else 'false'},true>"
    kind = "" if not fast else ("fast_" if weight_transposed else "persist_")
    return f"pair_gemm_{kind}kernel<{bn},{wr},{'true' if weight_transposed else 'false'}>"


def conv_pair_gemm(feat, kernel, nbmaps, nboffs, n_pairs, gather_col, weight_transposed=False):
    """Pass 1 of the two-pass convolution: z[p] = feat[nbmaps[p][gather_col]] @ (kernel[k(p)] or its transpose)."""
    L.require_device(feat, kernel, nbmaps, nboffs)
    feat, kernel = _f32(feat, "feat"), _f32(kernel, "kernel")
    nbmaps, nboffs = _i32(nbmaps, "nbmaps"), _i32(nboffs, "nboffs")
    if kernel.ndim != 3:
        raise ValueError("kernel must be [K, c_in, c_out]")
    k = kernel.shape[0]
    c_red, c_out = (kernel.shape[2], kernel.shape[1]) if weight_transposed else (kernel.shape[1], kernel.shape[2])
    if feat.shape[1] != c_red:
        raise ValueError("Input feature size and kernel size mismatch")
    z = torch.empty((int(n_pairs), c_out), dtype=torch.float32, device=feat.device)
    with _Timed("pair_gemm", name=pair_gemm_kernel_name(c_out, weight_transposed, c_red), pairs=int(n_pairs), c_red=c_red,
                c_out=c_out, k=k, n_rows=feat.shape[0]):
        L.check(L.load().ts_conv_pair_gemm(L.ptr(feat), feat.shape[0], c_red, L.ptr(kernel), k,
                                           1 if weight_transposed else 0, L.ptr(nbmaps), L.ptr(nboffs),
                                           int(n_pairs), int(gather_col), L.ptr(z), c_out, L.stream()),
                "ts_conv_pair_gemm")
    return z


def conv_gather_sum(z, pos, n_rows):
    """Pass 2: out[j] = sum_k z[pos[k, j]] (k ascending, -1 skipped)."""
    L.require_device(z, pos)
    z, pos = _f32(z, "z"), _i32(pos, "pos")
    k = pos.shape[0]
    if pos.shape != (k, n_rows):
        raise ValueError(f"position table shape {tuple(pos.shape)} != {(k, n_rows)}")
    out = torch.empty((n_rows, z.shape[1]), dtype=torch.float32, device=z.device)
    with _Timed("gather_sum", name=gather_sum_kernel_name(z.shape[1], k),
                pairs=z.shape[0], c_red=0, c_out=z.shape[1], k=k, n_rows=n_rows):
        L.check(L.load().ts_conv_gather_sum(L.ptr(z), z.shape[1], L.ptr(pos), k, n_rows, z.shape[0], L.ptr(out),
                                            L.stream()), "ts_conv_gather_sum")
    return out


def _f16(t, name):
    if t.dtype != torch.float16:
        raise TypeError(f"{name} must be float16 (got {t.dtype})")
    return t.contiguous()


def cast_weights_f16(kernel, want=(True, True)):
    """fp32 [K, Ci, Co] -> (w16 [K, Ci, Co], w16t [K, Co, Ci]) half copies in one pass (None where not wanted)."""
    L.require_device(kernel)
    kernel = _f32(kernel, "kernel")
    k, ci, co = kernel.shape
    w16 = torch.empty((k, ci, co), dtype=torch.float16, device=kernel.device) if want[0] else None
    w16t = torch.empty((k, co, ci), dtype=torch.float16, device=kernel.device) if want[1] else None
    L.check(L.load().ts_cast_weights_f16(L.ptr(kernel), k, ci, co, L.ptr(w16), L.ptr(w16t), L.stream()),
            "ts_cast_weights_f16")
    return w16, w16t


def conv_pair_gemm_f16(feat, w_rows, nbmaps, nboffs, n_pairs, gather_col, natural=False):
    """Pass 1 in half storage: z[p] = feat[nbmaps[p][gather_col]] @ W_k(p).  natural=False: w_rows [K, c_out, c_red] (one
    row per output column, contiguous in the reduction index: how the input gradient reads the half weight
    [K, C_in, C_out]); natural=True: w_rows [K, c_red, c_out] (how the forward pass reads the same tensor)."""
    L.require_device(feat, w_rows, nbmaps, nboffs)
    feat, w_rows = _f16(feat, "feat"), _f16(w_rows, "w_rows")
    nbmaps, nboffs = _i32(nbmaps, "nbmaps"), _i32(nboffs, "nboffs")
    if natural:
        k, c_red, c_out = w_rows.shape
    else:
        k, c_out, c_red = w_rows.shape
    if feat.shape[1] != c_red:
        raise ValueError("Input feature size and kernel size mismatch")
    z = torch.empty((int(n_pairs), c_out), dtype=torch.float16, device=feat.device)
    with _Timed("pair_gemm", name=f"pair_gemm_h_kernel<{128 if c_out % 128 == 0 else 96 if c_out % 96 == 0 else 64 if c_out % 64 == 0 else 32}>",
                pairs=int(n_pairs), c_red=c_red, c_out=c_out, k=k, esize=2, n_rows=feat.shape[0]):
        fn = L.load().ts_conv_pair_gemm_f16_nat if natural else L.load().ts_conv_pair_gemm_f16
        L.check(fn(L.ptr(feat), feat.shape[0], c_red, L.ptr(w_rows), k, L.ptr(nbmaps), L.ptr(nboffs), int(n_pairs),
                   int(gather_col), L.ptr(z), c_out, L.stream()), "ts_conv_pair_gemm_f16")
    return z


def conv_gather_sum_f16(z, pos, n_rows):
    """Pass 2 in half storage (fp32 accumulation)."""
    L.require_device(z, pos)
    z, pos = _f16(z, "z"), _i32(pos, "pos")
    k = pos.shape[0]
    if pos.shape != (k, n_rows):
        raise ValueError(f"position table shape {tuple(pos.shape)} != {(k, n_rows)}")
    out = torch.empty((n_rows, z.shape[1]), dtype=torch.float16, device=z.device)
    with _Timed("gather_sum", name=gather_sum_kernel_name(z.shape[1], k, half=True), pairs=z.shape[0], c_red=0,
                c_out=z.shape[1], k=k, n_rows=n_rows, esize=2):
        L.check(L.load().ts_conv_gather_sum_f16(L.ptr(z), z.shape[1], L.ptr(pos), k, n_rows, z.shape[0], L.ptr(out),
                                                L.stream()), "ts_conv_gather_sum_f16")
    return out


def conv_wgrad_f16(a_feat, b_feat, nbmaps, nboffs, kernel_volume, col_a, max_pairs):
    """grad_kernel[k] = sum_pairs a[pa]^T b[pb] from half rows -> fp32 [K, c_a, c_b]."""
    L.require_device(a_feat, b_feat, nbmaps, nboffs)
    a_feat, b_feat = _f16(a_feat, "a_feat"), _f16(b_feat, "b_feat")
    nbmaps, nboffs = _i32(nbmaps, "nbmaps"), _i32(nboffs, "nboffs")
    out = torch.empty((kernel_volume, a_feat.shape[1], b_feat.shape[1]), dtype=torch.float32, device=a_feat.device)
    pick = lambda c: 128 if c % 128 == 0 else 96 if c % 96 == 0 else 64 if c % 64 == 0 else 32  # noqa: E731
    with _Timed("conv_wgrad", name=f"wgrad_h_kernel<{pick(a_feat.shape[1])},{pick(b_feat.shape[1])}>", nboffs=nboffs,
                c_red=a_feat.shape[1], c_out=b_feat.shape[1], k=kernel_volume, esize=2, n_rows=a_feat.shape[0],
                n_rows_b=b_feat.shape[0]):
        lib = L.load()
        ws = L.workspace(lib.ts_conv_wgrad_workspace_bytes(int(max_pairs), a_feat.shape[1], b_feat.shape[1], kernel_volume),
                         a_feat.device)
        L.check(lib.ts_conv_wgrad_f16_det(L.ptr(a_feat), a_feat.shape[1], L.ptr(b_feat), b_feat.shape[1],
                                          L.ptr(nbmaps), L.ptr(nboffs), kernel_volume, int(col_a), int(max_pairs),
                                          L.ptr(out), L.ptr(ws), ws.numel(), L.stream()), "ts_conv_wgrad_f16_det")
    return out


def trilinear_map(points, vox_coords, stride):
    """8-corner voxel indices and trilinear weights of `voxel_to_point` (minkunet/utils.py:72-82)."""
    L.require_device(points, vox_coords)
    points = _f32(points, "points")
    vox_coords = _i32(vox_coords, "vox_coords")
    assert points.ndim == 2 and points.shape[1] == 4, points.shape
    n, m = points.shape[0], vox_coords.shape[0]
    lib = L.load()
    ws = L.workspace(lib.ts_trilinear_workspace_bytes(m), points.device)
    idx = torch.empty((n, 8), dtype=torch.int32, device=points.device)
    w = torch.empty((n, 8), dtype=torch.float32, device=points.device)
    L.check(lib.ts_trilinear_map(L.ptr(points), n, L.ptr(vox_coords), m, int(stride), L.ptr(idx), L.ptr(w),
                                 L.ptr(ws), ws.numel(), L.stream()), "ts_trilinear_map")
    return idx, w


_conv_impl = 0      # mirrors the library's selector (kernel names reported to bench.py)

# Optional per-launch timing (bench.py): HIP events on the launch stream around the conv kernels.
_prof = None
_prof_store = None


def profile_begin(expected_launches=0):
    """Start recording per-launch events.  `expected_launches` = bracketed launches until profile_end(): their events
    are created now instead of inside the recorded steps."""
    global _prof, _prof_store
    _prof = _prof_store = []
    lib = L.load()
    if expected_launches:
        check(lib.ts_prof_reserve(2 * int(expected_launches)), "ts_prof_reserve")
    lib.ts_prof_enable(1)           # the fused block calls bracket their own launches (csrc/block.hip)


def profile_pause(paused):
    """Stop / resume recording without dropping what was collected (bench.py samples every n-th step: two
    events per launch cost ~5 us of host time each, ~3 ms per step when every launch is bracketed)."""
    global _prof
    _prof = None if paused else _prof_store
    L.load().ts_prof_enable(0 if paused else 1)


class _Ms:
    """stands in for the (start, stop) event pair of a launch the library timed itself"""
    __slots__ = ("ms",)

    def __init__(self, ms):
        self.ms = ms

    def elapsed_time(self, _other):
        return self.ms


def profile_empty_bracket_us(reps=200):
    """Microseconds an event pair measures with nothing between its two records on the current stream (median of `reps`):
    the bracket's own share of every per-launch figure profile_end() returns."""
    import ctypes
    out = ctypes.c_double(0.0)
    L.check(L.load().ts_prof_empty_bracket_us(int(reps), L.stream(), ctypes.byref(out)), "ts_prof_empty_bracket_us")
    return float(out.value)


def profile_end():
    """[(kind, start, stop, meta)] of the bracketed launches: the Python wrappers' torch events and the records of the
    fused block calls (`ts_prof_collect`)."""
    global _prof, _prof_store
    out, _prof, _prof_store = _prof_store, None, None
    out = out or []
    lib = L.load()
    lib.ts_prof_enable(0)
    import ctypes
    cap = 1 << 16
    buf = (ctypes.c_double * (9 * cap))()
    n = int(lib.ts_prof_collect(buf, cap))
    if n < 0:
        raise BackendError("ts_prof_collect failed")
    for i in range(n):
        kind, ms, pairs, c_red, c_out, k, rows, esize, wt = buf[9 * i:9 * i + 9]
        c_red, c_out, k, pairs, rows, esize = int(c_red), int(c_out), int(k), int(pairs), int(rows), int(esize)
        half = esize == 2
        pick = lambda c: 128 if c % 128 == 0 else 96 if c % 96 == 0 else 64 if c % 64 == 0 else 32  # noqa: E731
        if int(kind) == 0:
            name = f"pair_gemm_h_kernel<{pick(c_out)}>" if half else pair_gemm_kernel_name(c_out, bool(wt), c_red)
            out.append(("pair_gemm", _Ms(ms), None, dict(name=name, pairs=pairs, c_red=c_red, c_out=c_out, k=k, esize=esize,
                                                         n_rows=rows)))
        elif int(kind) == 3:      # class-sorted implicit GEMM (csrc/conv_class.hip): `wt` carries the rows of Z'
            # (`wt` >= 0: rows of Z' of a pass-2 plan; < 0: minus the result rows a direct plan stores)
            out.append(("class_gemm", _Ms(ms), None, dict(name=f"class_gemm_kernel<{pick(c_out)}>", pairs=pairs, c_red=c_red,
                                                          c_out=c_out, k=k, esize=esize, n_rows=rows, z_rows=max(int(wt), 0),
                                                          out_rows=max(-int(wt), 0))))
        elif int(kind) == 4:      # ... finished inside the product: `wt` = Z' rows moved (written + read back), result rows = rows
            out.append(("class_gemm", _Ms(ms), None, dict(name=f"class_gemm_kernel<{pick(c_out)}>", pairs=pairs, c_red=c_red,
                                                          c_out=c_out, k=k, esize=esize, n_rows=rows, z_rows=int(wt), out_rows=rows)))
        elif int(kind) == 5:      # the ordered sum of the weight gradient's partial tiles as a launch of its own (second stream)
            out.append(("wgrad_reduce", _Ms(ms), None, dict(name="wgrad_reduce_seq_kernel" if rows else "wgrad_reduce_kernel", pairs=0,
                                                            c_red=c_red, c_out=c_out, k=k, esize=4, n_rows=0, bytes=float(wt))))
        elif int(kind) == 1:
            name = gather_sum_kernel_name(c_out, k, half)
            out.append(("gather_sum", _Ms(ms), None, dict(name=name, pairs=pairs, c_red=0, c_out=c_out, k=k, n_rows=rows,
                                                          esize=esize, side_bytes=float(wt))))
        else:
            name = f"wgrad_h_kernel<{pick(c_red)},{pick(c_out)}>" if half else conv_kernel_name(c_red, wgrad_cb=c_out)
            out.append(("conv_wgrad", _Ms(ms), None, dict(name=name, pairs=pairs, c_red=c_red, c_out=c_out, k=k,
                                                          esize=esize, n_rows=rows, n_rows_b=int(wt))))
    return out


class _NoTimer:
    __slots__ = ()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_TIMER = _NoTimer()


class _Timed:
    """`with _Timed(kind, name=..., ...)` brackets a launch with HIP events while bench.py profiles; otherwise the
    constructor returns a shared no-op context (this runs ~300 times per step)."""

    def __new__(cls, kind, **meta):
        if _prof is None:
            return _NO_TIMER
        return super().__new__(cls)

    def __init__(self, kind, **meta):
        self.kind, self.meta = kind, meta

    def __enter__(self):
        if _prof is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _prof is not None:
            self.e1.record()
            _prof.append((self.kind, self.e0, self.e1, self.meta))
        return False


def conv_kernel_name(c_out_or_ca, weight_transposed=False, wgrad_cb=None, n_out=0):
    """The template instantiation ts_conv_nbr / ts_conv_wgrad picks (mirrors the heuristic in csrc/conv.hip)."""
    if wgrad_cb is not None:
        pick = lambda c: 32 if c <= 32 else 64 if c <= 64 else 96 if c % 96 == 0 else 128  # noqa: E731
        tm, tn = pick(c_out_or_ca), pick(wgrad_cb)
        fast = c_out_or_ca % tm == 0 and wgrad_cb % tn == 0
        if fast and _conv_impl == 0:
            return f"wgrad_s_kernel<{tm},{tn}>"
        return f"wgrad_gemm_{'fast_' if fast else ''}kernel<{tm},{tn}>"
    c16 = (c_out_or_ca + 15) & ~15
    tiles64 = -(-n_out // 64)
    if c16 <= 64:
        o_tile = c16
    elif tiles64 * (-(-c16 // 128)) >= 1024:
        o_tile = 128 if (c16 % 128 == 0 or c16 > 256) else (c16 if c16 <= 128 else 64)
    else:
        o_tile = 64 if (c16 % 64 == 0 or c16 > 128) else c16
    if 64 < c16 <= 128 and c16 % 64 != 0:
        o_tile = c16
    bm = 128 if (o_tile <= 64 and tiles64 >= 2048) else 64
    if bm == 128:
        maxu = 4 if o_tile <= 32 else 8
    else:
        maxu = 2 if o_tile <= 32 else 4 if o_tile <= 64 else 8 if o_tile <= 128 else 16
    return f"conv_nbr_kernel<{bm},{maxu},{'true' if weight_transposed else 'false'}>"


def conv_nbr(in_feat, kernel, nbr, n_out, weight_transposed=False):
    """out[j] = sum_k in[nbr[k,j]] @ (kernel[k] or kernel[k]^T); every output row written once."""
    L.require_device(in_feat, kernel, nbr)
    in_feat, kernel, nbr = _f32(in_feat, "in_feat"), _f32(kernel, "kernel"), _i32(nbr, "nbr")
    if kernel.ndim != 3:
        raise ValueError("kernel must be [K, c_in, c_out]")
    k = kernel.shape[0]
    c_red, c_out = (kernel.shape[2], kernel.shape[1]) if weight_transposed else (kernel.shape[1], kernel.shape[2])
    if in_feat.shape[1] != c_red:
        raise ValueError("Input feature size and kernel size mismatch")
    if nbr.shape != (k, n_out):
        raise ValueError(f"neighbour table shape {tuple(nbr.shape)} != {(k, n_out)}")
    out = torch.empty((n_out, c_out), dtype=torch.float32, device=in_feat.device)
    with _Timed("conv_nbr", name=conv_kernel_name(c_out, weight_transposed, n_out=n_out), nbr=nbr, c_red=c_red, c_out=c_out, k=k,
                n_in=in_feat.shape[0], n_out=n_out):
        L.check(L.load().ts_conv_nbr(L.ptr(in_feat), in_feat.shape[0], c_red, L.ptr(kernel), k,
                                     1 if weight_transposed else 0, L.ptr(nbr), L.ptr(out), n_out, c_out,
                                     L.stream()), "ts_conv_nbr")
    return out


def conv_wgrad(a_feat, b_feat, nbmaps, nboffs, kernel_volume, col_a, max_pairs):
    """grad_kernel[k] = sum_pairs a[pa]^T b[pb]  ->  [K, c_a, c_b]; deterministic (partial tiles + ordered sum)."""
    L.require_device(a_feat, b_feat, nbmaps, nboffs)
    a_feat, b_feat = _f32(a_feat, "a_feat"), _f32(b_feat, "b_feat")
    nbmaps, nboffs = _i32(nbmaps, "nbmaps"), _i32(nboffs, "nboffs")
    out = torch.empty((kernel_volume, a_feat.shape[1], b_feat.shape[1]), dtype=torch.float32, device=a_feat.device)
    with _Timed("conv_wgrad", name=conv_kernel_name(a_feat.shape[1], wgrad_cb=b_feat.shape[1]), nboffs=nboffs,
                c_red=a_feat.shape[1], c_out=b_feat.shape[1], k=kernel_volume, n_rows=a_feat.shape[0],
                n_rows_b=b_feat.shape[0]):
        lib = L.load()
        ws = L.workspace(lib.ts_conv_wgrad_workspace_bytes(int(max_pairs), a_feat.shape[1], b_feat.shape[1], kernel_volume),
                         a_feat.device)
        L.check(lib.ts_conv_wgrad_det(L.ptr(a_feat), a_feat.shape[1], L.ptr(b_feat), b_feat.shape[1],
                                      L.ptr(nbmaps), L.ptr(nboffs), kernel_volume, int(col_a), int(max_pairs),
                                      L.ptr(out), L.ptr(ws), ws.numel(), L.stream()), "ts_conv_wgrad_det")
    return out


def image_plan(pix, pbatch, frame_end, frames, height, width, shift=0):
    """The FOV points of a batch in raster order of the pixels they project to, for one scale of the camera stack (unet2d.py:180-214;
    csrc/image.hip): dict(perm, paddr, run, err, n, hw, frames, shape) - `err` (int32[1], device) is non-zero if a point falls outside
    its sample's frames (the reference's indexing raises; reading it is a host sync, so the caller checks it lazily).
    pix [n, 2] float (row in the sample's stacked frames, col); pbatch [n] int32; frame_end [B] int32 cumulative frame counts."""
    L.require_device(pix, pbatch, frame_end)
    pix = _f32(pix, "pix")
    pbatch, frame_end = _i32(pbatch, "pbatch"), _i32(frame_end, "frame_end")
    n = int(pix.shape[0])
    dev = pix.device
    perm, paddr, run = (torch.empty(max(n, 1), dtype=torch.int32, device=dev) for _ in range(3))
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    lib = L.load()
    ws = L.workspace(lib.ts_image_plan_workspace_bytes(n), dev)
    L.check(lib.ts_image_plan(L.ptr(pix), L.ptr(pbatch), L.ptr(frame_end), n, frame_end.shape[0], int(frames), int(height), int(width),
                              int(shift), L.ptr(perm), L.ptr(paddr), L.ptr(run), L.ptr(err), L.ptr(ws), ws.numel(), L.stream()),
            "ts_image_plan")
    hs, ws_ = int(height) >> shift, int(width) >> shift
    return dict(perm=perm, paddr=paddr, run=run, err=err, n=n, hw=hs * ws_, frames=int(frames), shape=(hs, ws_), shift=int(shift))


def image_gather_forward(feat, plan):
    """out[n, c] = feat[first_frame(b_n) + row_n // H, c, (row_n % H) >> shift, col_n >> shift]  (unet2d.py:180-214) for the points
    of `plan`; feat [T, C, H >> shift, W >> shift] float32 NCHW.  Returns out [n, C] float32 in the original point order."""
    L.require_device(feat)
    feat = _f32(feat, "feat")
    t, c, hs, ws_ = feat.shape
    if (hs, ws_) != plan["shape"] or t != plan["frames"]:
        raise ValueError(f"feature stack {t}x{hs}x{ws_} does not match the plan's {plan['frames']}x{plan['shape'][0]}x{plan['shape'][1]}")
    n = plan["n"]
    out = torch.empty((n, c), dtype=torch.float32, device=feat.device)
    L.check(L.load().ts_image_gather_forward(L.ptr(feat), c, plan["hw"], L.ptr(plan["perm"]), L.ptr(plan["paddr"]), n, L.ptr(out),
                                             L.stream()), "ts_image_gather_forward")
    return out


def image_gather_backward(grad_out, plan, channels, into=None):
    """Adjoint of image_gather_forward as a segmented sum in raster order (no atomics, run-to-run identical).  `into` = the
    gradient the map already has from its other consumer, float32 contiguous [T, C, hs, ws]: the per-pixel sums are added to it IN
    PLACE (only pixels with points are touched) and it is returned; None: a zero-filled tensor is made first."""
    L.require_device(grad_out)
    grad_out = _f32(grad_out, "grad_out")
    n, c = grad_out.shape
    assert n == plan["n"] and c == channels
    hs, ws_ = plan["shape"]
    shape = (plan["frames"], c, hs, ws_)
    if into is None:
        out, acc = torch.empty(shape, dtype=torch.float32, device=grad_out.device), 0
    else:
        assert tuple(into.shape) == shape and into.dtype == torch.float32 and into.is_contiguous()
        out, acc = into, 1
    L.check(L.load().ts_image_gather_backward(L.ptr(grad_out), c, plan["hw"], L.ptr(plan["perm"]), L.ptr(plan["paddr"]),
                                              L.ptr(plan["run"]), n, L.ptr(out), out.numel(), acc, L.stream()),
            "ts_image_gather_backward")
    return out


def _channels_last_rows(t, what):
    """a [T, C, hs, ws] tensor whose MEMORY is [T, hs, ws, C] (torch.channels_last), or raise"""
    if t.dim() != 4 or not t.is_contiguous(memory_format=torch.channels_last):
        raise ValueError(f"{what}: expected a 4-d tensor in torch.channels_last memory format (shape {tuple(t.shape)}, strides {t.stride()})")
    return t


def image_gather_rows_forward(feat, plan):
    """image_gather_forward on a CHANNELS-LAST stack (any dtype): out [n, C] of feat's dtype, original point order; feat
    [T, C, hs, ws] with [T, hs, ws, C] memory - a pixel is one contiguous row (csrc/image.hip, ts_image_gather_rows_forward)."""
    L.require_device(feat)
    feat = _channels_last_rows(feat, "image_gather_rows_forward")
    t, c, hs, ws_ = feat.shape
    if (hs, ws_) != plan["shape"] or t != plan["frames"]:
        raise ValueError(f"feature stack {t}x{hs}x{ws_} does not match the plan's {plan['frames']}x{plan['shape'][0]}x{plan['shape'][1]}")
    n = plan["n"]
    out = torch.empty((n, c), dtype=feat.dtype, device=feat.device)
    L.check(L.load().ts_image_gather_rows_forward(L.ptr(feat), c, feat.element_size(), L.ptr(plan["perm"]), L.ptr(plan["paddr"]), n,
                                                  L.ptr(out), L.stream()), "ts_image_gather_rows_forward")
    return out


def image_gather_rows_backward(grad_out, plan, channels, dtype=None, into=None):
    """Adjoint of image_gather_rows_forward (segmented sum in raster order, no atomics, run-to-run identical) for float32 / float16
    maps.  `into` = the channels-last gradient the map already has from its other consumer: the per-pixel sums are added to it IN
    PLACE (only pixel rows with points are touched) and it is returned; None: a zero-filled channels-last tensor is made first."""
    L.require_device(grad_out)
    dtype = dtype or (into.dtype if into is not None else grad_out.dtype)
    if dtype not in (torch.float32, torch.float16):
        raise TypeError(f"image_gather_rows_backward: float32 / float16 maps only, got {dtype}")
    grad_out = grad_out.to(dtype).contiguous()
    n, c = grad_out.shape
    assert n == plan["n"] and c == channels
    hs, ws_ = plan["shape"]
    shape = (plan["frames"], c, hs, ws_)
    if into is None:
        out, acc = torch.empty(shape, dtype=dtype, device=grad_out.device, memory_format=torch.channels_last), 0
    else:
        assert tuple(into.shape) == shape and into.dtype == dtype
        out, acc = _channels_last_rows(into, "image_gather_rows_backward(into)"), 1
    L.check(L.load().ts_image_gather_rows_backward(L.ptr(grad_out), c, 1 if dtype == torch.float16 else 0, L.ptr(plan["perm"]),
                                                   L.ptr(plan["paddr"]), L.ptr(plan["run"]), n, L.ptr(out), out.numel(), acc,
                                                   L.stream()), "ts_image_gather_rows_backward")
    return out


def _pool_dtype(t, what):
    if t.dtype not in (torch.float32, torch.float16):
        raise TypeError(f"{what}: float32 / float16 only, got {t.dtype}")
    return 1 if t.dtype == torch.float16 else 0


def avgpool3s2_rows_forward(x):
    """AvgPool2d(kernel 3, stride 2, padding 1, count_include_pad) of a channels-last [T, C, H, W] float32 / float16 stack ->
    [T, C, Ho, Wo], channels-last (csrc/image.hip)."""
    L.require_device(x)
    half = _pool_dtype(x, "avgpool3s2_rows_forward")
    x = _channels_last_rows(x, "avgpool3s2_rows_forward")
    t, c, h, w = x.shape
    y = torch.empty((t, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    L.check(L.load().ts_avgpool3s2_rows_forward(L.ptr(x), t, h, w, c, half, L.ptr(y), L.stream()), "ts_avgpool3s2_rows_forward")
    return y


def avgpool3s2_rows_backward(grad_y, in_shape):
    """gradient of avgpool3s2_rows_forward with respect to its input of shape `in_shape` (channels-last, grad_y's dtype)"""
    L.require_device(grad_y)
    half = _pool_dtype(grad_y, "avgpool3s2_rows_backward")
    t, c, h, w = (int(v) for v in in_shape)
    if tuple(grad_y.shape) != (t, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1):
        raise ValueError(f"avgpool3s2_rows_backward: gradient of shape {tuple(grad_y.shape)} for an input of {(t, c, h, w)}")
    grad_y = grad_y.contiguous(memory_format=torch.channels_last)
    gx = torch.empty((t, c, h, w), dtype=grad_y.dtype, device=grad_y.device, memory_format=torch.channels_last)
    L.check(L.load().ts_avgpool3s2_rows_backward(L.ptr(grad_y), t, h, w, c, half, L.ptr(gx), L.stream()), "ts_avgpool3s2_rows_backward")
    return gx


def conv3x3c32_pack(weight, mode):
    """the packed MFMA operand of a Conv2d(32, 32, 3) weight (half, any strides): mode 0 forward, 1 data gradient"""
    L.require_device(weight)
    if weight.dtype != torch.float16 or tuple(weight.shape) != (32, 32, 3, 3):
        raise ValueError(f"conv3x3c32_pack: a float16 [32, 32, 3, 3] weight, got {weight.dtype} {tuple(weight.shape)}")
    lib = L.load()
    packed = torch.empty(lib.ts_conv3x3c32_packed_bytes(), dtype=torch.uint8, device=weight.device)
    s = weight.stride()
    L.check(lib.ts_conv3x3c32_pack(L.ptr(weight), s[0], s[1], s[2], s[3], int(mode), L.ptr(packed), L.stream()), "ts_conv3x3c32_pack")
    return packed


def conv3x3c32_rows(x, packed, bias, dilation):
    """Conv2d(32, 32, 3, stride 1, padding = dilation) of a channels-last float16 [T, 32, H, W] stack with a packed weight
    (conv3x3c32_pack) and an optional float32 bias [32]; with the data-gradient pack and x = grad_y: grad_x."""
    L.require_device(x, packed, bias)
    x = _channels_last_rows(x, "conv3x3c32_rows")
    t, c, h, w = x.shape
    if c != 32 or x.dtype != torch.float16:
        raise ValueError(f"conv3x3c32_rows: float16 with 32 channels, got {x.dtype} with {c}")
    if bias is not None:
        bias = _f32(bias, "bias")
    y = torch.empty_like(x, memory_format=torch.channels_last)
    L.check(L.load().ts_conv3x3c32_rows(L.ptr(x), L.ptr(packed), L.ptr(bias), t, h, w, int(dilation), L.ptr(y), L.stream()),
            "ts_conv3x3c32_rows")
    return y


def shuffle_cat_rows_takes(x, skip):
    """whether ts_shuffle_cat_rows_* takes this pair: channels-last float32 / float16 of one dtype, skip twice x's size, channel counts
    multiples of the 16-byte piece"""
    if (x.dim() != 4 or skip.dim() != 4 or x.dtype != skip.dtype or x.dtype not in (torch.float16, torch.float32) or not x.is_cuda
            or tuple(skip.shape[2:]) != (2 * x.shape[2], 2 * x.shape[3]) or skip.shape[0] != x.shape[0]):
        return False
    ve = 8 if x.dtype == torch.float16 else 4
    return (x.shape[1] % (4 * ve) == 0 and skip.shape[1] % ve == 0 and x.is_contiguous(memory_format=torch.channels_last)
            and skip.is_contiguous(memory_format=torch.channels_last))


def shuffle_cat_rows_forward(x, skip, scale=None):
    """concat(PixelShuffle(2)(x), skip) along the channels, times an optional float32 factor [T, C/4 + Cs] per frame and channel;
    channels-last in and out (csrc/shuffle_cat.hip)"""
    L.require_device(x, skip, scale)
    if not shuffle_cat_rows_takes(x, skip):
        raise ValueError(f"shuffle_cat_rows_forward: unsupported pair {tuple(x.shape)} {x.dtype} / {tuple(skip.shape)} {skip.dtype}")
    t, c, h, w = x.shape
    cs = skip.shape[1]
    if scale is not None:
        scale = _f32(scale, "scale")
        if scale.numel() != t * (c // 4 + cs):
            raise ValueError("shuffle_cat_rows_forward: scale must be [T, C/4 + Cs]")
    cat = torch.empty((t, c // 4 + cs, 2 * h, 2 * w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    L.check(L.load().ts_shuffle_cat_rows_forward(L.ptr(x), L.ptr(skip), L.ptr(scale), t, h, w, c, cs, int(x.dtype == torch.float16), L.ptr(cat),
                                                 L.stream()), "ts_shuffle_cat_rows_forward")
    return cat


def shuffle_cat_rows_backward(grad_cat, channels, scale=None):
    """the adjoint of shuffle_cat_rows_forward for x of `channels` channels: (grad_x [T, C, h, w], grad_skip [T, Cs, 2h, 2w])"""
    L.require_device(grad_cat, scale)
    grad_cat = _channels_last_rows(grad_cat, "shuffle_cat_rows_backward")
    t, cc, h2, w2 = grad_cat.shape
    c = int(channels)
    cs = cc - c // 4
    gx = torch.empty((t, c, h2 // 2, w2 // 2), dtype=grad_cat.dtype, device=grad_cat.device, memory_format=torch.channels_last)
    gs = torch.empty((t, max(cs, 0), h2, w2), dtype=grad_cat.dtype, device=grad_cat.device, memory_format=torch.channels_last)
    if not shuffle_cat_rows_takes(gx, gs) or cs <= 0:
        raise ValueError(f"shuffle_cat_rows_backward: unsupported gradient {tuple(grad_cat.shape)} {grad_cat.dtype} for {c} channels")
    if scale is not None:
        scale = _f32(scale, "scale")
        if scale.numel() != t * cc:
            raise ValueError("shuffle_cat_rows_backward: scale must be [T, C/4 + Cs]")
    L.check(L.load().ts_shuffle_cat_rows_backward(L.ptr(grad_cat), L.ptr(scale), t, h2 // 2, w2 // 2, c, cs, int(grad_cat.dtype == torch.float16),
                                                  L.ptr(gx), L.ptr(gs), L.stream()), "ts_shuffle_cat_rows_backward")
    return gx, gs


_conv3x3_takes = {}


def conv3x3_rows_takes(c_in, c_out):
    """whether csrc/conv2d_rows.hip's general kernel takes a Conv2d(c_in, c_out, 3, padding 1) forward AND backward"""
    key = (int(c_in), int(c_out))
    if key not in _conv3x3_takes:
        _conv3x3_takes[key] = _conv3x3_rows_takes(*key)
    return _conv3x3_takes[key]


def _conv3x3_rows_takes(c_in, c_out):
    lib = L.load()
    return (bool(lib.ts_conv3x3_rows_packed_bytes(int(c_in), int(c_out))) and bool(lib.ts_conv3x3_rows_packed_bytes(int(c_out), int(c_in)))
            and bool(lib.ts_conv3x3_wgrad_workspace_bytes(int(c_in), int(c_out))))


def conv3x3_rows_pack(weight, mode):
    """the packed MFMA operand of a Conv2d(c_in, c_out, 3) weight (half, any strides): mode 0 forward, 1 data gradient"""
    L.require_device(weight)
    if weight.dtype != torch.float16 or weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3):
        raise ValueError(f"conv3x3_rows_pack: a float16 [c_out, c_in, 3, 3] weight, got {weight.dtype} {tuple(weight.shape)}")
    lib = L.load()
    c_out, c_in = int(weight.shape[0]), int(weight.shape[1])
    n = lib.ts_conv3x3_rows_packed_bytes(c_in, c_out) if int(mode) == 0 else lib.ts_conv3x3_rows_packed_bytes(c_out, c_in)
    if n == 0:
        raise ValueError(f"conv3x3_rows_pack: the kernel does not take {c_in} -> {c_out} channels (mode {mode})")
    packed = torch.empty(n, dtype=torch.uint8, device=weight.device)
    s = weight.stride()
    L.check(lib.ts_conv3x3_rows_pack(L.ptr(weight), c_out, c_in, s[0], s[1], s[2], s[3], int(mode), L.ptr(packed), L.stream()),
            "ts_conv3x3_rows_pack")
    return packed


def conv3x3_rows(x, packed, bias, out_channels):
    """Conv2d(C, out_channels, 3, stride 1, padding 1) of a channels-last float16 [T, C, H, W] stack with a packed weight
    (conv3x3_rows_pack) and an optional float32 bias; with the data-gradient pack and x = grad_y: grad_x."""
    L.require_device(x, packed, bias)
    x = _channels_last_rows(x, "conv3x3_rows")
    t, c, h, w = x.shape
    lib = L.load()
    if x.dtype != torch.float16 or packed.numel() != lib.ts_conv3x3_rows_packed_bytes(c, int(out_channels)) or packed.numel() == 0:
        raise ValueError(f"conv3x3_rows: float16 rows and the packed operand of {c} -> {out_channels} channels")
    if bias is not None:
        bias = _f32(bias, "bias")
        if bias.numel() != int(out_channels):
            raise ValueError("conv3x3_rows: bias length")
    y = torch.empty((t, int(out_channels), h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    L.check(lib.ts_conv3x3_rows(L.ptr(x), c, L.ptr(packed), L.ptr(bias), t, h, w, L.ptr(y), int(out_channels), L.stream()),
            "ts_conv3x3_rows")
    return y


def conv1x1c32_pack(weight, mode):
    """the packed MFMA operand of a Conv2d(32, 32, 1) weight (half, any strides): mode 0 forward, 1 data gradient"""
    L.require_device(weight)
    if weight.dtype != torch.float16 or tuple(weight.shape) != (32, 32, 1, 1):
        raise ValueError(f"conv1x1c32_pack: a float16 [32, 32, 1, 1] weight, got {weight.dtype} {tuple(weight.shape)}")
    lib = L.load()
    packed = torch.empty(lib.ts_conv1x1c32_packed_bytes(), dtype=torch.uint8, device=weight.device)
    L.check(lib.ts_conv1x1c32_pack(L.ptr(weight), weight.stride(0), weight.stride(1), int(mode), L.ptr(packed), L.stream()), "ts_conv1x1c32_pack")
    return packed


def conv1x1c32_rows(x, packed, bias, slope=None):
    """Conv2d(32, 32, 1) of a channels-last float16 [T, 32, H, W] stack with a packed weight and an optional float32 bias, followed by
    LeakyReLU(slope) unless slope is None; with the data-gradient pack: grad_x (csrc/conv2d_rows.hip)"""
    L.require_device(x, packed, bias)
    x = _channels_last_rows(x, "conv1x1c32_rows")
    t, c, h, w = x.shape
    if c != 32 or x.dtype != torch.float16:
        raise ValueError(f"conv1x1c32_rows: float16 with 32 channels, got {x.dtype} with {c}")
    if bias is not None:
        bias = _f32(bias, "bias")
    y = torch.empty_like(x, memory_format=torch.channels_last)
    L.check(L.load().ts_conv1x1c32_rows(L.ptr(x), L.ptr(packed), L.ptr(bias), t * h * w, int(slope is not None), float(slope or 0.0), L.ptr(y),
                                        L.stream()), "ts_conv1x1c32_rows")
    return y


def conv1x1c32_wgrad(x, g, like_weight):
    """(grad_weight float16 like like_weight [32, 32, 1, 1], grad_bias float32 [32]) of Conv2d(32, 32, 1) from the channels-last float16
    input x and the gradient g at the layer's output"""
    L.require_device(x, g)
    x, g = _channels_last_rows(x, "conv1x1c32_wgrad(x)"), _channels_last_rows(g, "conv1x1c32_wgrad(g)")
    if x.shape != g.shape or x.shape[1] != 32 or x.dtype != torch.float16 or g.dtype != torch.float16 or tuple(like_weight.shape) != (32, 32, 1, 1):
        raise ValueError("conv1x1c32_wgrad: two float16 [T, 32, H, W] stacks of one shape and a [32, 32, 1, 1] weight")
    t, _, h, w = x.shape
    gw = torch.empty_like(like_weight, dtype=torch.float16)
    gb = torch.empty(32, dtype=torch.float32, device=x.device)
    lib = L.load()
    ws = L.workspace(lib.ts_conv3x3c32_wgrad_workspace_bytes(), x.device)
    L.check(lib.ts_conv1x1c32_wgrad(L.ptr(x), L.ptr(g), t, h, w, L.ptr(gw), gw.stride(0), gw.stride(1), L.ptr(gb), L.ptr(ws), ws.numel(), L.stream()),
            "ts_conv1x1c32_wgrad")
    return gw, gb


def conv3x3_wgrad(x, grad_y, like_weight, want_bias=True):
    """(grad_weight, grad_bias) of Conv2d(C_in, C_out, 3, stride 1, padding 1) from the channels-last float16 input x [T, C_in, H, W]
    and output gradient grad_y [T, C_out, H, W]: a float16 tensor with like_weight's shape and strides and a float32 [C_out] (None
    unless want_bias) - csrc/conv2d_rows.hip, deterministic"""
    L.require_device(x, grad_y)
    x, grad_y = _channels_last_rows(x, "conv3x3_wgrad(x)"), _channels_last_rows(grad_y, "conv3x3_wgrad(grad_y)")
    t, ci, h, w = x.shape
    co = grad_y.shape[1]
    lib = L.load()
    need = lib.ts_conv3x3_wgrad_workspace_bytes(ci, co)
    if (x.dtype != torch.float16 or grad_y.dtype != torch.float16 or tuple(grad_y.shape) != (t, co, h, w) or need == 0
            or tuple(like_weight.shape) != (co, ci, 3, 3)):
        raise ValueError(f"conv3x3_wgrad: float16 stacks {tuple(x.shape)} / {tuple(grad_y.shape)} and a weight like {tuple(like_weight.shape)}")
    gw = torch.empty_like(like_weight, dtype=torch.float16)          # (same strides as the weight)
    gb = torch.empty(co, dtype=torch.float32, device=x.device) if want_bias else None
    s = gw.stride()
    ws = L.workspace(need, x.device)
    L.check(lib.ts_conv3x3_wgrad(L.ptr(x), ci, L.ptr(grad_y), co, t, h, w, L.ptr(gw), s[0], s[1], s[2], s[3], L.ptr(gb), L.ptr(ws), ws.numel(),
                                 L.stream()), "ts_conv3x3_wgrad")
    return gw, gb


def conv3x3c32_wgrad(x, grad_y, like_weight, dilation, want_bias=False):
    """weight gradient of Conv2d(32, 32, 3, stride 1, padding = dilation) from the channels-last float16 input x and output gradient
    grad_y [T, 32, H, W]: a float16 tensor with like_weight's shape and strides (csrc/conv2d_rows.hip, deterministic); with want_bias
    the pair (grad_weight, grad_bias float32 [32]) - the column sums of grad_y come out of the same pass"""
    L.require_device(x, grad_y)
    x, grad_y = _channels_last_rows(x, "conv3x3c32_wgrad(x)"), _channels_last_rows(grad_y, "conv3x3c32_wgrad(grad_y)")
    if x.shape != grad_y.shape or x.shape[1] != 32 or x.dtype != torch.float16 or grad_y.dtype != torch.float16:
        raise ValueError("conv3x3c32_wgrad: two float16 [T, 32, H, W] stacks of one shape")
    t, _, h, w = x.shape
    gw = torch.empty_like(like_weight, dtype=torch.float16)          # (same strides as the weight)
    gb = torch.empty(32, dtype=torch.float32, device=x.device) if want_bias else None
    s = gw.stride()
    lib = L.load()
    ws = L.workspace(lib.ts_conv3x3c32_wgrad_workspace_bytes(), x.device)
    L.check(lib.ts_conv3x3c32_wgrad(L.ptr(x), L.ptr(grad_y), t, h, w, int(dilation), L.ptr(gw), s[0], s[1], s[2], s[3], L.ptr(gb), L.ptr(ws),
                                    ws.numel(), L.stream()), "ts_conv3x3c32_wgrad")
    return (gw, gb) if want_bias else gw


def set_conv_impl(impl):
    """0 = MFMA kernels (default: full-tile fp32 GEMMs as split-bf16 MFMAs), 1 = scalar cross-check kernels, 2 = guarded
    generic f32 MFMA staging only, 3 / 4 = f32 MFMA one workgroup per tile / persistent, 5 = f32 MFMA default choice."""
    global _conv_impl
    _conv_impl = int(impl)
    L.load().ts_set_conv_impl(int(impl))


def fuse_scan(points, pose0, pose):
    """fuse_multi_scan for one history scan (semantickitti_ms.py:403-417); points [n,4] f32."""
    L.require_device(points, pose0, pose)
    points, pose0, pose = _f32(points, "points"), _f32(pose0, "pose0"), _f32(pose, "pose")
    assert points.ndim == 2 and points.shape[1] == 4 and pose0.shape == (4, 4) and pose.shape == (4, 4)
    out = torch.empty_like(points)
    L.check(L.load().ts_fuse_scan(L.ptr(points), points.shape[0], L.ptr(pose0), L.ptr(pose), L.ptr(out), L.stream()),
            "ts_fuse_scan")
    return out


def fuse_scans(points, scan_idx, pose0, poses):
    """fuse_scan for the concatenated history scans of one sample: point i is transformed with poses[scan_idx[i]]."""
    L.require_device(points, scan_idx, pose0, poses)
    points, pose0, poses = _f32(points, "points"), _f32(pose0, "pose0"), _f32(poses, "poses")
    scan_idx = _i32(scan_idx, "scan_idx")
    assert points.ndim == 2 and points.shape[1] == 4 and pose0.shape == (4, 4) and poses.shape[1:] == (4, 4)
    out = torch.empty_like(points)
    L.check(L.load().ts_fuse_scans(L.ptr(points), L.ptr(scan_idx), points.shape[0], L.ptr(pose0), L.ptr(poses),
                                   poses.shape[0], L.ptr(out), L.stream()), "ts_fuse_scans")
    return out


def fuse_scans_batch(points, scan_idx, pose0s, poses):
    """fuse_scans for the history scans of a whole batch: point i is transformed with poses[scan_idx[i]] and the current-frame
    pose pose0s[scan_idx[i]] of its own sample (both [S, 4, 4])."""
    L.require_device(points, scan_idx, pose0s, poses)
    points, pose0s, poses = _f32(points, "points"), _f32(pose0s, "pose0s"), _f32(poses, "poses")
    scan_idx = _i32(scan_idx, "scan_idx")
    assert points.ndim == 2 and points.shape[1] == 4 and pose0s.shape == poses.shape and poses.shape[1:] == (4, 4)
    out = torch.empty_like(points)
    if points.shape[0]:
        L.check(L.load().ts_fuse_scans_batch(L.ptr(points), L.ptr(scan_idx), points.shape[0], L.ptr(pose0s), L.ptr(poses),
                                             poses.shape[0], L.ptr(out), L.stream()), "ts_fuse_scans_batch")
    return out


def fuse_sweeps(points, sweep_idx, params):
    """nuScenes multi-scan fuse for the concatenated selected sweeps of one sample (nuscenes_ms.py:280-318, 348-373):
    points [n,5] float32, sweep_idx [n] int32, params [S,28] float64 (taseg_amd.data.nuscenes.sweep_params).
    Returns (out [n,5] float32 = x', y', z', intensity, dt ; keep [n] bool = outside the ego box)."""
    L.require_device(points, sweep_idx, params)
    points, sweep_idx = _f32(points, "points"), _i32(sweep_idx, "sweep_idx")
    if params.dtype != torch.float64 or params.ndim != 2 or params.shape[1] != 28:
        raise TypeError("params must be float64 [S, 28]")
    params = params.contiguous()
    assert points.ndim == 2 and points.shape[1] == 5, points.shape
    out = torch.empty_like(points)
    keep = torch.empty(points.shape[0], dtype=torch.uint8, device=points.device)
    L.check(L.load().ts_fuse_sweeps(L.ptr(points), L.ptr(sweep_idx), points.shape[0], L.ptr(params), params.shape[0],
                                    L.ptr(out), L.ptr(keep), L.stream()), "ts_fuse_sweeps")
    return out, keep.bool()


def project_fov(points, proj, image_size, crop, row_offset=0.0):
    """TIAF camera projection of one scan (semantickitti_ms_mm.py:411-461): points [n,4] float32, proj [3,4] float64
    (P2 @ Tr), image_size = (width, height) of the camera image, crop = (HEIGHT, WIDTH) of the network input.
    Returns (pix [n,2] float32 = (row + row_offset, col), keep [n] bool)."""
    L.require_device(points, proj)
    points = _f32(points, "points")
    if proj.dtype != torch.float64 or tuple(proj.shape) != (3, 4):
        raise TypeError("proj must be float64 [3, 4]")
    proj = proj.contiguous()
    assert points.ndim == 2 and points.shape[1] == 4, points.shape
    n = points.shape[0]
    pix = torch.empty((n, 2), dtype=torch.float32, device=points.device)
    keep = torch.empty(n, dtype=torch.uint8, device=points.device)
    L.check(L.load().ts_project_fov(L.ptr(points), n, L.ptr(proj), int(image_size[0]), int(image_size[1]), int(crop[0]),
                                    int(crop[1]), float(row_offset), L.ptr(pix), L.ptr(keep), L.stream()), "ts_project_fov")
    return pix, keep.bool()


def project_cam(points, cam, image_size, crop_top, row_offset=0.0):
    """nuScenes TIAF camera projection (nuscenes_ms_mm.py:349-398): points [n,4] float32 in the lidar frame, cam [57] float64
    (taseg_amd.data.nuscenes_tiaf.camera_chain), image_size = (width, height) of the full-size camera image, crop_top rows cut
    off the half-resolution image.  Returns (pix [n,2] float32 = (row + row_offset, col) at half resolution, keep [n] bool)."""
    L.require_device(points, cam)
    points = _f32(points, "points")
    if cam.dtype != torch.float64 or cam.numel() != 57:
        raise TypeError("cam must be 57 float64 values")
    cam = cam.contiguous()
    assert points.ndim == 2 and points.shape[1] == 4, points.shape
    n = points.shape[0]
    pix = torch.empty((n, 2), dtype=torch.float32, device=points.device)
    keep = torch.empty(n, dtype=torch.uint8, device=points.device)
    L.check(L.load().ts_project_cam(L.ptr(points), n, L.ptr(cam), int(image_size[0]), int(image_size[1]), int(crop_top),
                                    float(row_offset), L.ptr(pix), L.ptr(keep), L.stream()), "ts_project_cam")
    return pix, keep.bool()


def voxel_coords(points, voxel_size, batch_idx=None, n_batch=1, shift=None):
    """int32(round(xyz / voxel_size)) minus the per-scan minimum (or a given shift).

    Returns (coords [n,4] int32 = x,y,z,b ; mins [n_batch,3] int32 actually subtracted).
    """
    L.require_device(points, batch_idx, shift)
    points = _f32(points, "points")
    n = points.shape[0]
    out = torch.empty((n, 4), dtype=torch.int32, device=points.device)
    mins = torch.empty((n_batch, 3), dtype=torch.int32, device=points.device) if shift is None else None
    if batch_idx is not None:
        batch_idx = _i32(batch_idx, "batch_idx")
    if shift is not None:
        shift = _i32(shift, "shift")
    L.check(L.load().ts_voxel_coords(L.ptr(points), n, points.shape[1], float(voxel_size), L.ptr(batch_idx),
                                     int(n_batch), L.ptr(shift), L.ptr(mins), L.ptr(out), L.stream()),
            "ts_voxel_coords")
    return out, (mins if shift is None else shift)


def segment_min3(points, seg, n_seg):
    """[n_seg, 3] float32: per-segment minimum of the first three columns of points [n, F]; seg [n] int64 segment index"""
    L.require_device(points, seg)
    points = _f32(points, "points")
    if seg.dtype != torch.int64:
        seg = seg.long()
    seg = seg.contiguous()
    out = torch.empty((int(n_seg), 3), dtype=torch.float32, device=points.device)
    L.check(L.load().ts_segment_min3(L.ptr(points), points.shape[0], points.shape[1], L.ptr(seg), int(n_seg), L.ptr(out),
                                     L.stream()), "ts_segment_min3")
    return out


def stage_keep_flags(points, scan_idx, cls, table, sample_of_scan, lo, pre_keep=None, neg_col=-1):
    """csrc/stage.hip: (keep [n] bool, sample [n] int64) of the history points of a batch - class-step table AND the optional
    pre-filter AND the clamp to the sample's current-scan minimum.  points [n, F] float32, scan_idx [n] int32, cls [n] int64,
    table [S, C] bool, sample_of_scan [S] int64, lo [B, 3] float32."""
    L.require_device(points, scan_idx, cls, table, sample_of_scan, lo, pre_keep)
    points, scan_idx, lo = _f32(points, "points"), _i32(scan_idx, "scan_idx"), _f32(lo, "lo")
    cls, sample_of_scan = cls.long().contiguous(), sample_of_scan.long().contiguous()
    table = table.contiguous()
    if table.dtype not in (torch.bool, torch.uint8) or table.ndim != 2:
        raise TypeError("table must be a bool / uint8 matrix")
    if pre_keep is not None:
        pre_keep = pre_keep.contiguous()
        if pre_keep.dtype not in (torch.bool, torch.uint8):
            raise TypeError("pre_keep must be bool / uint8")
    n = points.shape[0]
    keep = torch.empty(n, dtype=torch.bool, device=points.device)
    sample = torch.empty(n, dtype=torch.int64, device=points.device)
    L.check(L.load().ts_stage_keep_flags(L.ptr(points), n, points.shape[1], L.ptr(pre_keep), L.ptr(scan_idx), L.ptr(cls),
                                         L.ptr(table), table.shape[0], table.shape[1], int(neg_col), L.ptr(sample_of_scan),
                                         L.ptr(lo), lo.shape[0], L.ptr(keep), L.ptr(sample), L.stream()), "ts_stage_keep_flags")
    return keep, sample


def stage_layout(cur, cur_lab, cur_b, hist, hist_lab, hist_b, idx, cur_start, kept_start):
    """csrc/stage.hip: the fused clouds of a batch sample-major, current scan first -> (pts [Nc + Nk, F], labels int64, sample
    int64, sample int32, is_current bool)."""
    L.require_device(cur, cur_lab, cur_b, hist, hist_lab, hist_b, idx, cur_start, kept_start)
    cur, hist = _f32(cur, "cur"), _f32(hist, "hist")
    n_cur, n_kept, f = cur.shape[0], idx.shape[0], cur.shape[1]
    assert hist.shape[1] == f and cur_lab.dtype == hist_lab.dtype == torch.int64
    dev = cur.device
    pts = torch.empty((n_cur + n_kept, f), dtype=torch.float32, device=dev)
    lab = torch.empty(n_cur + n_kept, dtype=torch.int64, device=dev)
    sample = torch.empty(n_cur + n_kept, dtype=torch.int64, device=dev)
    sample32 = torch.empty(n_cur + n_kept, dtype=torch.int32, device=dev)
    is_cur = torch.empty(n_cur + n_kept, dtype=torch.bool, device=dev)
    L.check(L.load().ts_stage_layout(L.ptr(cur), L.ptr(cur_lab.contiguous()), L.ptr(cur_b.contiguous()), n_cur, L.ptr(hist),
                                     L.ptr(hist_lab.contiguous()), L.ptr(hist_b.contiguous()), L.ptr(idx.contiguous()), n_kept, f,
                                     L.ptr(cur_start.contiguous()), L.ptr(kept_start.contiguous()), L.ptr(pts), L.ptr(lab),
                                     L.ptr(sample), L.ptr(sample32), L.ptr(is_cur), L.stream()), "ts_stage_layout")
    return pts, lab, sample, sample32, is_cur


def stage_split_voxels(coords4, index, inverse, row_sample, n_samples):
    """csrc/stage.hip: after sparse_quantize on a whole batch -> (vox [m, 4] int32 = coords4[index], offset [B] int32 cumulative
    voxel counts, inverse_local [n] int64 = voxel index inside the point's own sample)."""
    L.require_device(coords4, index, inverse, row_sample)
    coords4, index, inverse = _i32(coords4, "coords4"), _i32(index, "index"), _i32(inverse, "inverse")
    row_sample = row_sample.long().contiguous()
    m, n = index.shape[0], inverse.shape[0]
    dev = coords4.device
    vox = torch.empty((m, 4), dtype=torch.int32, device=dev)
    start = torch.empty(n_samples + 1, dtype=torch.int64, device=dev)
    offset = torch.empty(n_samples, dtype=torch.int32, device=dev)
    local = torch.empty(n, dtype=torch.int64, device=dev)
    L.check(L.load().ts_stage_split_voxels(L.ptr(coords4), L.ptr(index), m, L.ptr(inverse), L.ptr(row_sample), n, int(n_samples),
                                           L.ptr(vox), L.ptr(start), L.ptr(offset), L.ptr(local), L.stream()),
            "ts_stage_split_voxels")
    return vox, offset, local


def sparse_quantize(coords):
    """np.unique-style voxel grouping of int coords [n,4]: (index [m], inverse [n]) int32, m via one sync."""
    L.require_device(coords)
    coords = _i32(coords, "coords")
    n = coords.shape[0]
    lib = L.load()
    ws = L.workspace(lib.ts_quantize_workspace_bytes(n), coords.device)
    index = torch.empty(max(n, 1), dtype=torch.int32, device=coords.device)
    inverse = torch.empty(max(n, 1), dtype=torch.int32, device=coords.device)
    cnt = torch.empty(1, dtype=torch.int32, device=coords.device)
    L.check(lib.ts_sparse_quantize(L.ptr(coords), n, L.ptr(index), L.ptr(inverse), L.ptr(cnt), L.ptr(ws),
                                   ws.numel(), L.stream()), "ts_sparse_quantize")
    m = _count_to_host(cnt, "sparse_quantize")
    return index[:m], inverse[:n]


def _tile_cols(c):
    return 128 if c % 128 == 0 else 96 if c % 96 == 0 else 64 if c % 64 == 0 else 32


def conv_class_plan(nbr, groups=3, direct=False):
    """Plan of the class-sorted implicit GEMM (csrc/conv_class.hip) for the table nbr [K, n] (input row feeding destination row j
    through offset k, or -1): the K offsets in `groups` groups of <= 9, destination rows sorted by their neighbour mask per group.
    groups = 3: submanifold 3x3x3 maps (build_kmap with in == out), pass 2 adds the three group rows through plan["pos"];
    direct (groups = 1): 2x2x2 strided maps, the sums are stored straight into the rows plan["rows"] names - no Z, no pass 2.
    Returns dict(src [K / groups, m_pad], tile_info [m_pad / 128, 2], n_tiles [2] (device: listed tiles, (tile, offset) steps),
    pos | rows, m_pad, n, K, groups, mirror).  No host sync."""
    L.require_device(nbr)
    nbr = _i32(nbr, "nbr")
    k, n = nbr.shape
    if direct:
        groups = 1
    lib = L.load()
    m_pad = int(lib.ts_conv_class_rows2(n, groups))
    dev = nbr.device
    src = torch.empty((k // groups, m_pad), dtype=torch.int32, device=dev)
    tile_info = torch.empty((max(m_pad // 128, 1), 2), dtype=torch.int32, device=dev)
    n_tiles = torch.empty(2, dtype=torch.int32, device=dev)          # (listed tiles, (tile, offset) steps)
    pos = None if direct else torch.empty((groups, n), dtype=torch.int32, device=dev)
    rows = torch.empty(m_pad, dtype=torch.int32, device=dev) if direct else None
    ws = L.workspace(lib.ts_conv_class_plan_workspace_bytes(n), dev)
    L.check(lib.ts_conv_class_plan(L.ptr(nbr), n, k, groups, L.ptr(src), L.ptr(tile_info), L.ptr(n_tiles), L.ptr(pos), L.ptr(rows),
                                   L.ptr(ws), ws.numel(), L.stream()), "ts_conv_class_plan")
    # mirror: the transposed product takes the slice of offset K-1-k (the submanifold map is its own transpose with the offsets
    # reversed); direct plans are built per direction
    return dict(src=src, tile_info=tile_info, n_tiles=n_tiles, pos=pos, rows=rows, m_pad=m_pad, n=n, K=k, groups=groups,
                mirror=0 if direct else 1, z_rows=0, map_id=None)


def conv_class_plan_pairs(nbmaps, nboffs, k, n_pairs):
    """Direct plan of the one-pair-per-destination direction of a strided map (destination = the map's INPUT rows: transposed
    forward, strided input gradient) straight from its rulebook - no sort, 2 launches.  n_pairs must equal the number of
    destination rows (every input row in exactly one pair)."""
    L.require_device(nbmaps, nboffs)
    nbmaps, nboffs = _i32(nbmaps, "nbmaps"), _i32(nboffs, "nboffs")
    lib = L.load()
    n = int(n_pairs)
    m_pad = int(lib.ts_conv_class_rows2(n, 1))
    dev = nbmaps.device
    src = torch.empty((k, m_pad), dtype=torch.int32, device=dev)
    tile_info = torch.empty((max(m_pad // 128, 1), 2), dtype=torch.int32, device=dev)
    n_tiles = torch.empty(2, dtype=torch.int32, device=dev)
    rows = torch.empty(m_pad, dtype=torch.int32, device=dev)
    ws = L.workspace(4 * (m_pad // 128) + 256, dev)
    L.check(lib.ts_conv_class_plan_pairs(L.ptr(nbmaps), L.ptr(nboffs), k, n, L.ptr(src), L.ptr(tile_info), L.ptr(n_tiles),
                                         L.ptr(rows), L.ptr(ws), ws.numel(), L.stream()), "ts_conv_class_plan_pairs")
    return dict(src=src, tile_info=tile_info, n_tiles=n_tiles, pos=None, rows=rows, m_pad=m_pad, n=n, K=k, groups=1, mirror=0,
                z_rows=0, map_id=None)


def conv_nbr_transposed(pos_in, nbmaps, k):
    """nbr_t [K, n_in]: the output row fed by input row i through offset k, or -1 (the table of a kernel map's transposed use)"""
    L.require_device(pos_in, nbmaps)
    pos_in, nbmaps = _i32(pos_in, "pos_in"), _i32(nbmaps, "nbmaps")
    n_in = pos_in.shape[1]
    out = torch.empty((k, n_in), dtype=torch.int32, device=pos_in.device)
    L.check(L.load().ts_conv_nbr_transposed(L.ptr(pos_in), L.ptr(nbmaps), k, n_in, L.ptr(out), L.stream()),
            "ts_conv_nbr_transposed")
    return out


def class_plan_struct(plan):
    """the TsClassPlan (include/taseg_hip.h) of a plan dict, cached in it (the dict keeps the tensors alive)"""
    st = plan.get("_struct")
    if st is None:
        mid = plan.get("map_id")
        st = L.TsClassPlan(L.ptr(plan["src"]), L.ptr(plan["tile_info"]), L.ptr(plan["n_tiles"]), L.ptr(plan["pos"]),
                           L.ptr(plan["rows"]), plan["n"], plan["m_pad"], int(plan.get("z_rows") or 0), plan["K"], plan["groups"],
                           plan["mirror"], None if mid is None else mid.data_ptr())
        plan["_struct"] = st
    return st


def conv_class_gemm(feat, kernel, plan, weight_transposed=False):
    """The product on a class plan.  Pass-2 plans: z' [m_pad, C] with one row per (destination row, group of offsets); the
    convolution is conv_gather_sum(z', plan["pos"], n).  Direct plans: the result [n, C] itself.  weight_transposed: the
    transposed product (feat = output gradients, kernel as stored)."""
    L.require_device(feat, kernel)
    feat, kernel = _f32(feat, "feat"), _f32(kernel, "kernel")
    k, c_in, c_out = kernel.shape
    c_red, cols = (c_out, c_in) if weight_transposed else (c_in, c_out)
    if feat.shape[1] != c_red:
        raise ValueError(f"conv_class_gemm: feat has {feat.shape[1]} channels, the product reduces over {c_red}")
    direct = plan["rows"] is not None
    zp = torch.empty((plan["n"] if direct else plan["m_pad"], cols), dtype=torch.float32, device=feat.device)
    with _Timed("class_gemm", name=f"class_gemm_kernel<{_tile_cols(cols)}>", pairs=int(plan.get("pairs", 0)), c_red=c_red,
                c_out=cols, k=k, esize=4, n_rows=feat.shape[0], z_rows=0 if direct else int(plan.get("z_rows") or plan["m_pad"]),
                out_rows=plan["n"] if direct else 0):
        L.check(L.load().ts_conv_class_gemm(L.ptr(feat), c_red, L.ptr(kernel), k, plan["groups"], cols, L.ptr(plan["src"]),
                                            plan["m_pad"], L.ptr(plan["tile_info"]), L.ptr(plan["n_tiles"]),
                                            1 if weight_transposed else 0, plan["mirror"], L.ptr(plan["rows"]), L.ptr(zp),
                                            L.stream()), "ts_conv_class_gemm")
    return zp


def _z_moves(plan):
    """Z' rows the finish-in-the-product form moves: the two outer groups' rows are written and read back once"""
    z = int(plan.get("z_rows") or plan["m_pad"])
    return 2 * max(0, z - (plan["n"] + 127) // 128 * 128)


def class_finish_pays(n, half=False):
    """does the finish inside the product (conv_class_conv) beat class GEMM + pass 2 on a map of n rows"""
    return bool(L.load().ts_conv_class_finish_pays(int(n), 1 if half else 0))


def conv_class_conv(feat, kernel, plan, weight_transposed=False, addend=None):
    """The whole convolution on a three-group class plan: out [n, C] (+ addend) in two launches of the product - the centre
    group's tiles add the outer groups' rows themselves; the same bits as conv_gather_sum(conv_class_gemm(...), plan["pos"], n)."""
    L.require_device(feat, kernel)
    feat, kernel = _f32(feat, "feat"), _f32(kernel, "kernel")
    k, c_in, c_out = kernel.shape
    c_red, cols = (c_out, c_in) if weight_transposed else (c_in, c_out)
    if feat.shape[1] != c_red:
        raise ValueError(f"conv_class_conv: feat has {feat.shape[1]} channels, the product reduces over {c_red}")
    if plan["rows"] is not None or plan["groups"] != 3:
        raise ValueError("conv_class_conv: a three-group pass-2 plan is needed")
    if addend is not None:
        addend = _f32(addend, "addend")
        if addend.shape != (plan["n"], cols):
            raise ValueError(f"conv_class_conv: addend shape {tuple(addend.shape)} != {(plan['n'], cols)}")
    zp = torch.empty((plan["m_pad"], cols), dtype=torch.float32, device=feat.device)
    out = torch.empty((plan["n"], cols), dtype=torch.float32, device=feat.device)
    with _Timed("class_gemm", name=f"class_gemm_kernel<{_tile_cols(cols)}>", pairs=int(plan.get("pairs", 0)), c_red=c_red,
                c_out=cols, k=k, esize=4, n_rows=feat.shape[0], z_rows=_z_moves(plan), out_rows=plan["n"]):
        L.check(L.load().ts_conv_class_conv(L.ptr(feat), c_red, L.ptr(kernel), k, 3, cols, L.ptr(plan["src"]), plan["m_pad"],
                                            L.ptr(plan["tile_info"]), L.ptr(plan["n_tiles"]), 1 if weight_transposed else 0,
                                            plan["mirror"], L.ptr(plan["pos"]), plan["n"], L.ptr(addend), L.ptr(zp), L.ptr(out),
                                            L.stream()), "ts_conv_class_conv")
    return out


def conv_class_conv_f16(feat, w16, plan, weight_transposed=False, addend=None):
    """conv_class_conv for IEEE-half rows (fp32 sums, the result rounded once)."""
    L.require_device(feat, w16)
    if feat.dtype != torch.float16 or w16.dtype != torch.float16 or (addend is not None and addend.dtype != torch.float16):
        raise TypeError("conv_class_conv_f16: half tensors expected")
    feat, w16 = feat.contiguous(), w16.contiguous()
    k, c_in, c_out = w16.shape
    c_red, cols = (c_out, c_in) if weight_transposed else (c_in, c_out)
    if feat.shape[1] != c_red:
        raise ValueError(f"conv_class_conv_f16: feat has {feat.shape[1]} channels, the product reduces over {c_red}")
    if plan["rows"] is not None or plan["groups"] != 3:
        raise ValueError("conv_class_conv_f16: a three-group pass-2 plan is needed")
    if addend is not None:
        addend = addend.contiguous()
        if addend.shape != (plan["n"], cols):
            raise ValueError(f"conv_class_conv_f16: addend shape {tuple(addend.shape)} != {(plan['n'], cols)}")
    zp = torch.empty((plan["m_pad"], cols), dtype=torch.float16, device=feat.device)
    out = torch.empty((plan["n"], cols), dtype=torch.float16, device=feat.device)
    with _Timed("class_gemm", name=f"class_gemm_kernel<{_tile_cols(cols)}>", pairs=int(plan.get("pairs", 0)), c_red=c_red,
                c_out=cols, k=k, esize=2, n_rows=feat.shape[0], z_rows=_z_moves(plan), out_rows=plan["n"]):
        L.check(L.load().ts_conv_class_conv_f16(L.ptr(feat), c_red, L.ptr(w16), k, 3, cols, L.ptr(plan["src"]), plan["m_pad"],
                                                L.ptr(plan["tile_info"]), L.ptr(plan["n_tiles"]), 1 if weight_transposed else 0,
                                                plan["mirror"], L.ptr(plan["pos"]), plan["n"], L.ptr(addend), L.ptr(zp),
                                                L.ptr(out), L.stream()), "ts_conv_class_conv_f16")
    return out


def conv_class_gemm_f16(feat, w16, plan, weight_transposed=False):
    """conv_class_gemm for IEEE-half rows: w16 = the half weight [K, C_in, C_out] as stored; result half (fp32 sums)."""
    L.require_device(feat, w16)
    if feat.dtype != torch.float16 or w16.dtype != torch.float16:
        raise TypeError("conv_class_gemm_f16: half tensors expected")
    feat, w16 = feat.contiguous(), w16.contiguous()
    k, c_in, c_out = w16.shape
    c_red, cols = (c_out, c_in) if weight_transposed else (c_in, c_out)
    if feat.shape[1] != c_red:
        raise ValueError(f"conv_class_gemm_f16: feat has {feat.shape[1]} channels, the product reduces over {c_red}")
    direct = plan["rows"] is not None
    zp = torch.empty((plan["n"] if direct else plan["m_pad"], cols), dtype=torch.float16, device=feat.device)
    with _Timed("class_gemm", name=f"class_gemm_kernel<{_tile_cols(cols)}>", pairs=int(plan.get("pairs", 0)), c_red=c_red,
                c_out=cols, k=k, esize=2, n_rows=feat.shape[0], z_rows=0 if direct else int(plan.get("z_rows") or plan["m_pad"]),
                out_rows=plan["n"] if direct else 0):
        L.check(L.load().ts_conv_class_gemm_f16(L.ptr(feat), c_red, L.ptr(w16), k, plan["groups"], cols, L.ptr(plan["src"]),
                                                plan["m_pad"], L.ptr(plan["tile_info"]), L.ptr(plan["n_tiles"]),
                                                1 if weight_transposed else 0, plan["mirror"], L.ptr(plan["rows"]), L.ptr(zp),
                                                L.stream()), "ts_conv_class_gemm_f16")
    return zp
