"""`torchsparse.nn.functional` on the HIP backend.

Operator names, argument meaning and results follow the reference
(TS/torchsparse/nn/functional/{conv,hash,query,count,voxelize,devoxelize,downsample}.py);
what differs is how the work is scheduled on the device:

* the kernel map of a convolution is built by ONE backend call (`ts_build_kmap`) that
  hashes, probes and compacts on device - no `torch.nonzero`, no `nbsizes.cpu()`;
* the convolution itself runs on the [K, N_out] neighbour table (output-stationary,
  every output row written once) instead of 3 x K gather / GEMM / scatter launches;
* `nbmaps` / `nbsizes` with the reference's exact contents and order are still produced
  (and are what the weight-gradient kernel walks), so rulebooks can be compared bit for bit.
"""
import os
from typing import List, Optional, Tuple, Union

import torch
from torch.autograd import Function

from ... import backend as B
from ... import planes as _planes
from ...options import options
from ..tensor import SparseTensor
from ..utils import make_ntuple
from .utils import get_kernel_offsets

__all__ = ["conv3d", "conv_geometry", "conv_block_ok", "conv_block_eval", "sphash", "sphashquery", "spcount", "spvoxelize", "spdevoxelize", "spdevoxelize_cat", "calc_ti_weights",
           "spdownsample", "KernelMap", "build_kernel_map", "build_pyramid", "point_linear"]

_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = torch.amp.custom_bwd(device_type="cuda")


# ------------------------------------------------------------------------------ hashing
def sphash(coords: torch.Tensor, offsets: Optional[torch.Tensor] = None) -> torch.Tensor:
    """hash.py:10-37: int64 FNV hash of [N,4] int coords, or the [K,N] table for K offsets."""
    assert coords.dtype == torch.int, coords.dtype
    assert coords.ndim == 2 and coords.shape[1] == 4, coords.shape
    if offsets is None:
        return B.hash_cuda(coords)
    assert offsets.dtype == torch.int, offsets.dtype
    assert offsets.ndim == 2 and offsets.shape[1] == 3, offsets.shape
    return B.kernel_hash_cuda(coords, offsets)


def sphashquery(queries: torch.Tensor, references: torch.Tensor) -> torch.Tensor:
    """query.py:8-33: position of every query hash in `references`, -1 where absent."""
    shape = queries.shape
    out = B.hash_query_cuda(queries.reshape(-1), references, None)
    return (out - 1).view(*shape)


def spcount(coords: torch.Tensor, num) -> torch.Tensor:
    """count.py:8-16."""
    return B.count_cuda(coords, int(num))


# ------------------------------------------------------------------------------ voxelize / devoxelize
class _Voxelize(Function):
    """fp32 kernels; under autocast / for half features the result is stored in half like the reference's
    `custom_fwd(cast_inputs=torch.half)` op (voxelize.py:13) - accumulation stays fp32."""

    @staticmethod
    def forward(ctx, feats, idx, counts):
        half = _amp_half(feats)
        idx = idx.int().contiguous()
        counts = counts.int().contiguous()
        out = B.voxelize_forward_cuda(feats.contiguous().float(), idx, counts)
        ctx.saved = (idx, counts, feats.shape[0], feats.dtype)
        return out.half() if half else out

    @staticmethod
    def backward(ctx, grad_out):
        idx, counts, n, dtype = ctx.saved
        return B.voxelize_backward_cuda(grad_out.contiguous().float(), idx, counts, n).to(dtype), None, None


def spvoxelize(feats: torch.Tensor, coords: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
    """voxelize.py:54-56: mean-pool point rows into voxel rows (`coords` = voxel index per point)."""
    return _Voxelize.apply(feats, coords, counts)


class _Devoxelize(Function):
    """fp32 kernels; half result under autocast / for half features (devoxelize.py:54), fp32 accumulation.  Half features
    with an inverse map or cell plan as `order` go through the half-storage kernels (same bits, no cast passes)."""

    @staticmethod
    def forward(ctx, feats, idx, weights, order=None):
        half = _amp_half(feats)
        idx = idx.int().contiguous()
        weights = weights.contiguous().float()
        stored_half = half and feats.dtype == torch.float16 and isinstance(order, tuple) and feats.shape[1] % 4 == 0 \
            and feats.shape[1] <= 1024
        ctx.saved = (idx, weights, feats.shape[0], order, feats.dtype, stored_half)
        if stored_half:
            out = torch.empty((idx.shape[0], feats.shape[1]), dtype=torch.float16, device=feats.device)
            B.devoxelize_forward_into(feats.contiguous(), idx, weights, out, 0)
            return out
        out = B.devoxelize_forward_cuda(feats.contiguous().float(), idx, weights)
        return out.half() if half else out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weights, m, order, dtype, stored_half = ctx.saved
        if stored_half and grad_out.dtype == torch.float16:
            g = grad_out.contiguous()
            return B.devoxelize_backward_from(g, 0, g.shape[1], idx, weights, m, order).to(dtype), None, None, None
        grad_out = grad_out.contiguous().float()
        if isinstance(order, tuple) and grad_out.shape[1] % 4 == 0 and grad_out.shape[1] <= 1024:
            # `order` is the inverse map (offsets, entries) of backend.devox_csr: gather per voxel, no atomics
            return B.devoxelize_backward_csr(grad_out, weights, order, m).to(dtype), None, None, None
        if isinstance(order, tuple):
            order = None
        if grad_out.shape[1] % 4 == 0 and grad_out.shape[1] <= 1024:
            return B.devoxelize_backward_runs(grad_out, idx, weights, m, order).to(dtype), None, None, None
        return B.devoxelize_backward_cuda(grad_out, idx, weights, m).to(dtype), None, None, None


def spdevoxelize(feats: torch.Tensor, coords: torch.Tensor, weights: torch.Tensor,
                 order: Optional[torch.Tensor] = None) -> torch.Tensor:
    """devoxelize.py:96-98: out[i] = sum_k weights[i,k] * feats[coords[i,k]].  `order` (optional, from
    `backend.devox_order`) only schedules the backward pass: points of the same interpolation cell are walked
    together so their contributions reach the voxel gradient as one atomic per run."""
    return _Devoxelize.apply(feats, coords, weights, order)


class _DevoxelizeCat(Function):
    """torch.cat([spdevoxelize(f_i, idx_i, w_i) for i], dim=1) as one node: every source is interpolated straight into its
    column block of the [N, sum C_i] result and the backward pass reads the gradient blocks in place - no concatenation
    copy, no contiguous copies of gradient slices (MinkUNet's z1 | z2 | z3 -> class head, minkunet.py:419-421).
    Half features (the AMP path) stay half on both sides of the kernels - float32 sums, one rounding, the bits of the
    float32 kernels between `.float()` and `.half()` - when every source has an inverse map or a cell plan."""

    @staticmethod
    def forward(ctx, maps, *feats):
        half = _amp_half(feats[0])
        n = maps[0][0].shape[0]
        cs = [f.shape[1] for f in feats]
        # (a backward pass through the half kernels walks the inverse maps / cell plans; a forward-only pass needs none)
        stored_half = half and all(f.dtype == torch.float16 for f in feats) \
            and (all(isinstance(o, tuple) for _, _, o in maps) or not any(ctx.needs_input_grad[1:]))
        out = torch.empty((n, sum(cs)), dtype=torch.float16 if stored_half else torch.float32, device=feats[0].device)
        col = 0
        for f, (idx, w, _order), c in zip(feats, maps, cs):
            B.devoxelize_forward_into(f.contiguous() if stored_half else f.contiguous().float(), idx, w, out, col)
            col += c
        ctx.maps, ctx.cs = maps, cs
        ctx.rows = [f.shape[0] for f in feats]
        ctx.dtypes = [f.dtype for f in feats]
        ctx.stored_half = stored_half
        return out.half() if half and not stored_half else out

    @staticmethod
    def backward(ctx, grad_out):
        g = grad_out.contiguous()
        if not (ctx.stored_half and g.dtype == torch.float16):
            g = g.float()
        grads, col = [], 0
        for i, ((idx, w, order), c) in enumerate(zip(ctx.maps, ctx.cs)):
            if ctx.needs_input_grad[1 + i]:
                grads.append(B.devoxelize_backward_from(g, col, c, idx, w, ctx.rows[i], order).to(ctx.dtypes[i]))
            else:
                grads.append(None)
            col += c
        return (None, *grads)


def spdevoxelize_cat(feats, maps) -> torch.Tensor:
    """Concatenation along the channels of spdevoxelize(feats[i], *maps[i][:2]); maps[i] = (coords [N, 8], weights
    [N, 8], order | inverse map | None).  Channel counts must be multiples of 4 (else: the plain ops)."""
    if all(f.is_cuda and f.shape[1] % 4 == 0 for f in feats):
        return _DevoxelizeCat.apply([(i.int().contiguous(), w.contiguous().float(), o) for i, w, o in maps], *feats)
    return torch.cat([spdevoxelize(f, i, w, o) for f, (i, w, o) in zip(feats, maps)], dim=1)


def calc_ti_weights(coords: torch.Tensor, idx_query: torch.Tensor, scale: float = 1) -> torch.Tensor:
    """devoxelize.py:10-48: trilinear weights [8, N] for float point coords and the [8, N] corner
    lookup result, masked where the corner voxel is absent and renormalised by (sum + 1e-8).

    Kept for API parity (elementwise torch ops); `voxel_to_point` uses the fused
    `backend.trilinear_map` which produces indices and weights in one pass.
    """
    with torch.no_grad():
        p = coords[:, :3]
        lo = torch.floor(p / scale) * scale if scale != 1 else torch.floor(p)
        hi = lo + scale
        up, dn = (hi - p).float(), (p - lo).float()   # weight of the lower / upper corner per axis
        sel = lambda bit, ax: (dn if bit else up)[:, ax]  # noqa: E731
        rows = [sel(k >> 2 & 1, 0) * sel(k >> 1 & 1, 1) * sel(k & 1, 2) for k in range(8)]
        w = torch.stack(rows, dim=0)
        if scale != 1:
            w /= scale ** 3
        w[idx_query == -1] = 0
        w /= torch.sum(w, dim=0) + 1e-8
    return w


# ------------------------------------------------------------------------------ downsample
def spdownsample(coords: torch.Tensor, stride=2, kernel_size=2, tensor_stride=1) -> torch.Tensor:
    """downsample.py:11-52: output coordinates of a strided convolution."""
    stride = make_ntuple(stride, ndim=3)
    kernel_size = make_ntuple(kernel_size, ndim=3)
    tensor_stride = make_ntuple(tensor_stride, ndim=3)
    if not all(stride[k] in (1, kernel_size[k]) for k in range(3)):
        # the overlapping-window branch (downsample.py:32-48) is never reached by the
        # MinkUNet family (kernel 2 / stride 2 and kernel 3 / stride 1 only)
        raise NotImplementedError("spdownsample: stride must be 1 or equal to kernel_size on every axis")
    step = [stride[k] * tensor_stride[k] for k in range(3)]
    return B.downsample(coords, step)


# ------------------------------------------------------------------------------ kernel map
class KernelMap:
    """What `input.kmaps[key]` holds.  Indexing / unpacking reproduces the reference's
    `[nbmaps, nbsizes, (n_in, n_out)]` list (conv.py:175-177); the device-resident neighbour
    tables are what the kernels consume.

      nbr    [K, n_out] int32   input row feeding output j through offset k (or -1)
      nbr_t  [K, n_in]  int32   output row fed by input i through offset k (or -1)
      nbmaps_buf [K*n_out, 2]   (in, out) pairs ordered by (k, out); first nboffs[K] rows valid
      nbsizes [K], nboffs [K+1] int32 on device
    """

    def __init__(self, tables, sizes):
        self.nbr = tables["nbr"]
        self.nbmaps_buf = tables["nbmaps"]
        self.nbsizes_dev = tables["nbsizes"]
        self.nboffs = tables["nboffs"]
        self.pos_out = tables["pos_out"]      # [K, n_out] row of nbmaps per (offset, output voxel) or -1
        self.pos_in = tables["pos_in"]        # [K, n_in]  row of nbmaps per (offset, input voxel) or -1
        self.sizes = sizes
        self._nbmaps = None
        self._total = None
        # duplicated coordinates in a map over ONE coordinate set (`dup`): an input row then has TWO pairs at one offset, which the
        # position table pos_in [K, n_in] of the list-form input gradient cannot hold (and the map is not its own transpose, which
        # the class plans' input gradient relies on) - such a map takes the scatter form of the input gradient and no plans
        self._dup_dev = tables.get("dup")     # device flag (None: a map between two coordinate sets), read with the pair total
        self._dup = None if self._dup_dev is not None else False
        self.cls = None                       # plan of the class-sorted implicit GEMM (csrc/conv_class.hip), large 3x3x3 maps
        self._cls_pending = None              # ... launched, not yet judged (accept_class_plans)
        self.direct = None                    # {"down", "up"}: direct plans of a 2x2x2 strided map (no Z, no pass 2)
        self._plans = {}                      # plans_for() answers (dropped when a plan is built)

    def build_class_plan(self, defer=False):
        """Plan of the class-sorted implicit GEMM for a SUBMANIFOLD 3x3x3 map (in == out; the caller knows): rows sorted by
        the neighbour mask of each z-plane of offsets.  Built with the map, on the stream that builds it; maps too small to
        gain (class_gemm_pays) go without.  defer=True only launches the builder: `accept_class_plans` then judges the plans of
        several maps with ONE host read (an index plan has three such maps)."""
        if (_CLASS_GEMM and self.cls is None and self._cls_pending is None and self.nbr.shape[0] == 27
                and self.sizes[0] == self.sizes[1] and self.sizes[0] >= _CLASS_MIN_ROWS and not self.dup):
            self._cls_pending = B.conv_class_plan(self.nbr)
        if not defer:
            accept_class_plans([self])
        return self.cls

    def _accept(self, tiles, steps):
        # is the plan worth walking?  Rows with LiDAR-like neighbour masks sort into near-uniform tiles (128 * steps ~ 1.1 P);
        # rows with unrelated masks would make every tile walk all nine offsets of its group with most rows absent (up to 3.7 P
        # row-products): two passes then
        cls, self._cls_pending = self._cls_pending, None
        cls["z_rows"], cls["steps"] = 128 * tiles, steps
        cls["map_id"], cls["pairs"] = self.nboffs, self.total          # the identity of the map the plan belongs to
        if 128 * steps <= _CLASS_MAX_WORK * self.total:
            self.cls = cls
            self._plans.clear()

    def build_direct_plans(self):
        """The two DIRECT plans of a 2x2x2 strided map (csrc/conv_class.hip): every destination row holds all its offsets in one
        tile, so the class GEMM stores the result rows themselves - no Z, no pass 2.  "down": destination = the map's output
        (coarse) rows, fed by its input (fine) rows - the strided forward and the transposed convolution's input gradient;
        "up": destination = the fine rows (each has exactly one pair) - the transposed forward and the strided
        convolution's input gradient (convolution_cuda.cu:21,34: transpose swaps the map's columns).  No host read."""
        k = self.nbr.shape[0]
        if _DIRECT_CONV and self.direct is None and k <= 9 and min(self.sizes) >= _DIRECT_MIN_ROWS and self.total > 0:
            n_in, n_out = self.sizes
            # "up": every input row sits in exactly one pair (total == n_in): the plan IS the rulebook order, no sort (2 launches);
            # otherwise the sorted builder on the inverse table.  "down" is only ever chosen above _DIRECT_DOWN_MIN_ROWS
            # destination rows (direct_conv_pays): smaller maps go without (the builder is a radix sort + 3 kernels per plan, on
            # the staging stream beside the training step)
            if self.total == n_in:
                up = B.conv_class_plan_pairs(self.nbmaps_buf, self.nboffs, k, self.total)
            else:
                up = B.conv_class_plan(B.conv_nbr_transposed(self.pos_in, self.nbmaps_buf, k), direct=True)
            down = B.conv_class_plan(self.nbr, direct=True) if (n_out >= _DIRECT_DOWN_MIN_ROWS or _DIRECT_FORCE) else None
            for plan in (down, up):
                if plan is not None:
                    plan["map_id"], plan["pairs"] = self.nboffs, self.total
            self.direct = {"down": down, "up": up}
            self._plans.clear()
        return self.direct

    def plans_for(self, transposed: bool, c_in: int, c_out: int, half: bool):
        """(forward plan, input-gradient plan) of a convolution over this map with these channel counts - either may be None
        (pair GEMM + pass 2).  Submanifold 3x3x3: the one mirrored plan for both where class_gemm_pays; 2x2x2 strided: the direct
        plans of the two directions where direct_conv_pays."""
        key = (transposed, c_in, c_out, half)
        hit = self._plans.get(key)            # (asked ~4 times per block and step: the answer only depends on the key)
        if hit is None:
            hit = self._choose_plans(transposed, c_in, c_out, half)
            self._plans[key] = hit
        return hit

    def _choose_plans(self, transposed, c_in, c_out, half):
        if not _dense_ok(c_in, c_out) or self.dup:
            return None, None
        if self.cls is not None and not transposed and class_gemm_pays(self.cls["n"], c_in, c_out, half):
            return self.cls, self.cls
        if self.direct is not None:
            d = self.direct
            fwd, dgrad = (d["up"], d["down"]) if transposed else (d["down"], d["up"])
            # (forward: reduces over c_in, writes c_out columns; input gradient: the other way round)
            return (fwd if fwd is not None and direct_conv_pays(fwd is d["up"], fwd["n"], c_in, c_out, half) else None,
                    dgrad if dgrad is not None and direct_conv_pays(dgrad is d["up"], dgrad["n"], c_out, c_in, half) else None)
        return None, None

    def class_rows(self) -> int:
        return 0 if self.cls is None else self.cls["z_rows"]

    @property
    def total(self) -> int:
        """number of pairs P: the one host read of a kernel map (sizes the pair-GEMM grid and its Z buffer);
        `build_pyramid` fills it for all maps of a forward pass with a single device->host copy"""
        if self._total is None:
            if self._dup is None:
                t, d = torch.stack([self.nboffs[-1], self._dup_dev.to(self.nboffs.dtype)]).tolist()      # (still one read)
                self._total, self._dup = int(t), bool(d)
            else:
                self._total = int(self.nboffs[-1].item())
        return self._total

    @property
    def dup(self) -> bool:
        """does the map's one coordinate set hold a coordinate twice?  (False for maps between two coordinate sets)"""
        if self._dup is None:
            if self._total is None:
                self.total
            else:
                self._dup = bool(self._dup_dev.item())
        return self._dup

    @property
    def nbmaps(self) -> torch.Tensor:
        """Exact [P, 2] int64 rulebook like the reference's."""
        if self._nbmaps is None:
            self._nbmaps = self.nbmaps_buf[:self.total].long()
        return self._nbmaps

    @property
    def nbsizes(self) -> torch.Tensor:
        return self.nbsizes_dev.long()

    def __getitem__(self, i):
        return (self.nbmaps, self.nbsizes, self.sizes)[i]

    def __iter__(self):
        return iter((self.nbmaps, self.nbsizes, self.sizes))

    def __len__(self):
        return 3


def accept_class_plans(kmaps):
    """judge the class plans launched by `build_class_plan(defer=True)` on these maps: one host read for all of them (the map's
    builder has read the pair totals the same way)"""
    pending = [km for km in kmaps if km._cls_pending is not None]
    if not pending:
        return
    counts = torch.stack([km._cls_pending["n_tiles"] for km in pending]).tolist() if len(pending) > 1 \
        else [pending[0]._cls_pending["n_tiles"].tolist()]
    for km, (tiles, steps) in zip(pending, counts):
        km._accept(tiles, steps)


# Class-sorted implicit GEMM (csrc/conv_class.hip) - where it beats pair GEMM + gather-sum (profiles/r03_class_gemm_layers.txt):
# 1.4-1.65x on the 178k-voxel stride-1 maps and the 32 / 64-wide stride-2 layers, 1.1-1.3x on the 84k-voxel 96-wide ones,
# 1.1x on 30k voxels x 64 channels, a loss on 30k x 128 and below.  TASEG_CLASS_GEMM=0 keeps every block on the two passes.
_CLASS_GEMM = options.class_gemm
_CLASS_MIN_ROWS = 16384          # <= 64 channels (module attributes: tests force them down to pin the class path at model level)
_CLASS_MIN_ROWS_96 = options.class_min_rows_96
_CLASS_MIN_ROWS_128 = options.class_min_rows_128
_CLASS_MAX_WORK = 1.6        # a class plan is used while its row-products stay under 1.6x the rulebook's pairs
_CLASS_MIN_ROWS_HALF = options.class_min_rows_half
# one-pass 2x2x2 strided / transposed convolutions on direct class plans (TASEG_DIRECT_CONV=0: pair GEMM + pass 2)
_DIRECT_CONV = options.direct_conv != "0"
_DIRECT_MIN_ROWS = options.direct_min_rows
_DIRECT_DOWN_MIN_ROWS = 8000        # smallest destination-row count any "down" plan is chosen at (direct_conv_pays)
_DIRECT_FORCE = options.direct_conv == "force"        # every fitting 2x2x2 product on its direct plan (tests, probes)


def class_conv(x, w, plan, f16, wt=False):
    """the convolution (wt: its transposed product) on a class plan: a direct plan's product IS the result; a three-group plan
    finishes inside the product where that pays (B.class_finish_pays), else through pass 2 - the same bits either way"""
    if plan["rows"] is not None:
        return (B.conv_class_gemm_f16 if f16 else B.conv_class_gemm)(x, w, plan, weight_transposed=wt)
    if plan["groups"] == 3 and B.class_finish_pays(plan["n"], f16):
        return (B.conv_class_conv_f16 if f16 else B.conv_class_conv)(x, w, plan, weight_transposed=wt)
    z = (B.conv_class_gemm_f16 if f16 else B.conv_class_gemm)(x, w, plan, weight_transposed=wt)
    return (B.conv_gather_sum_f16 if f16 else B.conv_gather_sum)(z, plan["pos"], plan["n"])


def class_gemm_pays(n_rows: int, c_in: int, c_out: int, half: bool = False) -> bool:
    if half:
        return n_rows >= _CLASS_MIN_ROWS_HALF
    if max(c_in, c_out) <= 64:
        return n_rows >= _CLASS_MIN_ROWS
    cols128 = any(c % 128 == 0 and c % 96 != 0 for c in (c_in, c_out))       # a direction on 128-column tiles (direct-rows pair GEMM)
    return n_rows >= (_CLASS_MIN_ROWS_128 if cols128 else _CLASS_MIN_ROWS_96)


def direct_conv_pays(up: bool, n_dest: int, c_red: int, c_cols: int, half: bool) -> bool:
    """Does the one-pass class GEMM on a direct plan beat pair GEMM + pass 2 for this product?  (tools/direct_probe.py on the bench
    rulebooks, profiles/r04_direct_conv_probe.txt.)  "up" plans (destination = fine rows, ONE offset per tile: a pair GEMM whose
    rows land in place) win 1.15-2.05x in fp32 down to ~30k destination rows of <= 64 channels / ~60k rows of wide ones and
    1.26-2.09x everywhere in half storage; "down" plans (destination = coarse rows, a tile walks up to 8 offsets) win 1.4-3.0x
    where tiles are many and rows narrow, and lose (0.2-0.9x) on the few, wide tiles of the deep levels - a coarse level has 128
    x fewer workgroups than its pairs have rows."""
    if _DIRECT_FORCE:
        return True
    wide = max(c_red, c_cols) > 64
    if up:
        return True if half else n_dest >= (60000 if wide else 20000)
    if half:
        return n_dest >= (60000 if wide else 8000) and max(c_red, c_cols) <= 96
    return (not wide) and n_dest >= 20000


def build_kernel_map(in_coords, out_coords, kernel_size, tensor_stride, dilation=1) -> KernelMap:
    offsets = get_kernel_offsets(kernel_size, stride=tensor_stride, dilation=dilation, device=in_coords.device)
    tables = B.build_kmap(in_coords, out_coords, offsets)
    k, n = offsets.shape[0], out_coords.shape[0]
    if (k & 1) and n > 0 and in_coords.shape == out_coords.shape and in_coords.data_ptr() == out_coords.data_ptr():
        # one coordinate set, odd kernel: the centre offset of row j must find j itself - anything else is a coordinate held twice
        # (the table keeps one row per coordinate, minkunet/utils.py / hash semantics of sphashquery)
        tables["dup"] = (tables["nbr"][k // 2] != torch.arange(n, device=out_coords.device, dtype=tables["nbr"].dtype)).any()
    return KernelMap(tables, (in_coords.shape[0], out_coords.shape[0]))


# ------------------------------------------------------------------------------ convolution
def _amp_half(feats: torch.Tensor) -> bool:
    """Should this op store its result in half?  The reference decorates its sparse ops with
    `custom_fwd(cast_inputs=torch.half)` (conv.py:19, voxelize.py:13, devoxelize.py:54): under torch.autocast inputs
    are cast to half and outputs are half.  Half features are also accepted outside autocast."""
    return feats.is_cuda and (feats.dtype == torch.float16 or torch.is_autocast_enabled("cuda"))


def _half_ok(*channels) -> bool:
    return all(c % 32 == 0 for c in channels)


class _NoCtx:
    __slots__ = ()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_CTX = _NoCtx()


def _no_autocast():
    """Context in which our kernels run: autocast off (they choose their own precision).  A real context manager
    only when autocast is on - this is entered ~200 times per step."""
    return torch.autocast("cuda", enabled=False) if torch.is_autocast_enabled("cuda") else _NO_CTX


class _SparseConv(Function):
    """conv.py:16-119 on the neighbour tables.  `transposed` swaps the roles of the two
    index columns exactly like convolution_cuda.cu:21,34.

    fp32: pair GEMM + gather-sum + weight gradient on the f32 MFMA kernels.  Half storage (autocast or half
    features; channel counts % 32 == 0): features, Z and the feature gradients are half, the fp32 master weight is
    cast once per call (ts_cast_weights_f16 writes the forward and the dgrad layout), every accumulation is fp32 and
    the weight gradient comes back in fp32."""

    @staticmethod
    def forward(ctx, feats, weight, kmap: KernelMap, transposed: bool, planes=None):
        if feats.shape[1] != weight.shape[1]:
            raise ValueError("Input feature size and kernel size mismatch")
        n_in, n_out = kmap.sizes
        rows_expected = n_out if transposed else n_in
        if feats.shape[0] != rows_expected:
            raise ValueError(f"conv3d{' (transposed)' if transposed else ''}: {feats.shape[0]} input rows but the "
                             f"kernel map has {rows_expected}")
        want_half = _amp_half(feats)
        half = want_half and _half_ok(weight.shape[1], weight.shape[2])
        gcol, table, rows = (1, kmap.pos_in, n_in) if transposed else (0, kmap.pos_out, n_out)
        # class plans (csrc/conv_class.hip): large submanifold 3x3x3 maps in place of pair GEMM + 27-way pass 2; 2x2x2 strided /
        # transposed maps in ONE pass (direct plans)
        cls, cls_d = kmap.plans_for(transposed, weight.shape[1], weight.shape[2], half)

        with _no_autocast():
            if half:
                fh = feats.contiguous().half()
                w16, _ = B.cast_weights_f16(weight.detach().float(), want=(True, False))
                if cls is not None:
                    out = class_conv(fh, w16, cls, True)
                else:
                    z = B.conv_pair_gemm_f16(fh, w16, kmap.nbmaps_buf, kmap.nboffs, kmap.total, gather_col=gcol, natural=True)
                    out = _scatter_pairs(z, kmap, False, rows) if (transposed and kmap.dup) else B.conv_gather_sum_f16(z, table, rows)
                ctx.save_for_backward(fh, w16)
            else:
                # pass 1: z[p] = feats[source row of pair p] @ W[k(p)];  pass 2: out[row] = sum_k z[pos[k, row]]
                f32, w32 = feats.contiguous().float(), weight.contiguous().float()
                if w32.data_ptr() != weight.data_ptr():
                    planes = None             # a converted copy: the planes belong to the parameter's own storage
                if cls is not None:
                    out = class_conv(f32, w32, cls, False)
                else:
                    _planes.hint(w32, planes)
                    z = B.conv_pair_gemm(f32, w32, kmap.nbmaps_buf, kmap.nboffs, kmap.total, gather_col=gcol)
                    # (a transposed product over a map with duplicated coordinates: two pairs may end in one row - scatter form)
                    out = _scatter_pairs(z, kmap, False, rows) if (transposed and kmap.dup) else B.conv_gather_sum(z, table, rows)
                ctx.save_for_backward(f32, w32)
                if want_half:
                    out = out.half()          # stem (C_in = 4 / 5): fp32 kernels, half result like the reference
        ctx.kmap, ctx.transposed, ctx.half, ctx.in_dtype = kmap, transposed, half, feats.dtype
        ctx.planes = None if half else planes
        ctx.cls = cls_d
        return out

    @staticmethod
    def backward(ctx, grad_out):
        feats, weight = ctx.saved_tensors
        kmap, transposed = ctx.kmap, ctx.transposed
        n_in, n_out = kmap.sizes
        k = weight.shape[0]
        grad_feats = grad_weight = None
        gcol = 0 if transposed else 1
        table, rows = (kmap.pos_in, n_in) if not transposed else (kmap.pos_out, n_out)
        with _no_autocast():
            if ctx.half:
                gh = grad_out.contiguous().half()
                if ctx.needs_input_grad[0] and ctx.cls is not None:
                    grad_feats = class_conv(gh, weight, ctx.cls, True, True).to(ctx.in_dtype)
                elif ctx.needs_input_grad[0]:
                    # d feats[i] = sum_k grad_out[partner(i, k)] @ W_k^T: rows of W_k (= w16) are the output columns
                    z = B.conv_pair_gemm_f16(gh, weight, kmap.nbmaps_buf, kmap.nboffs, kmap.total, gather_col=gcol)
                    grad_feats = (_scatter_pairs(z, kmap, transposed, rows) if kmap.dup else
                                  B.conv_gather_sum_f16(z, table, rows)).to(ctx.in_dtype)
                if ctx.needs_input_grad[1]:
                    grad_weight = B.conv_wgrad_f16(feats, gh, kmap.nbmaps_buf, kmap.nboffs, k,
                                                   col_a=1 if transposed else 0, max_pairs=kmap.total)
            else:
                g32 = grad_out.contiguous().float()
                if ctx.needs_input_grad[0] and ctx.cls is not None:
                    grad_feats = class_conv(g32, weight, ctx.cls, False, True).to(ctx.in_dtype)
                elif ctx.needs_input_grad[0]:
                    _planes.hint(weight, ctx.planes)
                    z = B.conv_pair_gemm(g32, weight, kmap.nbmaps_buf, kmap.nboffs, kmap.total, gather_col=gcol,
                                         weight_transposed=True)
                    grad_feats = (_scatter_pairs(z, kmap, transposed, rows) if kmap.dup else
                                  B.conv_gather_sum(z, table, rows)).to(ctx.in_dtype)
                if ctx.needs_input_grad[1]:
                    grad_weight = B.conv_wgrad(feats, g32, kmap.nbmaps_buf, kmap.nboffs, k,
                                               col_a=1 if transposed else 0, max_pairs=kmap.total)
        return grad_feats, grad_weight, None, None, None


def _scatter_pairs(z, kmap, transposed, rows):
    """input gradient of a map with duplicated coordinates: the rows of the per-pair product added into their input rows like the
    reference's scatter (convolution_cuda.cu:153-161) - the list form reads ONE pair per (offset, input row)"""
    total = kmap.total
    idx = kmap.nbmaps_buf[:total, 1 if transposed else 0].long()
    out = torch.zeros((rows, z.shape[1]), dtype=torch.float32, device=z.device)
    return out.index_add_(0, idx, z[:total].float()).to(z.dtype)


def build_pyramid(x: SparseTensor, num_levels: int = 4, kernel_size: int = 3, down_kernel: int = 2) -> None:
    """Build, up front, every coordinate set and kernel map a U-Net pass over `x` will ask for: the
    submanifold (kernel 3, stride 1) map at each of the num_levels + 1 strides and the strided (kernel 2,
    stride 2) map between consecutive strides - exactly the entries (same keys, same contents) that
    `conv3d` would create lazily (conv.py:144-177).  Doing it before the first convolution keeps the host
    reads (coordinate counts, pair totals) at the front of the step, where the device queue is short, instead
    of stalling the launch stream in the middle of the network; the pair totals of all maps come back in ONE
    device->host copy."""
    ks, dk, ones = make_ntuple(kernel_size, 3), make_ntuple(down_kernel, 3), (1, 1, 1)
    cur = x
    x.cmaps.setdefault(x.stride, x.coords)
    coords, stride = x.coords, x.stride
    maps = []
    for level in range(num_levels + 1):
        key = (stride, ks, ones, ones)
        if key not in x.kmaps:
            x.kmaps[key] = build_kernel_map(coords, coords, ks, stride, 1)
        maps.append(x.kmaps[key])
        if level == num_levels:
            break
        nxt = tuple(s * d for s, d in zip(stride, dk))
        if nxt not in x.cmaps:
            x.cmaps[nxt] = spdownsample(coords, dk, dk, stride)
        key = (stride, dk, dk, ones)
        if key not in x.kmaps:
            x.kmaps[key] = build_kernel_map(coords, x.cmaps[nxt], dk, stride, 1)
        maps.append(x.kmaps[key])
        coords, stride = x.cmaps[nxt], nxt
    pending = [m for m in maps if m._total is None]
    if pending:
        zero = pending[0].nboffs.new_zeros(())
        flags = [zero if m._dup_dev is None else m._dup_dev.to(zero.dtype) for m in pending]
        totals = torch.stack([m.nboffs[-1] for m in pending] + flags).tolist()      # one sync for all maps
        for m, t, d in zip(pending, totals, totals[len(pending):]):
            m._total = int(t)
            if m._dup is None:
                m._dup = bool(d)
    del cur


def _dense_ok(c_in: int, c_out: int) -> bool:
    """channel counts our pair-GEMM kernels take on their full-tile paths (fp32 split-bf16 and half storage)"""
    return c_in % 32 == 0 and c_out % 32 == 0


class _PointwiseConv(Function):
    """1x1x1 convolution (conv.py:135-140: `feats.matmul(weight)`) = a tall-skinny dense GEMM, N ~ 1e5 rows by
    C <= 384.  All three products run on our kernels with an identity rulebook (pair p = (p, p), one offset):
    forward and input gradient on the pair GEMM (its Z IS the result: one pair per output row, no pass 2), the weight
    gradient on the split-over-rows reduction.  hipBLASLt's heuristics pick tile-per-output kernels without a split
    over N for these shapes (fp32: 130-170 us per call; half: 330-490 us for a 178k x 128 x 96 product that moves
    80 MB).  Shapes outside the full-tile paths (channels not multiples of 32) keep the library GEMM for forward /
    input gradient.  Half storage (autocast / half features): half operands and results, fp32 accumulation, fp32
    weight gradient."""

    @staticmethod
    def forward(ctx, feats, weight, ident):
        half = _amp_half(feats)
        pairs, offs = ident
        c_in, c_out = weight.shape
        ours = _dense_ok(c_in, c_out)
        n = feats.shape[0]
        with _no_autocast():
            if half:
                f = feats.contiguous().half()
                if ours:
                    w16, _ = B.cast_weights_f16(weight.detach().float().view(1, c_in, c_out), want=(True, False))
                    out = B.conv_pair_gemm_f16(f, w16, pairs, offs, n, gather_col=0, natural=True)
                    w = w16
                else:
                    w = weight.detach().half()
                    out = f.matmul(w)
            else:
                f = feats.contiguous().float()
                w = weight.detach().float()
                out = B.conv_pair_gemm(f, w.view(1, c_in, c_out), pairs, offs, n, gather_col=0) if ours else f.matmul(w)
        ctx.save_for_backward(f, w)
        ctx.ident, ctx.half, ctx.in_dtype, ctx.ours = ident, half, feats.dtype, ours
        return out

    @staticmethod
    def backward(ctx, grad_out):
        feats, weight = ctx.saved_tensors
        pairs, offs = ctx.ident
        n = feats.shape[0]
        with _no_autocast():
            grad_out = grad_out.contiguous().to(feats.dtype)
            grad_feats = grad_weight = None
            if ctx.needs_input_grad[0]:
                if ctx.ours and ctx.half:       # weight = w16 [1, c_in, c_out]: rows = output columns of the input gradient
                    grad_feats = B.conv_pair_gemm_f16(grad_out, weight, pairs, offs, n, gather_col=0)
                elif ctx.ours:
                    grad_feats = B.conv_pair_gemm(grad_out, weight.view(1, *weight.shape[-2:]), pairs, offs, n,
                                                  gather_col=0, weight_transposed=True)
                else:
                    grad_feats = grad_out.matmul(weight.t())
                grad_feats = grad_feats.to(ctx.in_dtype)
            if ctx.needs_input_grad[1]:
                if ctx.half and _half_ok(feats.shape[1], grad_out.shape[1]):
                    grad_weight = B.conv_wgrad_f16(feats, grad_out, pairs, offs, 1, col_a=0, max_pairs=n)
                else:
                    grad_weight = B.conv_wgrad(feats.float(), grad_out.float(), pairs, offs, 1, col_a=0, max_pairs=n)
                grad_weight = grad_weight.view(weight.shape[-2:])
        return grad_feats, grad_weight, None


class _PointLinear(Function):
    """y = x W^T + b over per-point features (the class heads, minkunet.py:334-336: 480 -> 20 over ~2e5 points).  The
    output width is padded to 32 columns of zeros so that all three products run on our kernels with the identity
    rulebook like `_PointwiseConv` (the library needs 84 / 171 us for forward / input gradient in fp32 and several
    hundred in half for a product that only streams x once).  Half storage under autocast, fp32 otherwise."""

    @staticmethod
    def forward(ctx, x, weight, bias, ident):
        half = _amp_half(x)
        pairs, offs = ident
        o, c = weight.shape
        op = (o + 31) // 32 * 32
        n = x.shape[0]
        with _no_autocast():
            wpad = torch.zeros((1, c, op), dtype=torch.float32, device=x.device)      # W^T, zero columns beyond `o`
            wpad[0, :, :o] = weight.detach().float().t()
            if half:
                xs = x.contiguous().half()
                w16, _ = B.cast_weights_f16(wpad, want=(True, False))
                z = B.conv_pair_gemm_f16(xs, w16, pairs, offs, n, gather_col=0, natural=True)
                wsave = w16
            else:
                xs = x.contiguous().float()
                z = B.conv_pair_gemm(xs, wpad, pairs, offs, n, gather_col=0)
                wsave = wpad
            y = z[:, :o]
            y = y + bias.detach().to(z.dtype) if bias is not None else y.contiguous()
        ctx.save_for_backward(xs, wsave)
        ctx.ident, ctx.has_bias, ctx.half, ctx.o, ctx.in_dtype = ident, bias is not None, half, o, x.dtype
        return y

    @staticmethod
    def backward(ctx, grad_out):
        xs, wsave = ctx.saved_tensors
        pairs, offs = ctx.ident
        o, n = ctx.o, xs.shape[0]
        op = wsave.shape[2]
        with _no_autocast():
            gpad = torch.zeros((n, op), dtype=xs.dtype, device=xs.device)
            gpad[:, :o] = grad_out
            grad_x = grad_w = grad_b = None
            if ctx.needs_input_grad[0]:
                if ctx.half:
                    grad_x = B.conv_pair_gemm_f16(gpad, wsave, pairs, offs, n, gather_col=0)
                else:
                    grad_x = B.conv_pair_gemm(gpad, wsave, pairs, offs, n, gather_col=0, weight_transposed=True)
                grad_x = grad_x.to(ctx.in_dtype)
            if ctx.needs_input_grad[1]:
                if ctx.half:
                    gw = B.conv_wgrad_f16(xs, gpad, pairs, offs, 1, col_a=0, max_pairs=n)
                else:
                    gw = B.conv_wgrad(xs, gpad, pairs, offs, 1, col_a=0, max_pairs=n)
                grad_w = gw[0, :, :o].t().contiguous()
            if ctx.has_bias and ctx.needs_input_grad[2]:
                # column sums of a tall [n, <= 64] matrix: along the rows of the transposed copy (torch's dim-0 reduction
                # of a few columns runs on a handful of workgroups: 1 ms for 391k x 17)
                grad_b = grad_out.float().t().contiguous().sum(1)
        return grad_x, grad_w, grad_b, None


def point_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.functional.linear for [N, C] point features on the HIP kernels (training on a ROCm device, C a multiple of 32,
    at most 64 outputs); anything else goes to torch."""
    if x.is_cuda and x.dim() == 2 and x.shape[0] >= 4096 and x.dtype in (torch.float32, torch.float16) \
            and x.shape[1] % 32 == 0 and weight.shape[0] <= 64 and weight.requires_grad and torch.is_grad_enabled():
        return _PointLinear.apply(x, weight, bias, _identity_rows(x.shape[0], x.device))
    return torch.nn.functional.linear(x, weight, bias)


_ident_cache = {}


def _identity_rulebook(input: SparseTensor):
    return _identity_rows(input.feats.shape[0], input.feats.device)


def _identity_rows(n, dev):
    """(pairs [n,2] = (i, i), nboffs = [0, n]) for the 1x1x1 weight gradient; a handful of row counts per step,
    kept in a small module-level cache (NOT in input.kmaps, which mirrors the reference's dictionary)."""
    key = (n, dev)
    hit = _ident_cache.get(key)
    if hit is None:
        if len(_ident_cache) >= 32:
            _ident_cache.pop(next(iter(_ident_cache)))
        ar = torch.arange(n, dtype=torch.int32, device=dev)
        hit = (torch.stack([ar, ar], dim=1).contiguous(), torch.tensor([0, n], dtype=torch.int32, device=dev))
        _ident_cache[key] = hit
    return hit


class IdentityMap:
    """The identity rulebook of a 1x1x1 convolution over n rows (pair p = (p, p), one offset) with the attributes of a KernelMap
    that the block calls read: the shortcut branch of a residual block (minkunet.py:105-111) then runs as ONE block call per
    direction (csrc/block.hip, TsConvBlockOpts.natural) instead of a GEMM node + a BatchNorm node.  Cached per (n, device) like
    `_identity_rows`."""
    __slots__ = ("nbmaps_buf", "nboffs", "total", "pos_out", "pos_in", "sizes")
    dup = False                        # (one pair per row)

    def __init__(self, n, dev):
        pairs, offs = _identity_rows(n, dev)
        self.nbmaps_buf, self.nboffs, self.total = pairs, offs, n
        self.pos_out = self.pos_in = pairs         # never read by a natural call (any valid int32 pointer)
        self.sizes = (n, n)

    def plans_for(self, transposed, c_in, c_out, half):
        return None, None


_ident_maps = {}


def identity_map(n, dev) -> IdentityMap:
    key = (int(n), dev.type, dev.index)
    hit = _ident_maps.get(key)
    if hit is None:
        if len(_ident_maps) >= 32:
            _ident_maps.pop(next(iter(_ident_maps)))
        hit = _ident_maps[key] = IdentityMap(n, dev)
    return hit


def pointwise_block_ok(feats: torch.Tensor, weight: torch.Tensor, residual) -> bool:
    """Can the block call serve this 1x1x1 convolution + BatchNorm (identity rulebook, full-tile channel counts)?"""
    if not (feats.is_cuda and weight.dim() == 2 and feats.dim() == 2 and weight.is_cuda and weight.dtype == torch.float32):
        return False
    half = _amp_half(feats)
    c_in, c_out = weight.shape
    if not _dense_ok(c_in, c_out) or (half and not _half_ok(c_in, c_out)):
        return False
    if not half and feats.dtype != torch.float32:
        return False
    n = feats.shape[0]
    if c_out > 1024 or n <= 0 or feats.shape[1] != c_in:
        return False
    return residual is None or tuple(residual.shape) == (n, c_out)


def conv3d(input: SparseTensor, weight: torch.Tensor, kernel_size, bias: Optional[torch.Tensor] = None,
           stride: Union[int, List[int], Tuple[int, ...]] = 1, dilation: Union[int, Tuple[int, ...]] = 1,
           transposed: bool = False, planes: Optional[torch.Tensor] = None) -> SparseTensor:
    """conv.py:122-205 - same coordinate / kernel-map caching protocol as the reference:
    `cmaps[stride]` holds coordinates, `kmaps[(in_stride, kernel, stride, dilation)]` the map,
    a transposed convolution reuses the map of its mirror strided convolution."""
    kernel_size = make_ntuple(kernel_size, ndim=3)
    stride = make_ntuple(stride, ndim=3)
    dilation = make_ntuple(dilation, ndim=3)
    ones = (1, 1, 1)

    if kernel_size == ones and stride == ones and dilation == ones:
        out_stride, out_coords = input.stride, input.coords
        if input.feats.is_cuda and input.feats.dtype in (torch.float32, torch.float16) and weight.requires_grad:
            out_feats = _PointwiseConv.apply(input.feats, weight, _identity_rulebook(input))
        else:
            out_feats = input.feats.matmul(weight)   # plain dense GEMM -> rocBLAS/hipBLASLt
    else:
        kmap, out_coords, out_stride = conv_geometry(input, kernel_size, stride, dilation, transposed)
        out_feats = _SparseConv.apply(input.feats, weight, kmap, transposed, planes)
    if bias is not None:
        out_feats = out_feats + bias
    return _conv_output(input, out_feats, out_coords, out_stride)


def conv_geometry(input: SparseTensor, kernel_size, stride, dilation, transposed):
    """Coordinate / kernel-map protocol of conv3d (conv.py:144-199) for a kernel larger than 1x1x1: returns
    (kernel map, output coordinates, output stride), creating and caching `cmaps` / `kmaps` entries like the reference."""
    ones = (1, 1, 1)
    if not transposed:
        s_in = input.stride
        out_stride = (s_in[0] * stride[0], s_in[1] * stride[1], s_in[2] * stride[2])
        out_coords = input.cmaps.get(out_stride)
        key = (s_in, kernel_size, stride, dilation)
        if out_coords is not None:
            km = input.kmaps.get(key)
            if km is not None:                    # (every call but the first of a pass)
                return km, out_coords, out_stride
        elif stride == ones:
            out_coords = input.coords
        else:
            out_coords = spdownsample(input.coords, stride, kernel_size, input.stride)
        if key not in input.kmaps:
            km = build_kernel_map(input.coords, out_coords, kernel_size, input.stride, dilation)
            # (a class plan's input gradient relies on the map being its own transpose: only for a map over ONE coordinate set -
            # equal row counts of a user-provided cmaps entry do not make it one)
            if stride == ones and tuple(kernel_size) == (3, 3, 3) and out_coords.data_ptr() == input.coords.data_ptr():
                km.build_class_plan()     # a map made on demand (UNet3D of the TIAF models, user code): large ones get their class plan
            elif tuple(stride) == tuple(kernel_size) == (2, 2, 2):
                km.build_direct_plans()   # 2x2x2 strided map: one-pass plans of its two directions
            input.kmaps[key] = km
        return input.kmaps[key], out_coords, out_stride
    s_in = input.stride
    out_stride = (s_in[0] // stride[0], s_in[1] // stride[1], s_in[2] // stride[2])
    return input.kmaps[(out_stride, kernel_size, stride, dilation)], input.cmaps[out_stride], out_stride


def _conv_output(input: SparseTensor, out_feats, out_coords, out_stride) -> SparseTensor:
    output = SparseTensor(coords=out_coords, feats=out_feats, stride=out_stride)
    output.cmaps = input.cmaps
    output.cmaps.setdefault(out_stride, out_coords)
    output.kmaps = input.kmaps
    return output


import ctypes as _ctypes

_COMM_PRE, _COMM_POST = _ctypes.c_void_p(1), _ctypes.c_void_p(2)      # split-call sentinels (include/taseg_hip.h)


from ...rccl import c10d_sum as _c10d_sum  # noqa: E402


def _block_opts(plan_f, plan_d, planes, w16_current, addend, natural=False):
    """TsConvBlockOpts of one block call (the struct only holds pointers: the caller keeps the tensors alive over the call)"""
    LB = B.L
    pf = None if plan_f is None else _ctypes.pointer(B.class_plan_struct(plan_f))
    pd = None if plan_d is None else _ctypes.pointer(B.class_plan_struct(plan_d))
    return LB.TsConvBlockOpts(pf, pd, LB.ptr(planes), 1 if w16_current else 0, LB.ptr(addend), None, None, 0, 0, 0, 1 if natural else 0)


def _kcc(weight):
    """(K, C_in, C_out) of a convolution weight; a 1x1x1 weight is [C_in, C_out] (conv.py:135-140)"""
    return (1, weight.shape[0], weight.shape[1]) if weight.dim() == 2 else tuple(weight.shape)


class _ConvBlock(Function):
    """act(BN(conv(x)) [+ residual]) in training mode as ONE autograd node and one backend call per direction
    (csrc/block.hip).  Same launches and arithmetic as `_SparseConv` followed by `_BatchNormActTrain`; Z, the gradient
    w.r.t. the convolution output and the transposed half weight live in the per-stream workspace."""

    @staticmethod
    def forward(ctx, feats, weight, residual, bn_weight, bn_bias, kmap, transposed, bn_state, relu, comm, half, planes=None,
                passthrough=False, grad_dest=None, group=None):
        running_mean, running_var, nbt, momentum, eps = bn_state
        lib = B.L.load()
        L = B.L
        n_in, n_out = kmap.sizes
        k, c_in, c_out = _kcc(weight)
        natural = weight.dim() == 2              # 1x1x1 on the identity rulebook (IdentityMap): TsConvBlockOpts.natural
        gcol, table, rows = (1, kmap.pos_in, n_in) if transposed else (0, kmap.pos_out, n_out)
        dt = torch.float16 if half else torch.float32
        x = feats.contiguous().to(dt)
        w32 = weight.detach().contiguous().float()
        res = None if residual is None else residual.contiguous().to(dt)
        dev = x.device
        conv_out = torch.empty((rows, c_out), dtype=dt, device=dev)
        out = torch.empty((rows, c_out), dtype=dt, device=dev)
        stats = torch.empty((2, c_out), dtype=torch.float32, device=dev)
        mask = torch.empty(rows * (c_out // (8 if half else 4)), dtype=torch.uint8, device=dev) if relu else None
        if w32.data_ptr() != weight.data_ptr() or (planes is not None and planes.dtype != (torch.float16 if half else torch.int16)):
            planes = None                         # a converted copy: planes / half copy belong to the parameter's own storage
        # half storage: `planes` = the kept half copy of the weight (planes.half_for) - the call casts nothing
        w16 = (planes if planes is not None else torch.empty((k, c_in, c_out), dtype=torch.float16, device=dev)) if half else None
        # SyncBatchNorm transports: `comm` = the library-owned RCCL communicator (collective inline on this stream), or - `group`
        # given without one - torch.distributed runs the all-reduce between the two halves of a split call (csrc/block.hip)
        split = comm is None and group is not None
        pack = torch.empty(2 * c_out + 1, dtype=torch.float64, device=dev) if (comm is not None or split) else None
        total = kmap.total
        ws = L.workspace(lib.ts_conv_block_workspace_bytes(total, max(n_in, n_out), c_in, c_out, k, 1 if half else 0), dev)
        # everything the call may use beyond the rulebook, explicitly (include/taseg_hip.h TsConvBlockOpts): the class plans of
        # THIS kernel map (large submanifold maps: class-sorted implicit GEMM; 2x2x2 maps: direct one-pass plans), the pre-split
        # planes / the kept half copy of the weight
        plan_f, plan_d = kmap.plans_for(transposed, c_in, c_out, half)
        opts = _block_opts(plan_f, plan_d, None if half else planes, half and planes is not None, None, natural)

        def call(c):
            L.check(lib.ts_conv_block_forward(
                L.ptr(x), x.shape[0], c_in, L.ptr(w32), k, L.ptr(kmap.nbmaps_buf), L.ptr(kmap.nboffs), total, gcol,
                L.ptr(table), rows, c_out, L.ptr(res), L.ptr(bn_weight), L.ptr(bn_bias), L.ptr(running_mean),
                L.ptr(running_var), L.ptr(nbt), float(eps), float(momentum), 1 if relu else 0, 1 if half else 0, c,
                L.ptr(pack), L.ptr(conv_out), L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(out), L.ptr(mask), L.ptr(w16),
                _ctypes.byref(opts), L.ptr(ws), ws.numel(), L.stream()), "ts_conv_block_forward")

        if split:
            call(_COMM_PRE)                       # convolution + this rank's sums
            _c10d_sum(pack, group)
            call(_COMM_POST)                      # statistics over all ranks + elementwise pass
        else:
            call(comm)
        ctx.save_for_backward(x, w16 if half else w32, conv_out, stats, mask, bn_weight)
        for buf in (running_mean, running_var):      # written through raw pointers: move the version counters (batchnorm._bump)
            if buf is not None:
                torch.autograd.graph.increment_version(buf)
        ctx.kmap, ctx.transposed, ctx.half, ctx.comm = kmap, transposed, half, comm
        ctx.group = group if split else None
        ctx.planes = None if half else planes
        ctx.natural, ctx.wshape = natural, tuple(weight.shape)
        ctx.plan_d = plan_d
        ctx.grad_dest = grad_dest        # where the weight gradient is wanted (a gradient bucket's view), or None
        ctx.total_dev = None if pack is None else pack[2 * c_out:]
        ctx.in_dtype, ctx.res_dtype = feats.dtype, (None if residual is None else residual.dtype)
        # passthrough: the input leaves the node a second time (autograd aliases it); the shortcut of a residual block
        # consumes THAT tensor and its gradient comes back into this node, where it joins the input gradient's store
        return (out, feats) if passthrough else out

    @staticmethod
    def backward(ctx, grad_out, grad_pass=None):
        x, w, conv_out, stats, mask, bn_weight = ctx.saved_tensors
        kmap, transposed, half, comm = ctx.kmap, ctx.transposed, ctx.half, ctx.comm
        lib = B.L.load()
        L = B.L
        n_in, n_out = kmap.sizes
        k, c_in, c_out = _kcc(w)
        rows = conv_out.shape[0]
        dt = conv_out.dtype
        dev = x.device
        g = grad_out.contiguous().to(dt)
        gcol = 0 if transposed else 1
        table, drows = (kmap.pos_in, n_in) if not transposed else (kmap.pos_out, n_out)
        need = ctx.needs_input_grad
        grad_feat = torch.empty((drows, c_in), dtype=dt, device=dev) if need[0] else None
        grad_w = None
        if need[1]:
            dest = ctx.grad_dest
            if (dest is not None and dest.dtype == torch.float32 and dest.is_contiguous() and tuple(dest.shape) == ctx.wshape
                    and dest.device == dev):
                grad_w = dest.view_as(dest)      # a fresh alias of the bucket slot: autograd adopts it as p.grad, no copy
            else:
                grad_w = torch.empty(ctx.wshape, dtype=torch.float32, device=dev)
        grad_res = torch.empty_like(conv_out) if (ctx.res_dtype is not None and need[2]) else None
        gwb = torch.empty((2, c_out), dtype=torch.float32, device=dev)
        split = ctx.group is not None
        sums = torch.empty((2, c_out), dtype=torch.float64, device=dev) if (comm is not None or split) else None
        total = kmap.total
        ws = L.workspace(lib.ts_conv_block_workspace_bytes(total, max(n_in, n_out), c_in, c_out, k, 1 if half else 0), dev)

        addend = grad_pass.contiguous().to(dt) if (grad_pass is not None and grad_feat is not None) else None
        opts = _block_opts(None, ctx.plan_d if grad_feat is not None else None, None if half else ctx.planes, False, addend, ctx.natural)

        def call(c):
            L.check(lib.ts_conv_block_backward(
                L.ptr(g), L.ptr(mask), L.ptr(conv_out), L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(bn_weight),
                L.ptr(ctx.total_dev), c, L.ptr(sums), rows, c_out, 1 if half else 0, L.ptr(x), x.shape[0], c_in, L.ptr(w), k,
                L.ptr(kmap.nbmaps_buf), L.ptr(kmap.nboffs), total, gcol, L.ptr(table), drows, 1 if transposed else 0,
                L.ptr(grad_feat), L.ptr(grad_res), L.ptr(grad_w), L.ptr(gwb[0]), L.ptr(gwb[1]), _ctypes.byref(opts), L.ptr(ws),
                ws.numel(), L.stream()), "ts_conv_block_backward")

        if split:
            call(_COMM_PRE)                       # this rank's sums of the BatchNorm backward
            _c10d_sum(sums, ctx.group)
        call(_COMM_POST if split else comm)
        if grad_feat is not None and grad_feat.dtype != ctx.in_dtype:
            grad_feat = grad_feat.to(ctx.in_dtype)
        if grad_res is not None and grad_res.dtype != ctx.res_dtype:
            grad_res = grad_res.to(ctx.res_dtype)
        return grad_feat, grad_w, grad_res, gwb[0], gwb[1], None, None, None, None, None, None, None, None, None, None


def conv_block_eval(feats, weight, residual, bn_weight, bn_bias, mean, invstd, kmap, transposed, relu, half, planes=None):
    """act(BN_eval(conv(x)) [+ residual]) without a graph: ONE backend call (ts_conv_block_eval, csrc/block.hip) - the convolution
    of the training forward (class plans, pre-split planes / kept half weights) followed by one elementwise pass on the running
    statistics (mean, invstd = 1 / sqrt(running_var + eps)).  The evaluation branch of the segmentors runs 63 of these per pass
    (minkunet.py:435-455; ten passes per scan under test-time augmentation, R/train.py:474-503)."""
    lib = B.L.load()
    L = B.L
    n_in, n_out = kmap.sizes
    k, c_in, c_out = _kcc(weight)
    gcol, table, rows = (1, kmap.pos_in, n_in) if transposed else (0, kmap.pos_out, n_out)
    dt = torch.float16 if half else torch.float32
    x = feats.contiguous().to(dt)
    w32 = weight.detach().contiguous().float()
    res = None if residual is None else residual.contiguous().to(dt)
    dev = x.device
    out = torch.empty((rows, c_out), dtype=dt, device=dev)
    if w32.data_ptr() != weight.data_ptr() or (planes is not None and planes.dtype != (torch.float16 if half else torch.int16)):
        planes = None
    w16 = (planes if planes is not None else torch.empty((k, c_in, c_out), dtype=torch.float16, device=dev)) if half else None
    total = kmap.total
    ws = L.workspace(lib.ts_conv_block_workspace_bytes(total, max(n_in, n_out), c_in, c_out, k, 1 if half else 0), dev)
    plan_f, _ = kmap.plans_for(transposed, c_in, c_out, half)
    opts = _block_opts(plan_f, None, None if half else planes, half and planes is not None, None, weight.dim() == 2)
    L.check(lib.ts_conv_block_eval(
        L.ptr(x), x.shape[0], c_in, L.ptr(w32), k, L.ptr(kmap.nbmaps_buf), L.ptr(kmap.nboffs), total, gcol, L.ptr(table), rows,
        c_out, L.ptr(res), L.ptr(bn_weight), L.ptr(bn_bias), L.ptr(mean), L.ptr(invstd), 1 if relu else 0, 1 if half else 0,
        L.ptr(out), L.ptr(w16), _ctypes.byref(opts), L.ptr(ws), ws.numel(), L.stream()), "ts_conv_block_eval")
    return out


def conv_block_ok(feats: torch.Tensor, weight: torch.Tensor, kmap: "KernelMap", residual, rows: int) -> bool:
    """Can `_ConvBlock` serve this convolution + training BatchNorm?  (else: conv3d followed by bn_act)"""
    if not (feats.is_cuda and weight.dim() == 3 and feats.dim() == 2):
        return False
    half = _amp_half(feats)
    c_in, c_out = weight.shape[1], weight.shape[2]
    if half and not (_half_ok(c_in, c_out)):
        return False
    if not half and feats.dtype != torch.float32:
        return False
    if c_out % (8 if half else 4) != 0 or c_out > 1024 or rows <= 0 or kmap.total <= 0 or weight.shape[0] > 63 or kmap.dup:
        return False
    if feats.shape[1] != c_in:
        return False
    return residual is None or tuple(residual.shape) == (rows, c_out)
