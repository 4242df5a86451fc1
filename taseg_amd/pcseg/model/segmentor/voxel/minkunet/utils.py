"""Point <-> voxel glue of the MinkUNet family (reference pcseg/model/segmentor/voxel/minkunet/utils.py).

Same functions, arguments, caches and results as the reference; each is one or two fused
backend calls instead of the reference's chain of hash / unique / query / weight tensor ops:

  initial_voxelize  sphash + torch.unique + sphashquery  ->  ts_hash + ts_unique_i64 (sort, unique, inverse)
  voxel_to_point    get_kernel_offsets + 2 x sphash + sphashquery + calc_ti_weights + 2 transposes
                    ->  ts_trilinear_map (table build + 8 probes + weights per point, one pass)
"""
import torch

from taseg_amd import backend as B
from taseg_amd.torchsparse import PointTensor, SparseTensor
from taseg_amd.torchsparse.nn import functional as F

__all__ = ["initial_voxelize", "voxelize_index", "point_to_voxel", "voxel_to_point", "voxel_to_point_fov"]


def voxelize_index(z: PointTensor, init_res, after_res):
    """The index half of `initial_voxelize` (utils.py:11-26): stride-1 voxel coordinates ordered by ascending
    coordinate hash, the point->voxel map and the per-voxel point counts.  Touches no features, so a data
    stage can run it ahead of the forward pass.  Rescales `z.C` in place like the reference (:35)."""
    scaled = torch.cat([(z.C[:, :3] * init_res) / after_res, z.C[:, -1].view(-1, 1)], 1)
    cell = torch.floor(scaled)
    pc_hash = F.sphash(cell.int())
    sparse_hash, inverse = B.unique_i64(pc_hash)
    idx_query = inverse.long()
    counts = F.spcount(inverse, len(sparse_hash))
    coords = torch.round(F.spvoxelize(cell, idx_query, counts)).int()
    z.additional_features["idx_query"][1] = idx_query
    z.additional_features["counts"][1] = counts
    z.C = scaled
    return coords, idx_query, counts


def initial_voxelize(z: PointTensor, init_res, after_res) -> SparseTensor:
    """utils.py:11-36.  Re-voxelise the points of `z` on device: stride-1 voxels are ordered by
    ASCENDING COORDINATE HASH (torch.unique of the FNV hashes), features / coordinates are
    mean-pooled, and `z` gets the point->voxel map cached for later `point_to_voxel` calls."""
    coords, idx_query, counts = voxelize_index(z, init_res, after_res)
    out = SparseTensor(F.spvoxelize(z.F, idx_query, counts), coords, 1)
    out.cmaps.setdefault(out.stride, out.coords)
    return out


def point_to_voxel(x: SparseTensor, z: PointTensor) -> SparseTensor:
    """utils.py:41-65 (used by SPVCNN-style point branches; kept for API completeness)."""
    cache = z.additional_features
    if cache is None or cache.get("idx_query") is None or cache["idx_query"].get(x.s) is None:
        s = x.s[0]
        cell = torch.cat([torch.floor(z.C[:, :3] / s).int() * s, z.C[:, -1].int().view(-1, 1)], 1)
        idx_query = F.sphashquery(F.sphash(cell), F.sphash(x.C))
        counts = F.spcount(idx_query.int(), x.C.shape[0])
        cache["idx_query"][x.s] = idx_query
        cache["counts"][x.s] = counts
    else:
        idx_query, counts = cache["idx_query"][x.s], cache["counts"][x.s]
    return x._like(F.spvoxelize(z.F, idx_query, counts))


def _trilinear(x: SparseTensor, z: PointTensor, nearest: bool):
    idx_query, weights = B.trilinear_map(z.C.contiguous(), x.C, x.s[0])
    if nearest:
        weights[:, 1:] = 0.0
        idx_query[:, 1:] = -1
    return idx_query, weights


def voxel_to_point(x: SparseTensor, z: PointTensor, nearest: bool = False, features: bool = True) -> PointTensor:
    """utils.py:69-107.  Trilinear devoxelisation of x's features onto z's points; the 8-corner
    indices [N,8] and weights [N,8] are cached in `z` per voxel stride.  features=False only fills that cache and
    returns a PointTensor without features (MinkUNet's first call, minkunet.py:396: its z0.F is never read - the
    point-branch MLPs that consume it exist in SPVCNN only)."""
    cached = (z.idx_query is not None and z.weights is not None
              and z.idx_query.get(x.s) is not None and z.weights.get(x.s) is not None)
    if not cached:
        idx_query, weights = _trilinear(x, z, nearest)
        z.idx_query[x.s] = idx_query
        z.weights[x.s] = weights
    order = (z.additional_features.get("devox_order") or {}).get(x.s)     # backward walk order, if the plan built one
    feats = F.spdevoxelize(x.F, z.idx_query[x.s], z.weights[x.s], order) if features else None
    out = PointTensor(feats, z.C, idx_query=z.idx_query, weights=z.weights)
    out.additional_features = z.additional_features
    return out


def voxel_to_point_fov(x: SparseTensor, z: PointTensor, nearest: bool = False) -> PointTensor:
    """utils.py:150-170: same lookup against another cloud's voxels, nothing cached in `z`."""
    idx_query, weights = _trilinear(x, z, nearest)
    out = PointTensor(F.spdevoxelize(x.F, idx_query, weights), z.C)
    out.idx_query[x.s] = idx_query
    out.weights[x.s] = weights
    return out
