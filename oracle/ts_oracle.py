"""CPU ORACLE - test infrastructure, not product code.

A numpy restatement of the reference's algorithm for the TASeg hot path, used only as the
checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
taseg_amd/ imports it.  Every function cites the reference lines it follows
(TS/ = /root/reference/package/torchsparse.zip member torchsparse/, R/ = /root/reference/).
The CUDA sources are the semantic authority where the reference's CPU twin is buggy
(SURVEY.md fact 4).

Parity pin: tests/test_oracle_golden.py checks every function here against the golden vectors
in tests/golden/*.npz, which tests/golden/make_golden.py captured from the real reference
(its Python + its CPU extension) running in the build container.
"""
import numpy as np

FNV_OFFSET = np.uint64(14695981039346656037)
FNV_PRIME = np.uint64(1099511628211)
MASK60 = np.uint64(0x0FFFFFFFFFFFFFFF)


def _ft(a):
    """float32 like the reference, unless the caller hands in float64 arrays (the fp64 evaluation of the same
    algorithm that tests use as the yardstick for fp32 gradient noise)"""
    return np.float64 if getattr(a, "dtype", None) == np.float64 else np.float32


# --------------------------------------------------------------------------------- hashing
def sphash(coords, offsets=None):
    """TS/torchsparse/backend/hash/hash_cuda.cu:10-23 (K1) and :27-55 (K2, output [K, N], the
    row's own batch index).  coords int32 [N,4] (x,y,z,b); returns int64."""
    c = np.ascontiguousarray(coords, dtype=np.int32)
    if offsets is not None:
        off = np.asarray(offsets, dtype=np.int32)
        out = np.empty((off.shape[0], c.shape[0]), dtype=np.int64)
        for k in range(off.shape[0]):
            ck = c.copy()
            ck[:, :3] += off[k]
            out[k] = sphash(ck)
        return out
    h = np.full(c.shape[0], FNV_OFFSET, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for j in range(4):
            h ^= c[:, j].astype(np.uint32).astype(np.uint64)   # (unsigned int) cast of the int32
            h *= FNV_PRIME                                       # mod 2^64
    h = (h >> np.uint64(60)) ^ (h & MASK60)
    return h.astype(np.int64)


def sphashquery(queries, references, idx=None):
    """TS/torchsparse/nn/functional/query.py:8-33 + backend/others/query_cuda.cu:9-56 /
    query_cpu.cpp:12-37: position (or idx value) of each query hash among the references,
    -1 on a miss; duplicate reference keys keep the first (dense_hash_map::insert)."""
    q = np.asarray(queries, dtype=np.int64)
    r = np.asarray(references, dtype=np.int64)
    vals = np.arange(r.shape[0], dtype=np.int64) if idx is None else np.asarray(idx, dtype=np.int64)
    out = np.full(q.size, -1, dtype=np.int64)
    if r.size:
        order = np.argsort(r, kind="stable")
        rs = r[order]
        pos = np.searchsorted(rs, q.reshape(-1), side="left")
        pos_c = np.minimum(pos, rs.size - 1)
        hit = rs[pos_c] == q.reshape(-1)
        out[hit] = vals[order[pos_c[hit]]]
    return out.reshape(q.shape)


def spcount(idx, n):
    """TS/torchsparse/backend/others/count_cuda.cu:10-16: histogram of idx >= 0."""
    idx = np.asarray(idx)
    return np.bincount(idx[idx >= 0], minlength=n).astype(np.int32)[:n]


# --------------------------------------------------------------------------------- voxelize / devoxelize
def voxelize_forward(feat, idx, counts):
    """TS/.../voxelize/voxelize_cuda.cu:12-25: out[idx[i]] += feat[i] / counts[idx[i]]."""
    feat = np.asarray(feat, dtype=_ft(feat))
    idx = np.asarray(idx, dtype=np.int64)
    counts = np.asarray(counts)
    m = counts.shape[0]
    out = np.zeros((m, feat.shape[1]), dtype=feat.dtype)
    ok = (idx >= 0) & (idx < m)
    ok[ok] &= counts[idx[ok]] != 0
    np.add.at(out, idx[ok], feat[ok] / counts[idx[ok]].astype(feat.dtype)[:, None])
    return out


def voxelize_backward(gout, idx, counts, n):
    """TS/.../voxelize/voxelize_cuda.cu:28-41: gfeat[i] = gout[idx[i]] / counts[idx[i]]."""
    gout = np.asarray(gout, dtype=_ft(gout))
    idx = np.asarray(idx, dtype=np.int64)
    counts = np.asarray(counts)
    out = np.zeros((n, gout.shape[1]), dtype=gout.dtype)
    ok = (idx >= 0) & (idx < counts.shape[0])
    ok[ok] &= counts[idx[ok]] != 0
    out[ok] = gout[idx[ok]] / counts[idx[ok]].astype(gout.dtype)[:, None]
    return out


def devoxelize_forward(feat, idx, w):
    """TS/.../devoxelize/devoxelize_cuda.cu:11-33: out[i] = sum_k w[i,k] * feat[idx[i,k]] in k order."""
    feat = np.asarray(feat, dtype=_ft(feat))
    idx = np.asarray(idx, dtype=np.int64)
    w = np.asarray(w, dtype=np.float32)
    out = np.zeros((idx.shape[0], feat.shape[1]), dtype=feat.dtype)
    for k in range(8):
        ok = idx[:, k] >= 0
        out[ok] += w[ok, k, None] * feat[idx[ok, k]]
    return out


def devoxelize_backward(gout, idx, w, m):
    """TS/.../devoxelize/devoxelize_cuda.cu:37-57 (the CUDA adjoint; the CPU twin at
    devoxelize_cpu.cpp:35-59 is wrong): gfeat[idx[i,k]] += w[i,k] * gout[i]."""
    gout = np.asarray(gout, dtype=_ft(gout))
    idx = np.asarray(idx, dtype=np.int64)
    w = np.asarray(w, dtype=np.float32)
    out = np.zeros((m, gout.shape[1]), dtype=gout.dtype)
    for k in range(8):
        ok = idx[:, k] >= 0
        np.add.at(out, idx[ok, k], w[ok, k, None] * gout[ok])
    return out


def calc_ti_weights(coords, idx_query, scale=1):
    """TS/torchsparse/nn/functional/devoxelize.py:10-48.  coords float [N,>=3], idx_query [8,N];
    returns float32 [8,N]."""
    p = np.asarray(coords, dtype=np.float32)[:, :3]
    s = np.float32(scale)
    pf = np.floor(p / s) * s if scale != 1 else np.floor(p)
    pc = pf + s
    lo, hi = (p - pf).astype(np.float32), (pc - p).astype(np.float32)
    rows = []
    for k in range(8):
        bx, by, bz = (k >> 2) & 1, (k >> 1) & 1, k & 1
        fx = lo[:, 0] if bx else hi[:, 0]
        fy = lo[:, 1] if by else hi[:, 1]
        fz = lo[:, 2] if bz else hi[:, 2]
        rows.append((fx * fy) * fz)
    w = np.stack(rows, 0).astype(np.float32)
    if scale != 1:
        w = w / np.float32(scale ** 3)
    w[np.asarray(idx_query) == -1] = 0
    tot = np.zeros(w.shape[1], dtype=np.float32)
    for k in range(8):
        tot = tot + w[k]
    return (w / (tot + np.float32(1e-8))).astype(np.float32)


# --------------------------------------------------------------------------------- rulebook
def get_kernel_offsets(size, stride=1, dilation=1):
    """TS/torchsparse/nn/utils/kernel.py:11-32."""
    tup = lambda v: (v,) * 3 if isinstance(v, int) else tuple(v)  # noqa: E731
    size, stride, dilation = tup(size), tup(stride), tup(dilation)
    ax = [np.arange(-size[k] // 2 + 1, size[k] // 2 + 1) * stride[k] * dilation[k] for k in range(3)]
    if np.prod(size) % 2 == 1:
        rows = [[x, y, z] for z in ax[2] for y in ax[1] for x in ax[0]]
    else:
        rows = [[x, y, z] for x in ax[0] for y in ax[1] for z in ax[2]]
    return np.array(rows, dtype=np.int32)


def spdownsample(coords, stride=2, kernel_size=2, tensor_stride=1):
    """TS/torchsparse/nn/functional/downsample.py:25-51 (stride in {1, kernel} branch):
    trunc(c / s) * s, unique rows sorted by (b, x, y, z)."""
    tup = lambda v: (v,) * 3 if isinstance(v, int) else tuple(v)  # noqa: E731
    stride, tensor_stride = tup(stride), tup(tensor_stride)
    step = np.array([stride[k] * tensor_stride[k] for k in range(3)], dtype=np.int64)
    c = np.asarray(coords, dtype=np.int64).copy()
    c[:, :3] = np.trunc(c[:, :3] / step).astype(np.int64) * step
    u = np.unique(c[:, [3, 0, 1, 2]], axis=0)
    return u[:, [1, 2, 3, 0]].astype(np.int32)


def build_kmap(in_coords, out_coords, offsets):
    """TS/torchsparse/nn/functional/conv.py:160-176: results [K, N_out] (index of the input voxel at
    out + offset, -1 if none), nbsizes [K], nbmaps [P, 2] = (in, out) ordered by (k, out)."""
    refs = sphash(in_coords)
    queries = sphash(out_coords, offsets)
    results = sphashquery(queries, refs)
    nbsizes = (results != -1).sum(axis=1)
    kk, jj = np.nonzero(results != -1)
    nbmaps = np.stack([results[kk, jj], jj], axis=1).astype(np.int64)
    return results, nbmaps, nbsizes.astype(np.int64)


# --------------------------------------------------------------------------------- convolution
def conv_forward(feats, weight, nbmaps, nbsizes, sizes, transposed=False):
    """TS/.../convolution/convolution_cuda.cu:53-165 (per offset: gather, GEMM, scatter-add)."""
    feats = np.asarray(feats, dtype=_ft(feats))
    weight = np.asarray(weight, dtype=feats.dtype)
    n_out = sizes[0] if transposed else sizes[1]
    out = np.zeros((n_out, weight.shape[-1]), dtype=feats.dtype)
    a = 0
    for k in range(weight.shape[0]):
        b = a + int(nbsizes[k])
        if b > a:
            i = nbmaps[a:b, 1 if transposed else 0]
            o = nbmaps[a:b, 0 if transposed else 1]
            out[o] += feats[i] @ weight[k]      # the reference's non-atomic scatter: a row appears once per offset
        a = b
    return out


def conv_backward(feats, weight, gout, nbmaps, nbsizes, transposed=False):
    """TS/.../convolution/convolution_cuda.cu:167-278: grad_in (scatter of gout W^T) and
    grad_weight[k] = gather(in)^T gather(gout)."""
    feats = np.asarray(feats, dtype=_ft(feats))
    weight = np.asarray(weight, dtype=feats.dtype)
    gout = np.asarray(gout, dtype=feats.dtype)
    gin = np.zeros_like(feats)
    gw = np.zeros_like(weight)
    a = 0
    for k in range(weight.shape[0]):
        b = a + int(nbsizes[k])
        if b > a:
            i = nbmaps[a:b, 1 if transposed else 0]
            o = nbmaps[a:b, 0 if transposed else 1]
            gin[i] += gout[o] @ weight[k].T
            gw[k] = feats[i].T @ gout[o]
        a = b
    return gin, gw


# --------------------------------------------------------------------------------- dataset side
def ravel_hash(x):
    """TS/torchsparse/utils/quantize.py:9-21."""
    x = np.asarray(x)
    x = (x - x.min(axis=0)).astype(np.uint64)
    xmax = x.max(axis=0).astype(np.uint64) + np.uint64(1)
    h = np.zeros(x.shape[0], dtype=np.uint64)
    for k in range(x.shape[1] - 1):
        h += x[:, k]
        h *= xmax[k + 1]
    return h + x[:, -1]


def sparse_quantize(coords):
    """TS/torchsparse/utils/quantize.py:24-46 with voxel_size=1 on integer coords:
    (indices of first occurrence per voxel in ascending key order, inverse map)."""
    _, index, inverse = np.unique(ravel_hash(np.asarray(coords)), return_index=True, return_inverse=True)
    return index, inverse


def voxel_coords(points, voxel_size):
    """R/pcseg/data/dataset/semantickitti/semantickitti_voxel.py:119: np.round(xyz / vs).astype(int32)
    (float32 division, round-half-even); the min shift of :120 is left to the caller."""
    return np.round(np.asarray(points, dtype=np.float32)[:, :3] / np.float32(voxel_size)).astype(np.int32)


def fuse_scan(points, pose0, pose):
    """R/pcseg/data/dataset/semantickitti/semantickitti_ms.py:403-417, float32 throughout,
    products summed in index order like np.sum(axis=1) over a length-4 / length-3 axis."""
    points = np.asarray(points, dtype=np.float32)
    pose0 = np.asarray(pose0, dtype=np.float32)
    pose = np.asarray(pose, dtype=np.float32)
    h = np.concatenate([points[:, :3], np.ones_like(points[:, :1])], 1)
    pt = pose.T
    new = np.zeros((h.shape[0], 4), dtype=np.float32)
    for k in range(4):
        new = new + h[:, k:k + 1] * pt[k][None, :] if k else h[:, k:k + 1] * pt[k][None, :]
    nc = new[:, :3] - pose0[:3, 3]
    r0 = pose0[:3, :3]
    out = nc[:, 0:1] * r0[0][None, :]
    for k in (1, 2):
        out = out + nc[:, k:k + 1] * r0[k][None, :]
    return np.concatenate([out.astype(np.float32), points[:, 3:]], 1)


def history_mask(labels_raw, delta, flexible_steps, learning_map_inv):
    """R/.../semantickitti_ms.py:303-308: keep a history point when its (pseudo-)label's class has a
    non-zero step that divides |delta|."""
    mask = np.zeros(len(labels_raw), dtype=bool)
    for cls, step in enumerate(flexible_steps):
        if step and abs(delta) % step == 0:
            mask |= np.asarray(labels_raw) == learning_map_inv[cls]
    return mask


def append_time_flag(n_current, raw_ms):
    """R/.../semantickitti_ms.py:253-257: column 4 = 1 for the current scan's rows, 0 for history."""
    flag = np.zeros((len(raw_ms), 1), dtype=raw_ms.dtype)
    flag[:n_current, 0] = 1
    return np.concatenate([raw_ms[:, :4], flag, raw_ms[:, 4:]], axis=1)


# --------------------------------------------------------------------------------- model glue
def initial_voxelize_maps(coords_float4, pres, vres):
    """R/pcseg/model/segmentor/voxel/minkunet/utils.py:11-27: scaled float coords, ascending unique
    hashes, point->voxel map, counts."""
    c = np.asarray(coords_float4, dtype=np.float32)
    scaled = np.concatenate([(c[:, :3] * np.float32(pres)) / np.float32(vres), c[:, 3:4]], 1).astype(np.float32)
    cell = np.floor(scaled)
    pc_hash = sphash(cell.astype(np.int32))
    sparse_hash = np.unique(pc_hash)
    idx_query = sphashquery(pc_hash, sparse_hash)
    counts = spcount(idx_query, len(sparse_hash))
    return scaled, cell, sparse_hash, idx_query, counts


def trilinear_map(points_float4, vox_coords, stride):
    """R/.../minkunet/utils.py:72-82: idx_query [N,8] (int64, -1 = absent) and weights [N,8]."""
    p = np.asarray(points_float4, dtype=np.float32)
    s = int(stride)
    base = np.concatenate([np.floor(p[:, :3] / np.float32(s)).astype(np.int32) * s,
                           p[:, 3:4].astype(np.int32)], 1)
    off = get_kernel_offsets(2, s, 1)
    idx = sphashquery(sphash(base, off), sphash(vox_coords))            # [8, N]
    w = calc_ti_weights(p, idx, scale=s)
    return idx.T.copy(), w.T.copy()


def image_gather(feat, pix, pbatch, frame_end, shift=0):
    """R/pcseg/model/segmentor/voxel/minkunet/unet2d.py:180-209: the frames [start:end] of every sample are laid out
    as one tall NHWC image (`permute(0, 2, 3, 1)[start:end].reshape(-1, w, c)`) and indexed with
    (`row.long()`, `col.long()`), the 1/4-scale map with (`row // 4`, `col // 4`); per-sample results are
    concatenated in sample order (= point order for a collated, batch-sorted FOV cloud).
    feat [T, C, H >> shift, W >> shift], pix [n, 2] float, pbatch [n] int, frame_end [B] cumulative."""
    feat = np.asarray(feat)
    pix = np.asarray(pix).astype(np.int64)           # .long(): truncation of non-negative floats
    pbatch = np.asarray(pbatch)
    outs, start = [], 0
    for b, end in enumerate(np.asarray(frame_end).tolist()):
        tall = np.transpose(feat[start:end], (0, 2, 3, 1)).reshape(-1, feat.shape[3], feat.shape[1])
        p = pix[pbatch == b]
        outs.append(tall[p[:, 0] >> shift, p[:, 1] >> shift])
        start = end
    return np.concatenate(outs, 0)


# --------------------------------------------------------------------------- evaluation metrics
def fast_hist(pred, label, n):
    """R/train.py:35-40: confusion matrix (rows = label, columns = prediction) over the labels inside [0, n)."""
    pred, label = np.asarray(pred).reshape(-1), np.asarray(label).reshape(-1)
    keep = (label >= 0) & (label < n)
    return np.bincount(n * label[keep].astype(np.int64) + pred[keep].astype(np.int64), minlength=n * n)[:n * n].reshape(n, n)


def per_class_iu(hist):
    """R/train.py:43-44."""
    hist = np.asarray(hist, dtype=np.float64)
    return np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist) + 1e-9)


# --------------------------------------------------------------------------- nuScenes multi-scan fuse
def quaternion_rotation_matrix(q):
    """`pyquaternion.Quaternion(q).rotation_matrix` (pyquaternion 0.9.9, the quaternion class nuscenes-devkit depends
    on; absent from /root/reference and from this image - its published algorithm restated): q = (w, x, y, z) is
    normalised unless already unit (tolerance 1e-14), the matrix is the lower-right 3x3 of Q(q) . Qbar(q)^T."""
    q = np.asarray(q, dtype=np.float64).copy()
    n2 = float(np.dot(q, q))
    if abs(1.0 - n2) >= 1e-14:
        n = np.sqrt(n2)
        if n > 0:
            q = q / n
    w, x, y, z = q
    qm = np.array([[w, -x, -y, -z], [x, w, -z, y], [y, z, w, -x], [z, -y, x, w]])
    qb = np.array([[w, -x, -y, -z], [x, w, z, -y], [y, -z, w, x], [z, y, -x, w]])
    return np.dot(qm, qb.conj().transpose())[1:][:, 1:]


def nus_transform_point(raw_data, info0, info):
    """R/pcseg/data/dataset/nuscenes/nuscenes_ms.py:348-373: points of the lidar frame of `info` into the lidar frame of
    `info0` through ego and global poses.  info* = dict(lidar2ego_rotation q, lidar2ego_translation, ego2global_rotation q,
    ego2global_translation).  float64 rotation / translation, result stored back into the float32 array."""
    l2e_r_mat = quaternion_rotation_matrix(info0["lidar2ego_rotation"])
    e2g_r_mat = quaternion_rotation_matrix(info0["ego2global_rotation"])
    l2e_t, e2g_t = np.asarray(info0["lidar2ego_translation"]), np.asarray(info0["ego2global_translation"])
    l2e_r_s_mat = quaternion_rotation_matrix(info["lidar2ego_rotation"])
    e2g_r_s_mat = quaternion_rotation_matrix(info["ego2global_rotation"])
    l2e_t_s, e2g_t_s = np.asarray(info["lidar2ego_translation"]), np.asarray(info["ego2global_translation"])
    back = np.linalg.inv(e2g_r_mat).T @ np.linalg.inv(l2e_r_mat).T
    R = (l2e_r_s_mat.T @ e2g_r_s_mat.T) @ back
    T = (l2e_t_s @ e2g_r_s_mat.T + e2g_t_s) @ back
    T -= e2g_t @ back + l2e_t @ np.linalg.inv(l2e_r_mat).T
    raw_data[:, :3] = raw_data[:, :3] @ R + T
    return raw_data


def nus_select_sweeps(seq, index, multiscan, step):
    """nuscenes_ms.py:238-276: walk back from the current keyframe over the sweep list of its scene until the lidar has
    travelled more than multiscan * step metres (planar distance of each frame's sensor origin in the current lidar
    frame), pick for every multiple of `step` the frame whose distance is nearest (first one past it otherwise), add
    every keyframe on the way.  Returns ascending negative frame offsets.

    seq = dict(is_key [F], key_index [F], scene_tokens [F], local_indexes [F], global_indexes [K], s2l_r [F,3,3],
    s2l_t [F,3], keys = list of K pose dicts)."""
    g0 = int(seq["global_indexes"][index])
    info0 = seq["keys"][index]
    delta, total, dist = 0, [], []
    while len(dist) == 0 or dist[-1] <= multiscan * step:
        delta -= 1
        g = g0 + delta                     # (negative g wraps to the end of the list, as in the reference)
        if seq["scene_tokens"][g] != seq["scene_tokens"][g0]:
            dist.append(1000)
            break
        origin = np.zeros((1, 5), dtype=float)
        if not seq["is_key"][g]:
            origin[:, :3] = origin[:, :3] @ seq["s2l_r"][g].T
            origin[:, :3] += seq["s2l_t"][g]
        if seq["local_indexes"][g] != index:
            origin = nus_transform_point(origin, info0, seq["keys"][int(seq["local_indexes"][g])])
        total.append(delta)
        dist.append(np.linalg.norm(origin.reshape(-1)[:2], ord=2))
    cur, picked = 1, []
    for i in range(len(total)):
        if dist[i] - cur * step > 0 or ((dist[i] < dist[i + 1]) and
                                        (np.abs(dist[i] - cur * step) < np.abs(dist[i + 1] - cur * step))):
            picked.append(total[i])
            cur += 1
        if cur > multiscan:
            break
    picked += [d for d in total if seq["is_key"][g0 + d]]
    return sorted(set(picked))


def nus_multiscan_fuse(seq, index, sample_list, points, pseudo, labels, steps):
    """nuscenes_ms.py:280-346: per selected frame (oldest first) the ego-box filter on the RAW coordinates, the time
    delta in column 4, the rigid transform(s) into the current lidar frame, the class-step mask on the pseudo labels
    (`(position + 1) % step == 0`).  points / pseudo / labels: dicts keyed by the frame offset.  Returns the
    concatenated (raw [n,5] float32, labels [n], pseudo [n], mask [n]) - all AFTER the ego-box filter."""
    g0 = int(seq["global_indexes"][index])
    info0 = seq["keys"][index]
    raws, anns, pseudos, masks = [], [], [], []
    for pos, d in enumerate(sample_list):
        g = g0 + d
        raw = np.array(points[d], dtype=np.float32, copy=True).reshape(-1, 5)
        no_ego = ~((np.abs(raw[:, 0]) < 1.0) & (np.abs(raw[:, 1]) < 1.5))
        dt = seq["timestamps"][g0] / 1e6 - seq["timestamps"][g] / 1e6
        if seq["is_key"][g]:
            raw[:, 4] = dt
            raw = nus_transform_point(raw, info0, seq["keys"][int(seq["key_index"][g])])
            ann = np.asarray(labels[d]).reshape(-1)
        else:
            raw[:, :3] = raw[:, :3] @ seq["s2l_r"][g].T
            raw[:, :3] += seq["s2l_t"][g]
            raw[:, 4] = dt
            if seq["local_indexes"][g] != index:
                raw = nus_transform_point(raw, info0, seq["keys"][int(seq["local_indexes"][g])])
            ann = np.zeros(raw.shape[0], dtype=np.uint8)
        ps = np.asarray(pseudo[d]).reshape(-1)
        raw, ann, ps = raw[no_ego], ann[no_ego], ps[no_ego]
        m = np.zeros(len(ps), dtype=bool)
        for cls, st in enumerate(steps):
            if st == 0:
                continue
            if (pos + 1) % st == 0:
                m = m | (ps == cls)
        raws.append(raw)
        anns.append(ann)
        pseudos.append(ps)
        masks.append(m)
    return np.concatenate(raws, 0), np.concatenate(anns, 0), np.concatenate(pseudos, 0), np.concatenate(masks, 0)


# --------------------------------------------------------------------------- TIAF data stage (camera side)
def tiaf_fov_points(raw_data, proj_matrix, image_size, height, width, img_batch):
    """R/pcseg/data/dataset/semantickitti/semantickitti_ms_mm.py:411-461 (get_fov_points) without the image arrays:
    points in front of the camera whose projection falls inside the image and inside the (height, width) crop, with
    their pixel (row + height * img_batch, col) appended.  raw_data [n,4] float32, proj_matrix [3,4] float64 (P2 @ Tr),
    image_size = (W, H).  Returns (raw_fov [m,6] float32, keep [n] bool)."""
    raw_data = np.asarray(raw_data, dtype=np.float32)
    keep_mask = raw_data[:, 0] > 0
    xyz1 = np.concatenate([raw_data[:, :3][keep_mask], np.ones([keep_mask.sum(), 1], dtype=np.float32)], axis=1)
    uvz = (proj_matrix @ xyz1.T).T
    uv = uvz[:, :2] / np.expand_dims(uvz[:, 2], axis=1)
    frustum = (uv[:, 0] > 0) * (uv[:, 1] > 0) * (uv[:, 0] < image_size[0]) * (uv[:, 1] < image_size[1])
    keep_mask[keep_mask] = frustum
    uv = np.fliplr(uv)                                           # (row, col)
    frustum_uv = uv[frustum].astype(dtype=int)
    cropped = (frustum_uv[:, 0] < height) & (frustum_uv[:, 1] < width)
    keep_mask[keep_mask.nonzero()[0][~cropped]] = False
    frustum_uv = frustum_uv[cropped].astype(raw_data.dtype)
    frustum_uv[:, 0] += (height * img_batch)
    return np.concatenate([raw_data[keep_mask], frustum_uv], axis=-1), keep_mask


def tiaf_crop_image(image_u8_rgb, height, width):
    """semantickitti_ms_mm.py:432-447: float32, RGB -> BGR, / 255, top-left crop zero-padded to (height, width)."""
    image = np.array(image_u8_rgb, dtype=np.float32)
    image[..., [0, 1, 2]] = image[..., [2, 1, 0]]
    image = image / 255.
    r_max, c_max = min(height, image.shape[0]), min(width, image.shape[1])
    out = np.zeros((height, width, image.shape[2]), dtype=image.dtype)
    out[:r_max, :c_max] = image[:r_max, :c_max]
    return out


def kitti_ring_id(points):
    """semantickitti_ms_mm.py:131-141: beam index from the azimuth wrap-arounds of the scan order, clipped to 63."""
    scan_x, scan_y = points[:, 0], points[:, 1]
    yaw = -np.arctan2(scan_y, -scan_x)
    proj_x = 0.5 * (yaw / np.pi + 1.0)
    new_raw = np.nonzero((proj_x[1:] < 0.2) * (proj_x[:-1] > 0.8))[0] + 1
    proj_y = np.zeros_like(proj_x)
    proj_y[new_raw] = 1
    return np.clip(np.cumsum(proj_y), 0, 63)


# --------------------------------------------------------------------------- nuScenes TIAF data stage (camera side)
def view_points(points, view, normalize):
    """`nuscenes.utils.geometry_utils.view_points` of nuscenes-devkit (a dependency the reference imports,
    R/pcseg/data/dataset/nuscenes/nuscenes_ms_mm.py:15, without pinning a version; absent from /root/reference and from this
    image - its published algorithm, unchanged since devkit 1.0, restated): pad the view matrix to 4x4, apply it to the
    homogeneous points, keep three rows, divide by the third when `normalize`.  points [3, n]."""
    viewpad = np.eye(4)
    viewpad[:view.shape[0], :view.shape[1]] = view
    nbr_points = points.shape[1]
    points = np.concatenate((points, np.ones((1, nbr_points))))
    points = np.dot(viewpad, points)
    points = points[:3, :]
    if normalize:
        points = points / points[2:3, :].repeat(3, 0).reshape(3, nbr_points)
    return points


def nus_tiaf_fov_points(raw_data, annotated, cam, image_hw, height, img_batch, resize=0.5, crop_top=2):
    """nuscenes_ms_mm.py:329-401 (get_fov_points) without the image arrays: lidar points of a keyframe -> ego -> global ->
    camera ego -> camera (the calibrated-sensor and ego-pose records of the two sample_data entries), pinhole projection,
    in-image test, pixel at half resolution, the top `crop_top` rows cut, the row shifted by height * img_batch.
    raw_data [n,4] float32; cam = dict(lidar_cs_q, lidar_cs_t, lidar_pose_q, lidar_pose_t, cam_pose_q, cam_pose_t, cam_cs_q,
    cam_cs_t, intrinsic [3,3]); image_hw = (H, W) of the full-size image.  Returns (raw_fov [m,6] float64: x, y, z,
    intensity, row, col; labels [m]; keep [n])."""
    raw_data = np.asarray(raw_data, dtype=np.float32)
    pc = raw_data[:, :3].copy().T
    pc = quaternion_rotation_matrix(cam["lidar_cs_q"]) @ pc
    pc = pc + np.array(cam["lidar_cs_t"])[:, np.newaxis]
    pc = quaternion_rotation_matrix(cam["lidar_pose_q"]) @ pc
    pc = pc + np.array(cam["lidar_pose_t"])[:, np.newaxis]
    pc = pc - np.array(cam["cam_pose_t"])[:, np.newaxis]
    pc = quaternion_rotation_matrix(cam["cam_pose_q"]).T @ pc
    pc = pc - np.array(cam["cam_cs_t"])[:, np.newaxis]
    pc = quaternion_rotation_matrix(cam["cam_cs_q"]).T @ pc
    depths = pc[2, :]
    points = view_points(pc, np.array(cam["intrinsic"]), normalize=True).astype(np.float32)
    keep = depths > 0
    keep = np.logical_and(keep, points[0, :] > 0)
    keep = np.logical_and(keep, points[0, :] < image_hw[1])
    keep = np.logical_and(keep, points[1, :] > 0)
    keep = np.logical_and(keep, points[1, :] < image_hw[0])
    uv = points.T[:, :2][keep].astype(int)
    uv = np.ascontiguousarray(np.fliplr(uv))                     # (row, col) at full resolution
    uv[:, 0] = np.floor(resize * uv[:, 0])
    uv[:, 1] = np.floor(resize * uv[:, 1])
    crop = uv[:, 0] >= crop_top
    keep[keep] = crop
    uv = uv[crop]
    uv[:, 0] -= crop_top
    uv[:, 0] += height * img_batch
    return np.concatenate([raw_data[keep], uv], axis=-1), np.asarray(annotated).reshape(-1)[keep], keep


def nus_select_image_keyframes(keys, scene_of, index, multiscan_image, step_image, rng):
    """nuscenes_ms_mm.py:204-236 (head of multiscan_fuse_fov): keyframe offsets whose camera images join the stack - walk
    back over the KEYFRAME list of the scene until the lidar has travelled more than multiscan_image * step_image metres,
    pick per multiple of step_image the nearest keyframe, fill up from the passed-over ones with `rng.sample` (the
    reference uses the global `random.sample`), always add offset 0.  keys = list of pose dicts, scene_of[i] = scene of
    keyframe i."""
    if multiscan_image == 0:
        return [0]
    info0 = keys[index]
    delta, total, dist = 0, [], []
    while len(dist) == 0 or dist[-1] <= multiscan_image * step_image:
        delta -= 1
        if scene_of[index + delta] != scene_of[index]:
            dist.append(1000)
            break
        origin = nus_transform_point(np.zeros((1, 5), dtype=float), info0, keys[index + delta])
        total.append(delta)
        dist.append(np.linalg.norm(origin.reshape(-1)[:2], ord=2))
    cur, picked, passed = 1, [], []
    for i in range(len(total)):
        if dist[i] - cur * step_image > 0 or ((dist[i] < dist[i + 1]) and
                                              (np.abs(dist[i] - cur * step_image) < np.abs(dist[i + 1] - cur * step_image))):
            picked.append(total[i])
            cur += 1
        else:
            passed.append(total[i])
        if cur > multiscan_image:
            break
    if len(picked) < multiscan_image and len(passed) > 0:
        picked += rng.sample(passed, min(multiscan_image - len(picked), len(passed)))
    picked = list(set(picked))
    picked.append(0)
    picked.sort()
    return picked


def nus_tiaf_frame_cloud(keys, scene_of, stamps, index, delta, prev_delta, interval, points, labels, paint_dist):
    """nuscenes_ms_mm.py:246-300: the cloud one image keyframe projects - its own points (ego box cut on the raw
    coordinates, time delta to the current keyframe in column 4) plus those of up to `interval` earlier keyframes of the
    scene (not reaching back to the previous image keyframe `prev_delta`), moved into ITS lidar frame; then the paint
    radius.  points[i] / labels[i]: raw [n,5] float32 / mapped labels of keyframe i.  Returns (raw [m,5] float32, labels [m])."""
    i = index + delta
    raw = np.array(points[i], dtype=np.float32, copy=True).reshape(-1, 5)
    no_ego = ~((np.abs(raw[:, 0]) < 1.0) & (np.abs(raw[:, 1]) < 1.5))
    raw[:, 4] = stamps[index] / 1e6 - stamps[i] / 1e6
    ann = np.asarray(labels[i]).reshape(-1, 1)
    raw, ann = raw[no_ego], ann[no_ego]
    for meta in range(-interval, 0):
        j = i + meta
        if j < 0 or j >= len(keys) or scene_of[j] != scene_of[index]:
            continue
        if prev_delta is not None and delta + meta <= prev_delta:
            continue
        raw1 = np.array(points[j], dtype=np.float32, copy=True).reshape(-1, 5)
        no_ego1 = ~((np.abs(raw1[:, 0]) < 1.0) & (np.abs(raw1[:, 1]) < 1.5))
        raw1[:, 4] = stamps[index] / 1e6 - stamps[j] / 1e6
        ann1 = np.asarray(labels[j]).reshape(-1, 1)
        raw1, ann1 = raw1[no_ego1], ann1[no_ego1]
        raw1 = nus_transform_point(raw1, keys[i], keys[j])
        raw = np.concatenate([raw, raw1], axis=0)
        ann = np.concatenate([ann, ann1], axis=0)
    if paint_dist > 0:
        radius = np.linalg.norm(raw[:, :2], ord=2, axis=1, keepdims=False)
        raw, ann = raw[radius <= paint_dist], ann[radius <= paint_dist]
    return raw, ann.reshape(-1)
