"""Evaluation / test-time-augmentation side of the path: what the reference's trainer does
with the dictionary a segmentor's eval branch returns (R/train.py:452-611) and the label remap of the saved predictions
(R/tta_remap.py:95-155), as plain functions around the model - no trainer, logger or file-system layout of its own.

    ret = model(batch)                                  # eval mode: point_predict / point_labels / point_predict_logits
    hist += scan_confusions(ret, unique_label)          # validation: R/train.py:535-540 -> per_class_iu
    logits = accumulate_votes(ret, votes)               # TTA: one batch entry per vote (collate_batch_tta), summed
    write_prediction(path, vote_payload(logits, "semantickitti"))
"""
import os
from typing import Dict, Iterable, List, Sequence

import numpy as np
from taseg_amd.options import options

__all__ = ["accumulate_votes", "vote_payload", "write_prediction", "remap_lut", "remap_labels", "fast_hist",
           "fast_hist_crop", "per_class_iu", "scan_confusions", "evaluate"]


def accumulate_votes(ret_dict: Dict, votes: int) -> np.ndarray:
    """Sum of the per-vote point logits of ONE scan (R/train.py:474-477, 505-508): under TTA the dataset returns
    `votes` augmented copies of a scan as one batch (semantickitti_voxel_ms.py:66-72, collate_batch_tta), the eval
    branch un-voxelises each copy onto the scan's points, and the trainer adds the raw logits."""
    logits = ret_dict["point_predict_logits"]
    if len(logits) < votes:
        raise ValueError(f"eval dictionary holds {len(logits)} votes, {votes} expected")
    total = np.array(logits[0], copy=True)
    for count in range(1, votes):
        if logits[count].shape != total.shape:
            raise ValueError("votes must cover the same points (one scan per TTA batch)")
        total += logits[count]
    return total


def vote_payload(point_logits: np.ndarray, dataset: str = "semantickitti") -> np.ndarray:
    """What the trainer writes per scan: arg-max class as [n, 1] uint32 (.label, SemanticKITTI; R/train.py:499-503) or
    uint8 (nuScenes lidarseg .bin; :519-523, where class 0 must not occur)."""
    pred = np.expand_dims(np.argmax(point_logits, axis=1), axis=1)
    if dataset.startswith("nuscenes"):
        out = pred.astype(np.uint8)
        if (out == 0).any():
            raise ValueError("nuScenes submission must not contain the ignored class 0 (R/train.py:523)")
        return out
    return pred.astype(np.uint32)


def write_prediction(path: str, payload: np.ndarray) -> str:
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    payload.tofile(path)
    return path


def remap_lut(remapdict: Dict[int, int]) -> np.ndarray:
    """Lookup table of R/tta_remap.py:103-108 (size max key + 100, unknown labels map to 0)."""
    lut = np.zeros(max(remapdict.keys()) + 100, dtype=np.int32)
    lut[list(remapdict.keys())] = list(remapdict.values())
    return lut


def remap_labels(label: np.ndarray, lut: np.ndarray) -> np.ndarray:
    """R/tta_remap.py:147-154: the lower 16 bits (semantics) go through the table, the upper 16 (instance) stay."""
    label = np.asarray(label, dtype=np.uint32).reshape(-1)
    upper, lower = label >> 16, label & 0xFFFF
    return ((upper << 16) + lut[lower].astype(np.uint32)).astype(np.uint32)


def fast_hist(pred, label, n):
    """R/train.py:35-40."""
    pred, label = np.asarray(pred), np.asarray(label)
    k = (label >= 0) & (label < n)
    return np.bincount(n * label[k].astype(int) + pred[k], minlength=n ** 2)[:n ** 2].reshape(n, n)


def per_class_iu(hist):
    """R/train.py:43-44."""
    hist = np.asarray(hist)
    return np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist) + 1e-9)


def fast_hist_crop(output, target, unique_label):
    """R/train.py:47-52: confusion matrix over `unique_label` (the learning classes without the ignored one)."""
    unique_label = np.asarray(unique_label)
    hist = fast_hist(np.asarray(output).flatten(), np.asarray(target).flatten(), np.max(unique_label) + 2)
    hist = hist[unique_label + 1, :]
    return hist[:, unique_label + 1]


def scan_confusions(ret_dict: Dict, unique_label: Sequence[int]) -> np.ndarray:
    """Sum over the scans of a batch of the cropped confusion matrices (R/train.py:535-540).  Note the reference's
    index shift: `unique_label` holds class ids - 1 (np.arange(num_class - 1), :148-150) and predictions / labels are
    compared as they are, so row r of the result is class r + 1."""
    total = None
    for pred, label in zip(ret_dict["point_predict"], ret_dict["point_labels"]):
        h = fast_hist_crop(pred, label, unique_label)
        total = h if total is None else total + h
    return total


def _staged(model, batches: Iterable[Dict], prefetch: bool):
    """the batches with their index plan (coordinates only: coordinate pyramid, kernel maps, class plans, trilinear maps) built one
    batch AHEAD on a second stream / worker thread (data.stage.DevicePrefetcher) - the role of the reference's DataLoader workers;
    an evaluation pass then is forward + un-voxelisation only (bench.py --eval: 5.2 + 1.2 ms instead of 2.9 + 5.2 + 1.2 per batch)"""
    prepare = getattr(model, "prepare", None)
    if not prefetch or prepare is None:
        yield from batches
        return
    from ..data.stage import DevicePrefetcher
    it = iter(batches)
    done = object()
    import os
    # two batches staged ahead (each on a stream and thread of its own; the batches are taken from `batches` one after the other, in
    # order): beside a forward pass one index plan takes longer than the pass (bench.py --eval: 5.9-6.3 -> 5.4 ms fp32, 4.2-5.1 -> 3.5-3.8
    # ms under autocast).  TASEG_EVAL_STAGE_DEPTH=1: one.
    pf = DevicePrefetcher(lambda: next(it, done), lambda b: b if b is done else prepare(b), threaded=True,
                          depth=options.eval_stage_depth)
    try:
        while True:
            batch = pf.next()
            if batch is done:
                return
            pf.prefetch_early()
            yield batch
    finally:
        pf.close()


def evaluate(model, batches: Iterable[Dict], num_class: int, tta_votes: int = 0, dataset: str = "semantickitti",
             save_dir: str = None, prefetch: bool = True) -> Dict:
    """The loop body of Trainer.evaluate (R/train.py:465-540) over already collated, device-resident batches: validation
    (mIoU over the classes 1 .. num_class - 1) or, with tta_votes > 0, vote accumulation and optional writing of one
    prediction file per scan.  Returns {"iou", "miou", "hist"} or {"predictions": [...]}.  prefetch: stage the index plan of the
    next batch while this one runs (same results)."""
    import torch
    unique_label = np.arange(num_class - 1)
    hist, preds = np.zeros((num_class - 1, num_class - 1), dtype=np.int64), []
    was_training = model.training
    model.eval()
    try:
        # one batch in flight: the forward pass of batch i + 1 is issued before the host waits for the arrays of batch i (models
        # whose forward takes `defer`; see minkunet.PendingPredictions)
        import inspect
        can_defer = "defer" in inspect.signature(model.forward).parameters

        def results():
            pending = None
            for batch in _staged(model, batches, prefetch):
                with torch.no_grad():
                    cur = model(batch, defer=True) if can_defer else model(batch)
                if pending is not None:
                    yield pending.result()
                if can_defer:
                    pending = cur
                else:
                    yield cur
            if pending is not None:
                yield pending.result()

        for ret in results():
            if tta_votes:
                payload = vote_payload(accumulate_votes(ret, tta_votes), dataset)
                preds.append(payload)
                if save_dir is not None:
                    name = ret["name"][0]
                    if dataset.startswith("nuscenes"):
                        path = os.path.join(save_dir, name.split("/")[-1])
                    else:       # .../sequences/<seq>/velodyne/<frame>.bin -> sequences/<seq>/predictions/<frame>.label
                        seq_id, frame_id = name.split("/")[-3], name.split("/")[-1]
                        path = os.path.join(save_dir, "sequences", seq_id, "predictions", frame_id.replace("bin", "label"))
                    write_prediction(path, payload)
            else:
                hist += scan_confusions(ret, unique_label)
    finally:
        model.train(was_training)
    if tta_votes:
        return {"predictions": preds}
    iou = per_class_iu(hist)
    return {"iou": iou, "miou": float(np.nanmean(iou) * 100), "hist": hist}
