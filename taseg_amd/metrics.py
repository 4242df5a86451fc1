"""Segmentation metrics of the reference's evaluation loop, on device tensors.

`fast_hist` / `per_class_iu` follow R/train.py:35-45 (confusion matrix by one bincount over n * label + pred for the
labels inside [0, n); IoU = diag / (row sum + column sum - diag + 1e-9)); the reference evaluates them on numpy arrays
after copying every prediction to the host, here they stay on the device and the per-rank matrices are merged with
`taseg_amd.parallel.reduce_confusion`."""
import torch

__all__ = ["fast_hist", "per_class_iu", "mean_iou"]


def fast_hist(pred: torch.Tensor, label: torch.Tensor, n: int) -> torch.Tensor:
    """[n, n] int64 confusion matrix, rows = label, columns = prediction; labels outside [0, n) are skipped."""
    pred, label = pred.reshape(-1).long(), label.reshape(-1).long()
    keep = (label >= 0) & (label < n)
    return torch.bincount(n * label[keep] + pred[keep], minlength=n * n)[:n * n].reshape(n, n)


def per_class_iu(hist: torch.Tensor) -> torch.Tensor:
    hist = hist.double()
    diag = torch.diagonal(hist)
    return diag / (hist.sum(1) + hist.sum(0) - diag + 1e-9)


def mean_iou(hist: torch.Tensor, ignore_index: int = 0) -> float:
    """mIoU over the classes the trainer reports: every class but `ignore_index` (R/train.py:47-52 crops the matrix to
    `unique_label` = the learning classes, :576-584 averages their IoU)."""
    iu = per_class_iu(hist)
    keep = torch.ones_like(iu, dtype=torch.bool)
    if 0 <= ignore_index < iu.numel():
        keep[ignore_index] = False
    return float(iu[keep].mean())
