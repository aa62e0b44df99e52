"""numpy restatement of the reference's mIoU evaluation.

Test infrastructure only (see oracle/__init__.py).
"""
import numpy as np


def confusion_matrix(predictions, labels, num_classes, ignore_label=255):
    """libs/evaluator.py:143-154: weights = (label != ignore); ignored labels
    are remapped to class 0 with weight 0; cm[label, prediction] += weight.
    TF accumulates in float64, which is integer-exact; int64 here."""
    p = np.asarray(predictions).reshape(-1).astype(np.int64)
    l = np.asarray(labels).reshape(-1).astype(np.int64)
    keep = l != ignore_label
    l = np.where(keep, l, 0)
    cm = np.zeros((num_classes, num_classes), dtype=np.int64)
    np.add.at(cm, (l[keep], p[keep]), 1)
    return cm


def mean_iou(cm):
    """tf.contrib.metrics.streaming_mean_iou's value op (TF 1.8
    metrics_impl.mean_iou.compute_mean_iou), float32 arithmetic."""
    cm = np.asarray(cm)
    sum_over_row = cm.sum(axis=0).astype(np.float32)
    sum_over_col = cm.sum(axis=1).astype(np.float32)
    diag = np.diagonal(cm).astype(np.float32)
    den = sum_over_row + sum_over_col - diag
    num_valid = np.float32((den != 0).sum())
    den = np.where(den > 0, den, np.float32(1))
    iou = diag / den
    if num_valid > 0:
        return np.float32(iou.sum(dtype=np.float32) / num_valid)
    return np.float32(0)
