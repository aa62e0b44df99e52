"""Streaming mean IoU on the GPU.

Counterpart of tf.contrib.metrics.streaming_mean_iou as the reference uses it
(libs/evaluator.py:143-165): a num_classes x num_classes confusion matrix is
accumulated on the device as int64 (TensorFlow keeps float64 counts, which are
integer-exact), optionally summed across ranks, and the final value is computed
with TensorFlow's float32 formula.
"""
import ctypes

import numpy as np
import torch

from fss import _lib


def mean_iou_from_confusion(cm):
    """TF 1.8 metrics_impl.mean_iou's compute_mean_iou, float32 arithmetic:
    classes whose denominator is zero are left out of the mean."""
    cm = np.asarray(cm)
    sum_over_row = cm.sum(axis=0).astype(np.float32)
    sum_over_col = cm.sum(axis=1).astype(np.float32)
    diag = np.diagonal(cm).astype(np.float32)
    den = sum_over_row + sum_over_col - diag
    num_valid = np.float32(np.count_nonzero(den))
    den = np.where(den > 0, den, np.float32(1))
    iou = (diag / den).astype(np.float32)
    if num_valid > 0:
        return np.float32(iou.sum(dtype=np.float32) / num_valid)
    return np.float32(0)


class StreamingMeanIoU(object):
    def __init__(self, num_classes, ignore_label=255, device=None):
        if not 0 < num_classes <= 64:
            raise ValueError('num_classes must be in [1, 64]')
        self.num_classes = int(num_classes)
        self.ignore_label = int(ignore_label)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else device
        self.cm = torch.zeros((self.num_classes, self.num_classes), dtype=torch.int64,
                              device=self.device)

    def update(self, predictions_u8, labels_u8, stream_handle=None):
        """predictions / labels: uint8 CUDA tensors of equal element count.
        Labels equal to `ignore_label` carry weight 0; everything else counts."""
        if predictions_u8.dtype != torch.uint8 or labels_u8.dtype != torch.uint8:
            raise TypeError('predictions and labels must be uint8 tensors')
        if predictions_u8.numel() != labels_u8.numel():
            raise ValueError('predictions and labels differ in size')
        if not (predictions_u8.is_contiguous() and labels_u8.is_contiguous()):
            raise ValueError('predictions and labels must be contiguous')
        _lib.call('fss_confusion_accumulate', predictions_u8.data_ptr(), labels_u8.data_ptr(),
                  predictions_u8.numel(), self.num_classes, self.ignore_label,
                  self.cm.data_ptr(), stream_handle)

    def all_reduce(self):
        """Sums the confusion matrix over all ranks (RCCL when the process group
        runs on GPUs, gloo in CPU tests).  No-op without a process group; with one, the
        collective runs even for a single rank (identity), so the path is the same at any size."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            if dist.get_backend() == 'gloo':
                t = self.cm.cpu()
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                self.cm.copy_(t)
            else:
                dist.all_reduce(self.cm, op=dist.ReduceOp.SUM)

    def confusion_matrix(self):
        return self.cm.cpu().numpy()

    def result(self):
        return mean_iou_from_confusion(self.confusion_matrix())
