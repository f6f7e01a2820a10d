"""IoU / mIoU restatement (utils/metrics.py:6-19, workflows/validator.py:105-113).  TEST ONLY.
Parity: PINNED by tests/golden/metrics.npz."""
import numpy as np

from . import cref


def intersection_and_union(pred, target, K):
    inter, ap, at = cref.confusion_hist(pred, target, K)
    return inter, ap + at - inter


def miou(inter_sum, union_sum, synthia=False):
    iou = inter_sum / (union_sum + 1e-10)
    m = float(np.mean(iou))
    if synthia:
        iu13 = iou.copy()
        iu13[3:6] = 0
        return m * 19 / 16, float(np.mean(iu13)) * 19 / 13, iou
    return m, None, iou
