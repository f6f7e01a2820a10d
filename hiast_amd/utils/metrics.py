"""intersectionAndUnionGPU (reference: utils/metrics.py:6-19) on the HIP integer-histogram kernel.
Returns float32 tensors like the reference's histc (counts are accumulated in int64 on the device)."""
import torch

from hiast_amd import kernels as K


def intersection_union_counts(output, target, num_classes):
    """int64 (intersection, union) [K]; CUDA tensors -> HIP kernel, CPU tensors -> bincount."""
    output = output.reshape(-1).long()
    target = target.reshape(-1).long()
    if output.is_cuda:
        inter, ap, at = K.confusion_hist(output.contiguous(), target.contiguous(), num_classes)
    else:   # CPU-only validate (config 1): integer bincounts, same definition
        out = torch.where(target == 255, torch.full_like(output, 255), output)
        valid_o = (out >= 0) & (out < num_classes)
        valid_t = (target >= 0) & (target < num_classes)
        ap = torch.bincount(out[valid_o], minlength=num_classes)
        at = torch.bincount(target[valid_t], minlength=num_classes)
        same = (out == target) & valid_o
        inter = torch.bincount(out[same], minlength=num_classes)
    return inter, ap + at - inter


def intersectionAndUnionGPU(output, target, K_, ignore_index=255):
    assert output.dim() in (1, 2, 3) and output.shape == target.shape and ignore_index == 255
    inter, union = intersection_union_counts(output, target, K_)
    return inter.float(), union.float()
