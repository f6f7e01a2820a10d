"""LOSS registry (reference: sseg/models/modules/losses.py:32-41 'CE', 'SoftCE' with the
region-selection helper :75-89).  Same call signature
    LOSS[name](logits, labels, weights=None, ignore_index=255, refer_labels=None, region=...)
on FULL-RES logits [B,C,H,W].  Both go through the fused HIP loss kernel (identity upsample);
the trainers of this package use the low-res fused path (hiast_amd.functional.st_loss) instead
and never materialise full-res logits.

'MSE', 'KLDIV', 'BCEWithLogits' belong to the adversarial warm-up stage, which is outside the
self-training hot path; they are registered so that lookups fail with a clear
message rather than a KeyError."""
import torch

from hiast_amd import functional as HF
from hiast_amd.utils.registry.registries import LOSS


def _need_hip(t):
    if not t.is_cuda:
        raise RuntimeError("hiast_amd losses run on the HIP device only (got a %s tensor); "
                           "there is no CPU fallback" % t.device)


def _no_class_weights(weights):
    if weights is not None:
        raise NotImplementedError("per-class loss weights are not used by any HIAST config and are not implemented")


@LOSS.register("CE")
def ce(logits, labels, weights=None, ignore_index=255, refer_labels=None, region="confident"):
    """nn.CrossEntropyLoss(ignore_index=255) mean over non-ignored pixels (losses.py:35)."""
    _no_class_weights(weights)
    _need_hip(logits)
    if ignore_index != 255:
        raise NotImplementedError("ignore_index is fixed to 255")
    if refer_labels is not None:
        raise NotImplementedError("CE with refer_labels is not used by the self-training path")
    H, W = logits.shape[2:]
    loss, _, _, _ = HF.st_loss(logits, None, labels, (H, W), "ignored", 1.0, 0.0, 0.0, 0.0)
    return loss


@LOSS.register("SoftCE")
def soft_ce(logits, labels, weights=None, ignore_index=255, refer_labels=None, region="confident"):
    """-log_softmax(logits) * labels on `region` of refer_labels, divided by the number of
    non-zero elements (losses.py:61,75-89).  `labels` are probabilities [B,C,H,W]; they enter
    the kernel as log-probabilities (softmax(log q) == q up to rounding; q == 0 stays 0)."""
    _no_class_weights(weights)
    _need_hip(logits)
    if ignore_index != 255:
        raise NotImplementedError("ignore_index is fixed to 255")
    assert logits.shape == labels.shape
    H, W = logits.shape[2:]
    if refer_labels is None:
        raise NotImplementedError("SoftCE without refer_labels (plain mean) is not used by the self-training path")
    _, _, _, loss = HF.st_loss(logits, torch.log(labels), refer_labels, (H, W), region, 0.0, 0.0, 0.0, 1.0)
    return loss


def _out_of_scope(name):
    def fn(*a, **k):
        raise NotImplementedError("LOSS[%r] belongs to the adversarial warm-up stage, which is outside the "
                                  "self-training hot path implemented here" % name)
    return fn


for _n in ("MSE", "KLDIV", "BCEWithLogits"):
    LOSS.register(_n, _out_of_scope(_n))
