"""LOSS registry (reference: sseg/models/modules/losses.py:32-41 'CE', 'SoftCE' with the
region-selection helper :75-89).  Same call signature
    LOSS[name](logits, labels, weights=None, ignore_index=255, refer_labels=None, region=...)
on FULL-RES logits [B,C,H,W].  Both go through the fused HIP loss kernel (identity upsample);
the trainers of this package use the low-res fused path (hiast_amd.functional.st_loss) instead
and never materialise full-res logits.

'MSE', 'KLDIV', 'BCEWithLogits' (losses.py:10-30) serve the adversarial warm-up stage; they act on
the discriminator's B x 1 x H/32 x W/32 maps."""
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


def _plain(name, refer_labels):
    if refer_labels is not None:
        raise NotImplementedError("LOSS[%r] with refer_labels / region selection is not used by any trainer" % name)


@LOSS.register("MSE")
def mse(logits, labels, weights=None, ignore_index=255, refer_labels=None, region="ignore"):
    """nn.MSELoss() (losses.py:10-14); the discriminator maps it is applied to have B*16*32 elements, so the
    reduction is device plumbing, not a kernel of its own."""
    _need_hip(logits)
    _plain("MSE", refer_labels)
    return torch.nn.functional.mse_loss(logits.float(), labels.float())


@LOSS.register("KLDIV")
def kl_div(input_logits, target_logits, weights=None, ignore_index=255, refer_labels=None, region="confident"):
    """nn.KLDivLoss() default reduction = element mean of q (log q - log p) (losses.py:17-24)."""
    _need_hip(input_logits)
    _plain("KLDIV", refer_labels)
    logp = torch.log_softmax(input_logits.float(), dim=1)
    return torch.nn.functional.kl_div(logp, torch.softmax(target_logits.float(), dim=1), reduction="mean")


@LOSS.register("BCEWithLogits")
def bce_with_logits(logits, labels):
    """nn.BCEWithLogitsLoss() (losses.py:27-30)"""
    _need_hip(logits)
    return torch.nn.functional.binary_cross_entropy_with_logits(logits.float(), labels.float())
