"""LOSS registry (reference: sseg/models/modules/losses.py:32-41 'CE', 'SoftCE' with the
region-selection helper :75-89).  Same call signature
    LOSS[name](logits, labels, weights=None, ignore_index=255, refer_labels=None, region=...)
on FULL-RES logits [B,C,H,W].  The argument combinations the HIAST configs use go through the fused HIP loss kernel
(identity upsample); the rest of the signature (per-class weights, refer_labels + region with any loss, SoftCE's plain
mean, another ignore_index) is honoured by the general composition of device ops, formula by formula as the reference.
The trainers of this package use the low-res fused path (hiast_amd.functional.st_loss) and never materialise
full-res logits.

'MSE', 'KLDIV', 'BCEWithLogits' (losses.py:10-30) serve the adversarial warm-up stage; they act on
the discriminator's B x 1 x H/32 x W/32 maps."""
import torch

from hiast_amd import functional as HF
from hiast_amd.utils.registry.registries import LOSS


def _need_hip(t):
    if not t.is_cuda:
        raise RuntimeError("hiast_amd losses run on the HIP device only (got a %s tensor); "
                           "there is no CPU fallback" % t.device)


def _select(loss_tensor, refer_labels, ignore_index, region):
    """compute_loss_by_selected_pixel (losses.py:75-89): loss * region mask [B,1,H,W], sum / count of NON-ZERO elements.
    A per-pixel tensor [B,H,W] (CE) broadcasts against the mask to [B,B,H,W] there (out[i,j] = loss[j] * mask[i]) — kept."""
    if region == "ignored":
        mask = refer_labels == ignore_index
    elif region == "confident":
        mask = refer_labels != ignore_index
    elif region == "all":
        mask = torch.ones_like(refer_labels, dtype=torch.bool)
    else:
        raise ValueError("{} is not a valid region".format(region))
    t = loss_tensor * mask.unsqueeze(1)
    return t.sum() / (t != 0).sum()


@LOSS.register("CE")
def ce(logits, labels, weights=None, ignore_index=255, refer_labels=None, region="confident"):
    """nn.CrossEntropyLoss(ignore_index, weight) mean over non-ignored pixels (losses.py:32-36).  The combination every
    HIAST config uses (no class weights, ignore_index 255, no refer_labels) is the fused HIP loss kernel; per-class
    `weights`, another ignore_index, or refer_labels + region (losses.py:68-89) take the general composition on the
    device — the reference's formulas operation by operation (pinned by tests/golden/loss_registry.npz)."""
    _need_hip(logits)
    if weights is None and ignore_index == 255 and refer_labels is None:
        H, W = logits.shape[2:]
        loss, _, _, _ = HF.st_loss(logits, None, labels, (H, W), "ignored", 1.0, 0.0, 0.0, 0.0)
        return loss
    F = torch.nn.functional
    w = None if weights is None else torch.as_tensor(weights, dtype=torch.float32, device=logits.device)
    if refer_labels is None:
        return F.cross_entropy(logits.float(), labels.long(), weight=w, ignore_index=ignore_index)
    return _select(F.cross_entropy(logits.float(), labels.long(), weight=w, reduction="none"), refer_labels, ignore_index, region)


@LOSS.register("SoftCE")
def soft_ce(logits, labels, weights=None, ignore_index=255, refer_labels=None, region="confident"):
    """-log_softmax(logits) * labels on `region` of refer_labels, divided by the number of non-zero elements
    (losses.py:39-65,75-89); without refer_labels: sum / numel (:63-65).  `labels` are probabilities [B,C,H,W].  The
    combination the HIAST setting uses (refer_labels given, ignore_index 255, no class weights) is the fused HIP loss
    kernel — the probabilities enter it as log-probabilities (softmax(log q) == q up to rounding; q == 0 stays 0) —, the
    rest the general composition on the device.  `weights` scale the target IN PLACE, as the reference does (:57-59)."""
    _need_hip(logits)
    assert logits.shape == labels.shape
    if weights is None and ignore_index == 255 and refer_labels is not None:
        H, W = logits.shape[2:]
        _, _, _, loss = HF.st_loss(logits, torch.log(labels), refer_labels, (H, W), region, 0.0, 0.0, 0.0, 1.0)
        return loss
    if weights is not None:
        assert len(weights) == labels.shape[1]
        labels.mul_(torch.as_tensor(weights, dtype=labels.dtype, device=labels.device).view(1, -1, 1, 1))
    t = -torch.log_softmax(logits.float(), dim=1) * labels
    if refer_labels is None:
        return t.sum() / labels.numel()
    return _select(t, refer_labels, ignore_index, region)


@LOSS.register("MSE")
def mse(logits, labels, weights=None, ignore_index=255, refer_labels=None, region="ignore"):
    """nn.MSELoss() (losses.py:10-14); the discriminator maps it is applied to have B*16*32 elements, so the
    reduction is device plumbing, not a kernel of its own."""
    _need_hip(logits)
    if refer_labels is not None:
        return _select(torch.nn.functional.mse_loss(logits.float(), labels.float(), reduction="none"), refer_labels,
                       ignore_index, region)
    return torch.nn.functional.mse_loss(logits.float(), labels.float())


@LOSS.register("KLDIV")
def kl_div(input_logits, target_logits, weights=None, ignore_index=255, refer_labels=None, region="confident"):
    """nn.KLDivLoss() default reduction = element mean of q (log q - log p) (losses.py:17-24)."""
    _need_hip(input_logits)
    logp = torch.log_softmax(input_logits.float(), dim=1)
    q = torch.softmax(target_logits.float(), dim=1)
    if refer_labels is not None:
        return _select(torch.nn.functional.kl_div(logp, q, reduction="none"), refer_labels, ignore_index, region)
    return torch.nn.functional.kl_div(logp, q, reduction="mean")


@LOSS.register("BCEWithLogits")
def bce_with_logits(logits, labels):
    """nn.BCEWithLogitsLoss() (losses.py:27-30)"""
    _need_hip(logits)
    return torch.nn.functional.binary_cross_entropy_with_logits(logits.float(), labels.float())
