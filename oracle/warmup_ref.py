"""torch-CPU restatement of the adversarial warm-up losses (floating-point oracle).

TEST INFRASTRUCTURE ONLY.  Parity: PINNED by tests/golden/warmup.npz (outputs of the reference's
AdversarialWarmupSegmentor.forward + the two backward passes of BaseTrainer.update_model, see
tests/golden/make_golden.py::g_warmup).

Closed forms of sseg/models/segmentors/adversarial_warmup_segmentor.py:33-86 downstream of the segmentation net,
written on LOW-RES logits (the bilinear upsample of :36,:41 is part of the function), of
sseg/models/modules/discriminator.py:7-33 and of losses.py:10-30 ('MSE', 'BCEWithLogits').
"""
import math

import torch
import torch.nn.functional as F


def discriminator_input(logits_lr, size, entropy, dtype=torch.float64, interp_dtype=None):
    """D_preprocess_fun(F.interpolate(logits)) (:26-29,:36,:41,:71-76).  interp_dtype=torch.float32 keeps the
    bilinear source-coordinate arithmetic in fp32 as the reference (and the kernel) have it; the softmax / log2
    stage then runs in `dtype`."""
    z = F.interpolate(logits_lr.to(interp_dtype or dtype), size=size, mode="bilinear", align_corners=True).to(dtype)
    p = F.softmax(z, dim=1)
    if entropy:
        return -(p * torch.log2(p + 1e-30)) / math.log2(p.shape[1])
    return p


def discriminator(x, sd, prefix=""):
    """FCDiscriminator.forward (discriminator.py:19-29) from a state dict"""
    for name in ("conv1", "conv2", "conv3", "conv4", "classifier"):
        x = F.conv2d(x, sd[prefix + name + ".weight"].to(x.dtype), sd[prefix + name + ".bias"].to(x.dtype), stride=2,
                     padding=1)
        if name != "classifier":
            x = F.leaky_relu(x, 0.2)
    return x


def d_loss_fun(kind):
    if kind == "MSE":
        return lambda a, b: F.mse_loss(a, b)
    if kind == "BCEWithLogits":
        return lambda a, b: F.binary_cross_entropy_with_logits(a, b)
    raise KeyError(kind)


def warmup_losses(s_lr, t_lr, s_lbl, size, d_sd, d_loss="MSE", entropy_in=False, source_weight=1.0, adv_weight=0.05,
                  d_weight=1.0, ent_weight=3.0, dtype=torch.float64):
    """-> the reference's loss dict (:43-65).  d_sd: discriminator state dict (tensors may require grad)."""
    C = s_lr.shape[1]
    fun = d_loss_fun(d_loss)
    zs = F.interpolate(s_lr.to(dtype), size=size, mode="bilinear", align_corners=True)
    out = {"source_seg_loss": source_weight * F.cross_entropy(zs, s_lbl.long(), ignore_index=255)}
    x_t = discriminator_input(t_lr, size, entropy_in, dtype)
    frozen = {k: v.detach() for k, v in d_sd.items()}
    t_adv = discriminator(x_t, frozen)
    out["adv_loss"] = adv_weight * fun(t_adv, torch.zeros_like(t_adv))
    s_d = discriminator(discriminator_input(s_lr.detach(), size, entropy_in, dtype), d_sd)
    t_d = discriminator(x_t.detach(), d_sd)
    out["D_loss"] = d_weight * (fun(s_d, torch.zeros_like(s_d)) + fun(t_d, torch.ones_like(t_d))) / 2
    if ent_weight > 0:
        p = F.softmax(F.interpolate(t_lr.to(dtype), size=size, mode="bilinear", align_corners=True), dim=1)
        n, _, hh, ww = p.shape
        out["target_ent_loss"] = ent_weight * (-(p * torch.log2(p + 1e-30)).sum() / (n * hh * ww * math.log2(C)))
    return out
