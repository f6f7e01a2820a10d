"""torch-CPU restatement of the reference's self-training loss (floating-point oracle).

TEST INFRASTRUCTURE ONLY.  Parity: PINNED by tests/golden/losses.npz (outputs of the reference's
SelfTrainingSegmentor.compute_loss + autograd, see tests/golden/make_golden.py).

Closed forms of sseg/models/segmentors/self_training_segmentor.py:30-53,128-163 and
sseg/models/modules/losses.py:32-65,75-89, written on LOW-RES logits: the bilinear upsample of
SelfTrainingSegmentor.forward (:27) and the teacher softmax of
workflows/trainer/consistency_self_training_trainer.py:113-119 are part of the function, which
is what the fused HIP kernel computes.
"""
import torch
import torch.nn.functional as F

REGIONS = {"ignored": 0, "confident": 1, "all": 2}


def st_loss_sums(logits_lr, teacher_lr, plbl, size, region="ignored", dtype=torch.float64):
    """-> dict of the 4 numerators and 3 denominators (see include/hiast_hip.h, hiast_st_loss_fwd)."""
    z = F.interpolate(logits_lr.to(dtype), size=size, mode="bilinear", align_corners=True)
    C = z.shape[1]
    logp = torch.log_softmax(z, dim=1)
    p = logp.exp()
    conf = (plbl != 255)
    ign = ~conf
    y = plbl.clone().long()
    y[ign] = 0
    nll_y = -logp.gather(1, y.unsqueeze(1)).squeeze(1)
    s = {}
    s["ce"] = (nll_y * conf).sum()                                     # losses.py:35 numerator
    s["kld"] = (-(logp.sum(1)) / C * conf).sum()                       # self_training_segmentor.py:160-162
    s["ent"] = ((-(p * logp).sum(1)) * ign).sum()                      # :147-149
    s["n_conf"] = conf.sum().to(dtype)
    s["n_ign"] = ign.sum().to(dtype)
    if teacher_lr is not None:
        zt = F.interpolate(teacher_lr.float(), size=size, mode="bilinear", align_corners=True)
        q = F.softmax(zt, dim=1)                                       # fp32, as the trainer does
        mask = {"ignored": ign, "confident": conf, "all": torch.ones_like(conf)}[region]
        elem = (-logp.float() * q) * mask.unsqueeze(1)                 # losses.py:61, :87 in fp32
        s["cst"] = (-(logp) * q.to(dtype) * mask.unsqueeze(1)).sum()
        s["cst_cnt"] = (elem != 0).sum().to(dtype)                     # losses.py:89
    else:
        s["cst"] = torch.zeros((), dtype=dtype)
        s["cst_cnt"] = torch.zeros((), dtype=dtype)
    return s


def st_losses(logits_lr, teacher_lr, plbl, size, region="ignored", w_t=1.0, w_k=0.1, w_e=1.0,
              w_c=0.5, dtype=torch.float64):
    """The four loss values of compute_loss (0/0 -> NaN like the reference)."""
    s = st_loss_sums(logits_lr, teacher_lr, plbl, size, region, dtype)
    C = logits_lr.shape[1]
    return {
        "target_seg_loss": w_t * s["ce"] / s["n_conf"],
        "kld_confident_loss": w_k * s["kld"] / (C * s["n_conf"]),
        "ent_ignored_loss": w_e * s["ent"] / (C * s["n_ign"]),
        "cst_loss": w_c * s["cst"] / s["cst_cnt"],
    }


def registry_loss(kind, logits, labels, weights=None, ignore_index=255, refer_labels=None, region="confident"):
    """LOSS[kind](logits, labels, weights, ignore_index, refer_labels, region) of the reference on the FULL argument
    surface (sseg/models/modules/losses.py:10-41 registry entries, :44-65 SoftCELoss, :68-72 compute_loss,
    :75-89 compute_loss_by_selected_pixel).  Pinned by tests/golden/loss_registry.npz.

    refer_labels is None -> the criterion's own mean (CE: over labels != ignore_index, weighted mean with `weights`;
    SoftCE: sum / numel; MSE / KLDIV: element mean).  Otherwise the per-element loss tensor is multiplied by the
    region mask [B,1,H,W] — for CE the per-pixel tensor is [B,H,W], so the product BROADCASTS to [B,B,H,W]
    (out[i,j] = loss[j] * mask[i], :86-87) — and the sum is divided by the number of non-zero elements (:89).
    SoftCE scales the target by the class weights first (:57-59; the reference does so IN PLACE)."""
    if kind == "CE":
        if refer_labels is None:
            return F.cross_entropy(logits, labels, weight=weights, ignore_index=ignore_index)
        t = F.cross_entropy(logits, labels, weight=weights, reduction="none")            # [B,H,W]
    elif kind == "SoftCE":
        q = labels if weights is None else labels * weights.view(1, -1, 1, 1).to(labels.dtype)
        t = -F.log_softmax(logits, dim=1) * q                                              # [B,C,H,W]
        if refer_labels is None:
            return t.sum() / q.numel()
    elif kind == "MSE":
        if refer_labels is None:
            return F.mse_loss(logits, labels)
        t = F.mse_loss(logits, labels, reduction="none")
    elif kind == "KLDIV":
        lp, q = F.log_softmax(logits, dim=1), F.softmax(labels, dim=1)
        if refer_labels is None:
            return F.kl_div(lp, q, reduction="mean")
        t = F.kl_div(lp, q, reduction="none")
    else:
        raise ValueError(kind)
    if region == "ignored":
        mask = refer_labels == ignore_index
    elif region == "confident":
        mask = refer_labels != ignore_index
    elif region == "all":
        mask = torch.ones_like(refer_labels, dtype=torch.bool)
    else:
        raise ValueError("{} is not a valid region".format(region))
    t = t * mask.unsqueeze(1)
    return t.sum() / (t != 0).sum()
