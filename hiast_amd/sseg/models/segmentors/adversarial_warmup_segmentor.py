"""MODEL['AdversarialWarmupSegmentor'] (reference: sseg/models/segmentors/adversarial_warmup_segmentor.py:11-86):
source CE + adversarial loss on the target prediction + discriminator loss (+ MinEnt target entropy), the warm-up
stage that produces the checkpoint self-training starts from.

Same surface (`.seg_model`, `.D`, `forward(s_img, t_img=None, s_lbl=None)` -> loss dict in train mode /
{'logits'} in eval mode, loss names `source_seg_loss`, `adv_loss`, `D_loss`, `target_ent_loss`).  MI355X layout of
the step: both segmentation forwards stay LOW-RES; the source CE and the target entropy come out of the fused loss
kernel (K5-K8); the discriminator input (upsample -> softmax [-> self-information]) is one kernel (K15) whose
output is shared by the adversarial pass and the discriminator pass on the target (the reference recomputes it);
the adversarial pass runs the discriminator with detached weights so no weight gradient is formed only to be
zeroed (base_trainer.py:136 zeroes them before the D step)."""
import math

import torch
from torch import nn

from hiast_amd import functional as HF
from hiast_amd.sseg.models.modules.discriminator import build_discriminator
from hiast_amd.sseg.models.modules.seg_models import build_seg_model
from hiast_amd.sseg.models.segmentors.self_training_segmentor import upsample_logits
from hiast_amd.utils.registry.registries import LOSS, MODEL


@MODEL.register("AdversarialWarmupSegmentor")
class AdversarialWarmupSegmentor(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.seg_model = build_seg_model(cfg)
        assert cfg.model.discriminator.is_enabled
        self.D = build_discriminator(cfg.dataset.num_classes)
        if cfg.model.predictor.seg_loss.type != "CE":
            raise NotImplementedError("the fused loss implements seg_loss.type == 'CE'")
        self.D_loss_fun = LOSS[cfg.model.discriminator.D_loss.type]
        self._all_ignored = None

    def _ignored_map(self, B, size, device):
        m = self._all_ignored
        if m is None or tuple(m.shape) != (B,) + tuple(size) or m.device != device:
            m = self._all_ignored = torch.full((B,) + tuple(size), 255, dtype=torch.uint8, device=device)
        return m

    def forward(self, s_img, t_img=None, s_lbl=None, lowres=False):
        s_lr, _ = self.seg_model(s_img, need_feat=False)
        size = tuple(s_img.shape[2:])
        if not self.training:
            if lowres:
                return {"logits_lowres": s_lr, "backbone": None, "size": size}
            return {"logits": upsample_logits(s_lr, size)}

        cfg = self.cfg
        C = cfg.dataset.num_classes
        t_lr, _ = self.seg_model(t_img, need_feat=False)
        t_size = tuple(t_img.shape[2:])
        entropy_in = cfg.model.discriminator.is_entropy_input
        d_cfg = cfg.model.discriminator.D_loss
        losses = {}

        # source segmentation loss (:45): CE on the (virtually) upsampled source logits
        ce, _, _, _ = HF.st_loss(s_lr, None, s_lbl, size, "ignored", cfg.model.predictor.seg_loss.source_weight, 0.0,
                                 0.0, 0.0)
        losses["source_seg_loss"] = ce

        # adversarial loss (:48-50): target map labelled as source; gradient flows to the segmentation net only
        x_t = HF.discriminator_input(t_lr, t_size, entropy_in)
        frozen = {k: v.detach() for k, v in self.D.named_parameters()}
        t_adv = self.D(x_t, frozen)
        losses["adv_loss"] = d_cfg.adv_weight * self.D_loss_fun(t_adv, torch.zeros_like(t_adv))

        # discriminator loss (:54-59) on detached maps
        with torch.no_grad():
            x_s = HF.discriminator_input(s_lr.detach(), size, entropy_in)
        s_d = self.D(x_s)
        t_d = self.D(x_t.detach())
        losses["D_loss"] = d_cfg.weight * (self.D_loss_fun(s_d, torch.zeros_like(s_d)) +
                                           self.D_loss_fun(t_d, torch.ones_like(t_d))) / 2

        # MinEnt (:62-63, :79-86): -Σ p log2(p + 1e-30) / (N log2 C) over every target pixel
        w_ent = cfg.model.predictor.ent_loss.weight
        if w_ent > 0:
            ign = self._ignored_map(t_lr.shape[0], t_size, t_lr.device)
            _, _, ent, _ = HF.st_loss(t_lr, None, ign, t_size, "ignored", 0.0, 0.0, w_ent * C / math.log(C), 0.0)
            losses["target_ent_loss"] = ent
        return losses
