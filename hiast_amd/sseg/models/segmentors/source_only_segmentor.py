"""MODEL['SourceOnlySegmentor'] (reference: source_only_segmentor.py:9-24).  configs/validate.yaml
selects it, so its eval branch is part of the drop-in surface; its train branch is source-only CE."""
from torch import nn

from hiast_amd.sseg.models.modules.seg_models import build_seg_model
from hiast_amd.sseg.models.segmentors.self_training_segmentor import upsample_logits
from hiast_amd.utils.registry.registries import LOSS, MODEL


@MODEL.register("SourceOnlySegmentor")
class SourceOnlySegmentor(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.seg_model = build_seg_model(cfg)
        self.seg_loss_fun = LOSS[cfg.model.predictor.seg_loss.type]

    def forward(self, img, lbl=None, lowres=False):
        logits, feat = self.seg_model(img, need_feat=not lowres)
        if lowres:
            return {"logits_lowres": logits, "backbone": feat, "size": tuple(img.shape[2:])}
        logits = upsample_logits(logits, img.shape[2:])
        if self.training:
            return {"seg_loss": self.cfg.model.predictor.seg_loss.source_weight * self.seg_loss_fun(logits, lbl)}
        return {"logits": logits}
