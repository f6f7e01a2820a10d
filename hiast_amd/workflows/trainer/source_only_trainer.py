"""TRAINER['SourceOnlyTrainer'] (reference: workflows/trainer/source_only_trainer.py:7-24): supervised CE on the
source domain.  The model hands over low-res logits; the fused loss kernel does upsample + CE."""
import torch

from hiast_amd import functional as HF
from hiast_amd.sseg.datasets import utils as du
from hiast_amd.utils import utils
from hiast_amd.utils.registry.registries import TRAINER
from hiast_amd.workflows.trainer.base_trainer import BaseTrainer


@TRAINER.register("SourceOnlyTrainer")
class SourceOnlyTrainer(BaseTrainer):

    def train_on(self, s_img, s_lbl):
        utils.set_mode(self.model, True)
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            out = self.model(s_img, lowres=True)
        w = self.cfg.model.predictor.seg_loss.source_weight
        ce, _, _, _ = HF.st_loss(out["logits_lowres"], None, s_lbl, out["size"], "ignored", w, 0.0, 0.0, 0.0)
        return {"seg_loss": ce}

    def train(self):
        s = self.next_source_batch()
        return self.train_on(*du.to_device_batch(s["images"], s["labels"], self.device))
