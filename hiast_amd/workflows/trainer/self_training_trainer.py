"""TRAINER['SelfTrainingTrainer'] (reference: workflows/trainer/self_training_trainer.py:7-28):
one target batch with pseudo labels -> student forward -> 3-term loss.  The student hands over
low-res logits and the fused loss kernel does the rest."""
import torch

from hiast_amd.sseg.datasets import utils as du
from hiast_amd.utils import utils
from hiast_amd.utils.registry.registries import TRAINER
from hiast_amd.workflows.trainer.base_trainer import BaseTrainer


@TRAINER.register("SelfTrainingTrainer")
class SelfTrainingTrainer(BaseTrainer):

    def assert_cfg(self):
        assert self.cfg.dataset.target.pseudo_dir is not None, \
            "directory of pseudo labels should be given for self training"
        assert self.cfg.train.resume_from is not None, "self-training should resume_from one state_dict"

    def train_on(self, t_img, t_plbl):
        utils.set_mode(self.model, True)
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            out = self.model(t_img, lowres=True)
        return self.model.module.compute_loss_lowres(out["logits_lowres"], t_plbl, out["size"])

    def train(self):
        t = self.next_target_batch()
        return self.train_on(*du.to_device_batch(t["images"], t["labels"], self.device))
