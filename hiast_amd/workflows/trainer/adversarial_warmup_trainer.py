"""TRAINER['AdversarialWarmupTrainer'] (reference: workflows/trainer/adversarial_warmup_trainer.py:7-37): one source
batch with labels + one target batch per iteration through MODEL['AdversarialWarmupSegmentor']; generator step, then
discriminator step (base_trainer.py:127-141).

Two backward passes per forward do not fit torch DDP's one-reduction-per-iteration reducer, so on more than one GPU
this trainer keeps the bare module and all-reduces the gradients itself after each backward (flat buckets over
RCCL — the reference's apex DDP likewise delays its all-reduce to the end of backward)."""
import torch

from hiast_amd.sseg.datasets import utils as du
from hiast_amd.utils import utils
from hiast_amd.utils.registry.registries import TRAINER
from hiast_amd.workflows.trainer.base_trainer import BaseTrainer, _Bare


@TRAINER.register("AdversarialWarmupTrainer")
class AdversarialWarmupTrainer(BaseTrainer):
    manual_allreduce = True
    wgrad_overlap = False        # seg_model runs twice per step (source, target): two gradients per trunk weight

    def assert_cfg(self):
        assert self.cfg.model.discriminator.is_enabled, \
            "discriminator should be enabled for adversarial warmup training"
        assert self.cfg.train.resume_from is not None, "adversarial warmup training should resume_from one state_dict"

    def _wrap(self, model):
        return _Bare(model)

    def train_on(self, s_img, s_lbl, t_img):
        utils.set_mode(self.model, True)
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            return self.model(s_img, t_img, s_lbl)

    def train(self):
        s = self.next_source_batch()
        t = self.next_target_batch()
        s_img, s_lbl = du.to_device_batch(s["images"], s["labels"], self.device)
        t_img, _ = du.to_device_batch(t["images"], t["labels"], self.device)
        return self.train_on(s_img, s_lbl, t_img)
