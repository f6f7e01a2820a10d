"""TRAINER['ConsistencySelfTrainingTrainer'] — the HIAST setting
(reference: workflows/trainer/consistency_self_training_trainer.py:16-126): CopyPaste-augmented target
batches, EMA teacher on the weak view (no grad, eval), student on the strong view, 4-term loss, EMA
update every `iter_update` iterations.  Teacher and student both hand over LOW-RES logits; the teacher's
softmax happens inside the fused loss kernel."""
import os

import numpy as np
import torch

from hiast_amd import functional as HF
from hiast_amd.sseg.datasets import utils as du
from hiast_amd.sseg.datasets.preprocessor import CopyPaste
from hiast_amd.utils import utils
from hiast_amd.utils.registry.registries import DATASET, TRAINER
from hiast_amd.utils.result_recorder import ResultRecorder
from hiast_amd.workflows.trainer.base_trainer import BaseTrainer


@TRAINER.register("ConsistencySelfTrainingTrainer")
class ConsistencySelfTrainingTrainer(BaseTrainer):

    def assert_cfg(self):
        cfg = self.cfg
        assert cfg.dataset.target.pseudo_dir is not None, \
            "directory of pseudo labels should be given for self training"
        assert cfg.cst_training.is_enabled, "consistency training should be enabled"
        assert len(cfg.dataset.target.aug_type) in (1, 2), \
            "target domain dataset should have 1 or 2 augmentations for consistency training"
        assert cfg.preprocessor.type == "CopyPaste"

    def build_train_data_reader(self):
        t = self.cfg.dataset.target
        self.class_value = np.load(os.path.join(t.pseudo_dir, "..", "class_mean_probabilities.npy"))
        self.t_dataset = DATASET[t.type](self.cfg, t.json_path, t.image_dir, pseudo_dir=t.pseudo_dir,
                                         aug_type=t.aug_type, num_classes=self.cfg.dataset.num_classes)
        self.preprocessor = CopyPaste(self.cfg, self.t_dataset, self.class_value)
        self.t_dataset.set_preprocessor(self.preprocessor)
        self.t_sampler, self.t_loader = self._loader(self.t_dataset, self.cfg.train.batch_size, True, True)
        self.t_iter = iter(self.t_loader)

    def build_all_model(self):
        super().build_all_model()
        self.ema_model = utils.init_model(self.cfg, student_model=self.model).to(self.device)
        for p in self.ema_model.parameters():
            p.requires_grad = False
        self.ema_updater = utils.EmaUpdater()
        self.ema_model_recorder = ResultRecorder(self.cfg, self.gpu_index, None, None, "ema_model", self.logger)

    def after_update(self, current_iter):
        ema = self.cfg.cst_training.ema_model
        if current_iter % ema.iter_update == 0:
            self.ema_updater(self.ema_model, self.model, ema.gamma)

    def validate_all(self, current_iter):
        self.validate(self.model, self.model_recorder, current_iter)
        self.validate(self.ema_model, self.ema_model_recorder, current_iter, True)

    def train_on(self, t_weak_img, t_strong_img, t_plbl):
        if self.cfg.cst_training.cst_loss.type != "SoftCE":
            raise NotImplementedError("cst_loss.type %r" % self.cfg.cst_training.cst_loss.type)
        utils.set_mode(self.ema_model, False)
        # teacher forward on a side stream: its MFMA-bound convolutions co-run with the HBM-bound BatchNorm passes of
        # the student forward; the loss waits for both
        main = torch.cuda.current_stream()
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream(device=t_weak_img.device)
        side = self._side_stream
        side.wait_stream(main)
        if getattr(self, "_teacher_fwd", None) is None or self._teacher_fwd.model is not self.ema_model:
            # no-grad eval forward under autocast(amp_dtype) (HIAST_GRAPH_EVAL=1: replayed from a captured HIP graph)
            self._teacher_fwd = HF.GraphedEval(self.ema_model, self.amp_dtype, parts=1)
        with torch.cuda.stream(side):
            teacher_lr = self._teacher_fwd(t_weak_img)
        utils.set_mode(self.model, True)
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            out = self.model(t_strong_img, lowres=True)
        main.wait_stream(side)
        teacher_lr.record_stream(main)
        return self.model.module.compute_loss_lowres(out["logits_lowres"], t_plbl, out["size"], teacher_lr)

    def train(self):
        t = self.next_target_batch()
        img, plbl = t["images"], t["labels"]
        if isinstance(img, (list, tuple)):
            assert len(img) == 2 and torch.equal(plbl[0], plbl[1])
            weak, strong, plbl = img[0], img[1], plbl[0]
        else:
            weak = strong = img
        if strong is weak:
            weak, plbl = du.to_device_batch(weak, plbl, self.device)
            strong = weak
        else:
            (weak, strong), plbl = du.to_device_batch([weak, strong], plbl, self.device)
        return self.train_on(weak, strong, plbl)
