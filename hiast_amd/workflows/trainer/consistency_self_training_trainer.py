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


class StepLosses(dict):
    """the losses of an iteration whose backward pass has ALREADY run (GraphedTrainStep): `update_model` reads the marker from
    the object it is handed, so a `train()` without a following `update_model()` (tools, tests, an exception in between) leaves
    nothing behind that a later, unrelated `update_model()` could mistake for its own backward pass."""
    backward_done = True


class GraphedTrainStep:
    """train_on() + the backward pass of one HIAST iteration as ONE captured HIP graph (round 5).

    Why: an iteration is ~1000 kernel launches; through the DataLoader the trainer's host loop (36-40 ms) sat within 10 % of its
    device time (39.5 ms, profiles/r04_trainer_end_to_end.txt) — every further kernel gain would have been invisible.  What is
    captured: EMA-teacher forward (side stream), student forward, fused 4-term loss, backward incl. the grouped weight gradients
    on their side stream and the weight re-packing of both trunks (the launches of reference steps
    workflows/trainer/consistency_self_training_trainer.py:92-126 + the backward of base_trainer.py:127-131).  What stays
    eager: the batch's H2D copy + normalisation, the loss scaler's non-finite check, FusedAdam, the EMA update, the scheduler
    (the learning rate is a launch ARGUMENT of the optimiser kernel), validation, checkpoints.  Static shapes: a change of the
    batch's shape re-captures.  The first WARM iterations of a shape run eagerly (allocator, lazy initialisations).
    Gradients are bit-equal to the eager step's (tests/test_gpu_round5.py): same kernels, same order, same streams.
    Single process only — under DDP the reducer's hooks have to run, the step stays eager.
    Memory: the graph's private pool holds one iteration's activations for the life of the trainer, beside the blocks the WARM
    eager iterations left in the caching allocator (≈ twice the activation memory of an eager run: 2 x 9 GB at batch 8 of the 288 GB;
    HIAST_GRAPH_TRAIN=0 gives it back)."""
    WARM = 3

    def __init__(self, trainer):
        self.tr = trainer
        self.graph = None
        self.key = None
        self.seen = 0

    def _plans(self):
        nets = [self.tr.model.module, self.tr.ema_model]
        for net in nets:
            for m in net.modules():
                for ent in m.__dict__.get("_hiast_plans", {}).values():
                    yield ent[0]

    def _run(self, weak, strong, plbl):
        tr = self.tr
        losses = tr.train_on(weak, strong, plbl)
        g_loss = sum(torch.mean(v) for k, v in losses.items() if "D_" not in k)
        HF.enable_wgrad_overlap(tr.wgrad_overlap)
        try:
            (tr.scaler.scale(g_loss) if tr.scaler else g_loss).backward()
        finally:
            HF.enable_wgrad_overlap(False)
        HF.wgrad_stream_join()
        return losses

    def __call__(self, weak, strong, plbl):
        tr = self.tr
        key = (tuple(weak.shape), weak.dtype, tuple(strong.shape), tuple(plbl.shape), plbl.dtype, strong is weak)
        if key != self.key:
            self.key, self.graph, self.seen = key, None, 0
        if self.graph is None:
            self.seen += 1
            if self.seen <= self.WARM:                        # eager iterations of this shape
                tr.g_optimizer.zero_grad(set_to_none=True)
                return StepLosses(self._run(weak, strong, plbl))
            self.s_weak = weak.clone()
            self.s_strong = self.s_weak if strong is weak else strong.clone()
            self.s_plbl = plbl.clone()
            for plan in self._plans():
                plan.versions = None              # the re-pack launches must be IN the graph whatever moved last
            tr.g_optimizer.zero_grad(set_to_none=True)        # the graph's backward allocates the (from now on static) gradients
            torch.cuda.synchronize()
            # (no torch.cuda.empty_cache() here: round 6 tried it — hand the warm-up's blocks back before the graph allocates its
            # private pool — and the full GPU suite then died with a segmentation fault inside hipGraphLaunch at the replay of the
            # SECOND trainer of one process, gpurun_out r06_pytest_full.log; without it the suite has run clean since round 5)
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    losses = self._run(self.s_weak, self.s_strong, self.s_plbl)
            except Exception as e:      # a capture that cannot be made (an op that synchronises, a foreign stream): say so ONCE
                import warnings         # and keep training — the same launches, issued eagerly from here on
                warnings.warn("GraphedTrainStep: capture failed (%r); the iterations stay eager" % (e,))
                tr._graph_train = False
                torch.cuda.synchronize()
                # the failed capture RECORDED the re-pack launches without running them, yet marked every plan as up to date
                # (PackPlan.refresh) and published its buffers: the eager rerun must pack again, or this iteration would read the
                # packed weights from before the last optimiser / EMA step
                for plan in self._plans():
                    plan.versions = None
                    plan.published = False
                tr.g_optimizer.zero_grad(set_to_none=True)
                return StepLosses(self._run(weak, strong, plbl))
            self.graph = g
            self.losses = StepLosses((k, v.detach()) for k, v in losses.items())
        else:
            self.s_weak.copy_(weak, non_blocking=True)
            if self.s_strong is not self.s_weak:
                self.s_strong.copy_(strong, non_blocking=True)
            self.s_plbl.copy_(plbl, non_blocking=True)
        self.graph.replay()
        return self.losses


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
            self._side_stream = HF.new_stream(t_weak_img.device)
        side = self._side_stream
        side.wait_stream(main)
        if getattr(self, "_teacher_fwd", None) is None or self._teacher_fwd.model is not self.ema_model:
            # no-grad eval forward under autocast(amp_dtype) (HIAST_GRAPH_EVAL=1: replayed from a captured HIP graph)
            # (small batches are replayed automatically — unless the whole iteration is a captured graph already)
            self._teacher_fwd = HF.GraphedEval(self.ema_model, self.amp_dtype, parts=1, auto=not self.graph_train_enabled())
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
        if self.graph_train_enabled():
            # forward + backward from a captured HIP graph (GraphedTrainStep): update_model() finds the gradients in place
            if getattr(self, "_graphed_step", None) is None:
                self._graphed_step = GraphedTrainStep(self)
            return self._graphed_step(weak, strong, plbl)      # (StepLosses: carries the "backward has run" marker)
        return self.train_on(weak, strong, plbl)

    def graph_train_enabled(self):
        """on by default for 16-bit single-process training (HIAST_GRAPH_TRAIN=0, read once, keeps every iteration eager):
        through the DataLoader 45.0 -> 39.7 ms/iter = 177.7 -> 201.3 images/s on one box, the rate of a device-resident batch
        (profiles/r05_trainer_end_to_end.txt)"""
        on = self.__dict__.get("_graph_train")
        if on is None:
            # (HIAST_GRAPH_EVAL=1 replays the teacher forward from a graph of its own: a replay cannot be captured into another
            # graph, the iteration then stays eager)
            on = self._graph_train = (os.environ.get("HIAST_GRAPH_TRAIN", "1") == "1"
                                      and os.environ.get("HIAST_GRAPH_EVAL", "0") != "1" and not self.multi
                                      and self.amp_dtype is not None and not getattr(self, "manual_allreduce", False))
        return on
