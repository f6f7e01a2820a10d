"""BaseTrainer (reference: workflows/trainer/base_trainer.py:15-197) on PyTorch-ROCm without apex:
one process per GPU, torch.distributed 'nccl' (= RCCL over xGMI) with the reference's
tcp://127.0.0.1:port rendezvous (or the torchrun environment when present), torch DDP with bucketed
gradient all-reduce overlapped with backward (the reference's apex DDP delays it to the end),
torch SyncBatchNorm, torch.autocast(bf16) standing in for amp O1."""
import os

import numpy as np
import torch
import torch.distributed as dist

from hiast_amd import functional as HF
from torch.nn.parallel import DistributedDataParallel as DDP
from torch.utils.data import DataLoader, DistributedSampler

from hiast_amd.sseg.datasets import utils as du
from hiast_amd.utils import comm, metrics, utils
from hiast_amd.utils.registry.registries import DATASET
from hiast_amd.utils.result_recorder import ResultRecorder


def autocast_dtype(cfg):
    """apex opt levels -> autocast: O0 = fp32; O1/O2/O3 = 16-bit convolutions with fp32 accumulation and fp32 master
    weights (utils/utils.py:126-132).  cfg.train.amp_dtype picks the 16-bit type, both on the hand-written channels-last
    kernels (HIAST_FMT_FP16 / HIAST_FMT_BF16): 'fp16' (default) = the reference's apex-O1 arithmetic with dynamic loss
    scaling; 'bf16' = 8 exponent bits, no loss scaling.  Measured against the float64 oracle (tests/
    test_gpu_trainstep_oracle.py, profiles/r03_trainstep_oracle_*): fp16 keeps the trunk gradients at cos 0.94-0.97 of
    the truth, bf16 at 0.62-0.78 — the reference's type is also the more faithful one; 1 % slower (the loss-scale check)."""
    if cfg.train.apex_opt == "O0":
        return None
    kind = getattr(cfg.train, "amp_dtype", "fp16")
    if kind not in ("bf16", "fp16"):
        raise ValueError("train.amp_dtype must be 'bf16' or 'fp16', got %r" % (kind,))
    return torch.bfloat16 if kind == "bf16" else torch.float16


def make_grad_scaler(amp_dtype):
    """apex's dynamic loss scaling for fp16 (amp.initialize(..., opt_level='O1'): initial scale 2^16, halved on an
    overflow — that step is skipped —, doubled after 2000 clean steps); bf16 and fp32 need none"""
    if amp_dtype is torch.float16:
        return torch.amp.GradScaler("cuda", init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000)
    return None


def _worker_init(_worker_id):
    """DataLoader workers yield the CPU to the main process: beside 10-14 busy workers on a 16-CPU share the main
    process's launch loop (31 ms of Python per iteration when it runs alone) took 64-88 ms and bound the end-to-end rate"""
    try:
        os.nice(10)
    except OSError:
        pass


class _EpochChain(torch.utils.data.Sampler):
    """Batch sampler over a DistributedSampler that goes on into the next epoch instead of ending: the batches are those of
    `set_epoch(e)` + a fresh DataLoader iterator per epoch (base_trainer.py:95-110 of the reference restarts its iterator on
    StopIteration), without the restart — the prefetch queues of the persistent workers run dry at every epoch boundary and
    the first batch of the new iterator takes a whole batch time of ONE worker (~0.25 s at 1024x512 with CopyPaste: 4.5 ms
    per iteration on a 448-image set, 1.5 % at Cityscapes size).  HIAST_EPOCH_RESTART=1 keeps the restart."""

    def __init__(self, sampler, batch_size, drop_last):
        self.sampler, self.batch_size, self.drop_last = sampler, batch_size, drop_last

    def __iter__(self):
        while True:
            n = 0
            for b in torch.utils.data.BatchSampler(self.sampler, self.batch_size, self.drop_last):
                n += 1
                yield b
            if n == 0:          # fewer samples than one batch: nothing to chain
                return
            self.sampler.set_epoch(self.sampler.epoch + 1)

    def __len__(self):          # batches per epoch
        return len(torch.utils.data.BatchSampler(self.sampler, self.batch_size, self.drop_last))


class _Bare(torch.nn.Module):
    """`.module` indirection for a single process, so trainers can write model.module.* like under DDP"""

    def __init__(self, m):
        super().__init__()
        self.module = m

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def state_dict(self, *a, **k):
        return self.module.state_dict(*a, **k)


class BaseTrainer:
    manual_allreduce = False     # True: the trainer all-reduces gradients itself after each backward (no DDP wrapper)
    # Weight gradients on a side stream are only safe when every weight receives exactly ONE gradient per backward:
    # with two uses of a weight autograd sums the two tensors on the MAIN stream while the side stream may still be
    # writing them.  Trainers that run the segmentation net more than once per step switch it off.
    wgrad_overlap = True

    def __init__(self, cfg, gpu_index):
        self.cfg = cfg
        self.gpu_index = gpu_index
        self.assert_cfg()
        self.initialize()
        self.build_all_model()
        self.build_train_data_reader()
        self.build_val_data_reader()

    def assert_cfg(self):
        pass

    # ---------------------------------------------------------------- setup
    def initialize(self):
        utils.seed_everything(self.cfg.train.random_seed)
        if torch.cuda.is_available():
            utils.limit_cpu_threads()
        self.world = self.cfg.train.gpu_num
        # the N > 1 code path: more than one rank — or HIAST_DIST_REHEARSAL=1, a ONE-rank process group that runs DDP, the SyncBN
        # exchanges and the validation all-reduce through the real backend (RCCL on a one-GPU box; utils/comm.py rehearsal())
        self.multi = self.world > 1 or comm.rehearsal()
        self.logger = None
        if self.gpu_index == 0:
            os.makedirs(self.cfg.work_dir, exist_ok=True)
            self.logger = utils.init_logger(os.path.join(self.cfg.work_dir, "train.log"))
            self.checkpoint_dir_path = os.path.join(self.cfg.work_dir, "checkpoints")
            os.makedirs(self.checkpoint_dir_path, exist_ok=True)
        use_cuda = torch.cuda.is_available()
        if self.multi and not dist.is_initialized():
            # (timeout=comm.timeout(): a collective one rank never joins raises after HIAST_DIST_TIMEOUT_S instead of hanging)
            if "MASTER_ADDR" in os.environ and "RANK" in os.environ:
                dist.init_process_group(backend="nccl" if use_cuda else "gloo", timeout=comm.timeout())
            else:
                dist.init_process_group(backend="nccl" if use_cuda else "gloo",
                                        init_method="tcp://127.0.0.1:{}".format(self.cfg.train.port),
                                        world_size=self.world, rank=self.gpu_index, timeout=comm.timeout())
        if self.multi:
            comm.setup()        # communicators of the SyncBN sums and of the small exchanges, beside DDP's (utils/comm.py)
        if use_cuda:
            # HIAST_SAME_DEVICE=1 (functional tests of the N>1 path on a one-GPU box): every rank uses cuda:0
            self.device_index = 0 if os.environ.get("HIAST_SAME_DEVICE", "0") == "1" else self.gpu_index
            torch.cuda.set_device(self.device_index)
            self.device = torch.device("cuda", self.device_index)
            comm.apply_cu_reserve(self.world, self.device)     # HIAST_RESERVE_CUS=n at N > 1: a CU for the collectives' kernels
        else:
            raise RuntimeError("training runs on the HIP device; no GPU is visible and there is no CPU fallback")

    def _wrap(self, model):
        if self.multi:
            return DDP(model, device_ids=[self.device_index], gradient_as_bucket_view=True, bucket_cap_mb=32,
                       broadcast_buffers=False)
        return _Bare(model)

    def build_all_model(self):
        print("%% initialize main model")
        model = utils.init_model(self.cfg, resume_from=self.cfg.train.resume_from).to(self.device)
        self.g_optimizer, self.d_optimizer = utils.init_optimizers(self.cfg, model)
        self.schedulers = utils.init_schedulers(self.cfg, self.g_optimizer, self.d_optimizer)
        self.amp_dtype = autocast_dtype(self.cfg)
        self.scaler = make_grad_scaler(self.amp_dtype)
        self.model = self._wrap(model)
        self.model_recorder = ResultRecorder(self.cfg, self.gpu_index, self.g_optimizer, self.d_optimizer, "model",
                                             self.logger)

    def _loader(self, ds, batch_size, shuffle, drop_last):
        # workers hand over uint8 images / labels; ToTensor + Normalize run on the device (same bits, 4-8x fewer bytes
        # through the worker pipes and PCIe).  HIAST_HOST_TRANSFORM=1 keeps the reference's float32 / int64 batches.
        ds.device_transform = os.environ.get("HIAST_HOST_TRANSFORM", "0") != "1"
        sampler = DistributedSampler(ds, num_replicas=self.world, rank=self.gpu_index, shuffle=shuffle)
        nw = self.cfg.dataset.num_workers
        kw = dict(num_workers=nw, pin_memory=True, persistent_workers=nw > 0, worker_init_fn=_worker_init if nw > 0 else None)
        if shuffle and drop_last and os.environ.get("HIAST_EPOCH_RESTART", "0") != "1":
            # training loaders: one endless iterator over the epochs (same batches, no restart at the boundaries)
            return sampler, DataLoader(ds, batch_sampler=_EpochChain(sampler, batch_size, drop_last), **kw)
        return sampler, DataLoader(ds, batch_size, sampler=sampler, drop_last=drop_last, **kw)

    def build_train_data_reader(self):
        s = self.cfg.dataset.source
        if s.type is not None and s.json_path is not None and s.image_dir is not None:
            self.s_dataset = DATASET[s.type](self.cfg, s.json_path, s.image_dir, aug_type=s.aug_type,
                                             num_classes=self.cfg.dataset.num_classes)
            self.s_sampler, self.s_loader = self._loader(self.s_dataset, self.cfg.train.batch_size, True, True)
            self.s_iter = iter(self.s_loader)
        t = self.cfg.dataset.target
        if t.type is not None and t.json_path and t.image_dir is not None:
            self.t_dataset = DATASET[t.type](self.cfg, t.json_path, t.image_dir, pseudo_dir=t.pseudo_dir,
                                             aug_type=t.aug_type, num_classes=self.cfg.dataset.num_classes)
            self.t_sampler, self.t_loader = self._loader(self.t_dataset, self.cfg.train.batch_size, True, True)
            self.t_iter = iter(self.t_loader)

    def build_val_data_reader(self):
        v = self.cfg.dataset.val
        self.v_loader = None
        if v.type is not None and v.json_path and v.image_dir is not None:
            ds = DATASET[v.type](self.cfg, v.json_path, v.image_dir, num_classes=self.cfg.dataset.num_classes)
            _, self.v_loader = self._loader(ds, self.cfg.train.batch_size, False, False)

    def next_target_batch(self):
        try:
            return next(self.t_iter)
        except StopIteration:
            self.t_sampler.set_epoch(self.t_sampler.epoch + 1)
            self.t_iter = iter(self.t_loader)
            return next(self.t_iter)

    def next_source_batch(self):
        try:
            return next(self.s_iter)
        except StopIteration:
            self.s_sampler.set_epoch(self.s_sampler.epoch + 1)
            self.s_iter = iter(self.s_loader)
            return next(self.s_iter)

    # ---------------------------------------------------------------- loop
    def run(self):
        if self.gpu_index == 0:
            self.logger.info("=" * 100)
            self.logger.info(self.cfg)
            self.logger.info("=" * 100)
        self.model_recorder.reset_time_and_losses()
        for current_iter in range(1, self.cfg.train.total_iter + 1):
            self.step(current_iter)
        self.model_recorder.report_end_info()

    def step(self, current_iter):
        losses = self.train()
        self.update_model(self.g_optimizer, self.d_optimizer, losses)
        self.after_update(current_iter)
        for s in self.schedulers:
            s.step()
        self.model_recorder.record_losses(losses)
        if current_iter % self.cfg.train.iter_report == 0:
            self.model_recorder.report_losses(current_iter)
        if current_iter % self.cfg.train.iter_val == 0 and self.v_loader is not None:
            self.validate_all(current_iter)

    def after_update(self, current_iter):
        pass

    def validate_all(self, current_iter):
        self.validate(self.model, self.model_recorder, current_iter)

    def _sync_grads(self, optimizer):
        if self.manual_allreduce and self.multi:
            utils.all_reduce_grads([p for g in optimizer.param_groups for p in g["params"]], self.world)

    def update_model(self, g_optimizer, d_optimizer, losses):
        """base_trainer.py:127-141: g_loss = Σ mean(loss_i) over the non-'D_' losses -> generator step; then, when
        there is a 'D_loss', the discriminator step.  fp16 (amp_dtype) goes through the dynamic loss scaler like
        apex's amp.scale_loss (:129-131); bf16 autocast needs no loss scaling."""
        scaler = getattr(self, "scaler", None)
        if getattr(losses, "backward_done", False):
            pass        # train() already ran forward AND backward (GraphedTrainStep: one captured HIP graph); the gradients
                        # are in place — and static: they must not be set to None
        else:
            g_loss = sum(torch.mean(v) for k, v in losses.items() if "D_" not in k)
            g_optimizer.zero_grad(set_to_none=True)
            HF.enable_wgrad_overlap(self.wgrad_overlap)
            try:
                (scaler.scale(g_loss) if scaler else g_loss).backward()
            finally:
                HF.enable_wgrad_overlap(False)
            HF.wgrad_stream_join()      # single-process runs issue the trunk's weight gradients on a side stream
        self._sync_grads(g_optimizer)
        if scaler:
            scaler.step(g_optimizer)        # unscales, skips the step on inf / NaN gradients
        else:
            g_optimizer.step()
        if "D_loss" in losses:
            d_optimizer.zero_grad(set_to_none=True)
            d_loss = torch.mean(losses["D_loss"])
            (scaler.scale(d_loss) if scaler else d_loss).backward()
            self._sync_grads(d_optimizer)
            if scaler:
                scaler.step(d_optimizer)
            else:
                d_optimizer.step()
        if scaler:
            scaler.update()

    def train(self):
        raise NotImplementedError

    # ---------------------------------------------------------------- validation / checkpoints
    def validate(self, model, recorder, current_iter, is_ema=False):
        iou, miou = self.get_validate_result(model)
        recorder.record_and_report_metrics(miou, iou, current_iter)
        if self.gpu_index == 0:
            if not is_ema:
                self.save_checkpoint(model, current_iter, recorder.model_name, miou == recorder.best_miou)
            else:
                torch.save(model.state_dict(), os.path.join(self.checkpoint_dir_path,
                                                            "{}_last.pth".format(recorder.model_name)))

    @torch.no_grad()
    def get_validate_result(self, model):
        """base_trainer.py:160-186: image -> val.resize_size, logits -> label size, argmax, I/U;
        the two bilinear resamplings of the logits are kept (low-res -> input size -> label size); the
        second one is fused with the argmax (pass-1 kernel) so label-size logits are never stored."""
        from hiast_amd import functional as HF, kernels as K
        net = model.module if hasattr(model, "module") else model
        was_training = net.training
        model.eval()                 # the WRAPPER, like the reference (base_trainer.py:161): wrapper and net stay in step
        C = self.cfg.dataset.num_classes
        acc = torch.zeros(2, C, dtype=torch.int64, device=self.device)
        for data in self.v_loader:
            img, lbl = du.to_device_batch(data["images"], data["labels"], self.device)
            size = self.cfg.dataset.val.resize_size or tuple(img.shape[2:])
            img = HF.upsample_bilinear_ac(img, size) if tuple(size) != tuple(img.shape[2:]) else img
            with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
                out = net(img, lowres=True)
            logits = HF.upsample_bilinear_ac(out["logits_lowres"].float(), size)
            _, pred, _ = K.plabel_pass1(logits.contiguous(), lbl.shape[1], lbl.shape[2])
            inter, union = metrics.intersection_union_counts(pred.long(), lbl.long(), C)
            acc[0] += inter
            acc[1] += union
        model.train(was_training)    # the reference calls model.train() at the top of every iteration
        if self.multi:
            comm.all_reduce(acc, "aux")       # one 38-element all-reduce instead of two
        acc = acc.cpu().numpy().astype(np.float64)
        iou = acc[0] / (acc[1] + 1e-10)
        return iou, float(np.mean(iou))

    def save_checkpoint(self, model, it, model_name, is_best=False):
        ckpt = model.module.state_dict()
        d = self.checkpoint_dir_path
        if self.cfg.train.is_save_all:
            torch.save(ckpt, os.path.join(d, "{}_iter_{}.pth".format(model_name, it)))
        torch.save(ckpt, os.path.join(d, "{}_last.pth".format(model_name)))
        if is_best:
            torch.save(ckpt, os.path.join(d, "{}_best.pth".format(model_name)))
        mid = os.path.join(d, "{}_mid.pth".format(model_name))
        if it >= self.cfg.train.total_iter // 2 and not os.path.exists(mid):
            torch.save(ckpt, mid)
