"""Model / optimiser / distributed helpers (reference: utils/utils.py:19-170), without apex:
torch SyncBatchNorm + DistributedDataParallel over RCCL, torch.autocast instead of amp O1, and the
EMA teacher update as one multi-tensor HIP launch."""
import logging
import os
import random
import socket
import warnings

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

from hiast_amd.sseg.models.modules.schedulers import build_scheduler
from hiast_amd.utils.registry.registries import MODEL


def seed_everything(seed=888):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def limit_cpu_threads(default=4):
    """Cap torch's intra-op (OpenMP) pool in a process whose arithmetic runs on the device.  torch sizes the pool by the
    machine's logical CPUs (256 on an MI355X host) whatever the process may use; after every small CPU op those threads
    spin for a while, and beside the DataLoader workers they took the main thread's launch loop from 27 to 50-60 ms per
    iteration (end-to-end trainer 131 -> 154 images/s with the cap, `profiles/r02_trainer_end_to_end.txt`) — with eight
    ranks per node it is eight such pools.  HIAST_CPU_THREADS overrides the cap (0 = leave torch's default)."""
    n = int(os.environ.get("HIAST_CPU_THREADS", default))
    if n > 0 and torch.get_num_threads() > n:
        torch.set_num_threads(n)


def create_dir(path):
    if os.path.exists(path):
        warnings.warn("%s has existed" % path)
    else:
        os.makedirs(path)


def is_port_used(port, host="127.0.0.1"):
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.settimeout(1)
        return s.connect_ex((host, int(port))) == 0


def get_device(index=None):
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device() if index is None else index)
    return torch.device("cpu")


def freeze_bn(model):
    """requires_grad=False on every norm layer's affine parameters (utils.py:60-65); the layers
    still normalise with batch statistics in train() mode, exactly like the reference."""
    for m in model.modules():
        if isinstance(m, (nn.BatchNorm2d, nn.GroupNorm, nn.SyncBatchNorm)):
            for p in m.parameters():
                p.requires_grad = False


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def load_model(cfg, resume_from=None, student_model=None):
    """utils.py:68-89: build MODEL[cfg.model.type]; copy a (DDP-wrapped) student, or partially load a
    checkpoint (keys saved under DDP lose their 7-char 'module.' prefix)."""
    assert not (resume_from is not None and student_model is not None)
    model = MODEL[cfg.model.type](cfg)
    if student_model is not None:
        model.load_state_dict(_unwrap(student_model).state_dict())
        print("%% load model from student model")
    elif resume_from is not None:
        own = model.state_dict()
        saved = torch.load(resume_from, map_location="cpu")
        strip = len("module.") if "module" in next(iter(saved.keys())) else 0
        own.update({k[strip:]: v for k, v in saved.items() if k[strip:] in own})
        model.load_state_dict(own)
        print("%% load model from {}".format(resume_from))
    else:
        warnings.warn("not load model")
    return model


def init_model(cfg, resume_from=None, student_model=None):
    """utils.py:92-112 — SyncBN when more than one GPU takes part, then the BN freeze."""
    model = load_model(cfg, resume_from=resume_from, student_model=student_model)
    from hiast_amd.utils import comm
    if cfg.train.gpu_num > 1 or comm.rehearsal():      # (rehearsal: the one-rank run of the N > 1 path, utils/comm.py)
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
        print("%% convert BN to SyncBN")
    if cfg.model.is_freeze_bn:
        freeze_bn(model)
        print("%% freeze all BN layers")
    return model


def set_mode(module, training):
    """module.train(training) unless the module is already in that mode: train() / eval() walk every sub-module
    (~900 per network: 1.7 ms per call), and the trainers call them once per iteration"""
    training = bool(training)
    inner = module.module if hasattr(module, "module") else module     # DDP / _Bare wrapper: the net is what counts
    if module.training != training or inner.training != training:
        module.train(training)
    return module


class EmaUpdater:
    """update_ema_model (utils.py:115-123): parameters ema = ema*g + p*(1-g) in one HIP launch,
    buffers copied (one foreach copy)."""

    def __init__(self):
        self._plan = None
        self._bplan = None

    def _lists(self, ema_model, src):
        """parameter / buffer lists of the two networks, cached per pair: walking ~900 modules four times per update was
        7.5 ms of host time per training step (module.parameters() is a generator over named_modules())"""
        key = (id(ema_model), id(src))
        if getattr(self, "_key", None) != key:
            self._key = key
            self._ema_params = list(ema_model.parameters())
            self._src_params = list(src.parameters())
            self._ema_bufs = list(ema_model.buffers())
            self._src_bufs = list(src.buffers())
            self._plan = self._bplan = None
        return self._ema_params, self._src_params, self._ema_bufs, self._src_bufs

    def __call__(self, ema_model, model, gamma):
        from hiast_amd import kernels as K
        src = _unwrap(model)
        eparams, sparams, eb, sb = self._lists(ema_model, src)
        if not eparams[0].is_cuda:
            raise RuntimeError("EMA update runs on the HIP device only")
        if self._plan is None or not self._plan.matches(eparams, sparams):
            self._plan = K.EmaPlan([p.data for p in eparams], [p.data for p in sparams])
        K.ema_update(self._plan, gamma)
        # the kernel writes through raw pointers: tell autograd (and every cache keyed on Parameter._version — the
        # packed bf16 / split-plane trunk weights of ResNet.prepack) that the parameters have changed
        torch.autograd.graph.increment_version(eparams)
        if eb:      # ~300 BatchNorm buffers: one launch instead of one copy kernel per tensor
            if self._bplan is None or not self._bplan.matches(eb, sb):
                self._bplan = K.CopyPlan(eb, sb)
            K.multi_copy(self._bplan)
        return ema_model


_default_ema = EmaUpdater()


def update_ema_model(ema_model, model, gamma):
    return _default_ema(ema_model, model, gamma)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam (L2 weight decay, no amsgrad) whose step is ONE HIP launch over every parameter
    (hiast_adam_step) instead of a per-tensor / foreach sequence of elementwise kernels.  Same param_groups /
    state layout ('step', 'exp_avg', 'exp_avg_sq'), so LR schedulers and checkpoints are interchangeable.

    Dynamic loss scaling (fp16, torch.amp.GradScaler = apex's amp.scale_loss of base_trainer.py:129-131) is handled ON
    THE DEVICE (`_step_supports_amp_scaling`): scaler.step(opt) hands over its device scalars `grad_scale` / `found_inf`;
    the kernel unscales the gradients and skips the whole update — moments and step count included — on an overflow.
    The stock path (`scaler.unscale_` + `if not found_inf.item(): opt.step()`) waits for the device once per iteration."""
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._plan = None
        self._ctl = None        # device control block (element 0: number of steps SKIPPED on an overflow so far)
        self._attempts = {}     # id(param) -> steps attempted since the counts were last folded into state['step']
        self._part = None       # ids of the parameters that took part in the previous step

    def _skipped(self):
        return 0 if self._ctl is None else int(self._ctl[0].item())       # (synchronises)

    def _fold_steps(self):
        """state['step'] := steps actually applied (attempts minus the overflow steps the device skipped); the device
        counter restarts at 0.  Synchronises: called when the counts are read or saved, never inside a step."""
        if self._ctl is None:
            return
        k = float(self._skipped())
        for p, st in self.state.items():
            if "step" in st and self._attempts.get(id(p)):
                st["step"] = torch.tensor(float(st["step"]) - k)
        self._attempts = {}
        self._ctl.zero_()

    def applied_steps(self):
        """number of updates actually applied to the most-stepped parameter (overflow steps excluded); synchronises"""
        self._fold_steps()
        return max([int(st["step"]) for st in self.state.values() if "step" in st], default=0)

    def state_dict(self):
        self._fold_steps()      # torch's layout: per-parameter 'step' = applied steps
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._attempts = {}
        if self._ctl is not None:
            self._ctl.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        from hiast_amd import kernels as K
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grad_scale, found_inf = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
        by_hyper = {}
        for group in self.param_groups:
            key = (group["betas"], group["eps"], group["weight_decay"])
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    if self._attempts:      # a tensor joins later: its count must not inherit earlier skipped steps
                        self._fold_steps()
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                by_hyper.setdefault(key, []).append((p, group["lr"], st))
        dev = next((p.device for items in by_hyper.values() for p, _, _ in items), None)
        if dev is None:
            return loss
        # The device counts skipped (overflow) steps ONCE for all tensors, so between two folds every counted tensor must
        # have taken part in every step: when the set of participating parameters changes (a tensor sits out a step, or
        # comes back), the counts are folded first (one synchronisation, only on such a change).
        part = frozenset(id(p) for items in by_hyper.values() for p, _, _ in items)
        if self._part is not None and part != self._part and self._attempts:
            self._fold_steps()
        self._part = part
        if self._ctl is None:
            self._ctl = torch.zeros(8, dtype=torch.float32, device=dev)
        K.adam_prepare(self._ctl, grad_scale, found_inf)      # ONE decision per step, shared by every launch below
        for key, items in by_hyper.items():
            betas, eps, wd = key
            ps = [p for p, _, _ in items]
            for p, _, st in items:      # host side: ATTEMPTED steps; the device subtracts the ones it skipped (overflow)
                st["step"] += 1
                self._attempts[id(p)] = self._attempts.get(id(p), 0) + 1
            plan = self._plan.get(len(ps)) if isinstance(self._plan, dict) else None
            numels = tuple(p.numel() for p in ps)
            if plan is None or plan.numels != numels:
                plan = K.AdamPlan(numels, ps[0].device)
                self._plan = dict(self._plan or {})
                self._plan[len(ps)] = plan
            one = [1.0] * len(ps)       # bias corrections: formed on the device from (attempted - skipped) steps
            K.adam_step(plan, [p.data for p in ps], [p.grad.contiguous() for p in ps], [st["exp_avg"] for _, _, st in items],
                        [st["exp_avg_sq"] for _, _, st in items], [lr for _, lr, _ in items], one, one, betas[0], betas[1],
                        eps, wd, ctl=self._ctl, steps=[float(st["step"]) for _, _, st in items])
            # raw-pointer writes do not move Parameter._version by themselves; the packed-weight caches key on it
            torch.autograd.graph.increment_version(ps)
        return loss


def init_optimizers(cfg, model):
    """utils.py:135-154: generator optimiser over the segmentation net's three LR groups; Adam(lr=discriminator.lr)
    over model.D when the discriminator is enabled (adversarial warm-up)."""
    groups = _unwrap(model).seg_model.get_optimizer_params(cfg.train.lr)
    groups = [{"params": [p for p in g["params"] if p.requires_grad], "lr": g["lr"]} for g in groups]
    kind = cfg.train.optimizer
    if kind == "SGD":
        opt = torch.optim.SGD(groups, momentum=0.9, weight_decay=0.0005)
    elif kind == "Adam":
        on_device = all(p.is_cuda for g in groups for p in g["params"])
        opt = (FusedAdam if on_device else torch.optim.Adam)(groups, betas=(0.9, 0.999), weight_decay=0.0005)
    elif kind == "AdamW":
        opt = torch.optim.AdamW(groups, betas=(0.9, 0.999), weight_decay=0.0005)
    else:
        raise ValueError("%s is not a valid optimizer" % kind)
    d_opt = None
    if cfg.model.discriminator.is_enabled:
        d_params = list(_unwrap(model).D.parameters())
        on_device = all(p.is_cuda for p in d_params)
        d_opt = (FusedAdam if on_device else torch.optim.Adam)(d_params, lr=cfg.model.discriminator.lr, betas=(0.9, 0.999))
    return opt, d_opt


def init_schedulers(cfg, g_optimizer, d_optimizer=None):
    return [build_scheduler(cfg, o) for o in (g_optimizer, d_optimizer) if o is not None]


def all_reduce_grads(params, world_size, bucket_bytes=64 << 20):
    """average .grad over the ranks in flat buckets (one RCCL all-reduce per ~64 MB)"""
    from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors
    grads = [p.grad for p in params if p.grad is not None]
    bucket, size = [], 0

    def flush():
        if not bucket:
            return
        flat = _flatten_dense_tensors(bucket)
        dist.all_reduce(flat)
        flat.div_(world_size)
        for g, f in zip(bucket, _unflatten_dense_tensors(flat, bucket)):
            g.copy_(f)
        bucket.clear()

    for g in grads:
        bucket.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            flush()
            size = 0
    flush()


def all_reduce_average(tensor, world_size):
    if dist.is_available() and dist.is_initialized() and world_size > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
        tensor /= world_size
    return tensor


def init_logger(log_path):
    """[time-level] lines to <work_dir>/train.log (append) and to stderr, like the reference's
    init_logger_and_writer (utils.py:173-183) minus tensorboardX; handlers are attached to the named
    logger (not through logging.basicConfig, which only takes effect once per process)."""
    logger = logging.getLogger("UDA.trainer")
    logger.setLevel(logging.INFO)
    for h in list(logger.handlers):
        logger.removeHandler(h)
        h.close()
    fmt = logging.Formatter("[%(asctime)s-%(levelname)s]: %(message)s")
    fh = logging.FileHandler(log_path, mode="a")
    fh.setFormatter(fmt)
    sh = logging.StreamHandler()
    sh.setFormatter(fmt)
    logger.addHandler(fh)
    logger.addHandler(sh)
    logger.propagate = False
    return logger
