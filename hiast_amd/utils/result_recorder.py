"""Loss / metric bookkeeping (reference: utils/result_recorder.py).  The reference all-reduces every
loss scalar separately and blocks on .item() each iteration (C4 in SURVEY §2.3); here the losses of an
iteration are packed into ONE device vector, accumulated on the device, and all-reduced / read back
only at report time."""
import time

import numpy as np
import torch
import torch.distributed as dist

from hiast_amd.utils import comm


class ResultRecorder:

    def __init__(self, cfg, gpu_index, g_optimizer, d_optimizer, model_name, logger=None, writer=None):
        self.cfg, self.rank, self.opt, self.model_name, self.logger = cfg, gpu_index, g_optimizer, model_name, logger
        self.best_miou = -1.0
        self.best_iter = 0
        self.reset_time_and_losses()

    def reset_time_and_losses(self):
        self.t0 = time.time()
        self.names = None
        self.acc = None
        self.n = 0

    def record_losses(self, losses):
        names = sorted(losses)
        vec = torch.stack([losses[k].detach().float().mean() for k in names])
        if self.names != names:
            self.names, self.acc, self.n = names, torch.zeros_like(vec), 0
        self.acc += vec
        self.n += 1

    def report_losses(self, current_iter):
        if self.acc is None:
            return {}
        vec = self.acc / max(self.n, 1)
        if comm.multi():      # (N > 1, or the one-rank rehearsal: utils/comm.py)
            dist.all_reduce(vec)
            vec /= dist.get_world_size()
        vals = dict(zip(self.names, vec.cpu().tolist()))
        dt = time.time() - self.t0
        if self.logger is not None:
            lr = self.opt.param_groups[0]["lr"] if self.opt is not None else float("nan")
            msg = ", ".join("{}: {:.4f}".format(k, v) for k, v in vals.items())
            self.logger.info("[{}] iter {}/{}  lr {:.3e}  {:.3f} s/iter  {}".format(
                self.model_name, current_iter, self.cfg.train.total_iter, lr, dt / max(self.n, 1), msg))
        self.reset_time_and_losses()
        return vals

    def record_and_report_metrics(self, miou, iou, current_iter):
        if "SYNTHIA" in str(self.cfg.dataset.source.type):     # result_recorder.py:34-38
            miou = miou * 19 / 16
        if miou > self.best_miou:
            self.best_miou, self.best_iter = miou, current_iter
        if self.logger is not None:
            self.logger.info("[{}] iter {}  mIoU {:.4f} (best {:.4f} @ {})  IoU {}".format(
                self.model_name, current_iter, miou, self.best_miou, self.best_iter,
                {c: round(float(v), 4) for c, v in enumerate(np.asarray(iou))}))

    def report_end_info(self):
        if self.logger is not None:
            self.logger.info("[{}] finished, best mIoU {:.4f} @ iter {}".format(self.model_name, self.best_miou,
                                                                               self.best_iter))
