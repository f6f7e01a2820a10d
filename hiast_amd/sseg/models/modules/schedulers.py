"""LR schedules of the reference (sseg/models/modules/schedulers.py:7-14)."""
from torch.optim.lr_scheduler import CosineAnnealingLR, LambdaLR


def build_scheduler(cfg, optimizer):
    kind = cfg.train.lr_scheduler.type
    total = cfg.train.total_iter
    if kind == "Cosine":      # eta_min is 1e-3 of the BACKBONE lr for every param group
        return CosineAnnealingLR(optimizer, T_max=total, eta_min=cfg.train.lr * 0.001)
    if kind == "Poly":
        power = cfg.train.lr_scheduler.poly.power
        return LambdaLR(optimizer, lambda it: (1 - it / total) ** power)
    raise ValueError("%s is not a valid scheduler" % kind)
