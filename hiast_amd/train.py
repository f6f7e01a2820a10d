"""CLI: python -m hiast_amd.train --config_file ... --work_dir ... (reference: train.py).
Same flags and cfg priority (config_file < setting_file < args); one process per GPU.  Launch either
plainly (spawns torch.cuda.device_count() ranks itself, like the reference's mp.spawn) or under
torchrun (RANK/WORLD_SIZE in the environment)."""
import argparse
import os

import torch
import torch.multiprocessing as mp

from hiast_amd.utils.registry import register  # noqa: F401
from hiast_amd.utils import utils
from hiast_amd.utils.default_config import cfg
from hiast_amd.utils.registry.registries import SEG_MODEL, TRAINER


def main_worker(proc_idx, cfg):
    TRAINER[cfg.trainer](cfg, proc_idx).run()


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="UDA-Experiment Training")
    p.add_argument("--config_file", required=True)
    p.add_argument("--setting_file")
    p.add_argument("--resume_from")
    p.add_argument("--pseudo_save_dir")
    p.add_argument("--work_dir", required=True)
    p.add_argument("--seg_model", choices=list(SEG_MODEL.keys()))
    return p.parse_args(argv)


def update_cfg(cfg, args):
    cfg.merge_from_file(args.config_file)
    if args.setting_file:
        cfg.merge_from_file(args.setting_file)
    if args.work_dir:
        cfg.work_dir = args.work_dir
    if args.resume_from:
        cfg.train.resume_from = args.resume_from
    if args.pseudo_save_dir:
        cfg.dataset.target.pseudo_dir = args.pseudo_save_dir
    if args.seg_model:
        cfg.model.seg_model.type = args.seg_model
    world = int(os.environ.get("WORLD_SIZE", "0")) or torch.cuda.device_count()
    cfg.train.gpu_num = max(world, 1)
    cfg.train.batch_size //= cfg.train.gpu_num          # train.py:52-53: the yaml value is the GLOBAL batch
    assert cfg.train.batch_size > 0
    print("%% total gpu: {}, batch size (each gpu): {}".format(cfg.train.gpu_num, cfg.train.batch_size))
    while utils.is_port_used(cfg.train.port):
        cfg.train.port += 1
    cfg.freeze()
    return cfg


def main(argv=None):
    args = parse_args(argv)
    c = update_cfg(cfg, args)
    os.makedirs(c.work_dir, exist_ok=True)
    with open(os.path.join(c.work_dir, os.path.basename(args.config_file)), "w") as f:
        f.write(c.dump())
    if "RANK" in os.environ:                # torchrun: this process is one rank
        main_worker(int(os.environ.get("LOCAL_RANK", "0")), c)
    elif c.train.gpu_num > 1:
        mp.spawn(main_worker, nprocs=c.train.gpu_num, args=(c,))
    else:
        main_worker(0, c)


if __name__ == "__main__":
    main()
