"""CLI: python -m hiast_amd.generate_pseudo_labels --config_file ... (reference: generate_pseudo_labels.py).
Same flags; `--batch_size` works (the reference reads cfg.batch_size there and raises).  Under
torchrun (WORLD_SIZE > 1) the target images are sharded over the ranks."""
import argparse
import os

from hiast_amd.utils.registry import register  # noqa: F401
from hiast_amd.utils.default_config import cfg
from hiast_amd.utils.registry.registries import PSEUDO_POLICY, SEG_MODEL


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--config_file", required=True)
    p.add_argument("--setting_file")
    p.add_argument("--pseudo_resume_from")
    p.add_argument("--pseudo_save_dir")
    p.add_argument("--batch_size", type=int)
    p.add_argument("--seg_model", choices=list(SEG_MODEL.keys()))
    return p.parse_args(argv)


def update_cfg(cfg, args):
    cfg.merge_from_file(args.config_file)
    if args.setting_file:
        cfg.merge_from_file(args.setting_file)
    if args.pseudo_resume_from:
        cfg.pseudo_policy.resume_from = args.pseudo_resume_from
    if args.batch_size:
        cfg.pseudo_policy.batch_size = args.batch_size
    if args.pseudo_save_dir:
        cfg.pseudo_policy.save_dir = args.pseudo_save_dir
    if args.seg_model:
        cfg.model.seg_model.type = args.seg_model
    cfg.freeze()
    return cfg


def main(argv=None):
    c = update_cfg(cfg, parse_args(argv))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        from hiast_amd.utils import comm
        dist.init_process_group(backend="nccl", timeout=comm.timeout())
        comm.apply_cu_reserve()
    PSEUDO_POLICY[c.pseudo_policy.type](c).run()


if __name__ == "__main__":
    main()
