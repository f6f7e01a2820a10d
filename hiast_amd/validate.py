"""CLI: python -m hiast_amd.validate --config_file ... (reference: validate.py); `--device cpu|cuda` added."""
import argparse

import torch

from hiast_amd.utils.registry import register  # noqa: F401
from hiast_amd.utils.default_config import cfg
from hiast_amd.utils.registry.registries import SEG_MODEL
from hiast_amd.workflows.validator import Validator


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="UDA-Experiment Validation")
    p.add_argument("--config_file", required=True)
    p.add_argument("--setting_file")
    p.add_argument("--resume_from")
    p.add_argument("--color_mask_dir_path")
    p.add_argument("--seg_model", choices=list(SEG_MODEL.keys()))
    p.add_argument("--device", default=None)
    return p.parse_args(argv)


def update_cfg(cfg, args):
    cfg.merge_from_file(args.config_file)
    if args.setting_file:
        cfg.merge_from_file(args.setting_file)
    if args.resume_from:
        cfg.validate.resume_from = args.resume_from
    if args.color_mask_dir_path:
        cfg.validate.color_mask_dir_path = args.color_mask_dir_path
    if args.seg_model:
        cfg.model.seg_model.type = args.seg_model
    cfg.freeze()
    return cfg


def main(argv=None):
    args = parse_args(argv)
    c = update_cfg(cfg, args)
    v = Validator(c, device=torch.device(args.device) if args.device else None)
    return v.run()


if __name__ == "__main__":
    main()
