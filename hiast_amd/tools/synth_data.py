"""Synthetic target-domain data in the reference's on-disk layout (no dataset is reachable offline):
Cityscapes-style directory tree + `cityscapes_{train,val}.json` index files with
{"image_name", "mask_name", "has_target"} entries (the format of the reference's data/*.json), so the
DATASET / generator / trainer / validator code paths run unchanged.

Images are smooth random colour fields with a per-class bias, labels are piecewise-constant blobs in
{0..C-1, 255}.  Everything derives from numpy PCG64(seed)."""
import json
import os

import numpy as np
from PIL import Image


def _smooth(g, n, h, w, cells):
    coarse = g.standard_normal((n, cells[0], cells[1])).astype(np.float32)
    ys = np.linspace(0, cells[0] - 1, h)
    xs = np.linspace(0, cells[1] - 1, w)
    y0 = np.clip(np.floor(ys).astype(int), 0, cells[0] - 2)
    x0 = np.clip(np.floor(xs).astype(int), 0, cells[1] - 2)
    fy = (ys - y0).astype(np.float32)[None, :, None]
    fx = (xs - x0).astype(np.float32)[None, None, :]
    a = coarse[:, y0][:, :, x0]
    b = coarse[:, y0][:, :, x0 + 1]
    c = coarse[:, y0 + 1][:, :, x0]
    d = coarse[:, y0 + 1][:, :, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_sample(seed, h, w, num_classes=19, absent=()):
    g = np.random.Generator(np.random.PCG64(seed))
    fields = _smooth(g, num_classes, h, w, (5, 9))
    for c in absent:
        fields[c] = -1e9
    lbl = fields.argmax(0).astype(np.uint8)
    ign = _smooth(g, 1, h, w, (4, 7))[0] > 1.1
    lbl[ign] = 255
    palette = np.random.Generator(np.random.PCG64(12345)).integers(30, 226, size=(256, 3))
    base = palette[lbl].astype(np.float32)
    tex = _smooth(g, 3, h, w, (9, 17)).transpose(1, 2, 0) * 18.0
    img = np.clip(base + tex + g.standard_normal((h, w, 3)).astype(np.float32) * 4.0, 0, 255).astype(np.uint8)
    return img, lbl


def _write_one(args):
    image_dir, img_rel, lbl_rel, seed, h, w, num_classes, absent, upscale = args
    img, lbl = make_sample(seed, h // upscale, w // upscale, num_classes, absent)
    if upscale > 1:      # large frames: blocky upsample of a smaller sample (cheap to generate, same statistics)
        img = np.repeat(np.repeat(img, upscale, axis=0), upscale, axis=1)
        lbl = np.repeat(np.repeat(lbl, upscale, axis=0), upscale, axis=1)
    Image.fromarray(img).save(os.path.join(image_dir, img_rel), compress_level=1)
    Image.fromarray(lbl, mode="L").save(os.path.join(image_dir, lbl_rel), compress_level=1)


def write_cityscapes_like(root, split, n, h, w, seed=0, num_classes=19, absent=(), upscale=1, procs=0):
    """-> (json_path, image_dir); image_dir is what cfg.dataset.*.image_dir should point at.
    upscale / procs only speed up the writing of large benchmark sets (defaults keep the files byte-identical)."""
    image_dir = os.path.join(root, "data", "cityscapes")
    entries, jobs = [], []
    for i in range(n):
        stem = "synth_%06d_%06d" % (seed, i)
        img_rel = "leftImg8bit/%s/synth/%s_leftImg8bit.png" % (split, stem)
        lbl_rel = "gtFine/%s/synth/%s_gtFine_labelTrainIds.png" % (split, stem)
        for rel in (img_rel, lbl_rel):
            os.makedirs(os.path.dirname(os.path.join(image_dir, rel)), exist_ok=True)
        jobs.append((image_dir, img_rel, lbl_rel, seed * 100003 + i, h, w, num_classes, absent, upscale))
        entries.append({"image_name": img_rel, "mask_name": lbl_rel, "has_target": True})
    if procs > 1 and n > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(procs) as pool:
            pool.map(_write_one, jobs)
    else:
        for j in jobs:
            _write_one(j)
    json_path = os.path.join(root, "data", "cityscapes_%s.json" % split)
    with open(json_path, "w") as f:
        json.dump(entries, f)
    return json_path, image_dir


def synthetic_cfg(root, n_train=8, n_val=4, h=256, w=512, seed=1, num_classes=19, source_type="GTAV", upscale=1,
                  procs=0):
    """write a tiny dataset and return a cfg (hiast_amd.utils.default_config.CfgNode) pointing at it"""
    from hiast_amd.utils.default_config import get_default_cfg
    absent = (9, 14, 16) if source_type == "SYNTHIA" else ()
    tj, td = write_cityscapes_like(root, "train", n_train, h, w, seed, num_classes, absent, upscale, procs)
    vj, vd = write_cityscapes_like(root, "val", n_val, h, w, seed + 1, num_classes, absent, upscale, procs)
    c = get_default_cfg()
    c.dataset.num_classes = num_classes
    c.dataset.num_workers = 0
    c.dataset.source.type = source_type
    c.dataset.target.type = "Cityscapes"
    c.dataset.target.json_path = tj
    c.dataset.target.image_dir = td
    c.dataset.val.type = "Cityscapes"
    c.dataset.val.json_path = vj
    c.dataset.val.image_dir = vd
    c.dataset.val.resize_size = [h, w]
    c.model.type = "SelfTrainingSegmentor"
    c.model.predictor.ent_loss.weight = 1.0
    c.pseudo_policy.type = "IAS"
    c.pseudo_policy.batch_size = 2
    c.pseudo_policy.resize_size = [h, w]
    c.pseudo_policy.ias.alpha = 0.5
    c.pseudo_policy.save_dir = os.path.join(root, "pseudo", "pseudo_labels")
    c.validate.resize_sizes = [[h, w]]
    c.validate.batch_size = 2
    c.work_dir = os.path.join(root, "work")
    return c


def calibrate_bn(model, x):
    """Give a seeded / random-init network the BatchNorm running statistics of the data it will see: one training-mode
    forward of `x` with momentum 1 (running := batch statistics), in place.  A checkpoint whose running statistics do not
    belong to its weights lets the activations of an eval-mode forward grow block by block (nothing renormalises them) — in
    fp16, the reference's training type, past 65504 within the 33 blocks of a ResNet-101.  Trained checkpoints are
    calibrated by construction; synthetic ones need this."""
    import torch
    bns = [m for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
    mom = [m.momentum for m in bns]
    nbt = [None if m.num_batches_tracked is None else m.num_batches_tracked.clone() for m in bns]
    was = model.training
    with torch.no_grad():
        for m in bns:
            m.momentum = 1.0
        model.train()
        model(x, lowres=True) if "lowres" in model.forward.__code__.co_varnames else model(x)
        for m, v, n in zip(bns, mom, nbt):
            m.momentum = v
            if n is not None:
                m.num_batches_tracked.copy_(n)       # (the calibration forward is not a training iteration)
    model.train(was)
    return model
