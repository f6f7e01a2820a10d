"""Validator (reference: workflows/validator.py:13-115): multi-scale / flip softmax-sum test-time
evaluation, argmax, IoU accumulation, SYNTHIA 16/13-class rescale.  The device is a parameter
(`device=`) instead of the reference's hard-coded .cuda(), so BASELINE config 1 (CPU-only plumbing
run) works; on the HIP device the resamplings and the IoU histogram are the library's kernels."""
import os

import numpy as np
import torch
import tqdm
from PIL import Image
from torch.nn import functional as F
from torch.utils.data import DataLoader

from hiast_amd.utils import metrics, utils
from hiast_amd.utils.registry.registries import DATASET

PALETTE_19 = [128, 64, 128, 244, 35, 232, 70, 70, 70, 102, 102, 156, 190, 153, 153, 153, 153, 153, 250, 170, 30,
              220, 220, 0, 107, 142, 35, 152, 251, 152, 70, 130, 180, 220, 20, 60, 255, 0, 0, 0, 0, 142, 0, 0, 70,
              0, 60, 100, 0, 80, 100, 0, 0, 230, 119, 11, 32]
PALETTE_9 = [70, 130, 180, 220, 20, 60, 119, 11, 32, 0, 0, 142, 220, 220, 0, 250, 170, 30, 70, 70, 70, 244, 35, 232,
             128, 64, 128]


def _resample(x, size):
    if tuple(x.shape[2:]) == tuple(size):
        return x
    if x.is_cuda:
        from hiast_amd import functional as HF
        return HF.upsample_bilinear_ac(x.float(), size)
    return F.interpolate(x, size, mode="bilinear", align_corners=True)


class Validator:

    def __init__(self, cfg, device=None):
        self.cfg = cfg
        self.device = device if device is not None else utils.get_device()
        if self.device.type == "cuda":
            utils.limit_cpu_threads()
        self.initialize()

    def initialize(self):
        self.model = utils.load_model(self.cfg, resume_from=self.cfg.validate.resume_from).to(self.device)
        v = self.cfg.dataset.val
        ds = DATASET[v.type](self.cfg, v.json_path, v.image_dir, num_classes=self.cfg.dataset.num_classes)
        # on the HIP device the workers hand over uint8 frames and labels; ToTensor + Normalize run on the device
        # (hiast_normalize_u8: the same bits as the host transform) — as the trainers' loaders do since round 2.  A 2048 x 1024
        # frame is 6 + 2 MB through the worker pipes and PCIe instead of 25 + 17 MB of float32 / int64, and the workers no longer
        # spend their time on the float conversion (2048 x 1024, flip TTA, 8 workers: 47.9 -> 55.0 frames/s, 61.8 with the batched
        # mirror image below; profiles/r06_validator_synth.txt).  Workers persist across run() calls (a trainer validates every
        # iter_val iterations).  HIAST_HOST_TRANSFORM=1: the reference's float32 / int64 batches.
        ds.device_transform = self.device.type == "cuda" and os.environ.get("HIAST_HOST_TRANSFORM", "0") != "1"
        nw = self.cfg.dataset.num_workers
        self.v_loader = DataLoader(ds, self.cfg.validate.batch_size, num_workers=nw, pin_memory=self.device.type == "cuda",
                                   persistent_workers=nw > 0)
        d = self.cfg.validate.color_mask_dir_path
        if d is not None:
            assert not os.path.exists(d) or len(os.listdir(d)) == 0
            os.makedirs(d, exist_ok=True)

    def _tta_device(self, imgs, want_probs, want_label):
        """the softmax-sum TTA on the HIP device: every (scale, flip) forward hands over its LOW-RES head output and
        one kernel (hiast_tta_fused) does upsample -> softmax -> (+ flipped) -> resample to native -> Σ scales
        (-> argmax); no full-resolution logits or per-scale probability maps are stored"""
        from hiast_amd import kernels as K
        zs, zfs, sizes = [], [], []
        flip = bool(self.cfg.validate.is_flip)
        for size in self.cfg.validate.resize_sizes:
            assert len(size) == 2 and size[0] <= size[1], \
                "each resize_size is [height, width] with height <= width, such as [512, 1024]"
            x = _resample(imgs, size)
            # the view and its mirror image in ONE forward (round 6): an inference forward treats every image alone, and one or
            # two frames leave most of the chip idle.  Only where doubling the launch does not change which kernel runs (the
            # xconv kernels take 1x1 launches from 4096 rows on: frames whose 1/8-resolution map has >= 4096 pixels), so the
            # logits are bit for bit those of two forwards; HIAST_VAL_BATCH_FLIP=0: two forwards
            if (flip and (int(size[0]) // 8) * (int(size[1]) // 8) >= 4096
                    and os.environ.get("HIAST_VAL_BATCH_FLIP", "1") != "0"):
                z2 = self.model(torch.cat([x, torch.flip(x, dims=[3])], 0), lowres=True)["logits_lowres"].float()
                n = x.shape[0]
                zs.append(z2[:n].contiguous())
                zfs.append(z2[n:].contiguous())
            else:
                zs.append(self.model(x, lowres=True)["logits_lowres"].float().contiguous())
                if flip:
                    zfs.append(self.model(torch.flip(x, dims=[3]), lowres=True)["logits_lowres"].float().contiguous())
            sizes.append((int(size[0]), int(size[1])))
        H, W = imgs.shape[2:]
        return K.tta_fused(zs, zfs if self.cfg.validate.is_flip else None, sizes, H, W, want_probs, want_label)

    def _fused_tta_ok(self):
        """hiast_tta_fused is instantiated for C in {19, 16, 9, 2} and up to 8 scales; any other class count / scale list
        takes the general composition below (own resampling kernel + torch softmax), as the reference handles any"""
        from hiast_amd import kernels as K
        return K.tta_fused_supported(self.cfg.dataset.num_classes, len(self.cfg.validate.resize_sizes))

    def get_multi_scale_and_flip_logits(self, imgs, is_softmax=True):
        """validator.py:34-55: Σ over scales of softmax(model(resized)) (+ flipped), each resampled
        back to the native size"""
        if imgs.is_cuda and is_softmax and self._fused_tta_ok():
            return self._tta_device(imgs, True, False)[0]

        def pred(x):
            y = self.model(x)["logits"]
            return F.softmax(y, dim=1) if is_softmax else y
        total = None
        for size in self.cfg.validate.resize_sizes:
            assert len(size) == 2 and size[0] <= size[1], \
                "each resize_size is [height, width] with height <= width, such as [512, 1024]"
            x = _resample(imgs, size)
            r = pred(x)
            if self.cfg.validate.is_flip:
                r = r + torch.flip(pred(torch.flip(x, dims=[3])), dims=[3])
            r = _resample(r, imgs.shape[2:])
            total = r if total is None else total + r
        return total

    def colorize_mask(self, mask):
        C = self.cfg.dataset.num_classes
        if C not in (19, 9):
            raise NotImplementedError
        m = Image.fromarray(mask.astype(np.uint8)).convert("P")
        m.putpalette(PALETTE_19 if C == 19 else PALETTE_9)
        return m

    def save_color_mask(self, lbls_pred, img_paths):
        for l, p in zip(lbls_pred, img_paths):
            self.colorize_mask(l).save(os.path.join(self.cfg.validate.color_mask_dir_path, os.path.basename(p)))

    @torch.no_grad()
    def run(self):
        v = self.cfg.validate
        print("%% batch_size: {}".format(v.batch_size))
        print("%% num_classes: {}".format(self.cfg.dataset.num_classes))
        print("%% resize_sizes: {}".format(v.resize_sizes))
        print("%% is_flip: {}".format(v.is_flip))
        C = self.cfg.dataset.num_classes
        acc = torch.zeros(2, C, dtype=torch.int64, device=self.device)
        self.model.eval()
        for data in tqdm.tqdm(self.v_loader, desc="Validation", ncols=100):
            if self.device.type == "cuda":
                from hiast_amd.sseg.datasets import utils as du
                imgs, lbls = du.to_device_batch(data["images"], data["labels"], self.device)
                lbls = lbls.long()
            else:
                imgs = data["images"].to(self.device)
                lbls = data["labels"].to(self.device)
            if imgs.is_cuda and self._fused_tta_ok():     # fused tail: label map straight from the low-res head outputs
                pred = self._tta_device(imgs, False, True)[1].long()
            else:
                pred = self.get_multi_scale_and_flip_logits(imgs).argmax(dim=1)
            inter, union = metrics.intersection_union_counts(pred, lbls, C)
            acc[0] += inter
            acc[1] += union
            if v.color_mask_dir_path is not None:
                self.save_color_mask(pred.cpu().numpy(), data["image_paths"])
        acc = acc.cpu().numpy().astype(np.float64)
        iou = acc[0] / (acc[1] + 1e-10)
        miou = float(np.mean(iou))
        self.iou, self.miou, self.miou_13 = iou, miou, None
        if "SYNTHIA" in str(self.cfg.dataset.source.type):     # validator.py:108-113
            miou *= 19 / 16
            iu13 = iou.copy()
            iu13[3:6] = 0
            self.miou_13 = float(np.mean(iu13)) * 19 / 13
            self.miou = miou
            print("miou_16: {:.4f}, miou_13: {:.4f}, iou: {}".format(miou, self.miou_13,
                                                                   {c: round(float(x), 4) for c, x in enumerate(iou)}))
        else:
            print("miou: {:.4f}, iou: {}".format(miou, {c: round(float(x), 4) for c, x in enumerate(iou)}))
        return self.miou
