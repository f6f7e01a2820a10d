"""Dataset helpers (reference: sseg/datasets/utils.py:21-82)."""
import json
import os

import numpy as np
import torch

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def get_path_list(json_path, image_dir):
    """[{image_name, mask_name, ...}] index -> absolute image / label path lists (+ the city ids the
    reference derives for 'cityscapes_*' index files from the 5th path component, utils.py:28-30)."""
    with open(json_path) as f:
        entries = json.load(f)
    imgs = [os.path.join(image_dir, e["image_name"]) for e in entries]
    lbls = [os.path.join(image_dir, e["mask_name"]) for e in entries]
    if os.path.basename(json_path).split("_")[0] == "cityscapes":
        names = [p.split("/")[4] if len(p.split("/")) > 4 else "" for p in imgs]
        uniq = sorted(set(names))
        cities = np.array([uniq.index(n) for n in names], dtype=int)
    else:
        cities = [0 for _ in imgs]
    return imgs, lbls, cities


def _img_to_tensor(img, mean, std):
    """torchvision ToTensor + Normalize: uint8 HWC -> float32 CHW, /255, (x-mean)/std"""
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(img).transpose(2, 0, 1))).float().div(255)
    m = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)
    return (t - m) / s


def transform(img, lbl, mean=MEAN, std=STD, raw_u8=False):
    """utils.py:37-55: images normalised float32 CHW, labels int64; lists map element-wise.
    raw_u8: hand the uint8 HWC image and the uint8 label over untouched — the consumer normalises the image on the
    device (hiast_normalize_u8, same bits) and the fused loss reads uint8 labels directly: 4x (images) and 8x
    (labels) fewer bytes through the worker pipes, the pinned staging buffers and PCIe."""
    if raw_u8:
        def one_lbl(x):
            return torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint8))

        def one_img(i):
            return torch.from_numpy(np.ascontiguousarray(np.asarray(i), dtype=np.uint8))
    else:
        def one_lbl(x):
            return torch.from_numpy(np.ascontiguousarray(x)).long()

        def one_img(i):
            return _img_to_tensor(i, mean, std)
    img_t = [one_img(i) for i in img] if isinstance(img, (list, tuple)) else one_img(img)
    lbl_t = [one_lbl(l) for l in lbl] if isinstance(lbl, (list, tuple)) else one_lbl(lbl)
    return img_t, lbl_t


def to_device_batch(images, labels, device):
    """a collated batch -> device tensors: uint8 HWC images (device_transform datasets) are normalised on the device,
    float images pass through; labels keep their dtype (the fused loss takes uint8 or int64)"""
    def img(t):
        t = t.to(device, non_blocking=True)
        if t.dtype == torch.uint8:
            from hiast_amd import kernels as K
            t = K.normalize_u8(t, MEAN, STD)
        return t
    imgs = [img(t) for t in images] if isinstance(images, (list, tuple)) else img(images)
    lbls = ([l.to(device, non_blocking=True) for l in labels] if isinstance(labels, (list, tuple))
            else labels.to(device, non_blocking=True))
    return imgs, lbls


def preprocess_label(lbl, id_map, ignored_index=255):
    assert lbl.ndim == 2, "Only label with shape of [H, W] is valid"
    out = np.full(lbl.shape, ignored_index, dtype=np.uint8)
    for k, v in id_map.items():
        out[lbl == k] = v
    return out


def parse_resize_params(aug_type):
    parts = aug_type.split("-")
    assert len(parts) == 3, 'aug_type should be as "PRS-512-1024"'
    return [int(parts[1]), int(parts[2])]


def get_aug_fun(aug_type, build_aug_fun):
    assert isinstance(aug_type, (list, tuple))
    if len(aug_type) >= 2:
        return [build_aug_fun(a) for a in aug_type]
    if len(aug_type) == 1:
        return build_aug_fun(aug_type[0])
    return build_aug_fun(None)
