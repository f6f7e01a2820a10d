"""BaseDataset (reference: sseg/datasets/loader/base_dataset.py:15-178): same constructor,
__getitem__ dict keys ('images', 'labels', 'image_paths'[, 'copy_paste_mask']), pseudo-label file
naming (<stem>_pseudo_label.png), samples_with_class.json handling and load-failure fallback."""
import json
import os
import os.path as osp

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

from hiast_amd.sseg.datasets import augmentations, utils


class BaseDataset(Dataset):

    def __init__(self, cfg, json_path, image_dir, pseudo_dir=None, aug_type=[], num_classes=19):
        self.cfg = cfg
        self.pseudo_dir = pseudo_dir
        self.num_classes = num_classes
        self.preprocessor = None
        self.aug_fun = utils.get_aug_fun(aug_type, self.build_aug_fun)
        self.img_path_list, self.lbl_path_list, self.city_list = utils.get_path_list(json_path, image_dir)
        assert len(self.img_path_list) == len(self.lbl_path_list), "images and labels should have the same number"
        self.file_to_idx = {p.split("/")[-1]: i for i, p in enumerate(self.img_path_list)}
        if self.pseudo_dir is not None:
            root = self.pseudo_dir.split(self.pseudo_dir.split("/")[-1])[0]     # parent of the label dir
            self.samples_with_class = self.stat_samples_with_class(root)

    def __len__(self):
        return len(self.img_path_list)

    def __getitem__(self, index):
        if self.preprocessor is not None:
            return self.get_item_with_copy_paste(index)
        return self.original_get_item(index)

    def get_file_to_idx(self, file_name):
        return self.file_to_idx[file_name]

    def get_samples_with_class(self):
        return self.samples_with_class

    def get_aug(self):
        return self.aug_fun

    def get_city_list(self):
        return self.city_list

    def stat_samples_with_class(self, data_root):
        """per class: file names sorted by pixel count, lowest 10 % dropped (base_dataset.py:61-77)"""
        with open(osp.join(data_root, "samples_with_class.json")) as f:
            raw = {int(k): v for k, v in json.load(f).items()}
        out = {}
        for c in range(self.cfg.dataset.num_classes):
            names = [fn.split("/")[-1] for fn, _ in sorted(raw.get(c, []), key=lambda it: it[1])]
            out[c] = names[round(len(names) * 0.1):]
        return out

    def _safe_load(self, index):
        try:
            return self.load_data(index), index
        except Exception as e:   # same recovery as the reference: neighbour index
            print("## {} in loading {}: {}".format(repr(e), index, self.img_path_list[index]))
            index = index - 1 if index > 0 else index + 1
            return self.load_data(index), index

    def original_get_item(self, index):
        (img, lbl, path), index = self._safe_load(index)
        img, lbl = augmentations.aug(self.aug_fun, img, lbl, index)
        img, lbl = utils.transform(img, lbl, raw_u8=getattr(self, "device_transform", False))
        return {"images": img, "labels": lbl, "image_paths": path}

    def get_item_with_copy_paste(self, index):
        (img, lbl, path), index = self._safe_load(index)
        img, lbl, cp_mask = self.preprocessor.run(img, lbl)
        img, lbl = augmentations.aug(self.aug_fun, img, lbl)
        img, lbl = utils.transform(img, lbl, raw_u8=getattr(self, "device_transform", False))
        out = {"images": img, "labels": lbl, "image_paths": path}
        if cp_mask is not None:     # native-resolution map of the pasted classes (255 elsewhere); int64 as the reference,
            m = torch.from_numpy(cp_mask)           # uint8 for device_transform consumers (8x fewer bytes per sample)
            out["copy_paste_mask"] = m if getattr(self, "device_transform", False) else m.long()
        return out

    def _decoded_cache(self):
        """cfg.dataset.decoded_cache_dir (optional): decoded uint8 arrays are kept as raw files and re-read instead of
        decoding the PNG again (decoded_cache.py); created lazily, i.e. inside each DataLoader worker"""
        c = self.__dict__.get("_dcache", False)
        if c is False:
            d = getattr(self.cfg.dataset, "decoded_cache_dir", None)
            c = None
            if d:
                from hiast_amd.sseg.datasets.decoded_cache import DecodedCache
                c = DecodedCache(d, getattr(self.cfg.dataset, "decoded_cache_gb", 16.0))
            self.__dict__["_dcache"] = c
        return c

    def set_preprocessor(self, preprocessor):
        self.preprocessor = preprocessor
        print("%% use {}".format(type(preprocessor).__name__))

    def read_label(self, path):
        raise NotImplementedError

    def build_aug_fun(self, aug_type):
        raise NotImplementedError

    def load_data(self, index):
        """-> (uint8 HxWx3 image, uint8 HxW label, image path); with `pseudo_dir`, the label is the
        generator's <stem>_pseudo_label.png (base_dataset.py:158-178)"""
        img_path = self.img_path_list[index]
        cache = self._decoded_cache()
        dec_img = lambda p: np.array(Image.open(p).convert("RGB"), dtype=np.uint8)      # noqa: E731
        dec_lbl = lambda p: np.array(Image.open(p), dtype=np.uint8)                     # noqa: E731
        img = cache.load(img_path, dec_img) if cache else dec_img(img_path)
        if self.pseudo_dir is not None:
            stem = os.path.splitext(os.path.basename(img_path))[0]
            lp = os.path.join(self.pseudo_dir, stem + "_pseudo_label.png")
            lbl = cache.load(lp, dec_lbl) if cache else dec_lbl(lp)
        elif cache and self.lbl_path_list[index] is not None and os.path.exists(self.lbl_path_list[index]):
            lbl = cache.load(self.lbl_path_list[index], self.read_label,     # (the id-mapped label, as read_label returns it)
                             salt="%s/%d" % (type(self).__name__, self.num_classes))
        else:
            lbl = self.read_label(self.lbl_path_list[index])
        if lbl is None:
            lbl = np.full(img.shape[:2], 255, dtype=np.uint8)
        if lbl.shape != img.shape[:2]:
            lbl = np.asarray(Image.fromarray(lbl).resize((img.shape[1], img.shape[0]), Image.NEAREST))
        return img, lbl, img_path
