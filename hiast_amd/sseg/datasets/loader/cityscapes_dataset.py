"""DATASET['Cityscapes'] (reference: cityscapes_dataset.py:9-45): labelTrainIds PNGs, aug-string table."""
import numpy as np
from PIL import Image

from hiast_amd.sseg.datasets import augmentations, utils
from hiast_amd.sseg.datasets.loader.base_dataset import BaseDataset
from hiast_amd.utils.registry.registries import DATASET

# 19 -> 9 classes (Cityscapes -> Oxford RobotCar)
_TO_9 = {0: 8, 1: 7, 2: 6, 6: 5, 7: 4, 10: 0, 11: 1, 12: 1, 13: 3, 14: 3, 15: 3, 17: 2, 18: 2}


def common_aug(ds, aug_type, ms=None, oms=None, dacs=None, color=False, fda_source=False, fda_target=False):
    """the aug-string table of one dataset class (reference: <dataset>.build_aug_fun): `ms` / `oms` = min_max_height of
    the 'MS' / 'OMS' random sized crop, `dacs` = (h, w) of the 'DACS' resize, `color`: 'SCA' / 'CCA' available,
    `fda_source` / `fda_target`: 'FDA-Source' / 'FDA-Target' available"""
    if aug_type is None or aug_type == "":
        return None
    if aug_type == "MS" and ms is not None:
        return augmentations.flip_crop_resize(512, 1024, min_max_height=ms, w2h_ratio=2)
    if aug_type == "OMS" and oms is not None:      # Cityscapes -> Oxford RobotCar
        return augmentations.flip_crop_resize(768, 1024, min_max_height=oms, w2h_ratio=1280 / 960)
    if aug_type == "DACS" and dacs is not None:
        return augmentations.resize_crop(dacs[0], dacs[1], 512, 512)
    if aug_type == "SCA" and color:
        return augmentations.simple_color_aug()
    if aug_type == "CCA" and color:
        return augmentations.complex_color_aug()
    if "PRS" in aug_type:
        h, w = utils.parse_resize_params(aug_type)
        return augmentations.resize(h, w)
    if aug_type == "FDA-Source" and fda_source:    # transfer the SOURCE domain's style onto this dataset's images
        return augmentations.fda(ds.cfg.dataset.source.json_path, ds.cfg.dataset.source.image_dir)
    if aug_type == "FDA-Target" and fda_target:
        return augmentations.fda(ds.cfg.dataset.target.json_path, ds.cfg.dataset.target.image_dir)
    raise ValueError("aug_type is not valid")


@DATASET.register("Cityscapes")
class CityscapesDataset(BaseDataset):

    def read_label(self, path):
        assert self.num_classes in (9, 19)
        lbl = np.array(Image.open(path), dtype=np.uint8)
        if self.num_classes == 9:
            lbl = utils.preprocess_label(lbl, _TO_9)
        return lbl

    def build_aug_fun(self, aug_type):
        """cityscapes_dataset.py:22-45"""
        src = self.cfg.dataset.source.type
        if aug_type == "FDA-Source":
            assert src in ("GTAV", "SYNTHIA"), "FDA-Source for Cityscapes is only valid for GTAV/SYNTHIA to Cityscapes"
        if aug_type == "FDA-Target":
            assert src == "Oxford", "FDA-Target for Cityscapes is only valid for Cityscapes to Oxford RobotCar"
        return common_aug(self, aug_type, ms=(341, 1000), oms=(341, 1000), dacs=(512, 1024), color=True, fda_source=True,
                          fda_target=True)
