"""DATASET['Cityscapes'] (reference: cityscapes_dataset.py:9-45): labelTrainIds PNGs, aug-string table."""
import numpy as np
from PIL import Image

from hiast_amd.sseg.datasets import augmentations, utils
from hiast_amd.sseg.datasets.loader.base_dataset import BaseDataset
from hiast_amd.utils.registry.registries import DATASET

# 19 -> 9 classes (Cityscapes -> Oxford RobotCar)
_TO_9 = {0: 8, 1: 7, 2: 6, 6: 5, 7: 4, 10: 0, 11: 1, 12: 1, 13: 3, 14: 3, 15: 3, 17: 2, 18: 2}


def common_aug(aug_type, crop_hw=(512, 1024), w2h=2.0):
    if aug_type is None or aug_type == "":
        return None
    if aug_type == "MS":
        return augmentations.flip_crop_resize(crop_hw[0], crop_hw[1], min_max_height=(341, 1000), w2h_ratio=w2h)
    if aug_type == "SCA":
        return augmentations.simple_color_aug()
    if aug_type == "CCA":
        return augmentations.complex_color_aug()
    if "PRS" in aug_type:
        h, w = utils.parse_resize_params(aug_type)
        return augmentations.resize(h, w)
    raise ValueError("aug_type %r is not available in this build" % (aug_type,))


@DATASET.register("Cityscapes")
class CityscapesDataset(BaseDataset):

    def read_label(self, path):
        assert self.num_classes in (9, 19)
        lbl = np.array(Image.open(path), dtype=np.uint8)
        if self.num_classes == 9:
            lbl = utils.preprocess_label(lbl, _TO_9)
        return lbl

    def build_aug_fun(self, aug_type):
        if aug_type == "OMS":
            return augmentations.flip_crop_resize(768, 1024, min_max_height=(341, 1000), w2h_ratio=1280 / 960)
        return common_aug(aug_type)
