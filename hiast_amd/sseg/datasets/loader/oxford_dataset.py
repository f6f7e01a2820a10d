"""DATASET['Oxford'] (reference: oxford_dataset.py:9-39): Oxford RobotCar, 9 classes.  Annotated frames are
RGBA PNGs whose first channel carries the class id; unlabeled training frames have no PNG label."""
import numpy as np
from PIL import Image

from hiast_amd.sseg.datasets import augmentations, utils
from hiast_amd.sseg.datasets.loader.base_dataset import BaseDataset
from hiast_amd.sseg.datasets.loader.cityscapes_dataset import common_aug
from hiast_amd.utils.registry.registries import DATASET

# RobotCar annotation id -> 9 train ids (sky, person, two-wheel, automobile, sign, light, building, sidewalk, road)
_ID_MAP = {1: 0, 2: 1, 3: 2, 4: 3, 5: 4, 6: 5, 7: 6, 10: 7, 11: 8, 12: 8, 13: 8, 14: 8, 17: 8}


@DATASET.register("Oxford")
class OxfordDataset(BaseDataset):

    def read_label(self, path):
        assert self.num_classes == 9, "Oxford RobotCar is a 9-class target (Cityscapes -> Oxford RobotCar)"
        if not path.endswith(".png"):
            return None
        lbl = np.asarray(Image.open(path), dtype=np.uint8)
        if lbl.ndim == 3:
            lbl = lbl[:, :, 0]
        return utils.preprocess_label(lbl, _ID_MAP)

    def build_aug_fun(self, aug_type):
        """oxford_dataset.py:24-39"""
        return common_aug(self, aug_type, oms=(341, 900), color=True, fda_source=True)
