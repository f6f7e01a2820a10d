"""DATASET['SYNTHIA'] (reference: synthia_dataset.py:9-33): SYNTHIA-RAND-CITYSCAPES ids -> the 16
train ids (the reference maps all 19; classes 9, 14, 16 are excluded from its mIoU-16)."""
import numpy as np
from PIL import Image

from hiast_amd.sseg.datasets import utils
from hiast_amd.sseg.datasets.loader.base_dataset import BaseDataset
from hiast_amd.sseg.datasets.loader.cityscapes_dataset import common_aug
from hiast_amd.utils.registry.registries import DATASET

_ID_MAP = {3: 0, 4: 1, 2: 2, 21: 3, 5: 4, 7: 5, 15: 6, 9: 7, 6: 8, 16: 9, 1: 10, 10: 11, 17: 12, 8: 13, 18: 14, 19: 15,
           20: 16, 12: 17, 11: 18}


@DATASET.register("SYNTHIA")
class SYNTHIADataset(BaseDataset):

    def read_label(self, path):
        lbl = np.array(Image.open(path))
        if lbl.ndim == 3:
            lbl = lbl[..., 0]
        return utils.preprocess_label(lbl.astype(np.uint8), _ID_MAP)

    def build_aug_fun(self, aug_type):
        """synthia_dataset.py:20-33"""
        return common_aug(self, aug_type, ms=(341, 640), dacs=(760, 1280), fda_target=True)
