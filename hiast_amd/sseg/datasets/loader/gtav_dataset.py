"""DATASET['GTAV'] (reference: gtav_dataset.py:9-31): GTA5 label ids -> 19 train ids."""
import numpy as np
from PIL import Image

from hiast_amd.sseg.datasets import utils
from hiast_amd.sseg.datasets.loader.base_dataset import BaseDataset
from hiast_amd.sseg.datasets.loader.cityscapes_dataset import common_aug
from hiast_amd.utils.registry.registries import DATASET

_ID_MAP = {7: 0, 8: 1, 11: 2, 12: 3, 13: 4, 17: 5, 19: 6, 20: 7, 21: 8, 22: 9, 23: 10, 24: 11, 25: 12, 26: 13,
           27: 14, 28: 15, 31: 16, 32: 17, 33: 18}


@DATASET.register("GTAV")
class GTAVDataset(BaseDataset):

    def read_label(self, path):
        return utils.preprocess_label(np.array(Image.open(path), dtype=np.uint8), _ID_MAP)

    def build_aug_fun(self, aug_type):
        """gtav_dataset.py:18-31"""
        return common_aug(self, aug_type, ms=(341, 950), dacs=(720, 1280), fda_target=True)
