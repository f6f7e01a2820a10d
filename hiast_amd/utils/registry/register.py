"""Import-for-registration (utils/registry/register.py:3-9 of the reference): importing this module
fills LOSS / DATASET / SEG_MODEL / MODEL / PSEUDO_POLICY / PREPROCESSOR / TRAINER."""
from hiast_amd.sseg.models.modules import losses  # noqa: F401
from hiast_amd.sseg.datasets.loader import cityscapes_dataset, gtav_dataset, oxford_dataset, synthia_dataset  # noqa: F401
from hiast_amd.sseg.models.modules.seg_models.deeplab_v2 import DeepLab_V2  # noqa: F401
from hiast_amd.sseg.models.segmentors import adversarial_warmup_segmentor, self_training_segmentor, source_only_segmentor  # noqa: F401
from hiast_amd.workflows import pseudo_label_generator  # noqa: F401
from hiast_amd.sseg.datasets import preprocessor  # noqa: F401
from hiast_amd.workflows.trainer import self_training_trainer, consistency_self_training_trainer  # noqa: F401
from hiast_amd.workflows.trainer import adversarial_warmup_trainer, source_only_trainer  # noqa: F401,E402
