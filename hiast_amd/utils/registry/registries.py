"""The seven registries of the reference (utils/registry/registries.py:3-9), same names."""
from .registry import Registry

LOSS = Registry()
DATASET = Registry()
MODEL = Registry()
TRAINER = Registry()
PSEUDO_POLICY = Registry()
PREPROCESSOR = Registry()
SEG_MODEL = Registry()
