from .util import DummyLoss, dice_score  # noqa: F401
from .augment import (DualViewLoader, ViewAugmenter, apply_view, my_standardize_torch, my_strong_augmentations,  # noqa: F401
                      my_weak_augmentations, standardize_stats)
