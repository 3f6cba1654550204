"""Mirror of prob_utils/my_trainer/__init__.py:1-6 (the PUNet trainers; `PseudoTrainer` for plain UNets is out of scope)."""
from .default_trainer import DefaultTrainer  # noqa: F401
from .fused import FusedAdam  # noqa: F401
from .trainers import (AdaMatchTrainer, AdaMTTrainer, FixMatchTrainer, MeanTeacherTrainer, PseudoTrainerPUNet,  # noqa: F401
                       PUNetTrainer)
