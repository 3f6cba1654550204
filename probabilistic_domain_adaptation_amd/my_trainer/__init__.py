"""Mirror of prob_utils/my_trainer/__init__.py:1-6 (the PUNet trainers; `PseudoTrainer` for plain UNets is out of scope)."""
from .default_trainer import DefaultTrainer  # noqa: F401
from .fused import FusedAdam  # noqa: F401
from .trainers import (AdaMatchTrainer, AdaMTTrainer, FixMatchTrainer, MeanTeacherTrainer, PseudoTrainerPUNet,  # noqa: F401
                       PUNetTrainer)


class _NoOpLogger:
    """The reference's tensorboard loggers (e.g. mean_teacher_trainer.py:194-230) dump images and scalars and contain no
    compute; they are out of the hot-path scope.  These stand-ins keep the reference scripts' imports and
    `logger=MeanTeacherLogger` keyword working; the trainers expose what the loggers would have received as
    `trainer.last_loss`, `trainer.last_epoch_loss` and `trainer.last_validation`."""

    def __init__(self, trainer=None, save_root=None, **unused_kwargs):
        self.trainer, self.save_root = trainer, save_root

    def log_train(self, *args, **kwargs):
        pass

    def log_validation(self, *args, **kwargs):
        pass


class PUNetLogger(_NoOpLogger):
    pass


class MeanTeacherLogger(_NoOpLogger):
    pass


class FixMatchLogger(_NoOpLogger):
    pass


class AdaMTLogger(_NoOpLogger):
    pass


class AdaMatchLogger(_NoOpLogger):
    pass


class PseudoLogger(_NoOpLogger):
    pass
