from .util import dice_score  # noqa: F401
