from .probabilistic_unet import ProbabilisticUnet, l2_regularisation  # noqa: F401
from .unet import Unet  # noqa: F401
