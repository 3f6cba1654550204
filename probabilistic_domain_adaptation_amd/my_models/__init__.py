from .probabilistic_unet import ProbabilisticUnet, l2_regularisation  # noqa: F401
