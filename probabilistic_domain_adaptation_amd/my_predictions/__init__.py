"""Mirror of prob_utils/my_predictions/__init__.py (PUNet functions; `unet_prediction` for torch_em's UNet2d is out of scope)."""
from .punet_predictions import (mc_prediction, predict_image, pseudo_predict_image, punet_prediction,  # noqa: F401
                                punet_pseudo_prediction, standardize)
