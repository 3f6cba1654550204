"""Mirror of prob_utils/my_predictions/__init__.py: `punet_prediction`, `punet_pseudo_prediction`, `unet_prediction` (the last for this
package's plain `Unet`; the reference drives torch_em's UNet2d through it, which is out of scope) + the array-level functions."""
from .punet_predictions import (mc_prediction, predict_image, pseudo_predict_image, punet_prediction,  # noqa: F401
                                punet_pseudo_prediction, standardize, unet_predict_image, unet_prediction)
