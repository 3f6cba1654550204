from .probabilistic_unet import ProbabilisticUnet, l2_regularisation  # noqa: F401
from .unet import Unet  # noqa: F401


def clean_folder(folder_path):
    """prob_utils/my_models/utils.py:50-55 (re-exported by the reference's my_models package): remove the plain files of a
    folder (used by punet_pseudo_prediction before it writes).  Host-side file utility, kept for import compatibility."""
    import os
    for filename in os.listdir(folder_path):
        myfile = os.path.join(folder_path, filename)
        if os.path.isfile(myfile):
            os.remove(myfile)
