"""MI355X-native (gfx950 / CDNA4) Probabilistic U-Net self-training hot path.

Layout mirrors the reference's `prob_utils` package for the hot path only:
  my_models       ProbabilisticUnet, l2_regularisation        (prob_utils/my_models)
  my_trainer      PUNet / MeanTeacher / FixMatch / AdaMT / AdaMatch / PseudoPUNet trainers, FusedAdam
  my_predictions  MC-sample tiled prediction, pseudo-label + consensus prediction
  my_utils        dice_score
  csrc/           hand-written HIP kernels + the C ABI (include/punet_hip.h) -> libpunet_hip.so
  engine.py       host-side graph walker above the C ABI
"""
__all__ = ["my_models", "my_trainer", "my_predictions", "my_utils", "engine"]
