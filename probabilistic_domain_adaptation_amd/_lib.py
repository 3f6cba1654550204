"""ctypes binding of libpunet_hip.so (C ABI declared in include/punet_hip.h).

The product path has NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PUNET_HIP_LIB", os.path.join(_HERE, "libpunet_hip.so"))  # override: A/B of two builds

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_l = C.c_long
_sz = C.c_size_t

# name -> (restype, argtypes); every int-returning entry point is wrapped to raise on a non-zero code
_SIGS = {
    "punet_last_error": (C.c_char_p, []),
    "punet_build_info": (C.c_char_p, []),
    "punet_device_ok": (_i, []),
    "punet_conv_fwd": (_i, [_p, _i, _p, _i, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "punet_conv_dgrad": (_i, [_p, _i, _p, _i, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "punet_flip_transpose_weights": (_i, [_p, _p, _i, _i, _i, _p]),
    "punet_conv_dgrad_t": (_i, [_p, _i, _p, _i, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "punet_compute_units": (_i, []),
    "punet_split_weights_x6": (_i, [_p, _p, _l, _i, _i, _i, _i, _p]),
    "punet_split_weights_x6_batch": (_i, [_p, _p, _p, _i, _l, _i, _p]),
    "punet_conv3x3_x6": (_i, [_p, _i, _p, _i, _p, _l, _i, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "punet_gate_dwords": (_sz, [_i, _i, _i, _i]),
    "punet_upconv_eligible": (_i, [_i, _i, _i, _i]),
    "punet_upconv_min_cup_grad": (_i, []),
    "punet_conv_splitk_scratch_bytes": (_sz, []),
    "punet_conv3x3_splitk": (_i, [_p, _i, _p, _i, _p, _l, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p,
                                  _p, _sz, _p]),
    "punet_conv3x3_x6_gated": (_i, [_p, _i, _p, _i, _p, _l, _i, _i, _p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "punet_conv3x3_x6_pooled": (_i, [_p, _i, _p, _i, _p, _l, _i, _i, _p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "punet_conv3x3_f16": (_i, [_p, _i, _p, _i, _p, _l, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "punet_split_weights_f16": (_i, [_p, _p, _l, _i, _i, _i, _p, _p]),
    "punet_split_weights_f16_batch": (_i, [_p, _p, _p, _i, _l, _i, _p, _i, _p]),
    "punet_conv_wgrad_f16": (_i, [_p, _i, _p, _i, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _f, _i, _i, _p, _p, _p]),
    "punet_amax": (_i, [_p, _l, _p, _i, _p]),
    "punet_amax_bound": (_i, [_p, _f, _p, _f, _p, _p]),
    "punet_conv1x1_x6_gated": (_i, [_p, _i, _p, _i, _p, _l, _i, _i, _p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "punet_wgrad_workspace_floats": (_sz, [_i, _i, _i, _i]),
    "punet_wgrad_slices": (_i, [_i, _i, _i, _i, _i]),
    "punet_wgrad_x6_slices": (_i, [_i, _i, _i, _i, _i]),
    "punet_conv_wgrad": (_i, [_p, _i, _p, _i, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p]),
    "punet_conv_wgrad_x6": (_i, [_p, _i, _p, _i, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p]),
    "punet_conv3x3_small_fwd": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "punet_conv3x3_small_fwd_amax": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "punet_small_wgrad_workspace_floats": (_sz, [_i, _i]),
    "punet_conv3x3_small_wgrad": (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "punet_pool2x2_fwd": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "punet_pool2x2_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "punet_upsample2x_fwd": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "punet_upsample2x_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "punet_pool2x2_bwd_gated": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "punet_upsample2x_bwd_gated": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "punet_upconv_relayout": (_i, [_p, _i, _i, _i, _p, _p, _p]),
    "punet_upconv_gather_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "punet_upconv_gather_bwd": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "punet_upconv_wgrad_slices": (_i, [_i, _i, _i, _i, _i]),
    "punet_upconv_wgrad_workspace_floats": (_sz, [_i, _i, _i]),
    "punet_upconv_wgrad_x6": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _f, _p]),
    "punet_chan_sum_workspace_floats": (_sz, [_i, _i, _i]),
    "punet_chan_sum": (_i, [_p, _p, _p, _i, _i, _i, _f, _p]),
    "punet_latent_head_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "punet_latent_head_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "punet_reparam": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "punet_latent_kl": (_i, [_p, _p, _p, _i, _i, _p]),
    "punet_latent_kl_bwd": (_i, [_p, _p, _p, _p, _f, _p, _p, _i, _i, _p]),
    "punet_latent_kl_rows": (_i, [_p, _p, _p, _p, _p, _i, _i, _p]),
    "punet_latent_kl_bwd_ex": (_i, [_p, _p, _p, _p, _f, _i, _i, _p, _p, _i, _i, _p]),
    "punet_latent_split": (_i, [_p, _p, _p, _i, _i, _p]),
    "punet_concat_nhwc": (_i, [_p, _i, _p, _i, _p, _l, _p]),
    "punet_nchw_to_nhwc": (_i, [_p, _p, _i, _i, _l, _p]),
    "punet_nhwc_to_nchw": (_i, [_p, _p, _i, _i, _l, _p]),
    "punet_align_distribution": (_i, [_p, _p, _p, _p, _p, _l, _p]),
    "punet_fcomb_zbias": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "punet_fcomb_z_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "punet_dot_bias": (_i, [_p, _p, _p, _p, _l, _i, _p]),
    "punet_dot_bwd_workspace_floats": (_sz, [_i]),
    "punet_dot_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _i, _i, _p]),
    "punet_fcomb_sample": (_i, [_p, _p, _p, _p, _l, _p, _l, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _i, _p]),
    "punet_fcomb_sample_x6": (_i, [_p, _p, _p, _p, _l, _p, _l, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _i, _p]),
    "punet_fcomb_train_supported": (_i, [_i, _i, _i]),
    "punet_fcomb_train_workspace_floats": (_sz, [_i, _i, _i, _i]),
    "punet_fcomb_train_fwd_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "punet_fcomb_train_fwd_bwd_dice": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "punet_fcomb_train_reduce": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "punet_conv3x3_small_fwd_bf16s": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "punet_conv3x3_bf16s": (_i, [_p, _i, _p, _i, _p, _l, _i, _i, _p, _p, _i, _p, _p, _i, _i, _i, _i, _p]),
    "punet_conv_wgrad_bf16s": (_i, [_p, _i, _p, _i, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "punet_fcomb_train_fwd_bwd_bf16s": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "punet_gate_bits_bf16": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "punet_latent_head_bwd_bits": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "punet_upsample2x_fwd_bf16s": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "punet_latent_head_fwd_bf16s": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "punet_fcomb_sample_x6_bf16s": (_i, [_p, _p, _p, _p, _l, _p, _l, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _i, _p]),
    "punet_conv3x3_small_fwd_f16s": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "punet_conv3x3_f16s": (_i, [_p, _i, _p, _i, _p, _l, _i, _i, _p, _p, _i, _p, _p, _i, _i, _i, _i, _p, _p]),
    "punet_fcomb_train_fwd_bwd_f16s": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "punet_gate_bits_f16": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "punet_upsample2x_fwd_f16s": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "punet_latent_head_fwd_f16s": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "punet_fcomb_sample_x6_f16s": (_i, [_p, _p, _p, _p, _l, _p, _l, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _i, _p]),
    "punet_consensus": (_i, [_p, _i, _l, _f, _f, _i, _p, _p, _p]),
    "punet_loss_workspace_floats": (_sz, []),
    "punet_bce_fwd_bwd": (_i, [_p, _p, _p, _f, _p, _p, _p, _l, _p]),
    "punet_dice_fwd_bwd": (_i, [_p, _p, _p, _f, _p, _p, _p, _l, _p]),
    "punet_dice_score_sums": (_i, [_p, _p, _f, _i, _f, _i, _p, _p, _l, _p]),
    "punet_l2_reg": (_i, [_p, _p, _i, _i, _p, _p, _p, _p]),
    "punet_l2_reg_add_grad": (_i, [_p, _p, _p, _i, _p, _f, _p]),
    "punet_adam_ema": (_i, [_p, _p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _i, _f, _i, _p]),
    "punet_adam_hyper": (_i, [_p, _p, _p]),
    "punet_adam_ema_dev": (_i, [_p, _p, _p, _p, _p, _l, _p, _f, _f, _f, _p]),
    "punet_scale": (_i, [_p, _l, _f, _p]),
    "punet_standardize_workspace_doubles": (_sz, [_i]),
    "punet_standardize_stats": (_i, [_p, _p, _p, _i, _l, _p]),
    "punet_augment_view": (_i, [_p, _p, _p, _p, _i, _f, _f, _p, _p, _i, _i, _i, _i, _p]),
}



class PunetRtConfig(C.Structure):
    """include/punet_hip.h: PunetRtConfig"""
    _fields_ = [("input_channels", _i), ("num_classes", _i), ("levels", _i), ("latent_dim", _i), ("no_convs_fcomb", _i),
                ("num_filters", _i * 8)]


AFTER_UNET_CB = C.CFUNCTYPE(None, _p)  # void (*after_unet)(void* ctx)

_SIGS.update({
    "punet_rt_create": (_i, [C.POINTER(PunetRtConfig), C.POINTER(_p)]),
    "punet_rt_destroy": (None, [_p]),
    "punet_rt_set_arithmetic": (_i, [_p, _i]),
    "punet_rt_set_criterion": (_i, [_p, _i]),
    "punet_rt_set_upconv": (_i, [_p, _i]),
    "punet_rt_param_floats": (_l, [_p]),
    "punet_rt_unet_floats": (_l, [_p]),
    "punet_rt_num_entries": (_i, [_p]),
    "punet_rt_param_entry": (_i, [_p, _i, C.c_char_p, _i, C.POINTER(_l), C.POINTER(_l), C.POINTER(_i * 4)]),
    "punet_rt_tables_bytes": (_sz, [_p]),
    "punet_rt_init_tables": (_i, [_p, _p, _p]),
    "punet_rt_arena_bytes": (_sz, [_p, _i, _i, _i, _i, _i]),
    "punet_rt_last_arena_used": (_sz, [_p]),
    "punet_rt_pseudo_labels": (_i, [_p, _p, _p, _i, _i, _i, _p, _i, _f, _f, _i, _p, _p, _p, _p, _sz, _p]),
    "punet_rt_elbo_fwd_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _i, _p, _p, _p, AFTER_UNET_CB, _p, _p, _sz, _p]),
    "punet_rt_l2_reg_grad": (_i, [_p, _p, _p, _f, _p, _p, _sz, _p]),
    "punet_rt_supervised_step": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _f, _f, _f, _f, _f, _p, _p, _p, _sz, _p]),
    "punet_rt_joint_step": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _p, _p, _i, _i, _i, _p, _p, _i, _p, _f, _f, _i, _i, _f, _f,
                                 _f, _f, _f, _p, _p, _p, _p, _p, _sz, _p]),
    "punet_rt_self_training_step": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _p, _i, _p, _f, _f, _i, _i, _f, _f, _f, _f,
                                         _f, _p, _p, _p, _p, _p, _sz, _p]),
})

_NO_WRAP = {"punet_rt_destroy", "punet_rt_param_floats", "punet_rt_unet_floats", "punet_rt_num_entries", "punet_rt_tables_bytes",
            "punet_rt_arena_bytes", "punet_rt_last_arena_used", "punet_conv_splitk_scratch_bytes", "punet_upconv_eligible", "punet_upconv_min_cup_grad", "punet_last_error", "punet_build_info", "punet_device_ok", "punet_compute_units", "punet_wgrad_workspace_floats", "punet_gate_dwords",
            "punet_wgrad_slices", "punet_wgrad_x6_slices", "punet_upconv_wgrad_slices", "punet_upconv_wgrad_workspace_floats", "punet_small_wgrad_workspace_floats", "punet_chan_sum_workspace_floats",
            "punet_dot_bwd_workspace_floats", "punet_loss_workspace_floats", "punet_fcomb_train_supported",
            "punet_fcomb_train_workspace_floats", "punet_standardize_workspace_doubles"}

_lib = None
_fn = {}


def declared_symbols():
    return sorted(_SIGS)


def load():
    """dlopen the library and bind every symbol.  Raises RuntimeError when it is missing (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the product path.")
    import torch  # noqa: F401  - first: loads PyTorch-ROCm's bundled HIP runtime, which libpunet_hip.so links against
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        try:
            f = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"libpunet_hip.so does not export {name}") from e
        f.restype = res
        f.argtypes = args
        _fn[name] = f
    _lib = lib
    return lib


def last_error() -> str:
    load()
    return _fn["punet_last_error"]().decode()


def x6_opts(products: int = 6, shape: int = 16) -> int:
    """PUNET_X6_OPTS(products, shape) of punet_hip.h: the per-call options word of the split-plane kernels."""
    return int(products) | (int(shape) << 8)


def call(name, *args):
    """Call an entry point; int-returning ones raise RuntimeError(punet_last_error()) on failure."""
    if _lib is None:
        load()
    r = _fn[name](*args)
    if name not in _NO_WRAP and r != 0:
        raise RuntimeError(f"{name} failed ({r}): {last_error()}")
    return r
