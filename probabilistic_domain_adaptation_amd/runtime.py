"""ctypes face of the step-level native runtime (csrc/runtime.hip, include/punet_hip.h "Step-level runtime").

The graph walk of the Probabilistic U-Net - forward / sample / elbo and the whole backward - runs as native host code inside
libpunet_hip.so; this module only owns the handle, the two device buffers the runtime asks for (index tables, arena) and turns
tensors into pointers.  It is the "thin shim" form of the hot path: four library calls per self-training step.  The Python
engine (engine.py) covers the rest of the constructor space and enqueues the same kernels in the same order, so both give the
same bits (tests/test_gpu_runtime.py)."""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Optional

import torch

from . import _lib


def supported(layout) -> bool:
    """Does the native runtime cover this ParamLayout?  Mirrors EVERY precondition of punet_rt_create (csrc/runtime.hip), so that
    a configuration the runtime refuses takes the Python engine instead of raising: num_classes 1, num_filters[0] in
    {32, 64}, filters multiples of 8, Fcomb of 2 or 3 layers, 2..8 levels, latent_dim 1..64, input_channels + 1 <= 7."""
    return (layout.kind == "punet" and layout.num_classes == 1 and layout.nf[0] in (32, 64) and layout.no_convs_fcomb in (2, 3)
            and 2 <= len(layout.nf) <= 8 and all(f >= 8 and f % 8 == 0 for f in layout.nf)
            and 1 <= layout.latent_dim <= 64 and 1 <= layout.input_channels and layout.input_channels + 1 <= 7)


# While a hipGraph is being captured (my_trainer/fused.py::GraphedStep) every arena a call carves from is appended here: the graph
# bakes raw pointers into it, so the GraphedStep keeps these tensors alive for exactly as long as the graph itself lives.  A
# runtime whose arena has to grow afterwards simply drops its own reference: an arena no graph points into is freed at once, one
# that a graph points into goes away with that graph (the trainers' LRU evicts graphs) - nothing is retired for the life of the
# runtime any more (arenas are tens of GB).
CAPTURE_KEEP: Optional[list] = None


def _keep_for_capture(t: torch.Tensor):
    if CAPTURE_KEEP is not None and not any(t is k for k in CAPTURE_KEEP):
        CAPTURE_KEEP.append(t)


class NativeRuntime:
    _live = weakref.WeakSet()

    ARITHMETICS = {"x6": 6, "f16x3": 3, "f16": 1}  # conv_mode -> punet_rt_set_arithmetic

    def __init__(self, layout, device, conv_mode: str = "x6", dice: bool = False):
        _lib.load()
        if not supported(layout):
            raise ValueError("the native runtime covers num_classes = 1, num_filters[0] in {32, 64}, no_convs_fcomb in {2, 3}")
        self.layout, self.device = layout, torch.device(device)
        cfg = _lib.PunetRtConfig(layout.input_channels, layout.num_classes, len(layout.nf), layout.latent_dim, layout.no_convs_fcomb,
                                 (C.c_int * 8)(*(list(layout.nf) + [0] * (8 - len(layout.nf)))))
        h = C.c_void_p()
        _lib.call("punet_rt_create", C.byref(cfg), C.byref(h))
        self._h = h
        if conv_mode not in self.ARITHMETICS:
            raise ValueError(f"the native runtime computes in {sorted(self.ARITHMETICS)}, not '{conv_mode}'")
        self.conv_mode = conv_mode
        _lib.call("punet_rt_set_arithmetic", h, self.ARITHMETICS[conv_mode])
        self.dice = bool(dice)
        _lib.call("punet_rt_set_criterion", h, int(self.dice))
        # K5x (csrc/upconv.hip): the same switch the Python engine reads, so that the two graph walkers keep agreeing bit for bit
        _lib.call("punet_rt_set_upconv", h, int(os.environ.get("PUNET_UPCONV", "1") != "0"))
        if _lib.call("punet_rt_param_floats", h) != layout.numel:
            raise RuntimeError("native runtime and ParamLayout disagree about the flat parameter layout")
        self.tables = torch.empty(_lib.call("punet_rt_tables_bytes", h), dtype=torch.uint8, device=self.device)
        _lib.call("punet_rt_init_tables", h, self.tables.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.current_stream().synchronize()  # the tables come from host memory of the handle: copied before it can change
        self._arena: Optional[torch.Tensor] = None
        self._cb_keep = None
        NativeRuntime._live.add(self)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.call("punet_rt_destroy", h)
            except Exception:
                pass
            self._h = None

    # ------------------------------------------------------------------ layout (host only)
    def entries(self):
        """[(state-dict name, offset, numel, logical shape)] as the runtime lays the flat buffer out."""
        out = []
        for i in range(_lib.call("punet_rt_num_entries", self._h)):
            name = C.create_string_buffer(128)
            off, n, shape = C.c_long(), C.c_long(), (C.c_int * 4)()
            _lib.call("punet_rt_param_entry", self._h, i, name, 128, C.byref(off), C.byref(n), C.byref(shape))
            shp = tuple(shape) if shape[1] else (shape[0],)
            out.append((name.value.decode(), off.value, n.value, shp))
        return out

    @property
    def unet_numel(self) -> int:
        return _lib.call("punet_rt_unet_floats", self._h)

    # ------------------------------------------------------------------ device calls
    def arena_bytes(self, B, H, W, N=1, training=True) -> int:
        return _lib.call("punet_rt_arena_bytes", self._h, B, H, W, N, int(training))

    def arena(self, nbytes: int) -> torch.Tensor:
        """Grow-only arena (one per runtime: calls on one handle are sequential on one stream).  A captured graph that points
        into the current one holds its own reference (CAPTURE_KEEP above), so growing never pulls memory from under a graph and
        never keeps an arena that no graph uses."""
        if self._arena is None or self._arena.numel() < nbytes:
            self._arena = None  # (drop ours first: arenas are tens of GB)
            self._arena = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        _keep_for_capture(self._arena)
        return self._arena

    def pseudo_labels(self, flat, x_nhwc, eps, upper=0.9, lower=0.1, masking=False, want_logits=False, arena=None):
        """teacher.forward(x1, None, False) + N x sigmoid(sample()) + consensus -> (y, z[, logits])."""
        B, H, W, _ = x_nhwc.shape
        N = eps.shape[0]
        ar = arena if arena is not None else self.arena(self.arena_bytes(B, H, W, N, False))
        _keep_for_capture(ar)
        y = torch.empty(B, 1, H, W, dtype=torch.float32, device=self.device)
        z = torch.empty(B, 1, H, W, dtype=torch.float32, device=self.device)
        lg = torch.empty(N, B, 1, H, W, dtype=torch.float32, device=self.device) if want_logits else None
        _lib.call("punet_rt_pseudo_labels", self._h, flat.data_ptr(), x_nhwc.data_ptr(), B, H, W, eps.data_ptr(), N, float(upper),
                  float(lower), int(masking), y.data_ptr(), z.data_ptr(), lg.data_ptr() if lg is not None else None, ar.data_ptr(),
                  ar.numel(), torch.cuda.current_stream().cuda_stream)
        return (y, z, lg) if want_logits else (y, z)

    def elbo_fwd_bwd(self, flat, gflat, x_nhwc, target, consm, eps, beta=1.0, gscale=1.0, accumulate=False, after_unet=None,
                     want_logits=False, arena=None):
        """model.forward(x, y, True); elbo(y, consm); backward into gflat -> (rec [1], kl [1][, logits])."""
        B, H, W, _ = x_nhwc.shape
        ar = arena if arena is not None else self.arena(self.arena_bytes(B, H, W, 1, True))
        _keep_for_capture(ar)
        rec = torch.empty(1, dtype=torch.float32, device=self.device)
        kl = torch.empty(1, dtype=torch.float32, device=self.device)
        lg = torch.empty(B, 1, H, W, dtype=torch.float32, device=self.device) if want_logits else None
        cb = _lib.AFTER_UNET_CB(lambda ctx: after_unet()) if after_unet is not None else C.cast(None, _lib.AFTER_UNET_CB)
        self._cb_keep = cb
        _lib.call("punet_rt_elbo_fwd_bwd", self._h, flat.data_ptr(), gflat.data_ptr(), x_nhwc.data_ptr(), target.data_ptr(),
                  consm.data_ptr() if consm is not None else None, eps.data_ptr(), B, H, W, float(beta), float(gscale), int(accumulate),
                  rec.data_ptr(), kl.data_ptr(), lg.data_ptr() if lg is not None else None, cb, None, ar.data_ptr(), ar.numel(),
                  torch.cuda.current_stream().cuda_stream)
        return (rec, kl, lg) if want_logits else (rec, kl)

    def l2_reg_grad(self, flat, gflat, coef: float):
        """reg = l2(posterior) + l2(prior) + l2(fcomb.layers); gflat += coef * d reg -> reg [1]."""
        reg = torch.empty(1, dtype=torch.float32, device=self.device)
        ar = self.arena(1 << 20)
        _lib.call("punet_rt_l2_reg_grad", self._h, flat.data_ptr(), gflat.data_ptr() if gflat is not None else None, float(coef),
                  reg.data_ptr(), ar.data_ptr(), ar.numel(), torch.cuda.current_stream().cuda_stream)
        return reg
