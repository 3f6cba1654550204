"""Plain `Unet` with the reference's signature (prob_utils/my_models/unet.py:6-69) on the HIP kernels (forward only -
the reference trains its plain U-Nets with the third-party torch_em.model.UNet2d, which is out of scope)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib
from ..engine import Act, ParamLayout, PUNetEngine, _stream
from .probabilistic_unet import _attach


class Unet(nn.Module):
    def __init__(self, input_channels, num_classes, num_filters, initializers=None, apply_last_layer=True, padding=True,
                 device=None, seed=None):
        super().__init__()
        if not padding:
            raise NotImplementedError("padding=False is never used by the reference")
        self.input_channels, self.num_classes, self.num_filters = input_channels, num_classes, list(num_filters)
        self.padding, self.apply_last_layer = padding, apply_last_layer
        self.activation_maps = []
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device()) \
            if torch.cuda.is_available() else None
        if dev is None or dev.type != "cuda":
            raise RuntimeError("Unet (MI355X build) needs a ROCm GPU: there is no CPU fallback")
        self.layout = ParamLayout(input_channels, num_classes, self.num_filters, kind="unet", apply_last_layer=apply_last_layer)
        self.engine = PUNetEngine(self.layout, dev)
        self._flat = torch.zeros(self.layout.numel, dtype=torch.float32, device=dev)
        g = torch.Generator().manual_seed(seed if seed is not None else torch.seed() % (2 ** 31))
        for name, v in self.layout.views(self._flat).items():
            with torch.no_grad():
                t = torch.empty(tuple(v.shape))
                if name.endswith(".weight"):
                    nn.init.kaiming_normal_(t, mode="fan_in", nonlinearity="relu", generator=g)  # utils.py:16-21
                else:
                    nn.init.trunc_normal_(t, std=0.001, a=-0.002, b=0.002, generator=g)
                v.copy_(t)
            _attach(self, name, nn.Parameter(v, requires_grad=False))

    @property
    def device(self):
        return self._flat.device

    @torch.no_grad()
    def forward(self, x, val=False):
        """unet.py:50-69 -> [B, num_classes, H, W] logits (or the [B, nf[0], H, W] feature map without last layer)."""
        eng = self.engine
        xn = eng.to_nhwc(x)
        B, H, W, C = xn.shape
        eng.prepare_forward_weights(self._flat)
        feat = eng.unet(self._flat, Act(xn, B, H, W, C, needs_grad=False))
        if val:
            self.activation_maps.append(feat.t.permute(0, 3, 1, 2))
        if not self.apply_last_layer:
            return feat.t.permute(0, 3, 1, 2)
        c = self.layout.convs["last_layer"]
        K, CH = c.cout, c.cin
        if K == 1:
            out = torch.empty(B, 1, H, W, dtype=torch.float32, device=self.device)
            _lib.call("punet_dot_bias", feat.ptr, self._flat.data_ptr() + 4 * c.w_off, self._flat.data_ptr() + 4 * c.b_off,
                      out.data_ptr(), B * H * W, feat.C, _stream())
            return out
        # num_classes > 1 (no reference script): one dot product per class on its contiguous weight row (stored [CH][K])
        w = self._flat[c.w_off:c.w_off + CH * K].view(CH, K).t().contiguous()
        out = torch.empty(K, B, H, W, dtype=torch.float32, device=self.device)
        for k in range(K):
            _lib.call("punet_dot_bias", feat.ptr, w[k].data_ptr(), self._flat.data_ptr() + 4 * (c.b_off + k), out[k].data_ptr(),
                      B * H * W, feat.C, _stream())
        return out.permute(1, 0, 2, 3).contiguous()
