#!/usr/bin/env python3
"""How far is each conv arithmetic of this build from the EXACT result - and how far is the reference's own arithmetic?

For the layer shapes of the U-Net at nf=[64,128,256,512] (small spatial extent: the error of one output depends on K = 9 * cin,
not on the image size) the forward 3x3 convolution is evaluated
  * in fp64 on the CPU (the yardstick),
  * by PyTorch's CPU fp32 convolution (oneDNN) - the arithmetic of the reference's CPU path, i.e. of the oracle,
  * by the fp32-input MFMA kernel (`f32`), the six-product bf16 kernel (`x6`, the default), the three-product fp16 kernel
    (`f16x3`, what mixed_precision=True selects), `x3`, `f16` and `bf16`,
on post-ReLU-like activations (half zeros) with Kaiming-scaled weights.  Printed per shape: max |error| / max |exact| and the
RMS error / RMS exact.  The table answers "is x6 an fp32-accurate arithmetic" with a number: it is if its distance to the exact
result is no larger than the reference arithmetic's own.

  python tools/arithmetic_error_table.py            (needs the GPU; ~20 s)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_domain_adaptation_amd import _lib as L  # noqa: E402

SHAPES = [(64, 0, 64, 48, 64), (128, 0, 128, 32, 32), (256, 0, 256, 16, 32), (512, 0, 512, 8, 32), (512, 256, 256, 16, 32),
          (128, 64, 64, 48, 64)]


def S():
    return torch.cuda.current_stream().cuda_stream


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def errs(y, ref):
    d = (y.double() - ref)
    return d.abs().max().item() / ref.abs().max().item(), float(torch.sqrt((d * d).mean()) / torch.sqrt((ref * ref).mean()))


def main():
    L.load()
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(21)
    names = ["cpu fp32 (oneDNN)", "f32 MFMA", "x6", "f16x3", "x3", "f16", "bf16"]
    print(f"{'layer':>22} | " + " | ".join(f"{n:>19}" for n in names))
    print(f"{'':>22} | " + " | ".join(f"{'max / rms':>19}" for _ in names))
    worst = {n: 0.0 for n in names}
    for c0, c1, cout, H, W in SHAPES:
        B, cin = 2, c0 + c1
        x = torch.relu(torch.randn(B, cin, H, W, generator=g))  # post-ReLU-like: half zeros
        w = torch.randn(cout, cin, 3, 3, generator=g) * np.sqrt(2.0 / (cin * 9))
        b = 0.001 * torch.randn(cout, generator=g)
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        res = {"cpu fp32 (oneDNN)": F.conv2d(x, w, b, padding=1)}
        Wp = w.permute(2, 3, 1, 0).contiguous().cuda()
        nw = 9 * cin * cout
        X = nhwc(x)
        X0 = X[..., :c0].contiguous()
        X1 = X[..., c0:].contiguous() if c1 else None
        Bd = b.cuda()
        Y = torch.empty(B, H, W, cout, device="cuda")
        L.call("punet_conv_fwd", X0.data_ptr(), c0, X1.data_ptr() if c1 else None, c1, Wp.data_ptr(), Bd.data_ptr(), 0, Y.data_ptr(),
               cout, B, H, W, 3, 0, S())
        res["f32 MFMA"] = Y.permute(0, 3, 1, 2).cpu()
        for name, products in (("x6", 6), ("x3", 3), ("bf16", 1)):
            pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
            L.call("punet_split_weights_x6", Wp.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, S())
            Y = torch.empty(B, H, W, cout, device="cuda")
            L.call("punet_conv3x3_x6", X0.data_ptr(), c0, X1.data_ptr() if c1 else None, c1, pf.data_ptr(), nw, cout, 0, Bd.data_ptr(), 0,
                   Y.data_ptr(), cout, None, 0, 0, B, H, W, L.x6_opts(products, 16), S())
            res[name] = Y.permute(0, 3, 1, 2).cpu()
        for name, planes in (("f16x3", 2), ("f16", 1)):
            pf = torch.zeros(3 * nw, dtype=torch.bfloat16, device="cuda")
            aw, ay = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
            L.call("punet_split_weights_f16", Wp.data_ptr(), pf.data_ptr(), nw, cin, cout, 0, aw.data_ptr(), S())
            ax = torch.zeros(1, device="cuda")
            L.call("punet_amax", X.data_ptr(), X.numel(), ax.data_ptr(), 0, S())
            Y = torch.empty(B, H, W, cout, device="cuda")
            L.call("punet_conv3x3_f16", X0.data_ptr(), c0, X1.data_ptr() if c1 else None, c1, pf.data_ptr(), nw, cout, 0, Bd.data_ptr(), 0,
                   Y.data_ptr(), cout, None, None, None, None, 0, 0, B, H, W, planes, ax.data_ptr(), ax.data_ptr() if c1 else None,
                   aw.data_ptr(), ay.data_ptr(), S())
            res[name] = Y.permute(0, 3, 1, 2).cpu()
        torch.cuda.synchronize()
        cells = []
        for n in names:
            m, r = errs(res[n], ref)
            worst[n] = max(worst[n], m)
            cells.append(f"{m:9.2e} {r:9.2e}")
        print(f"{f'{c0}+{c1}->{cout} @{H}x{W} K={9 * cin}':>22} | " + " | ".join(cells))
    print(f"{'worst max':>22} | " + " | ".join(f"{worst[n]:19.2e}" for n in names))
    ratio = {n: worst[n] / worst["cpu fp32 (oneDNN)"] for n in names}
    print("worst max error relative to the CPU fp32 convolution's: " + ", ".join(f"{n} {ratio[n]:.2f}x" for n in names[1:]))


if __name__ == "__main__":
    main()
