#!/usr/bin/env python3
"""A/B in one process: fp32-MFMA conv vs bf16x6 conv (forward), per layer shape, TFLOP/s of algorithmic work."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from probabilistic_domain_adaptation_amd import _lib as L

SHAPES = [(64, 0, 64, 512), (128, 0, 128, 256), (256, 0, 256, 128), (512, 0, 512, 64), (64, 0, 128, 256), (128, 64, 64, 512),
          (512, 256, 256, 128)]
S = torch.cuda.current_stream().cuda_stream
B = 4
print("shape                 f32 TF   x6 TF   speedup   max|diff|/max")
for c0, c1, cout, hw in SHAPES:
    cin = c0 + c1
    x0 = torch.randn(B, hw, hw, c0, device="cuda")
    x1 = torch.randn(B, hw, hw, c1, device="cuda") if c1 else None
    w = torch.randn(3, 3, cin, cout, device="cuda") * 0.05
    b = torch.randn(cout, device="cuda")
    nw = 9 * cin * cout
    pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    L.call("punet_split_weights_x6", w.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, S)
    y1 = torch.empty(B, hw, hw, cout, device="cuda")
    y2 = torch.empty(B, hw, hw, cout, device="cuda")
    fl = 2.0 * B * hw * hw * cin * cout * 9

    def f32():
        L.call("punet_conv_fwd", x0.data_ptr(), c0, x1.data_ptr() if c1 else None, c1, w.data_ptr(), b.data_ptr(), 0, y1.data_ptr(),
               cout, B, hw, hw, 3, 1, S)

    def x6():
        L.call("punet_conv3x3_x6", x0.data_ptr(), c0, x1.data_ptr() if c1 else None, c1, pf.data_ptr(), nw, cout, 0, b.data_ptr(), 0,
               y2.data_ptr(), cout, None, 0, 1, B, hw, hw, 0, S)

    best = {"f": 1e9, "x": 1e9}
    for r in range(6):
        for k, fn in (("f", f32), ("x", x6)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            if r:
                best[k] = min(best[k], e0.elapsed_time(e1))
    d = (y1 - y2).abs().max().item() / y1.abs().max().item()
    print(f"{c0:4d}+{c1:<4d}->{cout:<4d}@{hw:<4d} {fl/best['f']/1e9:7.1f} {fl/best['x']/1e9:7.1f}   {best['f']/best['x']:5.2f}x   {d:.2e}")
