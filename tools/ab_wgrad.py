#!/usr/bin/env python3
"""A/B in one process: fp32-MFMA wgrad vs bf16x6 wgrad per layer shape (TFLOP/s of algorithmic work)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from probabilistic_domain_adaptation_amd import _lib as L

SHAPES = [(64, 64, 512), (128, 128, 256), (256, 256, 128), (512, 512, 64), (64, 128, 256), (512, 256, 128)]
S = torch.cuda.current_stream().cuda_stream
B = 4
print("shape               f32 TF   x6 TF  speedup  max|diff|/max")
for cin, cout, hw in SHAPES:
    x = torch.randn(B, hw, hw, cin, device="cuda")
    dz = torch.randn(B, hw, hw, cout, device="cuda")
    d1 = torch.empty(3, 3, cin, cout, device="cuda")
    d2 = torch.empty(3, 3, cin, cout, device="cuda")
    b1 = torch.empty(cout, device="cuda")
    ns = L.call("punet_wgrad_slices", cin, cout, B, hw, hw)
    ws = torch.empty(L.call("punet_wgrad_workspace_floats", cin, cout, 3, ns), device="cuda")
    fl = 2.0 * B * hw * hw * cin * cout * 9
    f32 = lambda: L.call("punet_conv_wgrad", x.data_ptr(), cin, dz.data_ptr(), cout, d1.data_ptr(), cin, 0, b1.data_ptr(), ws.data_ptr(),
                         ns, B, hw, hw, 3, 0, 1.0, S)
    x6 = lambda: L.call("punet_conv_wgrad_x6", x.data_ptr(), cin, dz.data_ptr(), cout, d2.data_ptr(), cin, 0, b1.data_ptr(),
                        ws.data_ptr(), ns, B, hw, hw, 0, 1.0, 0, S)
    best = {"f": 1e9, "x": 1e9}
    for r in range(6):
        for k, fn in (("f", f32), ("x", x6)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            if r:
                best[k] = min(best[k], e0.elapsed_time(e1))
    d = (d1 - d2).abs().max().item() / d1.abs().max().item()
    print(f"{cin:4d}->{cout:<4d}@{hw:<4d}   {fl/best['f']/1e9:7.1f} {fl/best['x']/1e9:7.1f}   {best['f']/best['x']:5.2f}x   {d:.2e}")
