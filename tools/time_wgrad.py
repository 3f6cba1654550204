#!/usr/bin/env python3
"""Time punet_conv_wgrad_x6 on the step's layer shapes (post-ReLU-like inputs); last line = total ms of the step's 45 launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from probabilistic_domain_adaptation_amd import _lib as L

# (cs, cout, hw, launches per step) of the Mean-Teacher student backward at nf = [64, 128, 256, 512], B = 4, 512^2
SHAPES = [(64, 64, 512, 9), (64, 128, 256, 3), (128, 64, 512, 1), (128, 128, 256, 9), (128, 256, 128, 3), (256, 128, 256, 1),
          (256, 256, 128, 9), (256, 512, 64, 3), (512, 256, 128, 1), (512, 512, 64, 6)]
S = torch.cuda.current_stream().cuda_stream
B = 4
total, ref = 0.0, None
out = []
for cs, cout, hw, cnt in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.relu(torch.randn(B, hw, hw, cs, device="cuda", generator=g))
    dz = torch.randn(B, hw, hw, cout, device="cuda", generator=g) * (torch.rand(B, hw, hw, cout, device="cuda", generator=g) > 0.5)
    dw = torch.empty(3, 3, cs, cout, device="cuda")
    db = torch.empty(cout, device="cuda")
    ns = L.call("punet_wgrad_x6_slices", cs, cout, B, hw, hw)
    ws = torch.empty(L.call("punet_wgrad_workspace_floats", cs, cout, 3, ns), device="cuda")
    fn = lambda: L.call("punet_conv_wgrad_x6", x.data_ptr(), cs, dz.data_ptr(), cout, dw.data_ptr(), cs, 0, db.data_ptr(), ws.data_ptr(), ns,
                        B, hw, hw, 0, 1.0, 0, S)
    best = 1e9
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if r:
            best = min(best, e0.elapsed_time(e1))
    total += best * cnt
    out.append(f"{cs}->{cout}@{hw}:{2.0 * B * hw * hw * cs * cout * 9 / best / 1e9:.0f}")
    chk = float(dw.double().abs().sum())
print("TF/s " + " ".join(out))
print(f"wgrad total {total:.3f} ms/step (45 launches incl. slab reduction), checksum {chk:.6e}")
