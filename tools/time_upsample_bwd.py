#!/usr/bin/env python3
"""Time punet_upsample2x_bwd on the three up-block shapes of the bench step (B = 4); last line = total us and a checksum."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from probabilistic_domain_adaptation_amd import _lib as L

S = torch.cuda.current_stream().cuda_stream
tot, chk, out = 0.0, 0.0, []
for H, C in ((64, 512), (128, 256), (256, 128)):
    g = torch.Generator(device="cuda").manual_seed(1)
    dy = torch.randn(4, 2 * H, 2 * H, C, device="cuda", generator=g)
    dx = torch.empty(4, H, H, C, device="cuda")
    fn = lambda: L.call("punet_upsample2x_bwd", dy.data_ptr(), None, dx.data_ptr(), 4, H, H, C, S)
    best = 1e9
    for r in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if r:
            best = min(best, e0.elapsed_time(e1) * 1e3)
    tot += best
    chk += float(dx.double().abs().sum())
    out.append(f"{H}x{H}x{C}: {best:.1f} us ({(dy.numel() + dx.numel()) * 4 / best / 1e6:.2f} TB/s)")
print(" | ".join(out))
print(f"upsample2x_bwd total {tot:.1f} us, checksum {chk:.6e}")
