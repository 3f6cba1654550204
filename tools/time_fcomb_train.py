#!/usr/bin/env python3
"""Time K7t (punet_fcomb_train_fwd_bwd + reduce) on the bench shape: B=4, 512x512, CH=64, one middle layer."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from probabilistic_domain_adaptation_amd import _lib as L

S = torch.cuda.current_stream().cuda_stream
B, H, W, CH, Ld, nmid = 4, 512, 512, 64, 6, 1
g = torch.Generator(device="cuda").manual_seed(1)
feat = torch.relu(torch.randn(B, H, W, CH, device="cuda", generator=g))
w1 = torch.randn(CH + Ld, CH, device="cuda", generator=g) / 8
wm = torch.randn(CH, CH, device="cuda", generator=g) / 8
bm = torch.randn(CH, device="cuda", generator=g) * 0.1
w3 = torch.randn(CH, device="cuda", generator=g) / 4
b3 = torch.zeros(1, device="cuda")
zb = torch.randn(B, CH, device="cuda", generator=g) * 0.1
y = (torch.rand(B, 1, H, W, device="cuda", generator=g) > 0.5).float()
cm = (torch.rand(B, 1, H, W, device="cuda", generator=g) > 0.3).float()
logits, dfeat, rec = torch.empty(B, 1, H, W, device="cuda"), torch.empty(B, H, W, CH, device="cuda"), torch.empty(1, device="cuda")
ws = torch.empty(L.call("punet_fcomb_train_workspace_floats", B, H * W, CH, nmid), device="cuda")
fn = lambda: L.call("punet_fcomb_train_fwd_bwd", feat.data_ptr(), w1.data_ptr(), zb.data_ptr(), wm.data_ptr(), bm.data_ptr(), w3.data_ptr(),
                    b3.data_ptr(), y.data_ptr(), cm.data_ptr(), 1.0, logits.data_ptr(), dfeat.data_ptr(), rec.data_ptr(), ws.data_ptr(), B,
                    H * W, CH, nmid, S)
best = 1e9
for r in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    if r:
        best = min(best, e0.elapsed_time(e1))
fl = 2.0 * B * H * W * CH * CH * 6
print(f"fcomb_train {best:.3f} ms ({fl / best / 1e9:.1f} TFLOP/s of fp32 MFMA work; MFMA-bound floor {fl / 157.3e9:.3f} ms), rec {rec.item():.6e} "
      f"dfeat |sum| {dfeat.double().abs().sum().item():.6e}")
