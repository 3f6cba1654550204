#!/usr/bin/env python3
"""Time K7 (fp32 MFMA) against its split-plane twin on the bench shape (B=4, 512x512, CH=64, one middle layer, N=8) and on the
inference shape (1024x1024, N=16); max |logit difference| between the two arithmetics."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from probabilistic_domain_adaptation_amd import _lib as L

S = torch.cuda.current_stream().cuda_stream
for (B, H, W, N) in ((4, 512, 512, 8), (1, 1024, 1024, 16)):
    CH, Ld, nmid = 64, 6, 1
    g = torch.Generator(device="cuda").manual_seed(1)
    feat = torch.relu(torch.randn(B, H, W, CH, device="cuda", generator=g))
    w1 = torch.randn(CH + Ld, CH, device="cuda", generator=g) / 8
    wm = torch.randn(CH, CH, device="cuda", generator=g) / 8
    bm = torch.randn(CH, device="cuda", generator=g) * 0.1
    w3 = torch.randn(CH, device="cuda", generator=g) / 4
    b3 = torch.zeros(1, device="cuda")
    zb = torch.randn(N, B, CH, device="cuda", generator=g) * 0.1
    out = {}
    for name in ("punet_fcomb_sample", "punet_fcomb_sample_x6"):
        y, z = torch.empty(B, 1, H, W, device="cuda"), torch.empty(B, 1, H, W, device="cuda")
        lg = torch.empty(N, B, 1, H, W, device="cuda")
        fn = lambda: L.call(name, feat.data_ptr(), w1.data_ptr(), zb.data_ptr(), wm.data_ptr(), 0, bm.data_ptr(), 0, w3.data_ptr(), b3.data_ptr(),
                            y.data_ptr(), z.data_ptr(), lg.data_ptr(), N, B, H * W, CH, nmid, 0.9, 0.1, 1, S)
        best = 1e9
        for r in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            if r:
                best = min(best, e0.elapsed_time(e1))
        out[name] = (best, lg.clone(), z.clone())
    d = (out["punet_fcomb_sample"][1] - out["punet_fcomb_sample_x6"][1]).abs().max().item()
    mm = int((out["punet_fcomb_sample"][2] != out["punet_fcomb_sample_x6"][2]).sum())
    print(f"B={B} {H}x{W} N={N}: fp32-MFMA {out['punet_fcomb_sample'][0]:.3f} ms, x6 {out['punet_fcomb_sample_x6'][0]:.3f} ms, "
          f"max |dlogit| {d:.2e}, mask pixels differing {mm}")
