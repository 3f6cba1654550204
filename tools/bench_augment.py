"""Time K13 (view generation) at the training batch shape and report it against its algorithmic bytes."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from probabilistic_domain_adaptation_amd.my_utils import augment

B, C, H, W = 4, 1, 512, 512
x = torch.randn(B, C, H, W, device="cuda") * 30 + 100
noise = torch.randn_like(x)
res = {}
for name, ks in (("no blur", 1), ("blur 11", 11), ("blur 23", 23), ("blur 49", 49)):
    p = np.tile(np.array([ks, 3.0, 0.2, 1.5], np.float32), (B, 1))
    for _ in range(3):
        augment.apply_view(x, p, noise)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 50
    for _ in range(n):
        augment.apply_view(x, p, noise)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    # stats: 4 B read; h pass: 4 read + 4 written; v pass: 4 read + 4 noise + 4 written
    byts = x.numel() * 24
    res[name] = {"ms": round(ms, 4), "algorithmic_GBps": round(byts / ms * 1e-6, 1)}
print(json.dumps({"shape": [B, C, H, W], "results": res}))
