import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probabilistic_domain_adaptation_amd import _lib as L
S = torch.cuda.current_stream().cuda_stream
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, W, cin) in [(4, 512, 512, 1), (4, 704, 704, 1), (4, 512, 512, 2), (4, 512, 512, 3), (4, 704, 704, 3), (4, 704, 704, 4)]:
    cout = 64
    x = torch.randn(B, H, W, cin, device="cuda"); w = torch.randn(3, 3, cin, cout, device="cuda"); b = torch.randn(cout, device="cuda")
    y = torch.empty(B, H, W, cout, device="cuda"); dz = torch.randn(B, H, W, cout, device="cuda")
    dw = torch.empty(3, 3, cin, cout, device="cuda"); db = torch.empty(cout, device="cuda")
    ws = torch.empty(L.call("punet_small_wgrad_workspace_floats", cin, cout), device="cuda")
    f = t(lambda: L.call("punet_conv3x3_small_fwd", x.data_ptr(), cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), cout, B, H, W, 1, S))
    g = t(lambda: L.call("punet_conv3x3_small_wgrad", x.data_ptr(), cin, dz.data_ptr(), cout, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, H, W, 0, 1.0, S))
    print(f"cin {cin} {H}x{W}: fwd {f:7.1f} us = {y.numel() * 4 / f / 1e6:5.2f} TB/s of stores; wgrad {g:7.1f} us = {dz.numel() * 4 / g / 1e6:5.2f} TB/s of dz reads")
