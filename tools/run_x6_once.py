#!/usr/bin/env python3
"""Launch the x6 forward conv a few times on one shape (for rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probabilistic_domain_adaptation_amd import _lib as L
cin, cout, hw, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 4
S = torch.cuda.current_stream().cuda_stream
x = torch.randn(B, hw, hw, cin, device="cuda"); w = torch.randn(3, 3, cin, cout, device="cuda") * 0.05
b = torch.randn(cout, device="cuda"); y = torch.empty(B, hw, hw, cout, device="cuda")
nw = 9 * cin * cout
pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
L.call("punet_split_weights_x6", w.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, S)
for _ in range(3):
    L.call("punet_conv3x3_x6", x.data_ptr(), cin, None, 0, pf.data_ptr(), nw, cout, 0, b.data_ptr(), 0, y.data_ptr(), cout, None, 0, 1, B, hw, hw, 0, S)
    L.call("punet_conv_fwd", x.data_ptr(), cin, None, 0, w.data_ptr(), b.data_ptr(), 0, y.data_ptr(), cout, B, hw, hw, 3, 1, S)
torch.cuda.synchronize()
