#!/usr/bin/env python3
"""Wave timeline of conv1x1_x6w_kernel (the 1x1 x6 GEMM of K5x): s_memtime stamps of one workgroup's phases 8..11.
Build the trace library first (build container):  tools/ab_variants.sh conv_x6.hip x6wtrace "-DX6W_TRACE"
then on the GPU box:  PUNET_HIP_LIB=$PWD/probabilistic_domain_adaptation_amd/libpunet_hip_x6wtrace.so python tools/trace_x6w.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda:0")
buf = torch.zeros(4 * 4 * 8, dtype=torch.int64, device=dev)
os.environ["PUNET_X6W_TRACE_PTR"] = hex(buf.data_ptr())
from probabilistic_domain_adaptation_amd import _lib as L  # noqa: E402

L.load()
S = torch.cuda.current_stream().cuda_stream
B, h, cup, cout = 4, 64, 512, 256
n9 = 9 * cout
x = torch.relu(torch.randn(B, h, h, cup, device=dev))
w1 = torch.randn(cup, n9, device=dev) * 0.02
p1 = torch.empty(3 * cup * n9, dtype=torch.bfloat16, device=dev)
L.call("punet_split_weights_x6", w1.data_ptr(), p1.data_ptr(), cup * n9, cup, n9, 1, 0, S)
z = torch.empty(B, h, h, n9, device=dev)
for _ in range(3):
    L.call("punet_conv1x1_x6_gated", x.data_ptr(), cup, None, 0, p1.data_ptr(), cup * n9, n9, 0, None, 0, z.data_ptr(), n9, None, None, None,
           0, 0, B, h, h, S)
torch.cuda.synchronize()
t = buf.cpu().view(4, 4, 8)
t0 = int(t[0, 0, 0])
names = ["top", "barrier1", "commit", "opbarrier", "kstep0", "splits"]
print("stamps in shader cycles (s_memtime) relative to wave 0's first stamp; workgroup 9, phases 8..11")
for wv in range(4):
    for ph in range(4):
        print(f"wave {wv} phase {8 + ph}: " + "  ".join(f"{names[i]} {int(t[wv, ph, i]) - t0:7d}" for i in range(6)))
