"""Probe (one-off, GPU box): do timing events recorded INSIDE a captured hipGraph give per-replay kernel durations on this
ROCm?  Decides how bench.py times the roofline kernel when the step is replayed from a graph."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probabilistic_domain_adaptation_amd import _lib as L

L.load()
dev = torch.device("cuda:0")
x = torch.ones(64 << 20, device=dev)
S = lambda: torch.cuda.current_stream().cuda_stream
L.call("punet_scale", x.data_ptr(), x.numel(), 1.0, S())
torch.cuda.synchronize()
e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        e0.record()
        L.call("punet_scale", x.data_ptr(), x.numel(), 1.0, S())
        e1.record()
        L.call("punet_scale", x.data_ptr(), x.numel() // 4, 1.0, S())
        e2.record()
    for i in range(3):
        g.replay()
        torch.cuda.synchronize()
        print("replay", i, "elapsed ms:", e0.elapsed_time(e1), e1.elapsed_time(e2))
    print("EVENTS_IN_GRAPH_OK")
except Exception as ex:
    print("EVENTS_IN_GRAPH_FAILED", type(ex).__name__, ex)
# external timing for comparison
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); L.call("punet_scale", x.data_ptr(), x.numel(), 1.0, S()); b.record(); torch.cuda.synchronize()
print("eager elapsed ms:", a.elapsed_time(b))
