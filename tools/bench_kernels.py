#!/usr/bin/env python3
"""Per-layer-shape micro-benchmark of the three MFMA kernels (forward, dgrad, wgrad) at the bench configuration
(batch 4, 512x512 input, nf=[64,128,256,512]).  HIP events on the launch stream, interleaved rounds in one process.
Usage: python tools/bench_kernels.py [--rounds 5] [--only fwd,dgrad,wgrad]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from probabilistic_domain_adaptation_amd import _lib as L

SHAPES = [  # (c0, c1, cout, HW, count per MT step fwd / bwd)
    (64, 0, 64, 512, 12, 8), (64, 0, 128, 256, 5, 3), (128, 0, 128, 256, 12, 8), (128, 0, 256, 128, 5, 3),
    (256, 0, 256, 128, 12, 8), (256, 0, 512, 64, 5, 3), (512, 0, 512, 64, 10, 6), (512, 256, 256, 128, 2, 1),
    (256, 128, 128, 256, 2, 1), (128, 64, 64, 512, 2, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--only", default="fwd,dgrad,wgrad")
    ap.add_argument("--batch", type=int, default=4)
    a = ap.parse_args()
    only = a.only.split(",")
    dev = torch.device("cuda:0")
    S = torch.cuda.current_stream().cuda_stream
    B = a.batch
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    print(f"{'shape':28s} {'fwd TF':>8s} {'dgrad TF':>9s} {'wgrad TF':>9s}   (ms: fwd dgrad wgrad)")
    for c0, c1, cout, hw, nf_, nb_ in SHAPES:
        cin = c0 + c1
        x0 = torch.randn(B, hw, hw, c0, device=dev)
        x1 = torch.randn(B, hw, hw, c1, device=dev) if c1 else None
        w = torch.randn(3, 3, cin, cout, device=dev) * 0.05
        b = torch.randn(cout, device=dev)
        y = torch.empty(B, hw, hw, cout, device=dev)
        dz = torch.randn(B, hw, hw, cout, device=dev)
        dx = torch.empty(B, hw, hw, c0, device=dev)
        dw = torch.empty(3, 3, cin, cout, device=dev)
        db = torch.empty(cout, device=dev)
        ns = L.call("punet_wgrad_slices", c0, cout, B, hw, hw)
        ws = torch.empty(L.call("punet_wgrad_workspace_floats", c0, cout, 3, ns), device=dev)
        fl_f = 2.0 * B * hw * hw * cin * cout * 9
        fl_d = 2.0 * B * hw * hw * c0 * cout * 9
        fl_w = fl_d

        def fwd():
            L.call("punet_conv_fwd", x0.data_ptr(), c0, x1.data_ptr() if c1 else None, c1, w.data_ptr(), b.data_ptr(), 0,
                   y.data_ptr(), cout, B, hw, hw, 3, 1, S)

        wt = torch.empty_like(w)
        L.call("punet_flip_transpose_weights", w.data_ptr(), wt.data_ptr(), cin, cout, 3, S)

        def dgrad():
            L.call("punet_conv_dgrad_t", dz.data_ptr(), cout, wt.data_ptr(), cin, 0, c0, dx.data_ptr(), x0.data_ptr(), 0, B, hw,
                   hw, 3, S)

        def wgrad():
            L.call("punet_conv_wgrad", x0.data_ptr(), c0, dz.data_ptr(), cout, dw.data_ptr(), cin, 0, db.data_ptr(),
                   ws.data_ptr(), ns, B, hw, hw, 3, 0, 1.0, S)

        fns = {"fwd": (fwd, fl_f, nf_), "dgrad": (dgrad, fl_d, nb_), "wgrad": (wgrad, fl_w, nb_)}
        res = {}
        for k in only:
            fns[k][0]()
        torch.cuda.synchronize()
        best = {k: 1e9 for k in only}
        for _ in range(a.rounds):
            for k in only:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fns[k][0]()
                e1.record()
                torch.cuda.synchronize()
                best[k] = min(best[k], e0.elapsed_time(e1))
        for k in only:
            res[k] = fns[k][1] / best[k] / 1e9
            tot[k][0] += fns[k][1] * fns[k][2]
            tot[k][1] += best[k] * fns[k][2]
        name = f"{c0}+{c1}->{cout} @{hw}"
        print(f"{name:28s} " + " ".join(f"{res.get(k, 0):9.1f}" for k in ("fwd", "dgrad", "wgrad")) + "   " +
              " ".join(f"{best.get(k, 0):7.3f}" for k in ("fwd", "dgrad", "wgrad")))
    print("step-weighted: " + ", ".join(f"{k} {tot[k][0] / tot[k][1] / 1e9:.1f} TF ({tot[k][1]:.1f} ms/step)"
                                        for k in only if tot[k][1] > 0))


if __name__ == "__main__":
    main()
