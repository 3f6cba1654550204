#!/usr/bin/env python3
"""A/B in ONE process (interleaved rounds, same device): persistent conv grid (k workgroups per CU) vs one workgroup per
tile, forward and dgrad(+gate), per layer shape.  Usage: python tools/ab_conv_grid.py [--modes 0,2,3,4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from probabilistic_domain_adaptation_amd import _lib as L

SHAPES = [(64, 64, 512), (128, 128, 256), (256, 256, 128), (512, 512, 64), (64, 128, 256)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="0,2,3,4")
    ap.add_argument("--rounds", type=int, default=6)
    a = ap.parse_args()
    modes = [int(m) for m in a.modes.split(",")]
    dev = torch.device("cuda:0")
    S = torch.cuda.current_stream().cuda_stream
    B = 4
    print("shape              " + "  ".join(f"fwd@{m:<2d} dgr@{m:<2d}" for m in modes) + "   (TFLOP/s, best of rounds)")
    for cin, cout, hw in SHAPES:
        x = torch.randn(B, hw, hw, cin, device=dev)
        w = torch.randn(3, 3, cin, cout, device=dev) * 0.05
        wt = torch.empty_like(w)
        L.call("punet_flip_transpose_weights", w.data_ptr(), wt.data_ptr(), cin, cout, 3, S)
        b = torch.randn(cout, device=dev)
        y = torch.empty(B, hw, hw, cout, device=dev)
        dz = torch.randn(B, hw, hw, cout, device=dev)
        dx = torch.empty(B, hw, hw, cin, device=dev)
        fl = 2.0 * B * hw * hw * cin * cout * 9

        def fwd():
            L.call("punet_conv_fwd", x.data_ptr(), cin, None, 0, w.data_ptr(), b.data_ptr(), 0, y.data_ptr(), cout, B, hw, hw, 3, 1, S)

        def dgr():
            L.call("punet_conv_dgrad_t", dz.data_ptr(), cout, wt.data_ptr(), cin, 0, cin, dx.data_ptr(), x.data_ptr(), 0, B, hw, hw,
                   3, S)

        best = {(m, k): 1e9 for m in modes for k in ("f", "d")}
        for r in range(a.rounds + 1):
            for m in modes:
                L.call("punet_set_conv_wg_per_cu", m)
                for k, fn in (("f", fwd), ("d", dgr)):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn()
                    e1.record()
                    torch.cuda.synchronize()
                    if r > 0:
                        best[(m, k)] = min(best[(m, k)], e0.elapsed_time(e1))
        print(f"{cin:4d}->{cout:<4d}@{hw:<4d}    " + "  ".join(f"{fl / best[(m, 'f')] / 1e9:6.1f} {fl / best[(m, 'd')] / 1e9:6.1f}" for m in modes))
    L.call("punet_set_conv_wg_per_cu", 3)


if __name__ == "__main__":
    main()
