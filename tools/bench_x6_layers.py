#!/usr/bin/env python3
"""Per-layer-shape timing of the three x6 kernels (forward, input gradient, weight gradient) at the bench configuration
(batch 4, 512x512 input, nf=[64,128,256,512]); HIP events on the launch stream, best of --rounds, one process.
The last columns weight every shape by how often a mean-teacher step launches it."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from probabilistic_domain_adaptation_amd import _lib as L

SHAPES = [  # (c0, c1, cout, HW, launches per MT step: fwd, bwd)
    (64, 0, 64, 512, 12, 8), (64, 0, 128, 256, 5, 3), (128, 0, 128, 256, 12, 8), (128, 0, 256, 128, 5, 3),
    (256, 0, 256, 128, 12, 8), (256, 0, 512, 64, 5, 3), (512, 0, 512, 64, 10, 6), (512, 256, 256, 128, 2, 1),
    (256, 128, 128, 256, 2, 1), (128, 64, 64, 512, 2, 1),
]


def timeit(fn, rounds):
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--slices-div", type=int, default=1, help="divide the split-K slice count of the weight gradient")
    ap.add_argument("--bits", type=int, default=1, help="1: input gradient gated by bits; 0: by the fp32 activation")
    ap.add_argument("--plane-in", action="store_true",
                    help="operands of the forward / input-gradient kernels as three stored bf16 planes (6 bytes per element, plane-major): "
                         "ONLY with a library built with -DX6B_PLANE_IN=1 (timing-only experiment); the weight gradient is skipped")
    a = ap.parse_args()
    S = torch.cuda.current_stream().cuda_stream
    B = a.batch
    tot_ms = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    tot_fl = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print(f"{'shape':22s} {'fwd TF':>8s} {'dgrad TF':>9s} {'wgrad TF':>9s}   ms: fwd dgrad wgrad")
    for c0, c1, cout, hw, nf_, nb_ in SHAPES:
        cin = c0 + c1
        if a.plane_in:  # planes of post-ReLU-like data: h = bf16(v), m / l = the next 8 + 8 bits' worth of magnitude
            mk = lambda c: torch.stack([t_ * s_ for t_, s_ in zip(torch.randn(3, B, hw, hw, c, device="cuda").unbind(0),
                                                                   (1.0, 2.0 ** -8, 2.0 ** -16))]).bfloat16()
            x0, x1 = mk(c0), (mk(c1) if c1 else None)
        else:
            x0 = torch.randn(B, hw, hw, c0, device="cuda")
            x1 = torch.randn(B, hw, hw, c1, device="cuda") if c1 else None
        w = torch.randn(3, 3, cin, cout, device="cuda") * 0.05
        b = torch.randn(cout, device="cuda")
        nw = 9 * cin * cout
        pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
        pd = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
        L.call("punet_split_weights_x6", w.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, S)
        L.call("punet_split_weights_x6", w.data_ptr(), pd.data_ptr(), nw, cin, cout, 3, 1, S)
        y = torch.empty(B, hw, hw, cout, device="cuda")
        dz = mk(cout) if a.plane_in else torch.randn(B, hw, hw, cout, device="cuda")
        dx = torch.zeros(B, hw, hw, c0, device="cuda")
        dw = torch.empty(3, 3, cin, cout, device="cuda")
        db = torch.empty(cout, device="cuda")
        ns = max(1, L.call("punet_wgrad_x6_slices", c0, cout, B, hw, hw) // a.slices_div)
        ws = torch.empty(L.call("punet_wgrad_workspace_floats", c0, cout, 3, ns), device="cuda")

        def fwd():
            L.call("punet_conv3x3_x6", x0.data_ptr(), c0, x1.data_ptr() if c1 else None, c1, pf.data_ptr(), nw, cout, 0,
                   b.data_ptr(), 0, y.data_ptr(), cout, None, 0, 1, B, hw, hw, 0, S)

        bits = torch.zeros(L.call("punet_gate_dwords", B, hw, hw, c0), dtype=torch.int32, device="cuda")

        def dgrad():  # gradient w.r.t. the first source, ReLU-gated (bit form of the gate, as the engine does)
            L.call("punet_conv3x3_x6_gated", dz.data_ptr(), cout, None, 0, pd.data_ptr(), nw, cin, 0, None, 0, dx.data_ptr(), c0,
                   None if a.bits else x0.data_ptr(), bits.data_ptr() if a.bits else None, None, 0, 0, B, hw, hw, 0, S)

        def wgrad():
            L.call("punet_conv_wgrad_x6", x0.data_ptr(), c0, dz.data_ptr(), cout, dw.data_ptr(), cin, 0, db.data_ptr(),
                   ws.data_ptr(), ns, B, hw, hw, 0, 1.0, 0, S)

        fl = {"fwd": 2.0 * B * hw * hw * cin * cout * 9, "dgrad": 2.0 * B * hw * hw * c0 * cout * 9,
              "wgrad": 2.0 * B * hw * hw * c0 * cout * 9}
        ms = {}
        for name, fn in (("fwd", fwd), ("dgrad", dgrad), ("wgrad", wgrad)):
            if a.plane_in and name == "wgrad":
                ms[name] = float("nan")
                continue
            fn()
            torch.cuda.synchronize()
            ms[name] = timeit(fn, a.rounds)
            n = nf_ if name == "fwd" else nb_
            tot_ms[name] += n * ms[name]
            tot_fl[name] += n * fl[name]
        print(f"{c0:4d}+{c1:<4d}->{cout:<4d}@{hw:<4d} " + " ".join(f"{fl[k] / ms[k] / 1e9:9.1f}" for k in ms) + "   " +
              " ".join(f"{ms[k]:.3f}" for k in ms), flush=True)
    print("per MT step (weighted): " + "  ".join(f"{k} {tot_ms[k]:.2f} ms @ {tot_fl[k] / tot_ms[k] / 1e9:.1f} TF" for k in tot_ms if tot_ms[k] > 0))


if __name__ == "__main__":
    main()
