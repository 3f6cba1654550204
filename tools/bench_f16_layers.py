#!/usr/bin/env python3
"""Forward / input-gradient / weight-gradient kernels of the fast modes per layer shape at the bench configuration (batch 4,
512x512 input, nf=[64,128,256,512]) on post-ReLU-like operands (half zeros): the bf16-plane arithmetics (x6, x3, bf16) next to the
fp16-plane ones (f16x3, f16) with and without the amax publication; HIP events, best of --rounds, one process."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from probabilistic_domain_adaptation_amd import _lib as L

SHAPES = [(64, 0, 64, 512, 12, 8), (128, 0, 128, 256, 12, 8), (256, 0, 256, 128, 12, 8), (512, 0, 512, 64, 10, 6),
          (512, 256, 256, 128, 2, 1), (128, 64, 64, 512, 2, 1)]


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
    ap.add_argument("--kinds", default="fwd,dgrad,wgrad")
    ap.add_argument("--plane-in", action="store_true",
                    help="operands as two STORED fp16 planes (h = fp16 of the scaled value, l = the residual; same 4 bytes per element as "
                         "fp32): ONLY with a library built with -DX6B_PLANE_IN=2 and two-plane variants (timing-only experiment)")
    ap.add_argument("--variants", default="x6,x3,bf16,f16x3,f16x3-noamax,f16",
                    help="(a library built with -DX6B_PLANE_IN reads NP x 2 bytes per element from these fp32 buffers: leave x6 out)")
    a = ap.parse_args()
    S = torch.cuda.current_stream().cuda_stream
    B = a.batch
    variants = a.variants.split(",")
    kinds = a.kinds.split(",")
    tot = {(k, v): [0.0, 0.0] for k in kinds for v in variants}
    for c0, c1, cout, hw, nf_, nb_ in SHAPES:
        cin = c0 + c1
        def planes(t):  # [2][...]: h, l of t * 2^k (amax at 2^14) as fp16, reinterpreted as the fp32 buffer the entry points expect
            sc = 2.0 ** (14 - int(torch.floor(torch.log2(t.abs().max())).item()))
            h = (t * sc).half()
            l = (t * sc - h.float()).half()
            return torch.stack([h, l]).contiguous().view(torch.float32)
        x0 = torch.relu(torch.randn(B, hw, hw, c0, device="cuda"))
        x1 = torch.relu(torch.randn(B, hw, hw, c1, device="cuda")) if c1 else None
        w = torch.randn(3, 3, cin, cout, device="cuda") * 0.05
        b = torch.randn(cout, device="cuda")
        nw = 9 * cin * cout
        P = {}
        for fam in ("bf", "f16"):
            pf = torch.zeros(3 * nw, dtype=torch.bfloat16, device="cuda")
            pd = torch.zeros(3 * nw, dtype=torch.bfloat16, device="cuda")
            aw = torch.zeros(1, device="cuda")
            if fam == "bf":
                L.call("punet_split_weights_x6", w.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, S)
                L.call("punet_split_weights_x6", w.data_ptr(), pd.data_ptr(), nw, cin, cout, 3, 1, S)
            else:
                L.call("punet_split_weights_f16", w.data_ptr(), pf.data_ptr(), nw, cin, cout, 0, aw.data_ptr(), S)
                L.call("punet_split_weights_f16", w.data_ptr(), pd.data_ptr(), nw, cin, cout, 1, aw.data_ptr(), S)
            P[fam] = (pf, pd, aw)
        y = torch.empty(B, hw, hw, cout, device="cuda")
        dz = torch.randn(B, hw, hw, cout, device="cuda") * (torch.rand(B, hw, hw, cout, device="cuda") > 0.5)
        dx = torch.zeros(B, hw, hw, c0, device="cuda")
        dw = torch.empty(3, 3, cin, cout, device="cuda")
        db = torch.empty(cout, device="cuda")
        ns = L.call("punet_wgrad_x6_slices", c0, cout, B, hw, hw)
        ws = torch.empty(L.call("punet_wgrad_workspace_floats", c0, cout, 3, ns), device="cuda")
        bits = torch.zeros(L.call("punet_gate_dwords", B, hw, hw, c0), dtype=torch.int32, device="cuda")
        slots = torch.zeros(8, device="cuda")
        for i, t in enumerate([x0, x1, dz]):
            if t is not None:
                L.call("punet_amax", t.data_ptr(), t.numel(), slots.data_ptr() + 4 * i, 0, S)
        if a.plane_in:
            x0, x1, dz = planes(x0), (planes(x1) if c1 else None), planes(dz)
        sp = lambda i: slots.data_ptr() + 4 * i
        row = f"{c0:4d}+{c1:<4d}->{cout:<4d}@{hw:<4d}"
        for kind in kinds:
            fl = 2.0 * B * hw * hw * (cin if kind == "fwd" else c0) * cout * 9
            n = nf_ if kind == "fwd" else nb_
            cells = []
            for v in variants:
                if v in ("x6", "x3", "bf16"):
                    opts = L.x6_opts({"x6": 6, "x3": 3, "bf16": 1}[v])
                    pf, pd, _ = P["bf"]
                    if kind == "fwd":
                        fn = lambda: L.call("punet_conv3x3_x6", x0.data_ptr(), c0, x1.data_ptr() if c1 else None, c1, pf.data_ptr(), nw, cout, 0,
                                            b.data_ptr(), 0, y.data_ptr(), cout, None, 0, 1, B, hw, hw, opts, S)
                    elif kind == "dgrad":
                        fn = lambda: L.call("punet_conv3x3_x6_gated", dz.data_ptr(), cout, None, 0, pd.data_ptr(), nw, cin, 0, None, 0,
                                            dx.data_ptr(), c0, None, bits.data_ptr(), None, 0, 0, B, hw, hw, opts, S)
                    else:
                        fn = lambda: L.call("punet_conv_wgrad_x6", x0.data_ptr(), c0, dz.data_ptr(), cout, dw.data_ptr(), cin, 0, db.data_ptr(),
                                            ws.data_ptr(), ns, B, hw, hw, 0, 1.0, opts, S)
                else:
                    planes = 1 if v == "f16" else 2
                    pf, pd, aw = P["f16"]
                    ao = None if v.endswith("noamax") else sp(5)
                    if kind == "fwd":
                        fn = lambda: L.call("punet_conv3x3_f16", x0.data_ptr(), c0, x1.data_ptr() if c1 else None, c1, pf.data_ptr(), nw, cout, 0,
                                            b.data_ptr(), 0, y.data_ptr(), cout, None, None, None, None, 0, 1, B, hw, hw, planes, sp(0),
                                            sp(1) if c1 else None, aw.data_ptr(), ao, S)
                    elif kind == "dgrad":
                        fn = lambda: L.call("punet_conv3x3_f16", dz.data_ptr(), cout, None, 0, pd.data_ptr(), nw, cin, 0, None, 0, dx.data_ptr(), c0,
                                            None, bits.data_ptr(), None, None, 0, 0, B, hw, hw, planes, sp(2), None, aw.data_ptr(), ao, S)
                    else:
                        fn = lambda: L.call("punet_conv_wgrad_f16", x0.data_ptr(), c0, dz.data_ptr(), cout, dw.data_ptr(), cin, 0, db.data_ptr(),
                                            ws.data_ptr(), ns, B, hw, hw, 0, 1.0, planes, 0, sp(0), sp(2), S)
                fn()
                torch.cuda.synchronize()
                ms = timeit(fn, a.rounds)
                tot[(kind, v)][0] += n * ms
                tot[(kind, v)][1] += n * fl
                cells.append(f"{v} {fl / ms / 1e9:7.1f}")
            print(f"{row} {kind:5s} TF: " + "  ".join(cells), flush=True)
    for kind in kinds:
        print(f"per MT step (weighted) {kind:5s}: " + "  ".join(f"{v} {tot[(kind, v)][0]:.2f} ms @ {tot[(kind, v)][1] / tot[(kind, v)][0] / 1e9:.0f} TF"
                                                               for v in variants))


if __name__ == "__main__":
    main()
