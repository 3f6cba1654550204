#!/usr/bin/env python3
"""Tile quantization across patch shapes: forward / input-gradient / weight-gradient rate of the main x6 layer of
every U-Net level (c -> c at H / 2^k) at a list of patch shapes, each as a fraction of the same layer's rate at 512 x 512 (where
every level fills whole rounds of the persistent grid).  HIP events on the launch stream, best of --rounds, one process.
  python tools/shape_sweep.py [--shapes 512,384,520x704,640,704,768,1024] [--batch 4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from probabilistic_domain_adaptation_amd import _lib as L

LEVELS = [(64, 1), (128, 2), (256, 4), (512, 8)]


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


SCRATCH = None


def layer(B, H, W, c, rounds, kinds, splitk=-1):
    """splitk: -1 = the entry points without scratch (no split-K tail); 0 = punet_conv3x3_splitk with the library's plan; 1 = the same
    entry point, split off; n >= 2 = n parts forced"""
    global SCRATCH
    S = torch.cuda.current_stream().cuda_stream
    if SCRATCH is None:
        SCRATCH = torch.zeros(L.call("punet_conv_splitk_scratch_bytes") // 4, dtype=torch.int32, device="cuda")
    x0 = torch.relu(torch.randn(B, H, W, c, device="cuda"))
    w = torch.randn(3, 3, c, c, device="cuda") * 0.05
    b = torch.randn(c, device="cuda")
    nw = 9 * c * c
    pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    pd = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    L.call("punet_split_weights_x6", w.data_ptr(), pf.data_ptr(), nw, c, c, 3, 0, S)
    L.call("punet_split_weights_x6", w.data_ptr(), pd.data_ptr(), nw, c, c, 3, 1, S)
    y = torch.empty(B, H, W, c, device="cuda")
    dz = torch.randn(B, H, W, c, device="cuda")
    dx = torch.zeros(B, H, W, c, device="cuda")
    dw = torch.empty(3, 3, c, c, device="cuda")
    db = torch.empty(c, device="cuda")
    ns = L.call("punet_wgrad_x6_slices", c, c, B, H, W)
    ws = torch.empty(L.call("punet_wgrad_workspace_floats", c, c, 3, ns), device="cuda")
    bits = torch.zeros(L.call("punet_gate_dwords", B, H, W, c), dtype=torch.int32, device="cuda")
    fns = {
        "fwd": lambda: L.call("punet_conv3x3_x6", x0.data_ptr(), c, None, 0, pf.data_ptr(), nw, c, 0, b.data_ptr(), 0, y.data_ptr(), c,
                              None, 0, 1, B, H, W, 0, S),
        "dgrad": lambda: L.call("punet_conv3x3_x6_gated", dz.data_ptr(), c, None, 0, pd.data_ptr(), nw, c, 0, None, 0, dx.data_ptr(), c,
                                None, bits.data_ptr(), None, 0, 0, B, H, W, 0, S),
        "wgrad": lambda: L.call("punet_conv_wgrad_x6", x0.data_ptr(), c, dz.data_ptr(), c, dw.data_ptr(), c, 0, db.data_ptr(),
                                ws.data_ptr(), ns, B, H, W, 0, 1.0, 0, S),
    }
    if splitk >= 0:
        so = L.x6_opts(6, 16) | (splitk << 16)  # PUNET_X6_SPLITK: 1 = off, n >= 2 = runs of n - 1 quads
        fns["fwd"] = lambda: L.call("punet_conv3x3_splitk", x0.data_ptr(), c, None, 0, pf.data_ptr(), nw, c, 0, b.data_ptr(), 0, y.data_ptr(), c,
                                    None, None, None, None, 0, 1, B, H, W, so, 0, None, None, None, None, SCRATCH.data_ptr(), 4 * SCRATCH.numel(), S)
        fns["dgrad"] = lambda: L.call("punet_conv3x3_splitk", dz.data_ptr(), c, None, 0, pd.data_ptr(), nw, c, 0, None, 0, dx.data_ptr(), c,
                                      None, bits.data_ptr(), None, None, 0, 0, B, H, W, so, 0, None, None, None, None, SCRATCH.data_ptr(),
                                      4 * SCRATCH.numel(), S)
    fl = 2.0 * B * H * W * c * c * 9
    out = {}
    for k in kinds:
        fns[k]()
        torch.cuda.synchronize()
        out[k] = fl / timeit(fns[k], rounds) / 1e9
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="512,384,520x704,640,704,768,1024")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--kinds", default="fwd,dgrad,wgrad")
    ap.add_argument("--splitk", type=int, default=0, help="-1: entry points without scratch; 0: the library's split-K plan; 1: off; n: forced")
    a = ap.parse_args()
    kinds = a.kinds.split(",")
    shapes = []
    for s in a.shapes.split(","):
        h, _, w = s.partition("x")
        shapes.append((int(h), int(w or h)))
    if (512, 512) not in shapes:
        shapes.insert(0, (512, 512))
    ref = {}
    cap = 2 * L.call("punet_compute_units")
    print(f"batch {a.batch}; persistent grid = {cap} workgroups; tiles of the forward / input-gradient launch = B * ceil(H/8) * ceil(W/32) * c/64; "
          f"every cell: TFLOP/s (fraction of the same layer's rate at 512 x 512)")
    print(f"{'patch':>9s} {'level':>16s} {'tiles':>6s} {'rounds':>6s}  " + "  ".join(f"{k:>14s}" for k in kinds))
    worst = {}
    for (H, W) in [(512, 512)] + [s for s in shapes if s != (512, 512)]:
        for c, d in LEVELS:
            h, w = H // d, W // d
            tiles = a.batch * ((h + 7) // 8) * ((w + 31) // 32) * (c // 64)
            r = layer(a.batch, h, w, c, a.rounds, kinds, a.splitk if (H, W) != (512, 512) or a.splitk == 0 else -1)
            if (H, W) == (512, 512):
                ref[c] = r
            cells = []
            for k in kinds:
                f = r[k] / ref[c][k]
                cells.append(f"{r[k]:7.1f} ({f:4.2f})")
                if (H, W) != (512, 512):
                    worst[k] = min(worst.get(k, 9.9), f)
            print(f"{H:4d}x{W:<4d} {c:4d}->{c:<4d}@{h}x{w:<5d} {tiles:6d} {tiles / cap:6.2f}  " + "  ".join(f"{x:>14s}" for x in cells), flush=True)
    print("worst fraction of the 512 x 512 rate: " + ", ".join(f"{k} {v:.2f}" for k, v in worst.items()))


if __name__ == "__main__":
    main()
