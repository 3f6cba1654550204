#!/usr/bin/env python3
"""Go / no-go probe: contracting the up-blocks' `up` source at LOW resolution.

The first conv of an UpConvBlock is conv3x3(cat[up(x), bridge]) (unet_blocks.py:51-57).  Bilinear interpolation acts per channel
and the conv's channel contraction per pixel, so they commute:   W_t . up(x)[q] = up(W_t . x)[q]   for every tap t, hence
    conv3x3(up(x))[p] = sum_t shift_t(up(z_t))[p],   z_t = W_t . x  (nine 1x1 convs at LOW resolution = one GEMM with N = 9 cout)
- a quarter of the multiply-adds of the `up` half (2/3 of the layer), and the up-sampled tensor never exists.  The same holds for
the input gradient (nine adjoint-interpolated maps of dz, then one GEMM) and the weight gradient (x^T . those maps).
This probe times what exists today for the three up-block shapes of the bench (batch 4, 512x512) - the x6 3x3 kernels on the `up`
source alone plus the up-sampling kernels - against the GEMM side of the rewrite on the kernels the library already has (fp32-input
MFMA 1x1 forward / input gradient / weight gradient, and the x6 1x1 forward); the interpolating gather kernels are HBM-bound and are
priced by their traffic.  HIP events, interleaved rounds.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from probabilistic_domain_adaptation_amd import _lib as L

# (Cup, Cbridge, Cout, low-res side)
BLOCKS = [(512, 256, 256, 64), (256, 128, 128, 128), (128, 64, 64, 256)]


def timeit(fn, rounds=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    L.load()
    dev = torch.device("cuda:0")
    S = torch.cuda.current_stream().cuda_stream
    B = 4
    OPTS = L.x6_opts(6, 16)
    tot_old, tot_new = 0.0, 0.0
    for cup, cb, cout, h in BLOCKS:
        H = 2 * h
        x = torch.relu(torch.randn(B, h, h, cup, device=dev))
        up = torch.empty(B, H, H, cup, device=dev)
        dz = torch.randn(B, H, H, cout, device=dev)
        y = torch.empty(B, H, H, cout, device=dev)
        dup = torch.empty(B, H, H, cup, device=dev)
        dx = torch.empty(B, h, h, cup, device=dev)
        bias = torch.zeros(cout, device=dev)
        w3 = torch.randn(3, 3, cup, cout, device=dev) * 0.02
        nw = 9 * cup * cout
        pf = torch.empty(3 * nw, dtype=torch.bfloat16, device=dev)
        pd = torch.empty(3 * nw, dtype=torch.bfloat16, device=dev)
        L.call("punet_split_weights_x6", w3.data_ptr(), pf.data_ptr(), nw, cup, cout, 3, 0, S)
        L.call("punet_split_weights_x6", w3.data_ptr(), pd.data_ptr(), nw, cup, cout, 3, 1, S)
        dw = torch.empty(3, 3, cup, cout, device=dev)
        ns = L.call("punet_wgrad_x6_slices", cup, cout, B, H, H)
        ws = torch.empty(L.call("punet_wgrad_workspace_floats", cup, cout, 3, ns), device=dev)
        old = {
            "upsample fwd": timeit(lambda: L.call("punet_upsample2x_fwd", x.data_ptr(), up.data_ptr(), B, h, h, cup, S)),
            "x6 3x3 fwd (up half)": timeit(lambda: L.call("punet_conv3x3_x6", up.data_ptr(), cup, None, 0, pf.data_ptr(), nw, cout, 0,
                                                             bias.data_ptr(), 0, y.data_ptr(), cout, None, 0, 1, B, H, H, OPTS, S)),
            "x6 3x3 dgrad (up half)": timeit(lambda: L.call("punet_conv3x3_x6", dz.data_ptr(), cout, None, 0, pd.data_ptr(), nw, cup, 0,
                                                               None, 0, dup.data_ptr(), cup, None, 0, 0, B, H, H, OPTS, S)),
            "x6 3x3 wgrad (up half)": timeit(lambda: L.call("punet_conv_wgrad_x6", up.data_ptr(), cup, dz.data_ptr(), cout, dw.data_ptr(),
                                                               cup, 0, None, ws.data_ptr(), ns, B, H, H, 0, 1.0, OPTS, S)),
            "upsample bwd": timeit(lambda: L.call("punet_upsample2x_bwd_gated", dup.data_ptr(), x.data_ptr(), None, dx.data_ptr(), B, h, h,
                                                    cup, S)),
        }
        # the rewrite's GEMMs: M = B h h pixels, K = cup, N = 9 cout
        n9 = 9 * cout
        w1 = torch.randn(cup, n9, device=dev) * 0.02
        z = torch.empty(B, h, h, n9, device=dev)
        dzt = torch.randn(B, h, h, n9, device=dev)
        b9 = torch.zeros(n9, device=dev)
        w1t = torch.empty(n9, cup, device=dev)
        L.call("punet_flip_transpose_weights", w1.data_ptr(), w1t.data_ptr(), cup, n9, 1, S)
        dw1 = torch.empty(cup, n9, device=dev)
        ns1 = L.call("punet_wgrad_slices", cup, n9, B, h, h)
        ws1 = torch.empty(L.call("punet_wgrad_workspace_floats", cup, n9, 1, ns1), device=dev)
        p1 = torch.empty(3 * cup * n9, dtype=torch.bfloat16, device=dev)
        L.call("punet_split_weights_x6", w1.data_ptr(), p1.data_ptr(), cup * n9, cup, n9, 1, 0, S)
        p1d = torch.empty(3 * cup * n9, dtype=torch.bfloat16, device=dev)
        L.call("punet_split_weights_x6", w1.data_ptr(), p1d.data_ptr(), cup * n9, cup, n9, 1, 1, S)
        new = {
            "f32 1x1 fwd GEMM": timeit(lambda: L.call("punet_conv_fwd", x.data_ptr(), cup, None, 0, w1.data_ptr(), b9.data_ptr(), 0,
                                                       z.data_ptr(), n9, B, h, h, 1, 0, S)),
            "x6 1x1 fwd GEMM": timeit(lambda: L.call("punet_conv1x1_x6_gated", x.data_ptr(), cup, None, 0, p1.data_ptr(), cup * n9, n9, 0,
                                                      b9.data_ptr(), 0, z.data_ptr(), n9, None, None, None, 0, 0, B, h, h, S)),
            "f32 1x1 dgrad GEMM": timeit(lambda: L.call("punet_conv_dgrad_t", dzt.data_ptr(), n9, w1t.data_ptr(), cup, 0, cup, dx.data_ptr(),
                                                         x.data_ptr(), 0, B, h, h, 1, S)),
            "f32 1x1 wgrad GEMM": timeit(lambda: L.call("punet_conv_wgrad", x.data_ptr(), cup, dzt.data_ptr(), n9, dw1.data_ptr(), cup, 0, None,
                                                         ws1.data_ptr(), ns1, B, h, h, 1, 0, 1.0, S)),
        }
        # the interpolating gather (forward: in place on the bridge half's convolution, with ReLU and gate bits) and its adjoint
        bits = torch.zeros(L.call("punet_gate_dwords", B, H, H, cout), dtype=torch.int32, device=dev) if cout % 64 == 0 else None
        new["gather fwd (relu + gate bits)"] = timeit(lambda: L.call("punet_upconv_gather_fwd", z.data_ptr(), y.data_ptr(),
                                                                       bits.data_ptr() if bits is not None else None, 1, B, h, h, cout, S))
        new["gather fwd (no bits)"] = timeit(lambda: L.call("punet_upconv_gather_fwd", z.data_ptr(), y.data_ptr(), None, 1, B, h, h, cout, S))
        new["gather bwd (adjoint)"] = timeit(lambda: L.call("punet_upconv_gather_bwd", dz.data_ptr(), dzt.data_ptr(), B, h, h, cout, S))
        new["x6 1x1 dgrad GEMM"] = timeit(lambda: L.call("punet_conv1x1_x6_gated", dzt.data_ptr(), n9, None, 0, p1d.data_ptr(), cup * n9, cup, 0,
                                                          None, 0, dx.data_ptr(), cup, x.data_ptr(), None, None, 0, 0, B, h, h, S))
        nsu = L.call("punet_upconv_wgrad_slices", cup, cout, B, h, h)
        wsu = torch.empty(L.call("punet_upconv_wgrad_workspace_floats", cup, cout, nsu), device=dev)
        new["x6 low-res wgrad (GEMM-TN)"] = timeit(lambda: L.call("punet_upconv_wgrad_x6", x.data_ptr(), cup, dzt.data_ptr(), cout, dw.data_ptr(), cup,
                                                                    wsu.data_ptr(), nsu, B, h, h, 0, 1.0, S))
        # z through the Infinity Cache instead of HBM: the GEMM and the gather image by image (or pair by pair) over ONE z buffer that
        # is small enough to stay on chip between its producer and its consumer and is overwritten before it is written back
        for per in (1, 2):
            if B % per:
                continue
            zs = torch.empty(per, h, h, n9, device=dev)

            def strips(per=per, zs=zs):
                for b0 in range(0, B, per):
                    L.call("punet_conv1x1_x6_gated", x[b0:].data_ptr(), cup, None, 0, p1.data_ptr(), cup * n9, n9, 0, b9.data_ptr(), 0,
                           zs.data_ptr(), n9, None, None, None, 0, 0, per, h, h, S)
                    L.call("punet_upconv_gather_fwd", zs.data_ptr(), y[b0:].data_ptr(), None, 1, per, h, h, cout, S)

            def whole():
                L.call("punet_conv1x1_x6_gated", x.data_ptr(), cup, None, 0, p1.data_ptr(), cup * n9, n9, 0, b9.data_ptr(), 0,
                       z.data_ptr(), n9, None, None, None, 0, 0, B, h, h, S)
                L.call("punet_upconv_gather_fwd", z.data_ptr(), y.data_ptr(), None, 1, B, h, h, cout, S)
            new[f"GEMM + gather, whole batch (z {z.numel() * 4 / 1e6:.0f} MB)"] = timeit(whole)
            new[f"GEMM + gather, {per} image(s) at a time (z {zs.numel() * 4 / 1e6:.0f} MB)"] = timeit(strips)
        gb_f = (z.numel() + 2 * y.numel()) * 4 / 1e9
        gb_b = (dz.numel() + dzt.numel()) * 4 / 1e9
        print(f"   gather traffic: forward {gb_f:.2f} GB = {gb_f / new['gather fwd (relu + gate bits)'] * 1e3:.2f} TB/s, "
              f"adjoint {gb_b:.2f} GB = {gb_b / new['gather bwd (adjoint)'] * 1e3:.2f} TB/s")
        gf = 2.0 * B * H * H * cup * cout * 9 / 1e9
        print(f"--- up block {cup}+{cb} -> {cout} @{H}x{H}: `up` half = {gf:.1f} GFLOP per direction, GEMM form {gf / 4:.1f}")
        for k, v in old.items():
            print(f"   today   {k:34s} {v:8.1f} us" + (f"  {gf / v * 1e3:7.1f} TF" if "3x3" in k else ""))
        for k, v in new.items():
            print(f"   rewrite {k:34s} {v:8.1f} us" + (f"  {gf / 4 / v * 1e3:7.1f} TF" if "GEMM" in k else ""))
        o_f = old["upsample fwd"] + old["x6 3x3 fwd (up half)"]
        o_b = old["x6 3x3 dgrad (up half)"] + old["x6 3x3 wgrad (up half)"] + old["upsample bwd"]
        n_f = min(new["f32 1x1 fwd GEMM"], new["x6 1x1 fwd GEMM"]) + new["gather fwd (relu + gate bits)"]
        # backward of the rewrite as built: the input gradient in the new form, the weight gradient as today (on the materialised
        # up-sampled tensor, whose forward interpolation then stays: counted)
        n_b = min(new["f32 1x1 dgrad GEMM"], new["x6 1x1 dgrad GEMM"]) + new["gather bwd (adjoint)"] + old["x6 3x3 wgrad (up half)"] \
            + old["upsample fwd"]
        print(f"   forward {o_f:8.1f} -> {n_f:8.1f} us   backward {o_b:8.1f} -> {n_b:8.1f} us")
        tot_old += 2 * o_f + o_b  # a Mean-Teacher step: teacher forward, student forward + backward
        tot_new += 2 * n_f + n_b
    print(f"per Mean-Teacher step (two forward passes + one backward): {tot_old / 1e3:.2f} ms today -> {tot_new / 1e3:.2f} ms "
          f"({(tot_old - tot_new) / 1e3:+.2f} ms)")


if __name__ == "__main__":
    main()
