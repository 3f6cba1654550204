"""Probe (round 4): do two INDEPENDENT kernel chains finish sooner when a hipGraph lets them overlap (fork / join over two
streams) than back to back on one stream?  The x6 kernels are persistent grids that fill every CU, so what overlap can recover is
the tail of each launch (slots idle while the last tiles finish) and the launch boundaries.  Chains: the U-Net forward and the prior
encoder forward of the bench model (what a teacher pass runs), B = 4, 512 x 512.

    python tools/overlap_probe.py [--size 512] [--batch 4] [--reps 20] [--mode x6]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--mode", default="x6")
    a = ap.parse_args()
    from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
    from probabilistic_domain_adaptation_amd.engine import Act
    dev = torch.device("cuda:0")
    nf = [64, 128, 256, 512]
    m = ProbabilisticUnet(1, 1, nf, 6, 3, 1.0, consensus_masking=True, device=dev, seed=1, conv_mode=a.mode)
    eng, flat = m.engine, m.flat_params
    B, H, W = a.batch, a.size, a.size
    x = torch.randn(B, H, W, 1, device=dev)
    eng.prepare_forward_weights(flat)

    def chain_unet():
        return eng.unet(flat, Act(x, B, H, W, 1, needs_grad=False), None, None)

    def chain_prior():
        return eng.latent(flat, "prior", Act(x, B, H, W, 1, needs_grad=False), None, None)

    for _ in range(2):  # size workspaces
        chain_unet(); chain_prior()
    torch.cuda.synchronize()

    def capture(fork):
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(dev)
        s2 = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            g.capture_begin()
            if fork:
                s2.wait_stream(s)
                with torch.cuda.stream(s2):
                    keep2 = chain_prior()
                keep1 = chain_unet()
                s.wait_stream(s2)
            else:
                keep2 = chain_prior()
                keep1 = chain_unet()
            g.capture_end()
        torch.cuda.current_stream().wait_stream(s)
        return g, (keep1, keep2)

    graphs = {"sequential": capture(False), "forked": capture(True)}
    res = {k: [] for k in graphs}
    for rnd in range(5):
        for k, (g, _) in graphs.items():
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                g.replay()
            torch.cuda.synchronize()
            res[k].append(1e3 * (time.perf_counter() - t0) / a.reps)
    for k, v in res.items():
        print(f"{k:11s} ms per (U-Net forward + prior encoder forward): " + " ".join(f"{t:.3f}" for t in v) + f"   min {min(v):.3f}")
    print(f"forked / sequential (min): {min(res['forked']) / min(res['sequential']):.4f}")


if __name__ == "__main__":
    main()
