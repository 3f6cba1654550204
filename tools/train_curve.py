#!/usr/bin/env python3
"""Training curves of the arithmetic modes on one fixed synthetic task: the supervised PUNet objective (PUNetTrainer's step) on 8
fixed 256x256 batches of 4 synthetic patches (labels = the patches' bright blobs, so that there is something to learn), the reference's architecture and Adam, 300 iterations -
x6 (parity arithmetic) against the fp16-operand modes (f16x3 = what mixed_precision=True selects, f16, f16s) and the bf16 ones.
Same initial weights, same batches, same latent draws in every mode: the curves differ by the arithmetic alone.
usage: python tools/train_curve.py [--iters 300] [--lr 1e-4] [--size 256] [--modes x6,f16x3,f16,f16s,bf16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import punet_oracle as O  # synthetic data and weights only (test infrastructure)
from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam
from probabilistic_domain_adaptation_amd.my_trainer import fused as FU


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--modes", default="x6,f16x3,f16,f16s,bf16")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    nf, B, S = [64, 128, 256, 512], 4, a.size
    cfg = O.PUNetConfig(1, 1, nf, 6, 3, 1.0, True, False)
    Pnp = O.synth_params(cfg.spec(), 3)
    data = []
    for i in range(8):
        x, _, _, _ = O.synth_batch(B, 1, S, S, 700 + i)
        data.append((x.to(dev), (x > 0.3).float().to(dev)))  # a LEARNABLE label: the bright blobs of the (blurred-noise) patch
    g = torch.Generator().manual_seed(1)
    eps = [torch.randn(1, B, 6, generator=g).to(dev) for _ in range(a.iters)]
    marks = sorted(set([0, 1, 2, 5, 10, 25, 50, 100, 150, 200, 250, a.iters - 1]) & set(range(a.iters)))
    curves = {}
    for mode in a.modes.split(","):
        m = ProbabilisticUnet(1, 1, nf, 6, 3, 1.0, consensus_masking=True, rl_swap=False, device=dev, seed=0)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in Pnp.items()})
        m.set_conv_mode(mode)
        opt = FusedAdam(m, lr=a.lr)
        losses = []
        for it in range(a.iters):
            x, y = data[it % len(data)]
            losses.append(FU.fused_supervised_step(m, opt, x, y, None, eps[it], 1e-5))
        torch.cuda.synchronize()
        curves[mode] = [float(v) for v in losses]
        with torch.no_grad():  # after training, on all 8 batches: Dice of the thresholded mean of 8 samples (punet_trainer.py:70-82)
            dices = []
            for bi, (x, y) in enumerate(data):
                e8 = torch.randn(8, B, 6, generator=torch.Generator().manual_seed(50 + bi)).to(dev)
                yy, _ = m.pseudo_labels(x, 8, 0.9, 0.1, True, e8)
                seg = (yy > 0.5).float()
                dices.append(float(2 * (seg * y).sum() / (seg.sum() + y.sum() + 1e-7)))
            dice = sum(dices) / len(dices)
        finite = all(v == v and abs(v) != float("inf") for v in curves[mode])
        print(f"{mode:6s} finite={finite} dice={dice:.4f} " + " ".join(f"it{it}:{curves[mode][it]:.1f}" for it in marks), flush=True)
    ref = curves.get("x6")
    if ref:
        for mode, c in curves.items():
            if mode == "x6":
                continue
            tail = range(a.iters - 50, a.iters)
            med = lambda v: sorted(v[i] for i in tail)[25]
            first = max(abs(p - q) / abs(q) for p, q in zip(c[:10], ref[:10]))
            print(f"{mode:6s} vs x6: loss within {first:.1e} over the first 10 iterations; median loss of the last 50 iterations {med(c):.0f} vs "
                  f"{med(ref):.0f} (max {max(c[i] for i in tail):.0f} vs {max(ref[i] for i in tail):.0f}: Adam at this lr spikes in every mode)")


if __name__ == "__main__":
    main()
