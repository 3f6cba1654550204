#!/usr/bin/env python3
"""Benchmark of the hot path: training patches/s of the PUNet Mean-Teacher step on 512x512 LIVECell-shaped synthetic
patches (BASELINE.json metric), N GPUs of one node, one process per GPU, RCCL all-reduce of the flat gradient.

  python bench.py [--gpus N] [--steps K] [--warmup W]
                  [--workload mean_teacher|punet_source|inference|fixmatch|adamt|adamatch] [--channels C] [--size S]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one Mean-Teacher iteration on a batch of 4 patches per GPU (mean_teacher_trainer.py:101-131):
teacher forward + N=8 Monte-Carlo samples + consensus masking, student forward, ELBO (BCE + KL) + L2 regulariser,
full backward, gradient all-reduce, Adam, EMA teacher update.  Inputs are resident in HBM before the timed region.
fp32 storage and fp32-accurate arithmetic everywhere (default conv arithmetic "x6": every fp32 product evaluated as six
bf16 MFMAs with fp32 accumulation, logits within 1e-3 of the reference, masks bit-exact); nothing is skipped or cached.
`--conv-mode f32` selects the fp32-input MFMA kernels (same tolerances); `--conv-mode x3|bf16` are the opt-in
reduced-precision fast modes (the reference scripts' mixed_precision=True analogue) - their lines say so in `dtype`
and are NOT the headline number.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel = the 3x3 MFMA
convolution, algorithmic FLOPs / HIP-event duration measured live in the timed region, against the matrix peak of the
arithmetic in use) and `cpu_baseline` (the CPU oracle = a port of the reference's path, timed on the host cores,
rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NF = [64, 128, 256, 512]
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide, "Peak BF16/FP16 MFMA ~2.5 PF dense"
# algorithmic GFLOP per patch at 512x512x1
GFLOP = {"mean_teacher": {8: 3233.4, 16: 3269.6}, "punet_source": 2539.4, "inference": {8: 694.0, 16: 730.2},
         "fixmatch": {8: 3233.4, 16: 3269.6}, "adamt": {8: 5772.8, 16: 5809.1}, "adamatch": {8: 5772.8, 16: 5809.1}}
# the same at 704x704x3 (BASELINE.json configs[3]); other shapes are scaled by the pixel count
GFLOP_704x3 = {"mean_teacher": {8: 6125.7, 16: 6194.2}, "fixmatch": {8: 6125.7, 16: 6194.2}, "punet_source": 4811.4,
               "inference": {8: 1314.3, 16: 1382.8}, "adamt": {8: 10937.1, 16: 11005.6}, "adamatch": {8: 10937.1, 16: 11005.6}}


def synth(B, H, W, seed, dev, C=1):
    g = torch.Generator().manual_seed(seed)
    k = torch.ones(1, 1, 5, 5) / 25.0
    blur = lambda t: torch.nn.functional.conv2d(torch.nn.functional.pad(t, (2, 2, 2, 2), mode="reflect"),
                                                k.expand(t.shape[1], 1, 5, 5), groups=t.shape[1])
    x = blur(torch.randn(B, C, H, W, generator=g))
    x = (x - x.mean(dim=(1, 2, 3), keepdim=True)) / (x.std(dim=(1, 2, 3), keepdim=True) + 1e-7)
    x1 = x + 0.1 * torch.randn(x.shape, generator=g)
    x2 = 1.5 * blur(x) + 0.25 * torch.randn(x.shape, generator=g)
    y = (blur(torch.randn(B, 1, H, W, generator=g)) * 5.0 > 0.3).float()
    return [t.contiguous().to(dev) for t in (x, x1, x2, y)]


def host_cores():
    """CPU threads this process may really use: affinity mask, cgroup quota, and the GPU box's per-GPU share (16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return int(os.environ.get("PUNET_CPU_THREADS", min(n, 16)))


def cpu_baseline(workload, n_samples, H, W, budget_s=12.0):
    """The CPU oracle (own restatement of the reference path, pinned by golden fixtures) on the host cores."""
    from oracle import punet_oracle as O
    torch.set_num_threads(host_cores())
    cfg = O.PUNetConfig(1, 1, NF, 6, 3, 1.0, consensus_masking=True, rl_swap=False)
    Pnp = O.synth_params(cfg.spec(), 1)

    def one(h, w):
        Ps, Pt = O.to_torch(Pnp, True), O.to_torch(Pnp)
        opt = O.AdamState(Ps, lr=1e-5)
        x, x1, x2, y = O.synth_batch(1, 1, h, w, 0)
        g = torch.Generator().manual_seed(2)
        et, es = torch.randn(n_samples, 1, 6, generator=g), torch.randn(1, 6, generator=g)
        t0 = time.time()
        if workload == "mean_teacher":
            O.mean_teacher_step(Ps, Pt, cfg, opt, x1, x2, et, es, True)
        elif workload == "punet_source":
            O.punet_step(Ps, cfg, opt, x2, y, es)
        else:
            O.mc_prediction(Ps, cfg, x1, et)
        return time.time() - t0

    one(64, 64)  # warm-up (page-in, thread pools)
    times = []
    while sum(times) < budget_s and len(times) < 2:
        times.append(one(H, W))
    per = sum(times) / len(times)
    return {"value": round(1.0 / per, 5), "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} x (batch 1, {H}x{W}, N={n_samples}) {workload} step of the CPU oracle "
                      f"(torch {torch.__version__} CPU fp32, {torch.get_num_threads()} threads), {per:.2f} s each"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="mean_teacher", choices=["mean_teacher", "punet_source", "inference", "fixmatch", "adamt", "adamatch"])
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--samples", type=int, default=8)
    ap.add_argument("--conv-mode", default=None, choices=["x6", "f32", "x3", "bf16"],
                    help="3x3 conv arithmetic (default: $PUNET_CONV_MODE or x6); x3/bf16 are reduced-precision fast modes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--kernel-timing", default="roofline", choices=["roofline", "all"],
                    help="HIP-event brackets inside the timed region: only the roofline kernel's launches (default; 73 per "
                         "step) or every MFMA launch (adds ~1 ms of bubbles per step)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI; the default) or gloo (rehearsal on one GPU)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})")
    local = local % max(torch.cuda.device_count(), 1)  # gloo rehearsal: several ranks may share one GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # "nccl" == RCCL on ROCm
        else:
            dist.init_process_group(a.backend)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU

    B, H, W, N = a.batch, a.size, a.size, a.samples
    model = ProbabilisticUnet(a.channels, 1, NF, 6, 3, 1.0, consensus_masking=True, rl_swap=False, device=dev, seed=1,
                              conv_mode=a.conv_mode)
    teacher = None
    if a.workload in ("mean_teacher", "adamt"):
        import copy
        teacher = copy.deepcopy(model)
    opt = FusedAdam(model, lr=1e-5)
    x, x1, x2, y = synth(B, H, W, 100 + rank, dev, a.channels)  # every rank its own shard of the synthetic stream
    xs, _, _, ys = synth(B, H, W, 500 + rank, dev, a.channels)  # source-domain batch of the joint trainers
    gen = torch.Generator(device=dev).manual_seed(7 + rank)
    model.generator = gen
    if teacher is not None:
        teacher.generator = gen

    def step():
        if a.workload == "mean_teacher":
            FU.fused_self_training_step(model, teacher, opt, x1, x2, N, True, None, None, 0.999, 1e-5)
        elif a.workload == "punet_source":
            FU.fused_supervised_step(model, opt, x2, y, None, None, 1e-5)
        elif a.workload == "fixmatch":  # fixmatch_trainer.py:67-95 (no EMA, pseudo-labels from the model itself)
            FU.fused_self_training_step(model, model, opt, x1, x2, N, True, None, None, None, 1e-5)
        elif a.workload == "adamt":  # adamt_trainer.py:89-128: one source + one target patch per "patch"
            FU.fused_joint_step(model, teacher, opt, xs, ys, x1, x2, N, True, None, None, None, 0.999, True, 1e-5)
        elif a.workload == "adamatch":  # adamatch_trainer.py:62-107
            FU.fused_joint_step(model, model, opt, xs, ys, x1, x2, N, True, None, None, None, None, True, 1e-5)
        else:
            FU.pseudo_labels(model, x1, N, True)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    eng = model.engine
    engines = [eng] + ([teacher.engine] if teacher is not None else [])
    prof = None if a.no_kernel_timing else []
    for e in engines:  # the warm-up runs with the same HIP-event brackets as the timed steps: the first use of ~1000 events
        e.prof = prof  # in a process costs tens of ms (seen as one 124 ms step in an otherwise flat 59 ms series)
        e.prof_kinds = {"conv3x3_fwd"} if a.kernel_timing == "roofline" else None
    for _ in range(a.warmup):
        step()
    sync()
    if prof is not None:
        prof.clear()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]  # per-step GPU timeline (diagnostic)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        step()
        marks[i + 1].record()
    t_enq = time.perf_counter() - t0  # host time to enqueue the K steps (the GPU runs behind; must stay below dt)
    sync()
    dt = time.perf_counter() - t0
    for e in engines:
        e.prof = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    if rank == 0:
        patches = world * B * a.steps
        exact = (H, W, a.channels) == (704, 704, 3)
        gf = (GFLOP_704x3 if exact else GFLOP)[a.workload]
        gf = gf[N] if isinstance(gf, dict) else gf
        if not exact and (H, W) != (512, 512):
            gf = gf * (H * W) / (512 * 512)
        out = {
            "metric": "training patches/sec (512x512, N=8 MC samples)" if a.workload != "inference" else "inference patches/sec",
            "value": round(patches / dt, 4), "unit": "patches/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"x6": "f32", "f32": "f32", "x3": "bf16x3 (fp32 storage, 16-bit significand products; reduced precision)",
                      "bf16": "bf16 (fp32 storage; reduced precision)"}[eng.conv_mode], "data": "synthetic",
            "config": {"workload": f"PUNet {a.workload} step: nf={NF}, latent 6, fcomb 3, BCE+KL, consensus masking, "
                                   f"{H}x{W}x{a.channels} patches, batch {B}/GPU, N={N} MC samples, Adam lr 1e-5, EMA 0.999",
                       "global_batch": world * B, "parallelism": f"dp{world}",
                       "algorithmic_gflop_per_patch": gf,
                       "model_tflops_per_gpu": round(gf * patches / dt / world / 1e3, 2),
                       "host_enqueue_ms_per_step": round(1e3 * t_enq / a.steps, 2),
                       "ms_each_step": [round(marks[i].elapsed_time(marks[i + 1]), 2) for i in range(a.steps)]},
        }
        out["config"]["conv_arithmetic"] = {
            "x6": "3x3 convs fwd+dgrad+wgrad: fp32 operands split into 3 bf16 planes, 6 bf16 MFMAs per product, fp32 "
                  "accumulate (fp32-accurate); 1x1 convs: fp32-input MFMA",
            "f32": "fp32-input MFMA everywhere",
            "x3": "REDUCED PRECISION fast mode: 3x3 convs with 2 bf16 planes per operand, 3 bf16 MFMAs per product "
                  "(16 significand bits), fp32 accumulate and storage; 1x1 convs fp32-input MFMA",
            "bf16": "REDUCED PRECISION fast mode: 3x3 convs in plain bf16 (1 MFMA per product), fp32 accumulate and "
                    "storage; 1x1 convs fp32-input MFMA"}[eng.conv_mode]
        if prof:
            agg = {}
            for kind, fl, e0, e1 in prof:
                d = agg.setdefault(kind, [0.0, 0.0, 0])
                d[0] += fl
                d[1] += e0.elapsed_time(e1) * 1e-3
                d[2] += 1
            kernels = {k: {"launches": v[2], "avg_ms": round(1e3 * v[1] / v[2], 4), "tflops": round(v[0] / v[1] / 1e12, 2),
                           "share_of_step": round(v[1] / dt, 3)} for k, v in agg.items()}
            k3 = agg.get("conv3x3_fwd")
            traffic, traffic_note = None, "not collected"
            try:  # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, tools/pmc_traffic.py)
                if a.workload == "mean_teacher" and (B, H, W, N) == (4, 512, 512, 8) and eng.conv_mode in ("x6", "f32"):
                    fn, key = {"x6": ("pmc_hbm_traffic_x6.json", "conv3x3_x6b_kernel<3, false>"),
                               "f32": ("pmc_hbm_traffic.json", "conv_mfma_kernel<3, false>")}[eng.conv_mode]
                    pm = json.load(open(os.path.join(ROOT, "profiles", "r01", fn)))
                    traffic = round(pm["kernels"][key]["hbm_bytes_per_launch"])
                    traffic_note = ("bytes per launch, rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE in separate "
                                    f"passes of this same workload (profiles/r01/{fn}), mean over the launches of the forward "
                                    "instantiation (algorithmic mean over the forward launches: 265 MB = 144 MB activations in + "
                                    "weight planes, 121 MB out); FETCH_SIZE includes the 1.33x halo re-reads of the 8x32-pixel tiles")
            except Exception:
                pass
            if k3:
                ach = k3[0] / k3[1] / 1e12
                if eng.x6:
                    # n bf16 MFMAs per product: the roofline of THIS arithmetic is the bf16 dense peak / n
                    n = {"x6": 6, "x3": 3, "bf16": 1}[eng.conv_mode]
                    peak = BF16_MFMA_PEAK_TFLOPS / n
                    kname = (f"conv3x3_x6b_kernel<{(n + 1) // 2 if n > 1 else 1} planes> (3x3 NHWC direct conv forward, "
                             f"v_mfma_f32_16x16x32_bf16; conv3x3_x6_kernel for channel counts that are not multiples of 32)")
                    pnote = (f"bf16 dense MFMA peak {BF16_MFMA_PEAK_TFLOPS} TFLOP/s / {n} MFMAs per product = {peak:.1f} TFLOP/s of "
                             f"algorithmic work; executed bf16 MFMA rate = {n} x achieved = {n * ach:.0f} TFLOP/s "
                             f"({n * ach / BF16_MFMA_PEAK_TFLOPS:.3f} of the bf16 peak; a register-only MFMA loop with "
                             f"random operands sustains ~1790 TFLOP/s on this part, profiles/r01/mfma_peak_microbench.log); "
                             f"for reference the fp32-input MFMA peak is {FP32_MFMA_PEAK_TFLOPS} TFLOP/s "
                             f"(frac {ach / FP32_MFMA_PEAK_TFLOPS:.3f})")
                else:
                    peak = FP32_MFMA_PEAK_TFLOPS
                    kname = "conv_mfma_kernel<3,false> (3x3 NHWC direct conv forward, fp32 MFMA)"
                    pnote = "fp32-input MFMA dense peak (PUNET_CONV_MODE=f32)"
                out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                                   "frac": round(ach / peak, 4), "traffic": traffic, "traffic_note": traffic_note,
                                   "kernel": kname, "flops_per_launch_avg": k3[0] / k3[2],
                                   "avg_launch_ms": round(1e3 * k3[1] / k3[2], 4), "peak_note": pnote}
            out["kernels"] = kernels
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(a.workload, N, H, W)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
