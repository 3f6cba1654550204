#!/usr/bin/env python3
"""Benchmark of the hot path: training patches/s of the PUNet Mean-Teacher step on 512x512 LIVECell-shaped synthetic
patches (BASELINE.json metric), N GPUs of one node, one process per GPU, RCCL all-reduce of the flat gradient.

  python bench.py [--gpus N] [--steps K] [--warmup W]
                  [--workload mean_teacher|punet_source|inference|fixmatch|adamt|adamatch] [--channels C] [--size S]
  N > 1, either way: `python bench.py --gpus N ...` by itself (it then starts its own N ranks as fresh child processes, relays
  rank 0's line and the ranks' exit code), or under a launcher:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one Mean-Teacher iteration on a batch of 4 patches per GPU (mean_teacher_trainer.py:101-131):
teacher forward + N=8 Monte-Carlo samples + consensus masking, student forward, ELBO (BCE + KL) + L2 regulariser,
full backward, gradient all-reduce, Adam, EMA teacher update.  Inputs are resident in HBM before the timed region.
fp32 storage and fp32-accurate arithmetic everywhere (default conv arithmetic "x6": every fp32 product evaluated as six
bf16 MFMAs with fp32 accumulation, logits within 1e-3 of the reference, masks bit-exact); nothing is skipped or cached.
`--conv-mode f32` selects the fp32-input MFMA kernels (same tolerances); `--conv-mode x3|bf16` are the opt-in
reduced-precision fast modes (the reference scripts' mixed_precision=True analogue) - their lines say so in `dtype`
and are NOT the headline number.

The timed steps are replays of the step captured once into a hipGraph (`--no-graph`: launched eagerly from Python; under
torch.distributed the capture is cut at the gradient all-reduces and replayed in segments behind a one-step bit-equality self-check).
After the headline the default command also times every other BASELINE.json config for a few steps (`config.other_configs`).
Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel = the 3x3 MFMA
convolution, algorithmic FLOPs / HIP-event duration measured live in this run - in K eagerly launched steps right after
the timed region, because events cannot sit inside a captured graph - against the matrix peak of the arithmetic in use)
and `cpu_baseline` (the CPU oracle = a port of the reference's path, timed on the host cores, rank 0, N=1).
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


def cpu_baseline(workload, n_samples, H, W, min_iters=3):
    """The CPU oracle (own restatement of the reference path, pinned by golden fixtures) on the host cores: 1 warm-up +
    `min_iters` timed iterations at batch 1 (~20-25 s of CPU work at 512x512)."""
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
        if workload in ("mean_teacher", "fixmatch"):
            O.mean_teacher_step(Ps, Pt, cfg, opt, x1, x2, et, es, True)
        elif workload == "punet_source":
            O.punet_step(Ps, cfg, opt, x2, y, es)
        elif workload in ("adamt", "adamatch"):
            O.adamt_step(Ps, Pt, cfg, opt, x, y, x1, x2, es, et, es, True, 0)
        else:
            O.mc_prediction(Ps, cfg, x1, et)
        return time.time() - t0

    one(64, 64)  # page-in, thread pools
    one(H, W)    # the warm-up iteration at size
    times = []
    while len(times) < min_iters:
        times.append(one(H, W))
    per = sum(times) / len(times)
    pinfo = " | ".join(l.strip() for l in torch.__config__.parallel_info().splitlines()
                       if any(k in l for k in ("ATen parallel backend", "OpenMP", "omp_get_max_threads", "MKL", "mkl_get_max_threads")))
    return {"value": round(1.0 / per, 5), "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 warm-up + {len(times)} timed x (batch 1, {H}x{W}, N={n_samples}) {workload} step of the CPU oracle "
                      f"(torch {torch.__version__} CPU fp32, {torch.get_num_threads()} threads), "
                      f"{', '.join(f'{t:.2f}' for t in times)} s; parallel_info: {pinfo}"}


def plain_unet_forward_config(dev, cpu=True, iters=20):
    """BASELINE.json configs[0]: the plain `Unet` (unet.py:6-69, apply_last_layer=True) forward on one 256x256x1 patch - the
    reference's own CPU-runnable case.  GPU: the HIP path behind this package's `Unet` (20 forwards between two
    synchronisations); CPU beside it: the oracle's restatement with the same weights (the `cpu_baseline` leg of this
    config), which also checks the GPU logits (max abs difference reported, bar 1e-3)."""
    from probabilistic_domain_adaptation_amd.my_models import Unet
    net = Unet(1, 1, NF, apply_last_layer=True, device=dev, seed=1)
    x = synth(1, 256, 256, 11, dev, 1)[0]
    for _ in range(3):
        y = net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        y = net(x)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / iters
    gf = 118.46  # unet forward at 256x256x1 (the 1x1 last layer adds 8 MFLOP)
    out = {"workload": "plain Unet forward (BASELINE.json configs[0]): one 256x256x1 patch, nf=[64, 128, 256, 512], eager launches "
                       "from Python (no graph)", "ms_per_step": round(ms, 3), "patches_per_s": round(1e3 / ms, 2),
           "model_tflops": round(gf / ms, 2)}
    if cpu:
        from oracle import punet_oracle as O
        torch.set_num_threads(host_cores())
        P = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        xc = x.cpu()
        with torch.no_grad():
            ref = O.plain_unet_forward(P, xc, len(NF))
            t0 = time.perf_counter()
            for _ in range(3):
                ref = O.plain_unet_forward(P, xc, len(NF))
            cms = 1e3 * (time.perf_counter() - t0) / 3
        out["cpu_oracle_ms"] = round(cms, 1)
        out["cpu_oracle_threads"] = torch.get_num_threads()
        out["max_abs_logit_difference_to_cpu_oracle"] = float((y.cpu() - ref).abs().max())
    return out


class Runner:
    """One workload on one rank: the fused step on static synthetic tensors, launched eagerly from Python or replayed from a
    hipGraph (FU.GraphedStep); fresh latent draws every step either way."""

    def __init__(self, workload, a, dev, rank, B, H, W, N, channels, conv_mode):
        import copy
        from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
        from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam
        from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
        self.FU, self.workload, self.N, self.B = FU, workload, N, B
        self.model = ProbabilisticUnet(channels, 1, NF, 6, 3, 1.0, consensus_masking=True, rl_swap=bool(getattr(a, "dice", False)),
                                       device=dev, seed=1, conv_mode=conv_mode)
        self.teacher = copy.deepcopy(self.model) if workload in ("mean_teacher", "adamt") else None
        self.opt = FusedAdam(self.model, lr=1e-5)
        x, x1, x2, y = synth(B, H, W, 100 + rank, dev, channels)  # every rank its own shard of the synthetic stream
        xs, _, _, ys = synth(B, H, W, 500 + rank, dev, channels)  # source-domain batch of the joint trainers
        self.t = {"x1": x1, "x2": x2, "y": y, "xs": xs, "ys": ys,
                  "eps_pl": torch.empty(N, B, 6, device=dev), "eps_s": torch.empty(1, B, 6, device=dev),
                  "eps_sup": torch.empty(1, B, 6, device=dev)}
        self.gen = torch.Generator(device=dev).manual_seed(7 + rank)
        self.graphed = None
        self.it = 0
        self.engines = [self.model.engine] + ([self.teacher.engine] if self.teacher is not None else [])
        self.initial_digest = self.checksum(self.model.flat_params)

    def body(self, s):
        FU, m, t, o, N = self.FU, self.model, self.teacher, self.opt, self.N
        w = self.workload
        if w == "mean_teacher":
            return FU.fused_self_training_step(m, t, o, s["x1"], s["x2"], N, True, s["eps_pl"], s["eps_s"], 0.999, 1e-5)
        if w == "punet_source":
            return FU.fused_supervised_step(m, o, s["x2"], s["y"], None, s["eps_s"], 1e-5)
        if w == "fixmatch":  # fixmatch_trainer.py:67-95 (no EMA, pseudo-labels from the model itself)
            return FU.fused_self_training_step(m, m, o, s["x1"], s["x2"], N, True, s["eps_pl"], s["eps_s"], None, 1e-5)
        if w == "adamt":  # adamt_trainer.py:89-128: one source + one target patch per "patch"
            return FU.fused_joint_step(m, t, o, s["xs"], s["ys"], s["x1"], s["x2"], N, True, s["eps_sup"], s["eps_pl"], s["eps_s"],
                                       0.999, True, 1e-5, self.it)
        if w == "adamatch":  # adamatch_trainer.py:62-107
            return FU.fused_joint_step(m, m, o, s["xs"], s["ys"], s["x1"], s["x2"], N, True, s["eps_sup"], s["eps_pl"], s["eps_s"],
                                       None, True, 1e-5)
        return FU.pseudo_labels(m, s["x1"], N, True, s["eps_pl"])

    def capture(self):
        w = self.workload
        ema = w in ("mean_teacher", "adamt")
        self.graphed = self.FU.GraphedStep(self.body, self.t, self.opt if w != "inference" else None, ema=ema,
                                           ema_momentum=0.999 if ema else 0.0, warmup_iteration=self.it if w == "adamt" else None,
                                           extra_state=[self.teacher.flat_params] if self.teacher is not None else [])

    def state_tensors(self):
        o = self.opt
        return [self.model.flat_params, o.exp_avg, o.exp_avg_sq, o.hstate, o.hyper] + \
               ([self.teacher.flat_params] if self.teacher is not None else [])

    @staticmethod
    def checksum(t):
        """Order-independent exact digest of a tensor's BITS (int64 sum of its 32-bit words, wrapping)."""
        return int(t.detach().contiguous().view(torch.int32).to(torch.int64).sum().item())

    def graph_matches_eager(self):
        """One eager step and one replayed step from the same state and the same latent draws: bit-identical weights, Adam
        moments and teacher?  Leaves the state as after ONE step (the replayed one).  Under torch.distributed both steps run
        their all-reduces, so every rank must call this."""
        if self.workload == "inference":
            return True
        ts = self.state_tensors()
        o = self.opt
        snap = [x.clone() for x in ts]
        mirrors = (o.step_count, o._iter_mirror, o._uploaded)

        def restore():
            with torch.no_grad():
                for c, x in zip(snap, ts):
                    x.copy_(c)
            o.step_count, o._iter_mirror, o._uploaded = mirrors
            o._uploaded = None  # the device state block was restored under the mirrors: upload again before the next step

        def digests():
            return [self.checksum(x) for x in (self.model.flat_params, o.exp_avg, o.exp_avg_sq)] + \
                   ([self.checksum(self.teacher.flat_params)] if self.teacher is not None else [])

        for k in ("eps_pl", "eps_s", "eps_sup"):
            self.t[k].normal_(generator=self.gen)
        done = False
        try:
            self.body(self.t)
            torch.cuda.synchronize()
            eager = digests()
            restore()
            self.graphed.step(warmup_iteration=self.it if self.workload == "adamt" else None)
            torch.cuda.synchronize()
            replay = digests()
            done = True
        finally:
            if not done:  # a step that raised half-way leaves no state behind; the caller decides what happens next
                restore()
        self.it += 1
        return eager == replay

    def step(self, graph):
        for k in ("eps_pl", "eps_s", "eps_sup"):  # fresh latent draws every step (device RNG, no host sync)
            self.t[k].normal_(generator=self.gen)
        if graph:
            self.graphed.step(warmup_iteration=self.it if self.workload == "adamt" else None)
        else:
            self.body(self.t)
        self.it += 1


class SynthRawLoader:
    """A loader of RAW (un-normalised, un-augmented) synthetic patches with labels, resident in pinned HOST memory as a disk-backed
    torch DataLoader would deliver them (the reference's loaders: LIVECell/livecell_mt.py:94-101): every batch crosses PCIe inside
    the loop, the views are built on the GPU (my_utils/augment.py::DualViewLoader, K13)."""

    def __init__(self, n_batches, B, H, W, C, seed):
        base = [synth(B, H, W, seed + 17 * i, "cpu", C) for i in range(min(n_batches, 8))]  # 8 distinct batches, cycled
        self.raw = [(40.0 + 25.0 * b[0]).clamp(0, 255).pin_memory() for b in base]  # camera-like intensities (standardised per sample later)
        self.y = [b[3].pin_memory() for b in base]
        self.n = n_batches

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield self.raw[i % len(self.raw)], self.y[i % len(self.y)]


def traffic_from_profiles(conv_mode):
    """roofline.traffic: HBM bytes per launch of the forward 3x3 instantiation from the newest committed PMC passes (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE, tools/pmc_traffic.py -> profiles/rNN/pmc_hbm_traffic*.json).  Returns (bytes or None, note); a file
    that cannot be read or has no entry for the kernel says so in the note instead of passing as "not collected"."""
    fn, stem = {"x6": ("pmc_hbm_traffic_x6.json", "conv3x3_x6b_kernel<3, false"),
                "f32": ("pmc_hbm_traffic.json", "conv_mfma_kernel<3, false")}[conv_mode]
    try:
        rdir = next(r for r in ("r05", "r04", "r03", "r02", "r01") if os.path.exists(os.path.join(ROOT, "profiles", r, fn)))
        pm = json.load(open(os.path.join(ROOT, "profiles", rdir, fn)))
        # the forward, bf16-plane, per-image instantiation: every template argument after the plane count is `false` (rounds 3 and 5
        # added arguments to the kernels' names: "<3, false>", "<3, false, false>", "<3, false, false, false>")
        key = next(k for k in pm["kernels"] if k.startswith(stem) and set(k[len(stem):].strip(">").replace(" ", "").split(",")) <= {"", "false"})
        traffic = round(pm["kernels"][key]["hbm_bytes_per_launch"])
        import hashlib
        hh = hashlib.sha256()
        for f_ in ("conv_x6.hip", "wgrad_x6.hip", "common.h"):
            hh.update(open(os.path.join(ROOT, "probabilistic_domain_adaptation_amd", "csrc", f_), "rb").read())
        same = pm.get("kernel_sources_sha16") == hh.hexdigest()[:16]
        note = (f"NOT measured in this run: read from profiles/{rdir}/{fn} ({key}), collected at commit {pm.get('git_head', 'unknown (pre-r03 file)')}; "
                f"kernel sources unchanged since that collection: {same if 'kernel_sources_sha16' in pm else 'unknown (pre-r03 file)'}; "
                "= rocprofv3 --pmc FETCH_SIZE (x2, gfx950 "
                "correction) + WRITE_SIZE in separate passes of this same command, bytes per launch, mean over the "
                "launches of the forward instantiation (algorithmic mean over the 73 forward launches of a step: "
                + ("250 MB with the up-blocks' entry convolutions in their low-resolution form, K5x" if rdir >= "r04"
                   else "265 MB = 144 MB activations in + weight planes, 121 MB out")
                + "); FETCH_SIZE includes the 1.33x halo re-reads of the 8x32-pixel tiles")
        return traffic, note
    except Exception as ex:
        return None, f"the committed PMC file for this kernel could not be used ({type(ex).__name__}: {ex})"


def fit_workload(a, dev, iterations, bare_ms_per_step=None, conv_mode=None):
    """What a user of the reference calls (LIVECell/livecell_mt.py:94-113): MeanTeacherTrainer.fit() with FusedAdam, graph=True and a
    DualViewLoader over raw patches (both views built by K13 on the GPU), two epochs of `iterations / 2` iterations, each followed
    by one validation pass and rank 0's checkpoints - against the bare step of the headline.  Returns the timings."""
    import shutil
    import tempfile
    from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam, MeanTeacherTrainer
    from probabilistic_domain_adaptation_amd.my_utils.augment import DualViewLoader, my_strong_augmentations, my_weak_augmentations
    B, H, W, N, C = a.batch, a.size, a.size, a.samples, a.channels
    per_epoch = max(iterations // 2, 1)
    model = ProbabilisticUnet(C, 1, NF, 6, 3, 1.0, consensus_masking=True, rl_swap=False, device=dev, seed=1, conv_mode=conv_mode)
    train = DualViewLoader(SynthRawLoader(per_epoch, B, H, W, C, 900), my_weak_augmentations(seed=1), my_strong_augmentations(seed=2), dev)
    val = DualViewLoader(SynthRawLoader(4, B, H, W, C, 1900), my_weak_augmentations(seed=3), my_strong_augmentations(seed=4), dev)
    root = tempfile.mkdtemp(prefix="punet_fit_")
    try:
        tr = MeanTeacherTrainer(name="bench-fit", save_root=root, train_loader=train, val_loader=None, model=model,
                                optimizer=FusedAdam(model, lr=1e-5), device=dev, do_consensus_masking=True, graph=True)
        tr.n_samples = N
        tr.fit(iterations=3)  # capture + first-use costs, not part of the measurement
        torch.cuda.synchronize()
        tr.val_loader = val
        times = {"train": 0.0, "val": 0.0, "ckpt": 0.0}
        inner_train, inner_val, inner_ckpt = tr._train_epoch_impl, tr._validate_impl, tr.save_checkpoint

        each = {"train": [], "val": [], "ckpt": []}

        def timed(key, fn):
            def w(*args, **kw):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = fn(*args, **kw)
                torch.cuda.synchronize()
                times[key] += time.perf_counter() - t0
                each[key].append(round(time.perf_counter() - t0, 4))
                return r
            return w

        host = [0.0, 0]
        tr._train_epoch_impl, tr._validate_impl, tr.save_checkpoint = timed("train", inner_train), timed("val", inner_val), timed("ckpt", inner_ckpt)
        it0 = tr._iteration
        t0 = time.perf_counter()
        tr.fit(iterations=2 * per_epoch)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        done = tr._iteration - it0
        # host time per iteration: the same loop once more without waiting for the GPU at its end (enqueue time only)
        tr._train_epoch_impl = inner_train
        tr.val_loader = None
        tr.save_root, tr.name = None, None
        tr.max_iteration = tr._iteration + min(per_epoch, 50)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        inner_train(type("P", (), {"update": lambda self, n: None})(), __import__("contextlib").nullcontext, None)
        t_enq = time.perf_counter() - t1
        n_enq = min(per_epoch, 50)
        torch.cuda.synchronize()
        ms_iter = 1e3 * times["train"] / done
        out = {"workload": f"MeanTeacherTrainer.fit(): {done} iterations in 2 epochs (FusedAdam, graph=True, DualViewLoader over raw "
                           f"{H}x{W}x{C} patches in pinned host memory, batch {B}, N={N}; weak / strong views built on the GPU by K13), each epoch "
                           f"followed by a validation pass over 4 batches and rank 0's best / latest checkpoints",
               "iterations": done, "train_loop_ms_per_iteration": round(ms_iter, 3),
               "train_loop_patches_per_s": round(1e3 * B / ms_iter, 3),
               "host_ms_per_iteration": round(1e3 * t_enq / n_enq, 3),
               "validation_s_per_epoch": round(times["val"] / 2, 3), "checkpoints_s_per_epoch": round(times["ckpt"] / 2, 3),
               "train_epoch_s": each["train"], "fit_wall_s": round(wall, 3), "whole_fit_patches_per_s": round(B * done / wall, 3),
               "final_loss": float(tr.last_loss) if tr.last_loss is not None else None}
        if bare_ms_per_step:
            out["bare_step_ms"] = round(bare_ms_per_step, 3)
            out["train_loop_rate_over_bare_step_rate"] = round(bare_ms_per_step / ms_iter, 4)
        return out
    finally:
        shutil.rmtree(root, ignore_errors=True)


def timed_region(runner, steps, graph, sync, dp, dev):
    """EXACTLY `steps` steps between two barriers + synchronisations; returns (seconds, max over ranks; host enqueue
    seconds; per-step GPU milliseconds)."""
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]  # per-step GPU timeline (diagnostic)
    sync()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        runner.step(graph)
        marks[i + 1].record()
    t_enq = time.perf_counter() - t0  # host time to enqueue the K steps (the GPU runs behind; must stay below dt)
    sync()
    dt = time.perf_counter() - t0
    if dp:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    return dt, t_enq, [round(marks[i].elapsed_time(marks[i + 1]), 2) for i in range(steps)]


def conv_peak_tflops(eng):
    """Matrix peak of the arithmetic the 3x3 kernels of `eng` compute in: n MFMAs per product on the bf16 / fp16 dense peak."""
    if not eng.x6:
        return FP32_MFMA_PEAK_TFLOPS, 1
    n = {"x6": 6, "x3": 3, "bf16": 1, "bf16s": 1, "f16x3": 3, "f16": 1, "f16s": 1}[eng.conv_mode]
    return BF16_MFMA_PEAK_TFLOPS / n, n


def instrumented_roofline(runner, steps, sync):
    """`steps` eagerly launched steps of `runner` with HIP events around every launch of the 3x3 forward convolution (the dominant
    kernel of every workload here): {bound, achieved, peak, unit, frac, ...} of that kernel, as the headline's `roofline`."""
    prof = []
    for e in runner.engines:
        e.prof, e.prof_kinds = prof, {"conv3x3_fwd"}
    try:
        runner.step(False)  # first use of the events
        sync()
        prof.clear()
        for _ in range(steps):
            runner.step(False)
        sync()
    finally:
        for e in runner.engines:
            e.prof = None
    fl = sum(p[1] for p in prof)
    sec = sum(p[2].elapsed_time(p[3]) for p in prof) * 1e-3
    if not prof or sec <= 0:
        return None
    peak, n = conv_peak_tflops(runner.model.engine)
    ach = fl / sec / 1e12
    return {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "kernel": "3x3 forward convolution (conv3x3_x6b_kernel)", "launches_per_step": len(prof) // steps,
            "avg_launch_ms": round(1e3 * sec / len(prof), 4), "flops_per_launch_avg": fl / len(prof),
            "timing": f"HIP events on the launch stream around every launch, {steps} eagerly launched steps after the timed ones"}


# BASELINE.json configs[4] asks for an HBM-bound roofline report; in this design the MC inference pass is bound by the matrix
# cores (one U-Net + prior encoder pass of 3x3 convolutions per tile; the N samples only re-run the 1x1 Fcomb head on resident
# features).  Algorithmic minimum traffic, SURVEY.md 8d: the weights once (78.1 MB of the 109.4 MB flat buffer: prior encoder +
# U-Net + Fcomb) + 4 B/px in + 5 B/px out (mean fp32 + mask u8).
def inference_hbm_report(H, W, ms_per_step):
    algo = 78.1e6 + 9.0 * H * W
    out = {"bound_asked_for": "hbm (BASELINE.json configs[4])", "bound_in_this_design": "mfma",
           "why": "one tile = 1.38 TFLOP of 3x3 convolutions over 78 MB of weights + 9 B/px of image data: arithmetic intensity "
                  f"{1382.8e9 * (H * W) / (1024 * 1024) / algo / 1e3:.1f} kFLOP/B against a ridge of 416.7 TFLOP/s / 8 TB/s = 52 FLOP/B - the pass "
                  "cannot be HBM-bound unless activations spill: they do (every layer's output makes one round trip), and still are not the bound",
           "algorithmic_min_bytes": int(algo), "hbm_peak_GBps": 8000.0,
           "time_at_hbm_peak_for_algorithmic_bytes_ms": round(algo / 8e12 * 1e3, 4),
           "frac_of_hbm_roofline_on_algorithmic_bytes": round(algo / 8e12 * 1e3 / ms_per_step, 5)}
    try:  # counted bytes of one inference step (all kernels), from the PMC passes under profiles/ (tools/collect_profiles.sh)
        rdir = next(r for r in ("r05", "r04") if os.path.exists(os.path.join(ROOT, "profiles", r, "pmc_hbm_traffic_inference_1024.json")))
        pm = json.load(open(os.path.join(ROOT, "profiles", rdir, "pmc_hbm_traffic_inference_1024.json")))
        if (H, W) == (1024, 1024):
            cb = pm["hbm_bytes_per_step"]
            out.update({"counted_hbm_bytes_per_step": int(cb), "counted_GBps_at_this_rate": round(cb / (ms_per_step * 1e-3) / 1e9, 1),
                        "counted_frac_of_hbm_peak": round(cb / (ms_per_step * 1e-3) / 8e12, 4),
                        "counted_note": f"NOT measured in this run: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE in separate passes over "
                                        f"this workload, summed over every kernel of a step, profiles/{rdir}/pmc_hbm_traffic_inference_1024.json "
                                        f"(collected at commit {pm.get('git_head', 'unknown')})"})
    except Exception as ex:
        out["counted_note"] = f"the committed counter file could not be used ({type(ex).__name__}: {ex})"
    return out


def gflop_per_patch(workload, N, H, W, channels):
    exact = (H, W, channels) == (704, 704, 3)
    gf = (GFLOP_704x3 if exact else GFLOP)[workload]
    gf = gf[N] if isinstance(gf, dict) else gf
    if not exact and (H, W) != (512, 512):
        gf = gf * (H * W) / (512 * 512)
    return gf


def launch_ranks(n, argv):
    """`python bench.py --gpus N` WITHOUT torch.distributed.run in front of it: this process becomes the launcher.  It builds the
    library once (in a child of its own, so that no rank compiles and this process loads nothing), starts the N ranks as FRESH
    child processes of this very file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment (what
    torch.distributed.run would set), relays rank 0's single JSON line and returns the ranks' exit code (the first non-zero one
    in rank order: 3 = replicas diverged, 4 = a rank's graph self-check raised).  The launcher never touches the GPU: no HIP
    call, no torch.cuda, no exec - the ranks are ordinary children (subprocess), and when one of them fails the others, which
    would wait in a collective for ever, are ended by their exact PIDs."""
    import socket
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "__graft_entry__.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout, file=sys.stderr)
        print("[bench launcher] the library build failed; no rank was started", file=sys.stderr)
        return 5
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PUNET_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL))
    print(f"[bench launcher] started {n} ranks (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}", file=sys.stderr, flush=True)
    # rank 0's stdout is the JSON line and nothing else (main() points fd 1 at stderr for everything else): read it to the end
    # on a thread so that a full pipe never blocks the rank, and watch all ranks meanwhile
    import threading
    captured = []
    reader = threading.Thread(target=lambda: captured.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * n
    failed = None
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                c = p.poll()
                if c is not None:
                    codes[i] = c
                    if c != 0 and failed is None:
                        failed = i
        if failed is not None:
            # the surviving ranks are (or will be) inside a collective the failed rank never joins: give them a moment to fail by
            # themselves (the process group notices a lost peer), then end exactly the processes started here
            deadline = time.time() + 20.0
            while time.time() < deadline and any(p.poll() is None for p in procs):
                time.sleep(0.2)
            for i, p in enumerate(procs):
                if p.poll() is None:
                    p.terminate()
                    try:
                        p.wait(10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        p.wait()
                if codes[i] is None:
                    codes[i] = p.returncode
            break
        time.sleep(0.05)
    reader.join(10)
    text = (captured[0] if captured else b"").decode(errors="replace")
    lines = [l for l in text.splitlines() if l.lstrip().startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    rc = codes[failed] if failed is not None else 0  # the code of the rank that failed FIRST (the others were ended here)
    if rc < 0:
        rc = 128 - rc  # a rank killed by signal s reports 128 + s, as a shell would
    if rc or not lines:
        print(f"[bench launcher] exit codes by rank: {codes}" + ("" if lines else "; rank 0 printed no JSON line"), file=sys.stderr)
        rc = rc or 6
    return rc


def stub_body(a):
    """`--stub-body` (tests/test_bench_launcher_cpu.py): a rank that does everything main() does AROUND the measurement - process
    group from the environment, a collective, one JSON line from rank 0 on the saved descriptor, the exit codes - and no GPU work,
    so that the launcher path runs on a machine without a GPU."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t)
    out = {"metric": "stub", "value": float(t.item()), "n_gpus": world, "world_size_seen": dist.get_world_size() if world > 1 else 1,
           "replicas_identical": True, "launched_by": "bench.py" if os.environ.get("PUNET_BENCH_LAUNCHED") else "external launcher"}
    fail = os.environ.get("PUNET_BENCH_STUB_FAIL")  # "<rank>:<code>": that rank exits with that code before the last barrier
    if fail and int(fail.split(":")[0]) == rank:
        sys.exit(int(fail.split(":")[1]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out if rank == 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="mean_teacher", choices=["mean_teacher", "punet_source", "inference", "fixmatch", "adamt", "adamatch", "fit"],
                    help="fit = MeanTeacherTrainer.fit() with a DualViewLoader and graph=True for --fit-iterations iterations next to the bare "
                         "mean_teacher step (the line's value stays the bare step; config.fit_loop holds the loop's rate and the ratio)")
    ap.add_argument("--fit-iterations", type=int, default=200)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--samples", type=int, default=8)
    ap.add_argument("--conv-mode", default=None, choices=["x6", "f32", "x3", "bf16", "bf16s", "f16x3", "f16", "f16s"],
                    help="3x3 conv arithmetic (default: $PUNET_CONV_MODE or x6); x3/bf16 are reduced-precision fast modes")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch the timed steps eagerly from Python instead of replaying the captured hipGraph")
    ap.add_argument("--graph", action="store_true",
                    help="(kept for compatibility; the segmented replay is now the default under torch.distributed too: it is used "
                         "when a one-step self-check shows it bit-identical to the eager step on EVERY rank, else eager with a note)")
    ap.add_argument("--runtime", default=None, choices=["native", "python"],
                    help="who walks the layer graph of a step: the step-level native runtime inside libpunet_hip.so (default where it "
                         "applies: four library calls per step) or the Python engine (~500 per-op calls); same kernels, same bits")
    ap.add_argument("--dice", action="store_true",
                    help="rl_swap=True: the Dice criterion every LIVECell / MitoEM script of the reference uses (recalled torch_em "
                         "definition, parity unpinned) instead of BCE (the criterion all parity gates use); not the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short runs of the other BASELINE.json configs (plain Unet forward, PUNet-source, AdaMT, AdaMatch 704x704x3, "
                         "inference 1024x1024 N=16) and of the f16x3 fast mode after the headline")
    ap.add_argument("--kernel-timing", default="roofline", choices=["roofline", "all"],
                    help="HIP-event brackets in the instrumented region: only the roofline kernel's launches (default; 73 per "
                         "step) or every MFMA launch (adds ~1 ms of bubbles per step)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI; the default) or gloo (rehearsal on one GPU)")
    ap.add_argument("--rehearse-rccl", action="store_true",
                    help="one-GPU rehearsal of the data-parallel path over the REAL backend: a one-rank RCCL group with the trainers' "
                         "world size forced to 2 (one-rank collectives are the identity; the gradient scale is 1/2), so that every "
                         "data-parallel code path of this file and of the fused steps runs through ProcessGroupNCCL; not a measurement")
    ap.add_argument("--stub-body", action="store_true", help=argparse.SUPPRESS)  # launcher test: ranks without GPU work (stub_body)
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: start the N ranks here (before anything in this process touches the GPU) and relay
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))

    # stdout carries the ONE JSON line and nothing else: native libraries write to file descriptor 1 behind Python's back (RCCL
    # prints a version banner there when its first communicator is built), so fd 1 is pointed at stderr for the whole run and
    # the line is written to the saved descriptor at the end
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    if a.runtime:
        os.environ["PUNET_RUNTIME"] = a.runtime
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} was started under a launcher with WORLD_SIZE={world}: the two must agree "
                         f"(without a launcher, `python bench.py --gpus {a.gpus}` starts its own ranks)")
    if a.stub_body:
        out = stub_body(a)
        if out is not None:
            os.write(line_fd, (json.dumps(out) + "\n").encode())
        return
    local = local % max(torch.cuda.device_count(), 1)  # gloo rehearsal: several ranks may share one GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dp = world > 1 or a.rehearse_rccl
    if a.rehearse_rccl:
        if world != 1 or a.backend != "nccl":
            raise SystemExit("--rehearse-rccl is a single-process run over the nccl (= RCCL) backend")
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_PORT=str(sk.getsockname()[1]))
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # "nccl" == RCCL on ROCm
        else:
            dist.init_process_group(a.backend)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if dp:
        dist.barrier()
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    if a.rehearse_rccl:
        FU._world = lambda: 2

    def sync():
        torch.cuda.synchronize()
        if dp:
            dist.barrier()
            torch.cuda.synchronize()

    B, H, W, N = a.batch, a.size, a.size, a.samples
    fit_iterations = a.fit_iterations if a.workload == "fit" else (120 if not a.no_other_configs else 0)
    if a.workload == "fit":
        a.workload = "mean_teacher"
    use_graph = not a.no_graph
    run = Runner(a.workload, a, dev, rank, B, H, W, N, a.channels, a.conv_mode)
    eng = run.model.engine
    graph_note = None
    graph_selfcheck = None
    if use_graph:
        try:
            run.capture()
        except Exception as ex:  # never lose the measurement to the capture: fall back to eager launches and say so
            graph_note = f"graph capture failed ({type(ex).__name__}: {ex}); the timed steps were launched eagerly"
            print("[bench] " + graph_note, file=sys.stderr)
            use_graph = False
            FU._CAPTURE = None
        if dp:
            # Data parallel: the capture is cut at the gradient all-reduces and the collectives run eagerly between the segment
            # replays.  Before that path is trusted with the measurement it must reproduce the eager step bit for bit, from the
            # same state and latent draws, on every rank (one extra eager + one replayed step; the state is restored in between).
            # (the self-check runs steps with collectives inside: every rank must enter it or none - agree on the capture first)
            cdev0 = dev if a.backend == "nccl" else "cpu"
            t = torch.tensor([1.0 if use_graph else 0.0], device=cdev0)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            all_captured = bool(t.item() == 1.0)
            ok = 0.0
            if all_captured:
                try:
                    ok = 1.0 if run.graph_matches_eager() else 0.0
                except Exception as ex:
                    # The self-check's steps contain collectives: a rank that raised half-way has left the others inside an
                    # all-reduce it will never join, so "fall back to eager" is not available here - continuing would hang on
                    # mismatched collectives.  The state was restored (graph_matches_eager's finally); leave with a non-zero
                    # code and let the launcher take the other ranks down.
                    print(f"[bench] rank {rank}: graph self-check raised {type(ex).__name__}: {ex}; exiting with code 4 "
                          "(collectives of the other ranks are outstanding)", file=sys.stderr, flush=True)
                    FU._CAPTURE = None
                    sys.exit(4)
            t = torch.tensor([ok], device=cdev0)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            graph_selfcheck = bool(t.item() == 1.0)
            if use_graph and not graph_selfcheck:
                graph_note = ("segmented graph replay did NOT reproduce the eager data-parallel step bit for bit on every rank (or "
                              "could not be captured everywhere): the timed steps were launched eagerly")
                print("[bench] " + graph_note, file=sys.stderr)
            use_graph = use_graph and graph_selfcheck
    for _ in range(a.warmup):
        run.step(use_graph)
    comm = [] if dp else None
    FU.COMM_EVENTS = comm  # HIP events around the exposed part of the gradient all-reduce (rank-local)
    comm_host = [0.0] if dp else None
    FU.COMM_HOST_S = comm_host  # host seconds inside the collective calls of the timed region
    # ---- the timed region: K steps, barrier + synchronize on both sides, max over ranks
    dt, t_enq, ms_each = timed_region(run, a.steps, use_graph, sync, dp, dev)
    FU.COMM_EVENTS = None
    FU.COMM_HOST_S = None
    exposed_ms = None
    if comm:
        exposed_ms = round(sum(e0.elapsed_time(e1) for e0, e1 in comm) / a.steps, 3)
    # ---- data parallel: the replicas must still be bit-identical after the timed steps (same initial weights, summed gradients,
    # rank-local Adam / EMA on identical inputs): MIN and MAX over ranks of exact digests of the student and the teacher
    replicas = None
    if dp:
        digests = [Runner.checksum(run.model.flat_params)] + ([Runner.checksum(run.teacher.flat_params)] if run.teacher is not None else [])
        cdev = dev if a.backend == "nccl" else "cpu"
        lo = torch.tensor(digests, dtype=torch.int64, device=cdev)
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        moved = Runner.checksum(run.model.flat_params) != run.initial_digest  # and the weights really changed
        replicas = {"identical": bool(torch.equal(lo, hi)), "student_digest": int(lo[0].item()),
                    "teacher_digest": int(lo[1].item()) if len(digests) > 1 else None, "weights_moved": bool(moved)}

    # ---- the instrumented region: the same K steps launched eagerly with HIP events around the roofline kernel's launches
    # on the launch stream (events cannot be recorded inside a captured hipGraph on this ROCm - hipErrorInvalidHandle,
    # profiles/r02/graph_event_probe.txt - so the kernel is timed in the eagerly launched twin of the replayed step; the
    # kernels and their order are the same, tests/test_gpu_graph.py shows the results are bit-identical)
    prof, dt_eager, ms_each_eager, t_enq_eager = None, None, None, None
    if not a.no_kernel_timing:
        prof = []
        for e in run.engines:
            e.prof = prof
            e.prof_kinds = {"conv3x3_fwd"} if a.kernel_timing == "roofline" else None
        run.step(False)  # the first use of ~1000 events in a process costs tens of ms: not in the measured steps
        sync()
        prof.clear()
        dt_eager, t_enq_eager, ms_each_eager = timed_region(run, a.steps, False, sync, dp, dev)
        for e in run.engines:
            e.prof = None

    if rank == 0:
        patches = world * B * a.steps
        gf = gflop_per_patch(a.workload, N, H, W, a.channels)
        out = {
            "metric": "training patches/sec (512x512, N=8 MC samples)" if a.workload != "inference" else "inference patches/sec",
            "value": round(patches / dt, 4), "unit": "patches/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"x6": "f32", "f32": "f32", "x3": "bf16x3 (fp32 storage, 16-bit significand products; reduced precision)",
                      "bf16": "bf16 (fp32 storage; reduced precision)",
                      "bf16s": "bf16 (bf16 activation storage on the no-gradient passes; reduced precision)",
                      "f16x3": "f16x3 (fp32 storage, two fp16 planes per operand with per-tensor power-of-two scaling, 3 MFMAs per "
                               "product: 22-23 significand bits; reduced precision)",
                      "f16": "f16 (fp32 storage, fp16 operands with per-tensor power-of-two scaling - the analogue of the reference's "
                             "fp16 autocast; reduced precision)",
                      "f16s": "f16 (fp16 operands AND fp16 activation storage, fp32 gradients scaled per tensor; reduced precision)"}[eng.conv_mode],
            "data": "synthetic",
            "config": {"workload": f"PUNet {a.workload} step: nf={NF}, latent 6, fcomb 3, {'Dice' if a.dice else 'BCE'}+KL, consensus masking, "
                                   f"{H}x{W}x{a.channels} patches, batch {B}/GPU, N={N} MC samples, Adam lr 1e-5, EMA 0.999",
                       "global_batch": world * B, "parallelism": f"dp{world}",
                       "launch": "hipGraph replay of the captured fused step (one graph launch per step"
                                 + (f"; {len(run.graphed.segments)} segments cut at the gradient all-reduces)" if dp else ")")
                                 if use_graph else (graph_note or ("eager: ~500 kernel launches per step, issued by the native runtime inside "
                                                                    "4-5 library calls" if FU.native_ok(run.model)
                                                                    else "eager: ~500 kernel launches per step from Python")),
                       "graph_walk": ("native runtime inside libpunet_hip.so (csrc/runtime.hip): pseudo-labels, ELBO forward/backward, "
                                      "regulariser, Adam+EMA = 4-5 library calls per step" if FU.native_ok(run.model)
                                      else "Python engine (engine.py): one library call per kernel"),
                       "algorithmic_gflop_per_patch": gf,
                       "model_tflops_per_gpu": round(gf * patches / dt / world / 1e3, 2),
                       "model_tflops_note": "algorithmic FLOPs of the layers AS WRITTEN in the reference / time; with K5x the up-blocks' `up` "
                                            "halves execute a quarter of their written multiply-adds (config.up_blocks): at the headline shape that is "
                                            "3/4 x 463.9 GFLOP per U-Net forward, input-gradient and weight-gradient pass of a batch = 1.39 of the "
                                            "12.93 TFLOP a step is counted with, i.e. the EXECUTED matrix rate is ~10.8 % below this figure",
                       "host_enqueue_ms_per_step": round(1e3 * t_enq / a.steps, 2),
                       "ms_each_step": ms_each},
        }
        if dp:
            if a.rehearse_rccl:
                out["rehearsal"] = ("NOT a data-parallel measurement: ONE GPU, a one-rank RCCL group with the trainers' world size forced "
                                    "to 2 (identity collectives, gradient scale 1/2) - the data-parallel code path over the real backend")
            out["replicas_identical"] = replicas["identical"]
            out["world_size_seen"] = dist.get_world_size()
            out["exposed_allreduce_ms_per_step"] = exposed_ms  # rank 0, HIP events (config.distributed.exposed_note)
            out["launched_by"] = "bench.py (own ranks)" if os.environ.get("PUNET_BENCH_LAUNCHED") else "external launcher"
            out["config"]["distributed"] = {
                "world_size_seen": dist.get_world_size(), "backend": dist.get_backend(),
                "replicas": replicas, "graph_selfcheck_bit_identical_to_eager_on_every_rank": graph_selfcheck,
                "allreduce": "flat fp32 gradient, U-Net slice launched under the encoders' backward + remainder after the regulariser",
                "exposed_allreduce_ms_per_step_rank0": exposed_ms,
                "host_ms_per_step_inside_collective_calls_rank0": round(1e3 * comm_host[0] / a.steps, 3),
                "host_launch_ms_per_step_rank0": round(1e3 * (t_enq - comm_host[0]) / a.steps, 3),
                "host_note": "host_enqueue_ms_per_step = launch work + time inside the collective calls; gloo BLOCKS the host in the "
                             "call until the gradient exists and has crossed the host, RCCL only enqueues",
                "exposed_note": "HIP events on the launch stream around the point where the step waits for the gradient all-reduce "
                                "(remainder + whatever of the overlapped U-Net slice is still in flight)"}
        out["config"]["conv_arithmetic"] = {
            "x6": "3x3 convs fwd+dgrad+wgrad: fp32 operands split into 3 bf16 planes, 6 bf16 MFMAs per product, fp32 "
                  "accumulate (fp32-accurate); 1x1 convs: fp32-input MFMA",
            "f32": "fp32-input MFMA everywhere",
            "x3": "REDUCED PRECISION fast mode: 3x3 convs with 2 bf16 planes per operand, 3 bf16 MFMAs per product "
                  "(16 significand bits), fp32 accumulate and storage; 1x1 convs fp32-input MFMA",
            "bf16": "REDUCED PRECISION fast mode: 3x3 convs in plain bf16 (1 MFMA per product), fp32 accumulate and "
                    "storage; 1x1 convs fp32-input MFMA",
            "bf16s": "REDUCED PRECISION fast mode: as bf16, plus bf16 STORAGE of the activations on the passes no gradient flows "
                     "through (teacher / weak pass, inference)",
            "f16x3": "REDUCED PRECISION fast mode (logits within 1e-4 of the reference's): 3x3 convs with 2 fp16 planes per operand "
                     "(h + l, products hh + hl + lh on v_mfma_f32_*_f16), every tensor scaled by an exact power of two from its amax "
                     "(published by the producing kernel), fp32 accumulate and storage; 1x1 convs six-product bf16 / fp32 MFMA",
            "f16": "REDUCED PRECISION fast mode (the reference's mixed_precision=True analogue: fp16 operands, fp32 accumulate): one "
                   "fp16 MFMA per product, per-tensor power-of-two scaling from amax instead of a global loss scale, fp32 storage",
            "f16s": "REDUCED PRECISION fast mode: as f16, plus fp16 STORAGE of every activation (unscaled, saturating - as under "
                    "autocast); gradients fp32 in memory, scaled per tensor in the kernels' loaders"}[eng.conv_mode]
        out["config"]["up_blocks"] = ("K5x: the up-blocks' entry convolutions contract their `up` source at low resolution (conv3x3(up(x)) = sum_t "
                                      "up(W_t . x): forward, input gradient and weight gradient; the up-sampled tensor is not materialised) - FLOPs "
                                      "above are counted for the layers as WRITTEN" if getattr(eng, "upconv", False)
                                      else "as written (bilinear x2 materialised, 3x3 convolution on the concat)")
        if prof:
            out["config"]["instrumented_eager_region"] = {
                "ms_per_step": round(1e3 * dt_eager / a.steps, 3), "host_enqueue_ms_per_step": round(1e3 * t_enq_eager / a.steps, 2),
                "ms_each_step": ms_each_eager,
                "note": "the same K steps launched eagerly right after the timed region, with HIP events around the roofline kernel"}
            agg = {}
            for kind, fl, e0, e1 in prof:
                d = agg.setdefault(kind, [0.0, 0.0, 0])
                d[0] += fl
                d[1] += e0.elapsed_time(e1) * 1e-3
                d[2] += 1
            kernels = {k: {"launches": v[2], "avg_ms": round(1e3 * v[1] / v[2], 4), "tflops": round(v[0] / v[1] / 1e12, 2),
                           "share_of_step": round(v[1] / dt_eager, 3)} for k, v in agg.items()}
            k3 = agg.get("conv3x3_fwd")
            traffic, traffic_note = None, "not collected for this workload / shape"
            if a.workload == "mean_teacher" and (B, H, W, N) == (4, 512, 512, 8) and eng.conv_mode in ("x6", "f32"):
                traffic, traffic_note = traffic_from_profiles(eng.conv_mode)
            if k3:
                ach = k3[0] / k3[1] / 1e12
                if eng.x6:
                    # n bf16 MFMAs per product: the roofline of THIS arithmetic is the bf16 dense peak / n
                    n = {"x6": 6, "x3": 3, "bf16": 1, "bf16s": 1, "f16x3": 3, "f16": 1, "f16s": 1}[eng.conv_mode]
                    peak = BF16_MFMA_PEAK_TFLOPS / n
                    et = "f16" if eng.f16 else "bf16"
                    kname = (f"conv3x3_x6b_kernel<{(n + 1) // 2 if n > 1 else 1} planes, {et}> (3x3 NHWC direct conv forward, "
                             f"v_mfma_f32_16x16x32_{et}; conv3x3_x6_kernel for channel counts that are not multiples of 32 and for "
                             f"the one-plane modes)")
                    pnote = (f"bf16 / fp16 dense MFMA peak {BF16_MFMA_PEAK_TFLOPS} TFLOP/s / {n} MFMAs per product = {peak:.1f} TFLOP/s of "
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
                                   "avg_launch_ms": round(1e3 * k3[1] / k3[2], 4), "peak_note": pnote,
                                   "timing": "HIP events on the launch stream around every launch of the kernel, live in this run, "
                                             "over the K eagerly launched steps of config.instrumented_eager_region",
                                   "population_note": ("since round 4 (K5x) the 768 / 384 / 192-channel concat convolutions of the up-blocks - "
                                                       "the launches this kernel ran fastest - are replaced by their bridge halves plus 1x1 "
                                                       "GEMMs and gathers: the mean rate over the remaining launches is ~0.02 lower than round "
                                                       "3's population at equal kernel code, while the step is 5 % faster"
                                                       if getattr(eng, "upconv", False) else None)}
            out["kernels"] = kernels
        if not a.no_other_configs and not dp and a.workload == "mean_teacher" and (B, H, W, a.channels) == (4, 512, 512, 1):
            # BASELINE.json configs[2] / configs[3] per GPU, timed in this same process (short: 2 warm-up + 5 steps each)
            others = []
            del run
            torch.cuda.empty_cache()
            # (configs[1] PUNet-source and configs[4] inference since round 4: every BASELINE.json config is driver-timed)
            for idx, wl, ch, size, b2, n2 in ((1, "punet_source", 1, 512, B, N), (2, "adamt", 1, 512, B, N),
                                              (3, "adamatch", 3, 704, B, N), (4, "inference", 1, 1024, 1, 16)):
                r2 = Runner(wl, a, dev, rank, b2, size, size, n2, ch, a.conv_mode)
                g2ok = use_graph
                if g2ok:
                    try:
                        r2.capture()
                    except Exception:
                        g2ok, FU._CAPTURE = False, None
                for _ in range(2):
                    r2.step(g2ok)
                d2, e2, each2 = timed_region(r2, 5, g2ok, sync, dp, dev)
                g2 = gflop_per_patch(wl, n2, size, size, ch)
                what = {"punet_source": f"{b2} labelled patches {size}x{size}x{ch} per step (supervised ELBO step, punet_trainer.py:19-50)",
                        "adamt": f"{b2} source + {b2} target patches {size}x{size}x{ch} per step, N={n2}",
                        "adamatch": f"{b2} source + {b2} target patches {size}x{size}x{ch} per step, N={n2}",
                        "inference": f"one {size}x{size}x{ch} tile per step, N={n2} MC samples, mean + consensus mask "
                                     f"(punet_predictions.py:29-33 / :114-124)"}[wl]
                unit = {"punet_source": "patches_per_s", "inference": "tiles_per_s"}.get(wl, "pairs_per_s")
                entry = {"workload": f"{wl} (BASELINE.json configs[{idx}], per GPU): {what}",
                         "ms_per_step": round(1e3 * d2 / 5, 3), unit: round(b2 * 5 / d2, 3),
                         "model_tflops": round(g2 * b2 * 5 / d2 / 1e3, 2), "host_enqueue_ms_per_step": round(1e3 * e2 / 5, 2),
                         "ms_each_step": each2}
                if not a.no_kernel_timing:  # the same roofline object as the headline's, for this config's dominant kernel
                    try:
                        entry["roofline"] = instrumented_roofline(r2, 2, sync)
                    except Exception as ex:
                        entry["roofline"] = {"error": f"{type(ex).__name__}: {ex}"}
                if wl == "inference":
                    entry["hbm_report"] = inference_hbm_report(size, size, 1e3 * d2 / 5)
                others.append(entry)
                del r2
                torch.cuda.empty_cache()
            try:  # configs[0]: the plain Unet forward on one 256x256x1 patch - the reference's CPU-runnable case, here on the GPU
                others.insert(0, plain_unet_forward_config(dev, cpu=not a.no_cpu_baseline))
            except Exception as ex:  # never lose the headline to an extra
                others.insert(0, {"workload": "plain Unet forward (BASELINE.json configs[0])", "error": f"{type(ex).__name__}: {ex}"})
            out["config"]["other_configs"] = others
            # the fast mode the trainers' mixed_precision=True selects (f16x3: logits within 1e-4 of the reference's), same headline
            # workload, same process: 2 warm-up + 5 steps (NOT the headline: reduced precision, its own roofline denominator)
            try:
                r3 = Runner("mean_teacher", a, dev, rank, B, H, W, N, a.channels, "f16x3")
                g3ok = use_graph
                if g3ok:
                    try:
                        r3.capture()
                    except Exception:
                        g3ok, FU._CAPTURE = False, None
                for _ in range(2):
                    r3.step(g3ok)
                d3, e3, each3 = timed_region(r3, 5, g3ok, sync, dp, dev)
                tf3 = gf * B * 5 / d3 / 1e3
                roof3 = None
                if not a.no_kernel_timing:
                    try:
                        roof3 = instrumented_roofline(r3, 2, sync)
                    except Exception as ex:
                        roof3 = {"error": f"{type(ex).__name__}: {ex}"}
                out["config"]["fast_mode_f16x3"] = {
                    "workload": "the headline Mean-Teacher step in the f16x3 arithmetic (two fp16 planes per operand, per-tensor power-of-two "
                                "scaling, 3 MFMAs per product; what mixed_precision=True selects) - REDUCED PRECISION, not the headline",
                    "ms_per_step": round(1e3 * d3 / 5, 3), "patches_per_s": round(B * 5 / d3, 3), "model_tflops": round(tf3, 2),
                    "frac_of_its_peak": round(tf3 / (BF16_MFMA_PEAK_TFLOPS / 3), 4), "peak": round(BF16_MFMA_PEAK_TFLOPS / 3, 1),
                    "ms_each_step": each3, "roofline": roof3}
                del r3
                torch.cuda.empty_cache()
            except Exception as ex:  # never lose the headline to an extra
                out["config"]["fast_mode_f16x3"] = {"error": f"{type(ex).__name__}: {ex}"}
        if fit_iterations and not dp and a.workload == "mean_teacher" and not a.dice:
            try:  # the API users call, against the bare step above
                try:
                    del run
                except NameError:
                    pass
                torch.cuda.empty_cache()
                out["config"]["fit_loop"] = fit_workload(a, dev, fit_iterations, 1e3 * dt / a.steps, a.conv_mode)
            except Exception as ex:  # never lose the headline to an extra
                out["config"]["fit_loop"] = {"error": f"{type(ex).__name__}: {ex}"}
        if not a.no_cpu_baseline and not dp and not a.dice:
            out["cpu_baseline"] = cpu_baseline(a.workload, N, H, W)
        sys.stdout.flush()
        os.write(line_fd, (json.dumps(out) + "\n").encode())
    if dp:
        dist.barrier()
        dist.destroy_process_group()
        if not replicas["identical"] or not replicas["weights_moved"]:
            print("[bench] FAILED: the data-parallel replicas diverged (or never moved): " + json.dumps(replicas), file=sys.stderr)
            sys.exit(3)


if __name__ == "__main__":
    main()
