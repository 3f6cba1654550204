"""Fused (autograd-free) training steps and the flat-buffer optimizer.

`FusedAdam` is torch.optim.Adam's arithmetic (defaults, LIVECell/livecell_mt.py:83) on the model's flat parameter buffer,
fused with the EMA teacher update (mean_teacher_trainer.py:52-55) in one kernel (K11 + K12).  The step functions are the
per-iteration bodies of the reference trainers with every tensor op replaced by a kernel launch and no host sync.

Data parallelism: one process per GPU; the flat gradient buffer is summed with ONE RCCL all-reduce (torch.distributed
backend "nccl" == RCCL on ROCm) and the 1/world_size scale is folded into the Adam kernel.  Per-rank loss semantics =
reference semantics at the per-GPU batch (reconstruction is a SUM over the local batch, KL a local batch mean);
gradients are averaged over ranks (DDP convention) - see DESIGN.md §6.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from ..my_models import ProbabilisticUnet


class FusedAdam(torch.optim.Optimizer):
    """Adam on a ProbabilisticUnet's flat buffer.  Usable (a) by the fused steps (gradients already in `grad_flat`) and
    (b) as a drop-in `optimizer.step()` after `loss.backward()` (gradients gathered from `.grad`)."""

    def __init__(self, model: ProbabilisticUnet, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps))
        self.model = model
        flat = model.flat_params
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.grad_flat = torch.zeros_like(flat)
        self.step_count = 0

    @property
    def lr(self) -> float:
        return self.param_groups[0]["lr"]

    def zero_grad(self, set_to_none: bool = True):
        for p in self.model.parameters():
            p.grad = None

    def gather_grads(self):
        """Copy autograd's per-parameter .grad into the flat gradient buffer (drop-in mode only)."""
        views = self.model.layout.views(self.grad_flat)
        with torch.no_grad():
            for p, v in zip(self.model.parameters(), views.values()):
                if p.grad is None:
                    v.zero_()
                else:
                    v.copy_(p.grad)

    def fused_step(self, teacher: Optional[ProbabilisticUnet] = None, ema_momentum: float = 0.999, grad_scale: float = 1.0):
        """One Adam step from grad_flat (+ EMA into `teacher` when given) - a single kernel launch."""
        g = self.param_groups[0]
        self.step_count += 1
        self.model.engine.adam_ema(self.model.flat_params, self.grad_flat, self.exp_avg, self.exp_avg_sq,
                                   teacher.flat_params if teacher is not None else None, self.step_count, g["lr"],
                                   g["betas"], g["eps"], grad_scale, ema_momentum, True)

    @torch.no_grad()
    def step(self, closure=None):
        self.gather_grads()
        self.fused_step()

    def state_dict(self):
        """torch.optim.Adam's layout (what the reference's checkpoints hold under "optimizer_state", LIVECell/livecell_mt.py:83
        + torch_em's save_checkpoint): state[i] = {step, exp_avg, exp_avg_sq} per parameter in state-dict order, moments in
        the parameter's logical [cout, cin, ky, kx] shape; param_groups with torch's Adam defaults.  A checkpoint written
        here resumes under torch.optim.Adam and vice versa."""
        lay = self.model.layout
        state = {}
        if self.step_count > 0:
            m_views, v_views = lay.views(self.exp_avg), lay.views(self.exp_avg_sq)
            for i, name in enumerate(m_views):
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": m_views[name].detach().cpu().contiguous(),
                            "exp_avg_sq": v_views[name].detach().cpu().contiguous()}
        groups = []
        for g in self.param_groups:
            d = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False}
            d.update({k: v for k, v in g.items() if k not in d and k != "params"})  # e.g. a scheduler's initial_lr
            d["params"] = list(range(len(lay.entries)))
            groups.append(d)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        """Accepts torch.optim.Adam's layout (above; also from a reference checkpoint) and this build's round-1 flat layout
        {step, exp_avg, exp_avg_sq}."""
        lay = self.model.layout
        if "state" in sd:
            st = sd["state"]
            if len(st) not in (0, len(lay.entries)):
                raise ValueError(f"optimizer state has {len(st)} parameter entries, the model has {len(lay.entries)}")
            g0 = sd["param_groups"][0] if sd.get("param_groups") else {}
            if g0.get("amsgrad") or g0.get("weight_decay"):
                raise ValueError("FusedAdam implements plain Adam (no amsgrad, no weight decay)")
            self.step_count = 0
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            if st:
                m_views, v_views = lay.views(self.exp_avg), lay.views(self.exp_avg_sq)
                steps = set()
                with torch.no_grad():
                    for i, name in enumerate(m_views):
                        e = st[i] if i in st else st[str(i)]
                        if tuple(e["exp_avg"].shape) != tuple(m_views[name].shape):
                            raise ValueError(f"optimizer state {i} ({name}): shape {tuple(e['exp_avg'].shape)} != "
                                             f"{tuple(m_views[name].shape)}")
                        m_views[name].copy_(e["exp_avg"])
                        v_views[name].copy_(e["exp_avg_sq"])
                        steps.add(int(float(e["step"])))
                if len(steps) != 1:
                    raise ValueError(f"per-parameter step counts differ ({sorted(steps)}); the fused kernel keeps one")
                self.step_count = steps.pop()
        else:
            self.step_count = sd["step"]
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, s_ in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in s_.items() if k != "params"})


def ema_update(model: ProbabilisticUnet, teacher: ProbabilisticUnet, momentum: float):
    """teacher = teacher * m + student * (1 - m) on the flat buffers (mean_teacher_trainer.py:52-55)."""
    model.engine.adam_ema(model.flat_params, None, None, None, teacher.flat_params, 0, 0.0, ema_momentum=momentum,
                          do_adam=False)


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def allreduce_grads(gflat: torch.Tensor, started: Optional[dict] = None) -> float:
    """Sum the flat gradient over ranks (RCCL ring all-reduce over xGMI); returns the scale to apply (1/world).
    `started` = what `elbo_grad_pass(..., overlap=True)` returned in st["allreduce"]: the U-Net slice is then already
    in flight (launched under the encoders' backward) and only the remainder is reduced here."""
    w = _world()
    if w > 1:
        if started:
            dist.all_reduce(gflat[started["numel"]:], op=dist.ReduceOp.SUM)
            started["work"].wait()
        else:
            dist.all_reduce(gflat, op=dist.ReduceOp.SUM)
    return 1.0 / w


def pseudo_labels(net: ProbabilisticUnet, x1, n_samples: int, masking: bool, eps=None, upper=0.9, lower=0.1):
    """sample_from_teacher / sample_from_weak_model (mean_teacher_trainer.py:72-88, fixmatch_trainer.py:37-54), fused:
    forward(x1, None, False) then K7.  Returns float [B,1,H,W] tensors (y soft mean, z consensus)."""
    with torch.no_grad():
        net.forward(x1, None, training=False)
        return net.sample_consensus(n_samples, upper, lower, masking, eps)


def elbo_grad_pass(model: ProbabilisticUnet, gflat, x, y, z, eps, gscale: float, accumulate: bool, overlap: bool = False):
    """forward(x, y, True); elbo(y, z); backward of gscale * (-elbo) into gflat.  Returns device scalars.
    overlap=True (the LAST gradient pass of a data-parallel step): the all-reduce of the U-Net slice of gflat is launched
    asynchronously as soon as that slice is final; finish with allreduce_grads(gflat, st["allreduce"])."""
    eng = model.engine
    B = x.shape[0]
    xn = eng.to_nhwc(x)
    H, W = xn.shape[1], xn.shape[2]
    yt = y.to(device=model.device, dtype=torch.float32).reshape(B, 1, H, W).contiguous()
    cm = None
    if model.consensus_masking is True and z is not None:
        cm = z.to(device=model.device, dtype=torch.float32).reshape(B, 1, H, W).contiguous()
    e = model._eps(1, B, eps)[0]
    st = eng.elbo_forward(model.flat_params, xn, yt, cm, e, "dice" if model.rl_swap else "bce", model.beta, gscale, True)
    started = {}
    hook = None
    if overlap and _world() > 1:
        n = model.layout.unet_numel

        def hook():
            started["numel"] = n
            started["work"] = dist.all_reduce(gflat[:n], op=dist.ReduceOp.SUM, async_op=True)
    eng.elbo_backward(model.flat_params, gflat, st, accumulate, after_unet=hook)
    st["allreduce"] = started
    return st["rec"][0], st["kl"][0], st


def fused_supervised_step(model, opt: FusedAdam, x, y, z=None, eps=None, reg_weight=1e-5):
    """PUNetTrainer / PseudoTrainerPUNet iteration (punet_trainer.py:24-36, punet_target_trainer.py:105-118)."""
    eng = model.engine
    rec, kl, st = elbo_grad_pass(model, opt.grad_flat, x, y, z, eps, 1.0, False, overlap=True)
    reg, norms = eng.l2_reg(model.flat_params)
    eng.l2_reg_add_grad(model.flat_params, opt.grad_flat, norms, reg_weight)
    scale = allreduce_grads(opt.grad_flat, st["allreduce"])
    opt.fused_step(None, grad_scale=scale)
    return rec + model.beta * kl + reg_weight * reg[0]


def fused_self_training_step(model, teacher, opt: FusedAdam, x1, x2, n_samples: int, masking: bool, eps_pl=None, eps_s=None,
                             momentum: Optional[float] = 0.999, reg_weight=1e-5, source_distribution=None):
    """Mean-Teacher iteration (teacher = EMA copy, mean_teacher_trainer.py:101-131) or FixMatch iteration
    (teacher is the model itself and momentum is None, fixmatch_trainer.py:67-95)."""
    eng = model.engine
    y, z = pseudo_labels(teacher, x1, n_samples, masking, eps_pl)
    ratio = None
    if source_distribution is not None:
        y, ratio = distribution_alignment(y, source_distribution, eng)
    rec, kl, st = elbo_grad_pass(model, opt.grad_flat, x2, y, z, eps_s, 1.0, False, overlap=True)
    reg, norms = eng.l2_reg(model.flat_params)
    eng.l2_reg_add_grad(model.flat_params, opt.grad_flat, norms, reg_weight)
    scale = allreduce_grads(opt.grad_flat, st["allreduce"])
    do_ema = momentum is not None and teacher is not model and bool(opt.lr)  # `if lr:` gate, mean_teacher_trainer.py:126
    opt.fused_step(teacher if do_ema else None, ema_momentum=momentum if do_ema else 0.0, grad_scale=scale)
    return rec + model.beta * kl + reg_weight * reg[0], y, z, ratio


def fused_joint_step(model, teacher, opt: FusedAdam, xs, ys, xt1, xt2, n_samples: int, masking: bool, eps_sup=None,
                     eps_pl=None, eps_s=None, momentum: Optional[float] = None, use_consensus=True, reg_weight=1e-5):
    """AdaMT (adamt_trainer.py:89-128) / AdaMatch (adamatch_trainer.py:62-107) iteration:
    loss = (supervised + target) / 2 with ONE optimizer step; two gradient passes at scale 0.5 accumulate."""
    eng = model.engine
    rec_s, kl_s, _ = elbo_grad_pass(model, opt.grad_flat, xs, ys, None, eps_sup, 0.5, False)
    y, z = pseudo_labels(teacher, xt1, n_samples, masking, eps_pl)
    rec_t, kl_t, st = elbo_grad_pass(model, opt.grad_flat, xt2, y, z if use_consensus else None, eps_s, 0.5, True, overlap=True)
    reg, norms = eng.l2_reg(model.flat_params)
    eng.l2_reg_add_grad(model.flat_params, opt.grad_flat, norms, reg_weight)  # (reg + reg) / 2
    scale = allreduce_grads(opt.grad_flat, st["allreduce"])
    ema = momentum is not None and teacher is not model
    opt.fused_step(teacher if ema else None, ema_momentum=momentum if ema else 0.0, grad_scale=scale)
    loss = 0.5 * (rec_s + model.beta * kl_s + rec_t + model.beta * kl_t) + reg_weight * reg[0]
    return loss, y, z


def distribution_alignment(y: torch.Tensor, source_distribution: torch.Tensor, engine=None):
    """fixmatch_trainer.py:77-84 on the device (K-align: exact integer count of (y >= 0.5), no torch.unique host sync):
    target class frequencies, ratio = source / target, rescale and clip.  Returns (aligned y, ratio [2])."""
    if engine is None:
        raise RuntimeError("distribution_alignment runs on the HIP engine of the model (no CPU fallback)")
    src = torch.as_tensor(source_distribution, dtype=torch.float32, device=y.device).contiguous()
    return engine.align_distribution(y, src)
