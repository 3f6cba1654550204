"""`ProbabilisticUnet` with the reference's API surface, computed by the gfx950 kernels.

Mirror of prob_utils/my_models/probabilistic_unet.py:217-371 (constructor kwargs, stateful
forward() -> sample() / reconstruct() / kl_divergence() / elbo(), attributes the trainers read) and of its
state-dict contract (same key names, same logical shapes, intended nf[i-1] -> nf[i] wiring).

Differences that are deliberate and documented:
  * all parameters live in ONE flat device buffer in kernel layout; the nn.Parameters registered under the
    reference's names are permuted views of it (state_dict()/load_state_dict()/optimizers work unchanged);
  * latent draws use an injectable standard-normal `eps` (the reference draws from torch's global generator,
    probabilistic_unet.py:302,307,349) so that results are reproducible against the CPU oracle;
  * elbo() runs the HIP forward AND backward eagerly and returns a scalar attached to autograd through a thin
    Function, so `backprop(loss)` of a torch_em-style trainer still fills `.grad` of every parameter;
  * there is no CPU fallback: constructing the model without a ROCm device raises.
"""
from __future__ import annotations

import copy
import math
from typing import Optional

import torch
import torch.nn as nn

from ..engine import ParamLayout, PUNetEngine


class _Node(nn.Module):
    """Container that reproduces the reference's module tree so parameter names match its state-dict."""


def _attach(root: nn.Module, dotted: str, param: nn.Parameter):
    parts = dotted.split(".")
    node = root
    for p in parts[:-1]:
        nxt = node._modules.get(p)
        if nxt is None:
            nxt = _Node()
            node.add_module(p, nxt)
        node = nxt
    node.register_parameter(parts[-1], param)


class _LatentSpace:
    """What the trainers may read from `prior_latent_space` / `posterior_latent_space`
    (Independent(Normal(mu, sigma), 1) in the reference, probabilistic_unet.py:141)."""

    def __init__(self, muls: torch.Tensor, engine: PUNetEngine):
        self.muls = muls
        self.loc, self.scale = engine.latent_split(muls)
        self.base_dist = self

    @property
    def mean(self):
        return self.loc

    def distribution(self):
        return torch.distributions.Independent(torch.distributions.Normal(self.loc, self.scale), 1)


class _ElboFn(torch.autograd.Function):
    """elbo value whose backward hands out the gradients the HIP backward already computed."""

    @staticmethod
    def forward(ctx, value, gflat, layout, *params):
        ctx.gflat, ctx.layout = gflat, layout
        return value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        # gflat holds d(rec + beta*kl)/d theta = d(-elbo)/d theta
        g = ctx.gflat * (-grad_out)
        return (None, None, None) + tuple(ctx.layout.views(g).values())


class ProbabilisticUnet(nn.Module):
    """A probabilistic UNet (https://arxiv.org/abs/1806.05034); same signature as the reference's
    (probabilistic_unet.py:227-237)."""

    def __init__(self, input_channels=1, num_classes=1, num_filters=[32, 64, 128, 192], latent_dim=6, no_convs_fcomb=4,
                 beta=10.0, consensus_masking=False, rl_swap=False, device=None, seed: Optional[int] = None,
                 conv_mode: Optional[str] = None):
        """conv_mode (extension): None = $PUNET_CONV_MODE or "x6" (fp32-accurate).  Opt-in fast modes (engine.PUNetEngine):
        "f16x3" (two fp16 planes, per-tensor power-of-two scaling: logits within 1e-4 of the reference's - what the trainers'
        mixed_precision=True selects), "f16" (one fp16 plane: the arithmetic of the reference's fp16 autocast), "x3" / "bf16" /
        "bf16s" (bf16 planes; bf16s also stores activations as bf16)."""
        super().__init__()
        self.input_channels = input_channels
        self.num_classes = num_classes
        self.num_filters = list(num_filters)
        self.latent_dim = latent_dim
        self.no_convs_per_block = 3
        self.no_convs_fcomb = no_convs_fcomb
        self.initializers = {'w': 'he_normal', 'b': 'normal'}
        self.beta = beta
        self.z_prior_sample = 0
        self.consensus_masking = consensus_masking
        self.rl_swap = rl_swap

        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device()) \
            if torch.cuda.is_available() else None
        if dev is None or dev.type != "cuda":
            raise RuntimeError("ProbabilisticUnet (MI355X build) needs a ROCm GPU: there is no CPU fallback")
        self.layout = ParamLayout(input_channels, num_classes, self.num_filters, latent_dim, no_convs_fcomb)
        self.engine = PUNetEngine(self.layout, dev, conv_mode)
        self._engines = {self.engine.conv_mode: self.engine}  # one engine per conv arithmetic, kept for the model's life
        self._native_rts = {}                                 # one native runtime per (conv arithmetic, criterion), likewise
        self._flat = torch.zeros(self.layout.numel, dtype=torch.float32, device=dev)
        self._views = self.layout.views(self._flat)
        for name, v in self._views.items():
            _attach(self, name, nn.Parameter(v, requires_grad=True))
        self.reset_parameters(seed)
        self._state = None
        self._features_nchw = None
        self.generator = None  # optional torch.Generator (device) for the latent draws

    # ------------------------------------------------------------------ parameters
    @property
    def flat_params(self) -> torch.Tensor:
        return self._flat

    @property
    def device(self):
        return self._flat.device

    def reset_parameters(self, seed: Optional[int] = None):
        """Same init families as the reference: He-normal weights + truncated-normal(1e-3) biases for the 3x3 stacks
        (utils.py:16-21), orthogonal weights for Fcomb and the latent heads (utils.py:24-28,
        probabilistic_unet.py:110-111)."""
        g = torch.Generator().manual_seed(seed if seed is not None else torch.seed() % (2 ** 31))
        with torch.no_grad():
            for name, v in self._views.items():
                shape = tuple(v.shape)
                if name.endswith(".weight"):
                    w = torch.empty(shape)
                    if name.startswith("fcomb.") or name.endswith("conv_layer.weight"):
                        nn.init.orthogonal_(w, generator=g)
                    else:
                        nn.init.kaiming_normal_(w, mode="fan_in", nonlinearity="relu", generator=g)
                    v.copy_(w)
                else:
                    b = torch.empty(shape)
                    nn.init.trunc_normal_(b, std=0.001, a=-0.002, b=0.002, generator=g)
                    v.copy_(b)

    def native_runtime(self):
        """The step-level native runtime of this model's configuration (runtime.NativeRuntime: the graph walk as host code
        inside libpunet_hip.so), created on first use; raises ValueError outside its domain (num_classes 1, num_filters[0] in
        {32, 64}, no_convs_fcomb 2 or 3) and for arithmetics other than "x6" (parity), "f16x3" and "f16"."""
        key = (self.engine.conv_mode, bool(self.rl_swap))
        rt = self._native_rts.get(key)
        if rt is None:
            from ..runtime import NativeRuntime
            rt = self._native_rts[key] = NativeRuntime(self.layout, self.device, key[0], dice=key[1])
        return rt

    def set_conv_mode(self, conv_mode: str):
        """Switch the 3x3 conv arithmetic of this model (engine.PUNetEngine: "x6" parity default, "f32", or a fast mode -
        "f16x3", "f16", "x3", "bf16", "bf16s").  An engine holds no weights; ONE engine (and, on first use, one native runtime)
        per arithmetic is built and then KEPT for the model's life, so switching back and forth - mixed_precision=True applied
        on entry to fit() and restored on exit - swaps between the same objects: `model.engine` has a stable identity per mode,
        a hipGraph captured against an engine's buffers stays valid while the model lives, nothing is re-uploaded.
        `model.engine` is therefore NOT the same object before and after a switch: hold `model`, not `model.engine`.
        `release_conv_mode_cache()` frees the engines of the modes not in use.  Cached forward state is dropped."""
        if conv_mode != self.engine.conv_mode:
            eng = self._engines.get(conv_mode)
            if eng is None:
                eng = self._engines[conv_mode] = PUNetEngine(self.layout, self.device, conv_mode)
            self.engine = eng
            self._state = None
            self._features_nchw = None
        return self

    def release_conv_mode_cache(self):
        """Drop the engines / native runtimes (workspaces, arenas, tables) of every conv arithmetic but the current one.  Graphs a
        trainer captured against them keep what they need alive themselves (GraphedStep holds its engines and buffers)."""
        cur = self.engine.conv_mode
        self._engines = {cur: self.engine}
        self._native_rts = {k: v for k, v in self._native_rts.items() if k[0] == cur}

    def __deepcopy__(self, memo):
        """deepcopy(model) -> teacher (mean_teacher_trainer.py:40): a new flat buffer with the same values."""
        m = ProbabilisticUnet(self.input_channels, self.num_classes, self.num_filters, self.latent_dim, self.no_convs_fcomb,
                              self.beta, self.consensus_masking, self.rl_swap, device=self.device, seed=0,
                              conv_mode=self.engine.conv_mode)
        with torch.no_grad():
            m._flat.copy_(self._flat)
        for p, q in zip(m.parameters(), self.parameters()):
            p.requires_grad_(q.requires_grad)
        m.train(self.training)
        return m

    def _apply(self, fn, recurse=True):
        # .to(device)/.cuda() would break the flat-buffer views; the model is built on its GPU and stays there
        probe = fn(torch.empty(0, device=self.device))
        if probe.device != self.device or probe.dtype != torch.float32:
            raise RuntimeError("the MI355X ProbabilisticUnet lives in one flat fp32 device buffer; build it on its device")
        return self

    # ------------------------------------------------------------------ reference API
    def _eps(self, n: int, batch: int, eps: Optional[torch.Tensor]) -> torch.Tensor:
        if eps is None:
            eps = torch.randn(n, batch, self.latent_dim, device=self.device, generator=self.generator)
        eps = eps.to(device=self.device, dtype=torch.float32).reshape(n, batch, self.latent_dim).contiguous()
        return eps

    def _b1hw(self, t, B, H, W):
        return t.to(device=self.device, dtype=torch.float32).reshape(B, self.num_classes, H, W).contiguous()

    def forward(self, patch, segm, training=True):
        """probabilistic_unet.py:285-293.  Returns None; stores posterior_latent_space (training only),
        prior_latent_space and unet_features.  With grad mode on and training=True the activations are kept for the
        backward that elbo() runs."""
        eng = self.engine
        x = eng.to_nhwc(patch)
        B, H, W = x.shape[0], x.shape[1], x.shape[2]
        if training:
            if segm is None:
                raise ValueError("forward(patch, None, training=True) is invalid: the posterior needs the segmentation")
            st = eng.train_forward(self._flat, x, self._b1hw(segm, B, H, W), record=torch.is_grad_enabled())
            self._state = {"x": x, "training": True, "train": st, "features": st["feat"], "prior_muls": st["muls_p"]}
            self.posterior_latent_space = _LatentSpace(st["muls_q"], eng)
        else:
            out = eng.forward_features(self._flat, x)
            self._state = {"x": x, "training": False, "features": out["features"], "prior_muls": out["prior_muls"]}
        self.prior_latent_space = _LatentSpace(self._state["prior_muls"], eng)
        self._features_nchw = None
        return None

    @property
    def unet_features(self):
        """[B, C, H, W] view of the U-Net features (probabilistic_unet.py:293); converted from NHWC on first access."""
        if self._state is None:
            raise AttributeError("unet_features: call forward() first")
        if self._features_nchw is None:
            self._features_nchw = self.engine.to_nchw(self._state["features"])
        return self._features_nchw

    def _ensure_features(self):
        if self._state is None:
            raise RuntimeError("call forward() first")

    def sample(self, testing=False, eps: Optional[torch.Tensor] = None):
        """probabilistic_unet.py:295-309: logits of one prior sample ([B,1,H,W])."""
        self._ensure_features()
        st = self._state
        B = st["x"].shape[0]
        z = self.engine.reparam(st["prior_muls"], self._eps(1, B, eps))
        self.z_prior_sample = z[0]
        _, _, logits = self.engine.sample(self._flat, st["features"], z, want_logits=True, want_consensus=False)
        return logits[0]

    def pseudo_labels(self, patch, n_samples: int, upper=0.9, lower=0.1, masking=False, eps: Optional[torch.Tensor] = None):
        """forward(patch, None, training=False) + sample_consensus(...) in one go: (mean of N sigmoids, consensus), float
        [B,1,H,W] - the body of sample_from_teacher / sample_from_weak_model (mean_teacher_trainer.py:72-88) and of the MC
        prediction (punet_predictions.py:29-33).  One call into the step-level native runtime where it covers this model
        (the forward state - unet_features, prior_latent_space - is then NOT kept: use forward() for that), else the engine."""
        from ..my_trainer.fused import native_ok
        x = self.engine.to_nhwc(patch)
        if native_ok(self):
            return self.native_runtime().pseudo_labels(self._flat, x, self._eps(n_samples, x.shape[0], eps), upper, lower, masking)
        self.forward(patch, None, training=False)
        return self.sample_consensus(n_samples, upper, lower, masking, eps)

    def sample_consensus(self, n_samples: int, upper=0.9, lower=0.1, masking=False, eps: Optional[torch.Tensor] = None):
        """Fused form of `[sigmoid(sample()) for _ in range(n)]` + thresholds + stack/sum
        (mean_teacher_trainer.py:74-86): returns (mean of sigmoids, consensus) as [B,1,H,W] float tensors."""
        self._ensure_features()
        st = self._state
        B = st["x"].shape[0]
        z = self.engine.reparam(st["prior_muls"], self._eps(n_samples, B, eps))
        y, c, _ = self.engine.sample(self._flat, st["features"], z, upper, lower, masking)
        return y, c

    def _posterior_muls(self):
        st = self._state
        if st is None or not st["training"]:
            raise RuntimeError("the posterior needs forward(patch, segm, training=True) first")
        return st["train"]["muls_q"]

    def reconstruct(self, use_posterior_mean=False, calculate_posterior=False, z_posterior=None,
                    eps: Optional[torch.Tensor] = None):
        """probabilistic_unet.py:311-322 (forward only): Fcomb on the posterior mean / a fresh posterior rsample / the
        supplied z.  (use_posterior_mean reads `posterior_latent_space.loc` in the reference, which an Independent does
        not have - AttributeError there; the intended base_dist.loc is used here.)"""
        self._ensure_features()
        st = self._state
        B = st["x"].shape[0]
        if use_posterior_mean:
            z_posterior = self.posterior_latent_space.loc
        elif calculate_posterior:
            z_posterior = self.engine.reparam(self._posterior_muls(), self._eps(1, B, eps))[0]
        z = z_posterior.to(device=self.device, dtype=torch.float32).reshape(1, B, self.latent_dim).contiguous()
        _, _, logits = self.engine.sample(self._flat, st["features"], z, want_logits=True, want_consensus=False)
        return logits[0]

    def kl_divergence(self, analytic=True, calculate_posterior=False, z_posterior=None, eps: Optional[torch.Tensor] = None):
        """probabilistic_unet.py:324-339 -> [B]: analytic KL(Q||P), or log q(z) - log p(z) at a posterior sample."""
        mq, mp = self._posterior_muls(), self._state["prior_muls"]
        if analytic:
            return self.engine.kl_rows(mq, mp, None)
        B = mq.shape[0]
        if calculate_posterior:
            z_posterior = self.engine.reparam(mq, self._eps(1, B, eps))[0]
        z = z_posterior.to(device=self.device, dtype=torch.float32).reshape(B, self.latent_dim).contiguous()
        return self.engine.kl_rows(mq, mp, z)

    def elbo(self, segm, consm=None, analytic_kl=True, reconstruct_posterior_mean=False, view=False,
             eps: Optional[torch.Tensor] = None):
        """probabilistic_unet.py:341-371: -(reconstruction_loss + beta * kl).

        Runs rsample, KL, Fcomb and the criterion on the state forward() stored; with grad mode on it also runs the whole
        backward (Fcomb, U-Net, both encoders) and returns a scalar wired into autograd (see _ElboFn).  `segm` is the
        criterion's target; the posterior saw forward()'s segm (:291) - they may differ, as in the reference."""
        st = self._state
        if st is None or not st["training"]:
            raise RuntimeError("elbo() needs forward(patch, segm, training=True) first")
        eng = self.engine
        B, H, W = st["x"].shape[0], st["x"].shape[1], st["x"].shape[2]
        tgt = self._b1hw(segm, B, H, W)
        cm = None
        if self.consensus_masking is True and consm is not None:
            cm = self._b1hw(consm, B, H, W)
        e = self._eps(1, B, eps)[0]
        record = torch.is_grad_enabled() and st["train"]["recorded"]
        es = eng.elbo_head(self._flat, st["train"], tgt, cm, e, "dice" if self.rl_swap else "bce", self.beta, 1.0, record,
                           analytic_kl=analytic_kl, posterior_mean=reconstruct_posterior_mean)
        if view:
            print(es["logits"].shape, segm.shape)
        self.kl = es["kl"][0]
        self.reconstruction = es["logits"]
        self.reconstruction_loss = es["rec"][0]
        self.mean_reconstruction_loss = self.reconstruction_loss / float(B * self.num_classes * H * W) if not self.rl_swap \
            else self.reconstruction_loss
        value = -(self.reconstruction_loss + self.beta * self.kl)
        if not record:
            return value
        gflat = torch.empty_like(self._flat)
        eng.elbo_backward(self._flat, gflat, es, accumulate=False)
        return _ElboFn.apply(value, gflat, self.layout, *self.parameters())

    # ------------------------------------------------------------------ state-dict compatibility
    def load_reference_state_dict(self, state_dict, strict=True):
        """Accepts a checkpoint's ["model_state"] / ["teacher_state"] (livecell_punet.py:101, livecell_mt.py:145)."""
        return self.load_state_dict(state_dict, strict=strict)


def l2_regularisation(m):
    """utils.py:32-40 (kept for drop-in trainers; the fused trainers use the K10 kernels instead)."""
    l2_reg = None
    for W in m.parameters():
        l2_reg = W.norm(2) if l2_reg is None else l2_reg + W.norm(2)
    return l2_reg
