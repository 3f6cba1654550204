"""CPU oracle for the Probabilistic U-Net self-training hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module.  The product path
(``probabilistic_domain_adaptation_amd``) never imports it and fails loudly when
the HIP extension is missing.

This is a *restatement* (own code, plain fp32 PyTorch CPU ops, functional style
over an explicit parameter dict) of the algorithm in the reference's
``prob_utils/my_models`` and ``prob_utils/my_trainer``.  Every function cites
the reference ``file:line`` it follows (paths relative to ``/root/reference``).

Pinning: the reference ships no tests or golden vectors, so this
oracle is pinned against the reference *itself*, imported in the build container
by ``tests/golden/gen_golden.py`` (in-memory stubs for the absent ``torch_em`` /
``torchvision``; in-memory channel fix for the ``input_dim`` bug at
``probabilistic_unet.py:50-51`` / ``unet.py:27-28``).  The committed fixtures in
``tests/golden/*.npz`` hold the reference's outputs; ``tests/test_oracle_golden.py``
checks this file against them on every run (the generator itself asserts oracle == live reference
while it writes them, so re-running ``gen_golden.py`` in the build container is the live re-check).

Parity status: BCE criterion (``rl_swap=False``) fully pinned.  The Dice
criterion (``rl_swap=True``) lives in the un-vendored, un-pinned third-party
``torch_em.loss.dice.DiceLossWithLogits`` -> "parity unpinned" for Dice only
(recalled definition, see ``dice_loss_with_logits``).

Parameter naming / ordering = the reference's ``state_dict()``
with the *intended* ``nf[i-1] -> nf[i]`` wiring.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------
# parameter inventory (state-dict contract M0)
# --------------------------------------------------------------------------------------
def encoder_conv_indices(n_blocks: int, convs_per_block: int = 3) -> List[List[int]]:
    """nn.Sequential indices of the convs of ``Encoder`` (probabilistic_unet.py:44-63):
    per block [pool if i != 0] + convs_per_block * [conv, relu]."""
    idx, out = 0, []
    for i in range(n_blocks):
        if i != 0:
            idx += 1  # AvgPool2d (probabilistic_unet.py:53-54)
        blk = []
        for _ in range(convs_per_block):
            blk.append(idx)
            idx += 2  # conv + ReLU
        out.append(blk)
    return out


def param_spec(input_channels: int = 1, num_classes: int = 1, num_filters: Sequence[int] = (64, 128, 256, 512),
               latent_dim: int = 6, no_convs_fcomb: int = 3) -> "OrderedDict[str, Tuple[int, ...]]":
    """Names and shapes in ``ProbabilisticUnet.state_dict()`` order
    (module registration order probabilistic_unet.py:251-283: unet, prior, posterior, fcomb)."""
    nf = list(num_filters)
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(name, cin, cout, k):
        spec[name + ".weight"] = (cout, cin, k, k)
        spec[name + ".bias"] = (cout,)

    # Unet (unet.py:26-43); DownConvBlock layers indices: pool shifts by one (unet_blocks.py:16-24)
    for i, co in enumerate(nf):
        ci = input_channels if i == 0 else nf[i - 1]
        off = 0 if i == 0 else 1
        base = f"unet.contracting_path.{i}.layers."
        conv(base + str(off + 0), ci, co, 3)
        conv(base + str(off + 2), co, co, 3)
        conv(base + str(off + 4), co, co, 3)
    out = nf[-1]
    for j, i in enumerate(range(len(nf) - 2, -1, -1)):
        ci = out + nf[i]
        out = nf[i]
        base = f"unet.upsampling_path.{j}.conv_block.layers."
        conv(base + "0", ci, out, 3)
        conv(base + "2", out, out, 3)
        conv(base + "4", out, out, 3)
    # prior / posterior (probabilistic_unet.py:259-274)
    for net, extra in (("prior", 0), ("posterior", num_classes)):
        for i, blk in enumerate(encoder_conv_indices(len(nf))):
            ci = (input_channels + extra) if i == 0 else nf[i - 1]
            for k, li in enumerate(blk):
                conv(f"{net}.encoder.layers.{li}", ci if k == 0 else nf[i], nf[i], 3)
        conv(f"{net}.conv_layer", nf[-1], 2 * latent_dim, 1)
    # fcomb (probabilistic_unet.py:168-177)
    conv("fcomb.layers.0", nf[0] + latent_dim, nf[0], 1)
    for k in range(no_convs_fcomb - 2):
        conv(f"fcomb.layers.{2 * (k + 1)}", nf[0], nf[0], 1)
    conv("fcomb.last_layer", nf[0], num_classes, 1)
    return spec


def l2_reg_param_names(spec) -> List[str]:
    """Tensors covered by the regulariser: posterior, prior, fcomb.layers
    (mean_teacher_trainer.py:115-116; NOT fcomb.last_layer, NOT unet)."""
    return [n for n in spec if n.startswith("posterior.") or n.startswith("prior.") or n.startswith("fcomb.layers.")]


def synth_params(spec, seed: int = 1, last_layer_gain: float = 8.0) -> "OrderedDict[str, np.ndarray]":
    """Deterministic synthetic weights (numpy PCG64) shared by the fixture generator and all tests.
    He-normal conv weights (the reference's init family, utils.py:16-21) and small biases; the
    final fcomb layer is scaled and biased so that sigmoid outputs straddle the 0.9 / 0.1
    consensus thresholds (random init gives an all-zero mask)."""
    rng = np.random.default_rng(seed)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in spec.items():
        if name.endswith(".weight"):
            fan_in = shape[1] * shape[2] * shape[3]
            w = rng.standard_normal(shape, dtype=np.float32) * np.float32(math.sqrt(2.0 / fan_in))
            if name.startswith("fcomb.last_layer"):
                w = w * np.float32(last_layer_gain)
            if name.endswith("conv_layer.weight"):
                w = w * np.float32(0.5)
            out[name] = w.astype(np.float32)
        else:
            b = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.05)
            if name.startswith("fcomb.last_layer"):
                b = b + np.float32(0.3)
            out[name] = b.astype(np.float32)
    return out


def synth_batch(batch: int, channels: int, height: int, width: int, seed: int = 0):
    """Seeded synthetic LIVECell-shaped batch: x, weak view x1, strong view x2, labels y."""
    g = torch.Generator().manual_seed(seed)
    k = torch.ones(1, 1, 5, 5) / 25.0

    def blur(t):
        c = t.shape[1]
        return F.conv2d(F.pad(t, (2, 2, 2, 2), mode="reflect"), k.expand(c, 1, 5, 5), groups=c)

    x = blur(torch.randn(batch, channels, height, width, generator=g))
    x = (x - x.mean(dim=(1, 2, 3), keepdim=True)) / (x.std(dim=(1, 2, 3), keepdim=True) + 1e-7)
    x1 = x + 0.1 * torch.randn(x.shape, generator=g)
    x2 = 1.5 * blur(x) + 0.25 * torch.randn(x.shape, generator=g)
    y = (blur(torch.randn(batch, 1, height, width, generator=g)) * 5.0 > 0.3).float()
    return x.contiguous(), x1.contiguous(), x2.contiguous(), y.contiguous()


def to_torch(params_np, requires_grad: bool = False) -> Params:
    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v)).clone().requires_grad_(requires_grad))
                       for k, v in params_np.items())


# --------------------------------------------------------------------------------------
# model forward
# --------------------------------------------------------------------------------------
# Optional ReLU gates (tests only): when RELU_GATES maps a conv name to a 0/1 tensor, that layer's ReLU is applied as
# pre_activation * gate instead of max(pre_activation, 0).  GPU gradient-parity tests pass the gates of the GPU's own
# forward, so that a pre-activation within fp32 rounding of 0 (which may legitimately resolve to either side) does not
# turn into an O(1) difference of individual gradients.  Default None = the reference's plain ReLU.
RELU_GATES = None
# Filled whenever RELU_GATES is in force: per layer, how many of the imposed gates differ from the oracle's
# OWN ReLU decision (y > 0), and how far from zero the oracle's pre-activation is where they do - a handful of entries within
# rounding of zero is what the substitution is for; a systematic gate error of the GPU forward would show up here as a count.
RELU_GATE_REPORT: dict = {}


def _conv_relu(P: Params, name: str, x: torch.Tensor, relu: bool = True) -> torch.Tensor:
    w = P[name + ".weight"]
    y = F.conv2d(x, w, P[name + ".bias"], stride=1, padding=w.shape[-1] // 2)
    if relu and RELU_GATES is not None and name in RELU_GATES:
        g = RELU_GATES[name]
        with torch.no_grad():
            diff = (y > 0) != (g > 0)
            n = int(diff.sum())
            RELU_GATE_REPORT[name] = (n, float(y[diff].abs().max()) if n else 0.0, float(y.abs().max()), y.numel())
        return y * g.to(y.dtype)
    return F.relu(y) if relu else y


def relu_gate_report(max_fraction: float = 3e-4, max_rel_preact: float = 1e-4, clear: bool = True) -> str:
    """Summary of RELU_GATE_REPORT; raises AssertionError when any layer has more than `max_fraction` of its gates differing
    from the oracle's own ReLU, or a differing gate whose oracle pre-activation exceeds `max_rel_preact` of the layer's largest
    one (i.e. a gate that rounding cannot explain).  Tests print the returned line."""
    tot = sum(v[0] for v in RELU_GATE_REPORT.values())
    nel = sum(v[3] for v in RELU_GATE_REPORT.values())
    worst_rel, worst_frac, bad = 0.0, 0.0, []
    for k, (n, pre, ymax, numel) in RELU_GATE_REPORT.items():
        rel = pre / (ymax + 1e-30)
        worst_rel, worst_frac = max(worst_rel, rel), max(worst_frac, n / max(numel, 1))
        if n > max(1, max_fraction * numel) or rel > max_rel_preact:
            bad.append(f"{k}: {n} of {numel} gates differ, |pre-activation| up to {pre:.3e} (layer max {ymax:.3e})")
    line = (f"ReLU gates imposed on the oracle: {tot} of {nel} differ from its own ReLU over {len(RELU_GATE_REPORT)} layers "
            f"(worst layer fraction {worst_frac:.2e}; largest |pre-activation| at a differing gate {worst_rel:.2e} of its layer's maximum)")
    if clear:
        RELU_GATE_REPORT.clear()
    assert not bad, "gate differences that rounding does not explain:\n  " + "\n  ".join(bad)
    return line


def _pool(x):
    # unet_blocks.py:17 / probabilistic_unet.py:54
    return F.avg_pool2d(x, kernel_size=2, stride=2, padding=0, ceil_mode=True)


def unet_features(P: Params, x: torch.Tensor, n_blocks: int, taps: Optional[dict] = None, prefix: str = "unet.") -> torch.Tensor:
    """Unet.forward with apply_last_layer=False (unet.py:50-69, unet_blocks.py:7-59)."""
    blocks = []
    for i in range(n_blocks):
        off = 0 if i == 0 else 1
        if i != 0:
            x = _pool(x)
        for k in (0, 2, 4):
            x = _conv_relu(P, f"{prefix}contracting_path.{i}.layers.{off + k}", x)
        if taps is not None:
            taps[f"unet.down{i}"] = x
        if i != n_blocks - 1:
            blocks.append(x)
    for j in range(n_blocks - 1):
        up = F.interpolate(x, mode="bilinear", scale_factor=2, align_corners=True)  # unet_blocks.py:51
        bridge = blocks[-j - 1]
        assert up.shape[3] == bridge.shape[3]  # unet_blocks.py:55
        x = torch.cat([up, bridge], 1)  # unet_blocks.py:56
        for k in (0, 2, 4):
            x = _conv_relu(P, f"{prefix}upsampling_path.{j}.conv_block.layers.{k}", x)
        if taps is not None:
            taps[f"unet.up{j}"] = x
    return x


def plain_unet_spec(input_channels=1, num_classes=1, num_filters=(64, 128, 256, 512), apply_last_layer=True):
    """state_dict() of the reference's standalone Unet (unet.py:16-46), intended nf[i-1] -> nf[i] wiring."""
    full = param_spec(input_channels, num_classes, num_filters)
    spec = OrderedDict((k[5:], v) for k, v in full.items() if k.startswith("unet."))
    if apply_last_layer:
        spec["last_layer.weight"] = (num_classes, num_filters[0], 1, 1)
        spec["last_layer.bias"] = (num_classes,)
    return spec


def plain_unet_forward(P: Params, x: torch.Tensor, n_blocks: int, apply_last_layer=True) -> torch.Tensor:
    """Unet.forward(x, val=False) (unet.py:50-69) incl. the optional 1x1 last layer (:66-67)."""
    x = unet_features(P, x, n_blocks, None, prefix="")
    if apply_last_layer:
        x = F.conv2d(x, P["last_layer.weight"], P["last_layer.bias"])
    return x


def encoder(P: Params, net: str, x: torch.Tensor, n_blocks: int, taps: Optional[dict] = None) -> torch.Tensor:
    """Encoder.forward (probabilistic_unet.py:44-69)."""
    for i, blk in enumerate(encoder_conv_indices(n_blocks)):
        if i != 0:
            x = _pool(x)
        for li in blk:
            x = _conv_relu(P, f"{net}.encoder.layers.{li}", x)
        if taps is not None:
            taps[f"{net}.enc{i}"] = x
    return x


def latent_space(P: Params, net: str, x: torch.Tensor, segm: Optional[torch.Tensor], n_blocks: int,
                 latent_dim: int, taps: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """AxisAlignedConvGaussian.forward (probabilistic_unet.py:113-142) -> (mu, sigma)."""
    if segm is not None:
        x = torch.cat((x, segm), dim=1)  # :118  order [patch, segm]
    enc = encoder(P, net, x, n_blocks, taps)
    enc = torch.mean(enc, dim=2, keepdim=True)  # :126
    enc = torch.mean(enc, dim=3, keepdim=True)  # :127
    mls = F.conv2d(enc, P[f"{net}.conv_layer.weight"], P[f"{net}.conv_layer.bias"])  # :130
    mls = mls[:, :, 0, 0]
    mu, log_sigma = mls[:, :latent_dim], mls[:, latent_dim:]  # :136-137
    return mu, torch.exp(log_sigma)  # :141


def fcomb(P: Params, feat: torch.Tensor, z: torch.Tensor, no_convs_fcomb: int) -> torch.Tensor:
    """Fcomb.forward (probabilistic_unet.py:200-214): broadcast z over H, W; cat [features, z]; 1x1 convs."""
    b, _, h, w = feat.shape
    zt = z[:, :, None, None].expand(b, z.shape[1], h, w)
    x = torch.cat((feat, zt), dim=1)  # :212
    x = _conv_relu(P, "fcomb.layers.0", x)
    for k in range(no_convs_fcomb - 2):
        x = _conv_relu(P, f"fcomb.layers.{2 * (k + 1)}", x)
    return _conv_relu(P, "fcomb.last_layer", x, relu=False)  # :214


def kl_diag_gaussians(mu_q, sig_q, mu_p, sig_p) -> torch.Tensor:
    """torch.distributions.kl for Independent(Normal) (probabilistic_unet.py:332) -> [B]."""
    var_ratio = (sig_q / sig_p).pow(2)
    t1 = ((mu_q - mu_p) / sig_p).pow(2)
    return (0.5 * (var_ratio + t1 - 1 - var_ratio.log())).sum(-1)


def dice_loss_with_logits(logits: torch.Tensor, target: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """[RECALLED - parity unpinned] torch_em.loss.dice.DiceLossWithLogits: sigmoid, flatten to
    [C, N*H*W], per channel 1 - 2*sum(p*t)/max(sum(p^2)+sum(t^2), eps), summed over channels.
    Source is not in /root/reference (environment.yaml:18, un-pinned)."""
    p = torch.sigmoid(logits)
    c = p.shape[1]
    pf = p.transpose(0, 1).reshape(c, -1)
    tf = target.transpose(0, 1).reshape(c, -1)
    num = (pf * tf).sum(-1)
    den = (pf * pf).sum(-1) + (tf * tf).sum(-1)
    return (1.0 - 2.0 * num / den.clamp(min=eps)).sum()


class PUNetConfig:
    def __init__(self, input_channels=1, num_classes=1, num_filters=(64, 128, 256, 512), latent_dim=6,
                 no_convs_fcomb=3, beta=1.0, consensus_masking=False, rl_swap=False):
        self.input_channels = input_channels
        self.num_classes = num_classes
        self.num_filters = list(num_filters)
        self.latent_dim = latent_dim
        self.no_convs_fcomb = no_convs_fcomb
        self.beta = beta
        self.consensus_masking = consensus_masking
        self.rl_swap = rl_swap

    @property
    def n_blocks(self):
        return len(self.num_filters)

    def spec(self):
        return param_spec(self.input_channels, self.num_classes, self.num_filters, self.latent_dim,
                          self.no_convs_fcomb)


def punet_forward(P: Params, cfg: PUNetConfig, patch, segm, training: bool, taps: Optional[dict] = None):
    """ProbabilisticUnet.forward (probabilistic_unet.py:285-293) -> dict of the stored state."""
    st = {}
    if training:
        st["posterior"] = latent_space(P, "posterior", patch, segm, cfg.n_blocks, cfg.latent_dim, taps)
    st["prior"] = latent_space(P, "prior", patch, None, cfg.n_blocks, cfg.latent_dim, taps)
    st["features"] = unet_features(P, patch, cfg.n_blocks, taps)
    return st


def punet_sample(P: Params, cfg: PUNetConfig, st, eps: torch.Tensor) -> torch.Tensor:
    """ProbabilisticUnet.sample (probabilistic_unet.py:295-309); z = mu + sigma*eps is what
    Normal.rsample()/sample() compute from a standard-normal draw eps."""
    mu, sig = st["prior"]
    return fcomb(P, st["features"], mu + sig * eps, cfg.no_convs_fcomb)


def kl_sampled(mu_q, sig_q, mu_p, sig_p, z) -> torch.Tensor:
    """kl_divergence(analytic=False) (probabilistic_unet.py:334-338): log q(z) - log p(z) of the two
    Independent(Normal) -> [B] (Normal.log_prob = -(z-mu)^2/(2 var) - log sigma - log sqrt(2 pi))."""
    def lp(mu, sig):
        return (-((z - mu) ** 2) / (2 * sig ** 2) - sig.log() - math.log(math.sqrt(2 * math.pi))).sum(-1)
    return lp(mu_q, sig_q) - lp(mu_p, sig_p)


def punet_elbo(P: Params, cfg: PUNetConfig, st, segm, consm, eps: torch.Tensor, analytic_kl: bool = True,
               reconstruct_posterior_mean: bool = False):
    """ProbabilisticUnet.elbo (probabilistic_unet.py:341-371).  Returns (elbo, dict of scalars/tensors).
    reconstruct_posterior_mean: the reference reads `.loc` of an Independent there (:318), which raises
    AttributeError on torch 2.x; the evident intent (base_dist.loc = mu_q) is what this restates."""
    mu_q, sig_q = st["posterior"]
    mu_p, sig_p = st["prior"]
    z = mu_q + sig_q * eps  # :349 rsample
    if analytic_kl:
        kl = kl_diag_gaussians(mu_q, sig_q, mu_p, sig_p).mean()  # :351-353
    else:
        kl = kl_sampled(mu_q, sig_q, mu_p, sig_p, z).mean()  # :334-338
    recon = fcomb(P, st["features"], mu_q if reconstruct_posterior_mean else z, cfg.no_convs_fcomb)  # :356-358
    if cfg.consensus_masking and consm is not None:
        lg, tg = recon * consm, segm * consm  # :363-364 (masks the LOGITS)
    else:
        lg, tg = recon, segm
    if cfg.rl_swap:
        rl = dice_loss_with_logits(lg, tg)  # :347 (recalled)
        rec_sum, rec_mean = rl, rl
    else:
        rl = F.binary_cross_entropy_with_logits(lg, tg, reduction="none")  # :348
        rec_sum, rec_mean = rl.sum(), rl.mean()  # :368-369
    elbo = -(rec_sum + cfg.beta * kl)  # :371
    return elbo, {"kl": kl, "reconstruction": recon, "reconstruction_loss": rec_sum,
                  "mean_reconstruction_loss": rec_mean, "z_posterior": z}


def l2_regularisation(P: Params, names: Sequence[str]) -> torch.Tensor:
    """utils.py:32-40: sum of un-squared 2-norms, biases included."""
    return sum(P[n].norm(2) for n in names)


def training_loss(P: Params, cfg: PUNetConfig, x, y, consm, eps, reg_weight: float = 1e-5, taps=None, target=None,
                  analytic_kl: bool = True, reconstruct_posterior_mean: bool = False):
    """Body shared by all trainers (e.g. mean_teacher_trainer.py:113-117):
    forward(x, y, True); elbo(y, z); loss = -elbo + 1e-5 * (l2(post) + l2(prior) + l2(fcomb.layers)).
    target: the segm handed to elbo() when it differs from forward()'s (the API allows it; no trainer does it)."""
    st = punet_forward(P, cfg, x, y, True, taps)
    elbo, aux = punet_elbo(P, cfg, st, y if target is None else target, consm, eps, analytic_kl, reconstruct_posterior_mean)
    reg = l2_regularisation(P, l2_reg_param_names(P))
    loss = -elbo + reg_weight * reg
    aux.update(reg=reg, loss=loss, state=st)
    return loss, aux


# --------------------------------------------------------------------------------------
# self-training pieces
# --------------------------------------------------------------------------------------
def sample_consensus(P: Params, cfg: PUNetConfig, x1, eps_n: torch.Tensor, upper=0.9, lower=0.1,
                     masking: bool = False, return_logits: bool = False):
    """MeanTeacherTrainer.sample_from_teacher (mean_teacher_trainer.py:72-88) ==
    FixMatchTrainer.sample_from_weak_model (fixmatch_trainer.py:37-54).
    eps_n: [N, B, L] standard-normal draws, one per MC sample."""
    with torch.no_grad():
        st = punet_forward(P, cfg, x1, None, False)
        n = eps_n.shape[0]
        logits = [punet_sample(P, cfg, st, eps_n[i]) for i in range(n)]
        samples = [torch.sigmoid(l) for l in logits]  # :74
        one, zero = torch.tensor(1.0), torch.tensor(0.0)
        cons = [torch.where((s >= upper) + (s <= lower), one, zero) for s in samples]  # :75-81
        y = torch.stack(samples, 0).sum(0) / n  # :82
        c = torch.stack(cons, 0).sum(0) / n  # :83
        if masking:
            c = torch.where(c == 1, 1, 0)  # :85-86  (int64)
    if return_logits:
        return y, c, torch.stack(logits, 0), st
    return y, c


def consensus_from_logits(logits_n: torch.Tensor, upper=0.9, lower=0.1, masking=False):
    """The reduction half of sample_from_teacher on given logits [N,B,1,H,W] (K8 unit parity)."""
    s = torch.sigmoid(logits_n)
    n = logits_n.shape[0]
    conf = torch.where((s >= upper) + (s <= lower), torch.tensor(1.0), torch.tensor(0.0))
    y = s.sum(0) / n
    c = conf.sum(0) / n
    if masking:
        c = torch.where(c == 1, 1, 0)
    return y, c


def distribution_alignment(y: torch.Tensor, source_distribution: torch.Tensor):
    """FixMatchTrainer._train_epoch_impl (fixmatch_trainer.py:77-84)."""
    y_binary = torch.where(y >= 0.5, 1, 0)
    _, target_distribution = torch.unique(y_binary, return_counts=True)
    target_distribution = target_distribution / target_distribution.sum()
    ratio = source_distribution / target_distribution
    return torch.where(y < 0.5, y * ratio[0], y * ratio[1]).clip(0, 1), ratio


def momentum_update(P_student: Params, P_teacher: Params, momentum: float) -> None:
    """_momentum_update (mean_teacher_trainer.py:52-55; adamt_trainer.py:40-43 passes the warm-up value)."""
    with torch.no_grad():
        for k in P_teacher:
            P_teacher[k].copy_(P_teacher[k] * momentum + P_student[k] * (1.0 - momentum))


def adamt_momentum(iteration: int, momentum: float) -> float:
    """adamt_trainer.py:41."""
    return min(1 - 1 / (iteration + 1), momentum)


def dice_score(segmentation: np.ndarray, groundtruth: np.ndarray, threshold_seg=None, threshold_gt=None) -> float:
    """my_utils/util.py:17-44."""
    assert segmentation.shape == groundtruth.shape
    seg = segmentation if threshold_seg is None else segmentation > threshold_seg
    gt = groundtruth if threshold_gt is None else groundtruth > threshold_gt
    nom = 2 * np.sum(gt * seg)
    denom = np.sum(gt) + np.sum(seg)
    return float(nom) / float(denom + 1e-7)


class AdamState:
    """torch.optim.Adam defaults as used at e.g. LIVECell/livecell_mt.py:83 (no amsgrad, no weight decay)."""

    def __init__(self, P: Params, lr=1e-5, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.betas, self.eps, self.step = lr, betas, eps, 0
        self.m = {k: torch.zeros_like(v) for k, v in P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in P.items()}

    def apply(self, P: Params, grads: Dict[str, torch.Tensor]) -> None:
        b1, b2 = self.betas
        self.step += 1
        bc1 = 1 - b1 ** self.step
        bc2 = 1 - b2 ** self.step
        with torch.no_grad():
            for k, p in P.items():
                g = grads[k]
                self.m[k].lerp_(g, 1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
                p.addcdiv_(self.m[k], denom, value=-self.lr / bc1)


def grads_of(loss: torch.Tensor, P: Params) -> Dict[str, torch.Tensor]:
    names = list(P.keys())
    gs = torch.autograd.grad(loss, [P[n] for n in names], allow_unused=True)
    return {n: (g if g is not None else torch.zeros_like(P[n])) for n, g in zip(names, gs)}


def mean_teacher_step(P_s: Params, P_t: Params, cfg: PUNetConfig, opt: AdamState, x1, x2, eps_teacher, eps_student,
                      masking: bool, momentum: float = 0.999, reg_weight: float = 1e-5):
    """One iteration of MeanTeacherTrainer._train_epoch_impl (mean_teacher_trainer.py:101-131)."""
    y, z = sample_consensus(P_t, cfg, x1, eps_teacher, masking=masking)
    loss, aux = training_loss(P_s, cfg, x2, y, z, eps_student, reg_weight)
    grads = grads_of(loss, P_s)
    opt.apply(P_s, grads)
    if opt.lr:
        momentum_update(P_s, P_t, momentum)
    return {"y": y, "z": z, "loss": loss.detach(), "aux": aux, "grads": grads}


def fixmatch_step(P: Params, cfg: PUNetConfig, opt: AdamState, x1, x2, eps_weak, eps_student, masking: bool,
                  source_distribution=None, reg_weight: float = 1e-5):
    """One iteration of FixMatchTrainer._train_epoch_impl (fixmatch_trainer.py:67-95)."""
    y, z = sample_consensus(P, cfg, x1, eps_weak, masking=masking)
    ratio = None
    if source_distribution is not None:
        y, ratio = distribution_alignment(y, source_distribution)
    loss, aux = training_loss(P, cfg, x2, y, z, eps_student, reg_weight)
    grads = grads_of(loss, P)
    opt.apply(P, grads)
    return {"y": y, "z": z, "loss": loss.detach(), "aux": aux, "grads": grads, "ratio": ratio}


def adamt_step(P_s: Params, P_t: Params, cfg: PUNetConfig, opt: AdamState, xs, ys, xt1, xt2, eps_sup, eps_teacher,
               eps_student, masking: bool, iteration: int, momentum: float = 0.999, reg_weight: float = 1e-5):
    """One iteration of AdaMTTrainer._train_epoch_impl (adamt_trainer.py:89-128)."""
    sup_loss, _ = training_loss(P_s, cfg, xs, ys, None, eps_sup, reg_weight)
    y, z = sample_consensus(P_t, cfg, xt1, eps_teacher, masking=masking)
    tgt_loss, aux = training_loss(P_s, cfg, xt2, y, z, eps_student, reg_weight)
    loss = (sup_loss + tgt_loss) / 2  # :118
    grads = grads_of(loss, P_s)
    opt.apply(P_s, grads)
    momentum_update(P_s, P_t, adamt_momentum(iteration, momentum))  # :41, ungated :127-128
    return {"y": y, "z": z, "loss": loss.detach(), "aux": aux, "grads": grads}


def punet_step(P: Params, cfg: PUNetConfig, opt: AdamState, x, y, eps, consm=None, reg_weight: float = 1e-5):
    """PUNetTrainer._train_epoch_impl (punet_trainer.py:24-36); with consm: PseudoTrainerPUNet
    (punet_target_trainer.py:105-118)."""
    loss, aux = training_loss(P, cfg, x, y, consm, eps, reg_weight)
    grads = grads_of(loss, P)
    opt.apply(P, grads)
    return {"loss": loss.detach(), "aux": aux, "grads": grads}


def mc_prediction(P: Params, cfg: PUNetConfig, raw, eps_n):
    """_custom_punet_prediction (punet_predictions.py:29-33): mean of N sigmoid(sample(testing=True))."""
    with torch.no_grad():
        st = punet_forward(P, cfg, raw, None, False)
        n = eps_n.shape[0]
        return torch.stack([torch.sigmoid(punet_sample(P, cfg, st, eps_n[i])) for i in range(n)], 0).sum(0) / n


def punet_validate(P: Params, cfg: PUNetConfig, batches, eps_elbo, eps_samples, reg_weight: float = 1e-5):
    """PUNetTrainer._validate_impl (punet_trainer.py:52-97) -> (metric_val, loss_val): per batch the no-grad training
    loss, then the mean of 8 sigmoid(sample(testing=False)) (:70-78) and 1 - dice_score against y (:79-82), averaged over
    the loader (:88-90).  eps_elbo[i]: [B, L]; eps_samples[i]: [n, B, L]."""
    metric = loss_val = 0.0
    with torch.no_grad():
        for (x, y), e1, en in zip(batches, eps_elbo, eps_samples):
            loss, aux = training_loss(P, cfg, x, y, None, e1, reg_weight)
            st = aux["state"]
            pred = torch.stack([torch.sigmoid(punet_sample(P, cfg, st, en[i])) for i in range(en.shape[0])], 0).sum(0) / en.shape[0]
            metric += 1.0 - dice_score(pred.numpy().squeeze(), y.numpy().squeeze())
            loss_val += loss.item()
    return metric / len(batches), loss_val / len(batches)


def mean_teacher_validate(P_s: Params, P_t: Params, cfg: PUNetConfig, batches, eps_teacher, eps_elbo, eps_model,
                          masking: bool, reg_weight: float = 1e-5, use_consensus: bool = True):
    """MeanTeacherTrainer._validate_impl (mean_teacher_trainer.py:142-191) -> dict(metric, loss, gt_metric, samples, y, z)
    (samples / y / z of the last batch): teacher pseudo-labels on x1 (:153), no-grad student loss on x2 with elbo(y, z)
    (:157-161), sample_from_model (:163, :90-93), 1 - dice vs the pseudo-label (:164-166) and vs the true gt (:168-170).
    The same body serves FixMatchTrainer (fixmatch_trainer.py:111-160; P_t = P_s), AdaMTTrainer (adamt_trainer.py:139-189)
    and AdaMatchTrainer (adamatch_trainer.py:118-168), whose elbo(y) drops the consensus until training has switched
    `consensus_masking_is_activated` on (:138) - `use_consensus=False`."""
    metric = loss_val = gt_metric = 0.0
    out = {}
    with torch.no_grad():
        for (x, x1, x2, gt), et, e1, em in zip(batches, eps_teacher, eps_elbo, eps_model):
            y, z = sample_consensus(P_t, cfg, x1, et, masking=masking)
            loss, aux = training_loss(P_s, cfg, x2, y, z if use_consensus else None, e1, reg_weight)
            st = aux["state"]
            samples = torch.stack([torch.sigmoid(punet_sample(P_s, cfg, st, em[i])) for i in range(em.shape[0])], 0).sum(0) / em.shape[0]
            pred = samples.numpy().squeeze()
            metric += 1.0 - dice_score(pred, y.numpy().squeeze())
            gt_metric += 1.0 - dice_score(pred, gt.numpy().squeeze())
            loss_val += loss.item()
            out.update(samples=samples, y=y, z=z)
    n = len(batches)
    out.update(metric=metric / n, loss=loss_val / n, gt_metric=gt_metric / n)
    return out


def standardize_np(raw: np.ndarray, eps: float = 1e-7) -> np.ndarray:
    """[RECALLED - parity unpinned] torch_em.transform.raw.standardize with default arguments: float32,
    (raw - mean) / (std + eps) over the whole array (population std)."""
    raw = raw.astype("float32")
    return (raw - raw.mean()) / (raw.std() + eps)


def halo_tiles(shape, block_shape, halo):
    """[RECALLED - parity unpinned] the blocking of torch_em.util.prediction.predict_with_halo (called at
    punet_predictions.py:41-49 with block (384, 384), halo (64, 64)): blocks in row-major order; every block is read as
    the FULL block_shape + 2 * halo window around its origin and the part of the window outside the image is filled by
    np.pad(mode="reflect") of the part inside.  Yields (y0, y1, x0, x1) of the block clipped to the image."""
    for y0 in range(0, shape[0], block_shape[0]):
        for x0 in range(0, shape[1], block_shape[1]):
            yield y0, min(y0 + block_shape[0], shape[0]), x0, min(x0 + block_shape[1], shape[1])


def load_halo_block(image: np.ndarray, y0, x0, block_shape, halo) -> np.ndarray:
    """[RECALLED] one outer tile of predict_with_halo: clip the window to the image, reflect-pad what is missing."""
    H, W = image.shape
    ys, ye = y0 - halo[0], y0 + block_shape[0] + halo[0]
    xs, xe = x0 - halo[1], x0 + block_shape[1] + halo[1]
    data = image[max(ys, 0):min(ye, H), max(xs, 0):min(xe, W)]
    pad = ((max(-ys, 0), max(ye - H, 0)), (max(-xs, 0), max(xe - W, 0)))
    if any(p for pr in pad for p in pr):
        data = np.pad(data, pad, mode="reflect")
    return data


def tiled_prediction(P: Params, cfg: PUNetConfig, image: np.ndarray, eps_tiles, block_shape=(384, 384), halo=(64, 64)):
    """punet_prediction's tiling=True branch (punet_predictions.py:40-50) with the recalled predict_with_halo semantics:
    per tile standardise (its default preprocess), `_custom_punet_prediction` (:29-33), crop the halo, write the block.
    eps_tiles[t]: [N, 1, L] draws of tile t (row-major block order)."""
    out = np.zeros(image.shape, dtype="float32")
    for t, (y0, y1, x0, x1) in enumerate(halo_tiles(image.shape, block_shape, halo)):
        tile = standardize_np(load_halo_block(image, y0, x0, block_shape, halo))
        pred = mc_prediction(P, cfg, torch.from_numpy(tile)[None, None], eps_tiles[t])[0, 0].numpy()
        out[y0:y1, x0:x1] = pred[halo[0]:halo[0] + (y1 - y0), halo[1]:halo[1] + (x1 - x0)]
    return out


def pseudo_prediction(P: Params, cfg: PUNetConfig, image: np.ndarray, eps_n, upper=0.9, lower=0.1):
    """punet_pseudo_prediction's per-image body (punet_predictions.py:103-124): standardise the whole image, forward, N
    sigmoid(sample(testing=True)); mean -> float32 map; per-sample numpy mask (s >= upper) + (s <= lower), consensus =
    (sum / N == 1) -> uint8."""
    with torch.no_grad():
        patch = torch.from_numpy(standardize_np(image))[None, None]
        st = punet_forward(P, cfg, patch, None, False)
        n = eps_n.shape[0]
        samples = [torch.sigmoid(punet_sample(P, cfg, st, eps_n[i])) for i in range(n)]
        mean = (torch.stack(samples, 0).sum(0) / n).numpy().squeeze()
        masks = [(s.numpy().squeeze() >= upper) + (s.numpy().squeeze() <= lower) for s in samples]
        cons = np.stack(masks, 0).sum(0) / n
        return mean, np.where(cons == 1, 1, 0).astype("uint8")


# ----------------------------------------------------------------------------------------------------------------------
# View generation.  ``my_standardize`` follows prob_utils/my_utils/util.py:9-14 and is pinned by torch.
# The three random transforms live in the un-vendored, un-pinned ``torch_em.transform.raw`` / ``torchvision`` ->
# "parity unpinned": recalled definitions (torchvision GaussianBlur = reflect padding + depthwise conv with the outer
# product of exp(-0.5 (i/sigma)^2)/sum taps; AdditiveGaussianNoise = x + N(0, scale); RandomContrast = mean + alpha (x - mean)).
# The call sites that fix the parameter ranges are LIVECell/livecell_fm.py:43-67 and MitoEM/common.py:50-68.
# ----------------------------------------------------------------------------------------------------------------------
def my_standardize(x: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """util.py:9-14 on one item: (x - mean) / (std + eps), torch's Bessel-corrected std."""
    x = x.clone().float()
    x -= x.mean()
    x /= (x.std() + eps)
    return x


def gaussian_taps(ksize: int, sigma: float) -> torch.Tensor:
    half = (ksize - 1) * 0.5
    t = torch.linspace(-half, half, steps=ksize)
    pdf = torch.exp(-0.5 * (t / sigma).pow(2)) if ksize > 1 else torch.ones(1)
    return pdf / pdf.sum()


def augment_view(x: torch.Tensor, params, noise: Optional[torch.Tensor], contrast_mean: float = 0.0,
                 standardize: bool = True, eps: float = 1e-7) -> torch.Tensor:
    """x [B,C,H,W]; params[b] = (ksize, sigma, noise_scale, alpha); noise [B,C,H,W] ~ N(0,1) or None."""
    out = []
    for b in range(x.shape[0]):
        ks, sigma, nscale, alpha = int(params[b][0]), float(params[b][1]), float(params[b][2]), float(params[b][3])
        v = my_standardize(x[b], eps) if standardize else x[b].clone().float()
        if ks > 1:
            k1 = gaussian_taps(ks, sigma)
            k2 = torch.outer(k1, k1)[None, None].expand(v.shape[0], 1, ks, ks)
            r = ks // 2
            v = F.conv2d(F.pad(v[None], (r, r, r, r), mode="reflect"), k2, groups=v.shape[0])[0]
        if noise is not None and nscale != 0.0:
            v = v + nscale * noise[b]
        v = contrast_mean + alpha * (v - contrast_mean)
        out.append(v)
    return torch.stack(out)
