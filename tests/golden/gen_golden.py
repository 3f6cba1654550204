#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Runs only in the build container (needs /root/reference; never on the GPU box).  Nothing of the
reference is copied: it is imported in place, with in-memory stubs for the absent third-party
``torch_em`` / ``torchvision`` symbols and the in-memory channel fix of SURVEY.md §8c, and only
its *outputs* on deterministic synthetic inputs / weights are stored (small .npz files).

Inputs and weights are NOT stored: they are re-synthesised by ``oracle.punet_oracle.synth_params``
/ ``synth_batch`` (numpy PCG64 / torch CPU generator, seeded), which tests call with the seeds
recorded in each fixture; a checksum of every synthesised tensor is stored to catch RNG drift.

Usage:  python tests/golden/gen_golden.py            (writes tests/golden/*.npz)
"""
import contextlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import punet_oracle as O  # noqa: E402


# ---------------------------------------------------------------- stubs for absent third-party modules
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Dice:
    def __init__(self, *a, **k):
        raise NotImplementedError("torch_em absent: Dice parity unpinned")


class _DefaultTrainer:  # attribute holder; the real fit() loop is third-party
    def __init__(self, name=None, train_loader=None, val_loader=None, model=None, loss=None, optimizer=None,
                 metric=None, device=None, lr_scheduler=None, log_image_interval=100, logger=None, **kw):
        self.__dict__.update(name=name, train_loader=train_loader, val_loader=val_loader, model=model, loss=loss,
                             optimizer=optimizer, metric=metric, device=torch.device(device or "cpu"),
                             lr_scheduler=lr_scheduler, log_image_interval=log_image_interval, logger=None,
                             _iteration=0, max_iteration=10 ** 9)


class _Logger:
    def __init__(self, trainer, save_root, **kw):
        pass


te = _mod("torch_em")
te.loss = _mod("torch_em.loss")
te.loss.dice = _mod("torch_em.loss.dice", DiceLossWithLogits=_Dice)
te.trainer = _mod("torch_em.trainer", DefaultTrainer=_DefaultTrainer)
te.trainer.logger_base = _mod("torch_em.trainer.logger_base", TorchEmLogger=_Logger)
te.transform = _mod("torch_em.transform")
te.transform.raw = _mod("torch_em.transform.raw", _normalize_torch=lambda x, **k: x,
                        standardize=lambda raw, *a, **k: O.standardize_np(raw))  # [recalled] third-party; see oracle


def _predict_with_halo(input_, model, gpu_ids, block_shape, halo, output=None, prediction_function=None, **kw):
    """[RECALLED, stated as such in the oracle] stand-in for the un-vendored torch_em.util.prediction.predict_with_halo:
    only the blocking / padding / per-block standardise / halo crop are ours (oracle.halo_tiles, load_halo_block,
    standardize_np); the per-block arithmetic is the REFERENCE's own `_custom_punet_prediction` (punet_predictions.py:29-33)
    passed in as prediction_function, running the reference model."""
    for (y0, y1, x0, x1) in O.halo_tiles(input_.shape, block_shape, halo):
        tile = O.standardize_np(O.load_halo_block(input_, y0, x0, block_shape, halo))
        pred = prediction_function(model, torch.from_numpy(tile)[None, None].to(gpu_ids[0]))
        pred = pred.detach().cpu().numpy()[0, 0]
        output[y0:y1, x0:x1] = pred[halo[0]:halo[0] + (y1 - y0), halo[1]:halo[1] + (x1 - x0)]
    return output


te.util = _mod("torch_em.util")
te.util.prediction = _mod("torch_em.util.prediction", predict_with_halo=_predict_with_halo, predict_with_padding=None)
IMAGES = {}   # path -> array "read" by the imageio stand-in
WRITTEN = {}  # path -> array "written"
iio = _mod("imageio")
iio.v3 = _mod("imageio.v3", imread=lambda path, **k: IMAGES[path],
              imwrite=lambda path, arr, **k: WRITTEN.__setitem__(path, np.asarray(arr).copy()))
tv = _mod("torchvision")
tv.utils = _mod("torchvision.utils", make_grid=lambda *a, **k: None)
import matplotlib  # noqa: E402

matplotlib.use("Agg")
import prob_utils.my_models as mm  # noqa: E402
import prob_utils.my_trainer as mt  # noqa: E402
from prob_utils.my_models.utils import init_weights  # noqa: E402
from prob_utils.my_utils.util import dice_score as ref_dice_score  # noqa: E402


def fix_channels(model, nf):
    """Repairs probabilistic_unet.py:50-51 / unet.py:27-28 in memory (forward code untouched)."""
    for i in range(1, len(nf)):
        for seq, idx in ((model.unet.contracting_path[i].layers, 1),
                         (model.prior.encoder.layers, 6 + (i - 1) * 7 + 1),
                         (model.posterior.encoder.layers, 6 + (i - 1) * 7 + 1)):
            c = nn.Conv2d(nf[i - 1], nf[i], 3, stride=1, padding=1)
            init_weights(c)
            seq[idx] = c
    return model


# ---------------------------------------------------------------- eps injection
class EpsQueue:
    """Replaces Independent.rsample/sample so every latent draw uses a recorded standard-normal eps."""

    def __init__(self):
        self.q = []
        self.used = []

    def push(self, eps):
        self.q.append(eps)

    def draw(self, dist):
        eps = self.q.pop(0)
        self.used.append(eps)
        return dist.base_dist.loc + dist.base_dist.scale * eps


EPS = EpsQueue()
torch.distributions.Independent.rsample = lambda self, sample_shape=torch.Size(): EPS.draw(self)
torch.distributions.Independent.sample = lambda self, sample_shape=torch.Size(): EPS.draw(self).detach()


def make_ref_model(cfg: O.PUNetConfig, seed: int, fixed: bool):
    m = mm.ProbabilisticUnet(cfg.input_channels, cfg.num_classes, cfg.num_filters, cfg.latent_dim,
                             cfg.no_convs_fcomb, cfg.beta, consensus_masking=cfg.consensus_masking,
                             rl_swap=cfg.rl_swap)
    if fixed:
        fix_channels(m, cfg.num_filters)
    spec = cfg.spec()
    sd = m.state_dict()
    assert list(sd.keys()) == list(spec.keys()), "state-dict order differs from oracle spec"
    for k in sd:
        assert tuple(sd[k].shape) == tuple(spec[k]), (k, sd[k].shape, spec[k])
    P = O.synth_params(spec, seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    return m, P


def eps_draws(n, b, l, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, b, l, generator=g)


def csum(a):
    a = np.asarray(a, dtype=np.float64)
    return np.array([a.sum(), np.abs(a).sum()], dtype=np.float64)


def sample_idx(n, k=64, seed=7):
    return np.random.default_rng(seed).integers(0, n, size=k)


def tensor_digest(t):
    a = t.detach().cpu().numpy().astype(np.float32).ravel()
    idx = sample_idx(a.size)
    return np.concatenate([[np.sqrt((a.astype(np.float64) ** 2).sum())], a[idx].astype(np.float64)])


def gen_model_case(name, nf, fixed, B, C, H, W, n_list, full_grads, seeds=(0, 1, 2), K=1):
    out = {}
    cfg = O.PUNetConfig(C, K, nf, 6, 3, 1.0, consensus_masking=True, rl_swap=False)
    out["num_classes"] = np.array([K], dtype=np.int64)
    m, Pnp = make_ref_model(cfg, seeds[1], fixed)
    x, x1, x2, y = O.synth_batch(B, C, H, W, seeds[0])
    out["meta"] = np.array([B, C, H, W, seeds[0], seeds[1], seeds[2], int(fixed)] + list(nf), dtype=np.int64)
    out["chk_x"] = csum(x)
    out["chk_x1"] = csum(x1)
    out["chk_x2"] = csum(x2)
    out["chk_y"] = csum(y)
    out["chk_params"] = np.stack([csum(v) for v in Pnp.values()])
    Po = O.to_torch(Pnp)

    # ---- inference / teacher path: forward(x1, None, False) + N samples
    nmax = max(n_list)
    eps_n = eps_draws(nmax, B, 6, seeds[2])
    with torch.no_grad():
        m.forward(x1, None, training=False)
        out["prior_mu"] = m.prior_latent_space.base_dist.loc.numpy().copy()
        out["prior_sigma"] = m.prior_latent_space.base_dist.scale.numpy().copy()
        out["features_chk"] = csum(m.unet_features)
        out["features_crop"] = m.unet_features[:, :, :8, :8].numpy().copy()
        logits = []
        for i in range(nmax):
            EPS.push(eps_n[i])
            logits.append(m.sample(testing=(i % 2 == 0)).numpy().copy())
        out["logits"] = np.stack(logits)
    # consensus via the real trainer method (MeanTeacherTrainer.sample_from_teacher)
    for n in n_list:
        for masking in (False, True):
            tr = mt.MeanTeacherTrainer(name="g", train_loader=[], val_loader=None, model=m, optimizer=None,
                                       device="cpu", do_consensus_masking=masking)
            tr.n_samples = n
            for i in range(n):
                EPS.push(eps_n[i])
            with torch.no_grad():
                ys, zs = tr.sample_from_teacher(x1)
            out[f"y_n{n}"] = ys.numpy().copy()
            out[f"z_n{n}_{'mask' if masking else 'weight'}"] = zs.numpy().copy()
            # oracle cross-check at generation time
            yo, zo = O.sample_consensus(Po, cfg, x1, eps_n[:n], masking=masking)
            assert (zo.numpy() == zs.numpy()).all(), "oracle consensus differs from reference"
            assert np.abs(yo.numpy() - ys.numpy()).max() < 1e-6

    # ---- training path: forward(x2, y_soft, True); elbo(y_soft, z) ; reg ; grads
    y_soft = torch.from_numpy(out[f"y_n{n_list[-1]}"])
    eps_s = eps_draws(1, B, 6, seeds[2] + 100)[0]
    for mode, consm in (("none", None), ("weight", torch.from_numpy(out[f"z_n{n_list[-1]}_weight"])),
                        ("mask", torch.from_numpy(out[f"z_n{n_list[-1]}_mask"]))):
        m.zero_grad()
        m.forward(x2, y_soft, training=True)
        EPS.push(eps_s)
        elbo = m.elbo(y_soft, consm)
        reg = mm.l2_regularisation(m.posterior) + mm.l2_regularisation(m.prior) + mm.l2_regularisation(m.fcomb.layers)
        loss = -elbo + 1e-5 * reg
        loss.backward()
        out[f"train_{mode}_scalars"] = np.array([loss.item(), m.reconstruction_loss.item(),
                                                  m.mean_reconstruction_loss.item(), m.kl.item(), reg.item()],
                                                 dtype=np.float64)
        out[f"train_{mode}_post_mu"] = m.posterior_latent_space.base_dist.loc.detach().numpy().copy()
        out[f"train_{mode}_post_sigma"] = m.posterior_latent_space.base_dist.scale.detach().numpy().copy()
        out[f"train_{mode}_recon"] = m.reconstruction.detach().numpy().copy()
        gd = np.stack([tensor_digest(p.grad) for p in m.parameters()])
        out[f"train_{mode}_grad_digest"] = gd
        if full_grads and mode == "mask":
            out["train_mask_grads_flat"] = np.concatenate([p.grad.numpy().ravel() for p in m.parameters()])
        # oracle cross-check
        Pg = O.to_torch(Pnp, requires_grad=True)
        lo, aux = O.training_loss(Pg, cfg, x2, y_soft, consm, eps_s)
        go = O.grads_of(lo, Pg)
        assert abs(lo.item() - loss.item()) <= 1e-5 * abs(loss.item()) + 1e-6, (lo.item(), loss.item())
        for (k, g), p in zip(go.items(), m.parameters()):
            d = (g - p.grad).abs().max().item()
            assert d <= 1e-4 * (p.grad.abs().max().item() + 1e-6) + 1e-7, (k, d)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"[gen_golden] wrote {name}.npz ({os.path.getsize(os.path.join(HERE, name + '.npz')) / 1e6:.2f} MB)")


class _Progress:
    def update(self, n):
        pass


def gen_trainer_case(name, nf, B, H, W, n_samples, seeds=(3, 4, 5)):
    """Drive the reference's real trainer loops for one iteration each."""
    out = {}
    C = 1
    cfg = O.PUNetConfig(C, 1, nf, 6, 3, 1.0, consensus_masking=True, rl_swap=False)
    x, x1, x2, gt = O.synth_batch(B, C, H, W, seeds[0])
    xs, _, _, ys = O.synth_batch(B, C, H, W, seeds[0] + 50)
    out["meta"] = np.array([B, C, H, W, seeds[0], seeds[1], seeds[2], n_samples] + list(nf), dtype=np.int64)
    eps_t = eps_draws(n_samples, B, 6, seeds[2])
    eps_s = eps_draws(1, B, 6, seeds[2] + 100)[0]
    eps_sup = eps_draws(1, B, 6, seeds[2] + 200)[0]
    lr = 1e-3  # larger than the scripts' 1e-5 so the update is visible in fp32 digests

    def digest_model(mod):
        return np.stack([tensor_digest(p) for p in mod.parameters()])

    def run(trainer, push):
        captured = {}

        def backprop(loss):
            captured["loss"] = loss.item()
            loss.backward()
            captured["grads"] = np.stack([tensor_digest(p.grad) for p in trainer.model.parameters()])
            trainer.optimizer.step()

        push()
        trainer._train_epoch_impl(_Progress(), contextlib.nullcontext, backprop)
        return captured

    # ---- Mean-Teacher (separate), masking on  (mean_teacher_trainer.py:95-140)
    for masking in (True, False):
        m, Pnp = make_ref_model(cfg, seeds[1], True)
        opt = torch.optim.Adam(m.parameters(), lr=lr)
        tr = mt.MeanTeacherTrainer(name="g", train_loader=[(x, x1, x2, gt)], val_loader=None, model=m, optimizer=opt,
                                   device="cpu", do_consensus_masking=masking)
        tr.n_samples = n_samples
        cap = run(tr, lambda: [EPS.push(e) for e in list(eps_t) + [eps_s]])
        tag = "mt_mask" if masking else "mt_weight"
        out[f"{tag}_loss"] = np.array([cap["loss"]])
        out[f"{tag}_grads"] = cap["grads"]
        out[f"{tag}_student"] = digest_model(tr.model)
        out[f"{tag}_teacher"] = digest_model(tr.teacher)
        # oracle cross-check of the whole step
        Ps, Pt = O.to_torch(Pnp, True), O.to_torch(Pnp)
        ost = O.AdamState(Ps, lr=lr)
        r = O.mean_teacher_step(Ps, Pt, cfg, ost, x1, x2, eps_t, eps_s, masking)
        assert abs(r["loss"].item() - cap["loss"]) <= 1e-5 * abs(cap["loss"]), (r["loss"].item(), cap["loss"])
        for (k, v), p, pt in zip(Ps.items(), tr.model.parameters(), tr.teacher.parameters()):
            assert (v.detach() - p.detach()).abs().max().item() < 2e-2 * lr, k  # entries with |g|~adam eps are ill-conditioned
            assert (Pt[k] - pt.detach()).abs().max().item() < 2e-2 * lr, k

    # ---- FixMatch (separate) with distribution alignment (fixmatch_trainer.py:61-109)
    m, Pnp = make_ref_model(cfg, seeds[1], True)
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    src_dist = [0.7, 0.3]
    tr = mt.FixMatchTrainer(name="g", train_loader=[(x, x1, x2, gt)], val_loader=None, model=m, optimizer=opt,
                            device="cpu", do_consensus_masking=True, source_distribution=src_dist)
    tr.n_samples = n_samples
    cap = run(tr, lambda: [EPS.push(e) for e in list(eps_t) + [eps_s]])
    out["fm_loss"] = np.array([cap["loss"]])
    out["fm_grads"] = cap["grads"]
    out["fm_student"] = digest_model(tr.model)
    out["fm_src_dist"] = np.array(src_dist)
    P = O.to_torch(Pnp, True)
    ost = O.AdamState(P, lr=lr)
    r = O.fixmatch_step(P, cfg, ost, x1, x2, eps_t, eps_s, True, torch.tensor(src_dist))
    assert abs(r["loss"].item() - cap["loss"]) <= 1e-5 * abs(cap["loss"]), (r["loss"].item(), cap["loss"])
    out["fm_y_aligned"] = r["y"].numpy().copy()

    # ---- AdaMT (joint mean teacher)  (adamt_trainer.py:83-137); iteration 0 => momentum 0
    for it0 in (0, 5):
        m, Pnp = make_ref_model(cfg, seeds[1], True)
        opt = torch.optim.Adam(m.parameters(), lr=lr)
        tr = mt.AdaMTTrainer(source_train_loader=[(xs, ys)], target_train_loader=[(x, x1, x2, gt)], name="g",
                             val_loader=None, model=m, optimizer=opt, device="cpu", do_consensus_masking=True)
        tr.n_samples = n_samples
        tr._iteration = it0
        # make teacher differ from student so EMA is visible
        with torch.no_grad():
            for p in tr.teacher.parameters():
                p.mul_(0.5)
        Pt_np = {k: v * np.float32(0.5) for k, v in Pnp.items()}
        cap = run(tr, lambda: [EPS.push(e) for e in [eps_sup] + list(eps_t) + [eps_s]])
        out[f"adamt_it{it0}_loss"] = np.array([cap["loss"]])
        out[f"adamt_it{it0}_grads"] = cap["grads"]
        out[f"adamt_it{it0}_student"] = digest_model(tr.model)
        out[f"adamt_it{it0}_teacher"] = digest_model(tr.teacher)
        Ps, Pt = O.to_torch(Pnp, True), O.to_torch(Pt_np)
        ost = O.AdamState(Ps, lr=lr)
        r = O.adamt_step(Ps, Pt, cfg, ost, xs, ys, x1, x2, eps_sup, eps_t, eps_s, True, it0)
        assert abs(r["loss"].item() - cap["loss"]) <= 1e-5 * abs(cap["loss"]), (r["loss"].item(), cap["loss"])
        for (k, v), pt in zip(Pt.items(), tr.teacher.parameters()):
            assert (v - pt.detach()).abs().max().item() < 2e-2 * lr, k

    # ---- AdaMatch (joint FixMatch)  (adamatch_trainer.py:56-116)
    m, Pnp = make_ref_model(cfg, seeds[1], True)
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    tr = mt.AdaMatchTrainer(source_train_loader=[(xs, ys)], target_train_loader=[(x, x1, x2, gt)], name="g",
                            val_loader=None, model=m, optimizer=opt, device="cpu", do_consensus_masking=True)
    tr.n_samples = n_samples
    cap = run(tr, lambda: [EPS.push(e) for e in [eps_sup] + list(eps_t) + [eps_s]])
    out["adamatch_loss"] = np.array([cap["loss"]])
    out["adamatch_grads"] = cap["grads"]
    out["adamatch_student"] = digest_model(tr.model)

    # ---- PUNet source trainer (punet_trainer.py:19-50) and pseudo-label trainer (punet_target_trainer.py:92-)
    m, Pnp = make_ref_model(cfg, seeds[1], True)
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    tr = mt.PUNetTrainer(name="g", train_loader=[(xs, ys)], val_loader=None, model=m, optimizer=opt, device="cpu")
    cap = run(tr, lambda: EPS.push(eps_sup))
    out["punet_loss"] = np.array([cap["loss"]])
    out["punet_grads"] = cap["grads"]
    out["punet_student"] = digest_model(tr.model)

    m, Pnp = make_ref_model(cfg, seeds[1], True)
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    zmask = (torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(9)) > 0.3).long()
    tr = mt.PseudoTrainerPUNet(name="g", train_loader=[(xs, ys, zmask)], val_loader=None, model=m, optimizer=opt,
                               device="cpu")
    cap = run(tr, lambda: EPS.push(eps_sup))
    out["pseudo_loss"] = np.array([cap["loss"]])
    out["pseudo_grads"] = cap["grads"]
    out["pseudo_student"] = digest_model(tr.model)

    # ---- dice_score (my_utils/util.py:17-44)
    a = torch.sigmoid(torch.randn(2, 32, 32, generator=torch.Generator().manual_seed(11))).numpy()
    b = (torch.rand(2, 32, 32, generator=torch.Generator().manual_seed(12)) > 0.5).float().numpy()
    out["dice_inputs_a"] = a
    out["dice_inputs_b"] = b
    out["dice_values"] = np.array([ref_dice_score(a, b), ref_dice_score(a, b, threshold_seg=0.5),
                                   ref_dice_score(a, b, threshold_seg=0.5, threshold_gt=0.5)])
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"[gen_golden] wrote {name}.npz ({os.path.getsize(os.path.join(HERE, name + '.npz')) / 1e6:.2f} MB)")


class _CaptureLogger:
    """Stands where the tensorboard logger would: receives what `_validate_impl` computed (metric, loss, ...)."""

    def __init__(self):
        self.calls = []

    def log_validation(self, *a, **k):
        self.calls.append((a, k))


def gen_api_case(name, nf, B, H, W, n_samples, seeds=(13, 14, 15)):
    """Model-API corners and the validation / prediction loops, produced by the reference's own methods:
    reconstruct / kl_divergence / elbo option space (probabilistic_unet.py:311-371), PUNetTrainer._validate_impl
    (punet_trainer.py:52-97), MeanTeacherTrainer._validate_impl + sample_from_model (mean_teacher_trainer.py:90-93,
    142-191), punet_pseudo_prediction and punet_prediction (punet_predictions.py:15-136; file I/O, standardize and
    predict_with_halo replaced by in-memory stand-ins, see the stubs above)."""
    import tempfile
    from prob_utils.my_predictions.punet_predictions import punet_prediction, punet_pseudo_prediction
    out = {}
    C, L = 1, 6
    cfg = O.PUNetConfig(C, 1, nf, L, 3, 1.0, consensus_masking=True, rl_swap=False)
    x, x1, x2, gt = O.synth_batch(B, C, H, W, seeds[0])
    xb, x1b, x2b, gtb = O.synth_batch(B, C, H, W, seeds[0] + 1)
    out["meta"] = np.array([B, C, H, W, seeds[0], seeds[1], seeds[2], n_samples] + list(nf), dtype=np.int64)
    m, Pnp = make_ref_model(cfg, seeds[1], True)
    Po = O.to_torch(Pnp)

    def E(n, k):
        return eps_draws(n, B, L, seeds[2] + k)

    # ---------------- model API: forward(x2, gt, True) then the option space -------------------------------------
    y_other = torch.roll(gt, shifts=3, dims=3)  # an elbo() target that differs from forward()'s segm
    consm = (torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(21)) > 0.4).float()
    with torch.no_grad():
        m.forward(x2, gt, training=True)
        out["post_mu"] = m.posterior_latent_space.base_dist.loc.numpy().copy()
        out["post_sigma"] = m.posterior_latent_space.base_dist.scale.numpy().copy()
        out["kl_analytic"] = m.kl_divergence(analytic=True).numpy().copy()
        z_given = m.posterior_latent_space.base_dist.loc + 0.7 * m.posterior_latent_space.base_dist.scale * E(1, 1)[0]
        out["z_given"] = z_given.numpy().copy()
        out["kl_sampled_z"] = m.kl_divergence(analytic=False, z_posterior=z_given).numpy().copy()
        EPS.push(E(1, 2)[0])
        out["kl_sampled_calc"] = m.kl_divergence(analytic=False, calculate_posterior=True).numpy().copy()
        out["recon_z"] = m.reconstruct(z_posterior=z_given).numpy().copy()
        EPS.push(E(1, 3)[0])
        out["recon_calc"] = m.reconstruct(calculate_posterior=True).numpy().copy()
        try:
            m.reconstruct(use_posterior_mean=True)
            out["recon_mean_raises"] = np.array([0])
        except AttributeError:  # Independent has no .loc (probabilistic_unet.py:318)
            out["recon_mean_raises"] = np.array([1])
        so = O.punet_forward(Po, cfg, x2, gt, True)
        assert np.abs(O.kl_sampled(*so["posterior"], *so["prior"], z_given).numpy() - out["kl_sampled_z"]).max() < 1e-5
        assert np.abs(O.fcomb(Po, so["features"], z_given, 3).numpy() - out["recon_z"]).max() < 1e-5
    for tag, kw, tgt in (("sampledkl", dict(analytic_kl=False), gt), ("othertarget", dict(), y_other)):
        m.zero_grad()
        m.forward(x2, gt, training=True)
        EPS.push(E(1, 4)[0])
        elbo = m.elbo(tgt, consm, **kw)
        reg = mm.l2_regularisation(m.posterior) + mm.l2_regularisation(m.prior) + mm.l2_regularisation(m.fcomb.layers)
        loss = -elbo + 1e-5 * reg
        loss.backward()
        out[f"elbo_{tag}_scalars"] = np.array([loss.item(), m.reconstruction_loss.item(), m.mean_reconstruction_loss.item(),
                                               m.kl.item(), reg.item()], dtype=np.float64)
        out[f"elbo_{tag}_grad_digest"] = np.stack([tensor_digest(p.grad) for p in m.parameters()])
        Pg = O.to_torch(Pnp, requires_grad=True)
        lo, _ = O.training_loss(Pg, cfg, x2, gt, consm, E(1, 4)[0], target=tgt, **kw)
        assert abs(lo.item() - loss.item()) <= 1e-5 * abs(loss.item()) + 1e-6, (tag, lo.item(), loss.item())
        for (k, g), p_ in zip(O.grads_of(lo, Pg).items(), m.parameters()):
            assert (g - p_.grad).abs().max().item() <= 1e-4 * (p_.grad.abs().max().item() + 1e-6) + 1e-7, (tag, k)
    m.zero_grad()

    # ---------------- PUNetTrainer._validate_impl over two batches ------------------------------------------------
    batches = [(x, gt), (xb, gtb)]
    tr = mt.PUNetTrainer(name="g", train_loader=[], val_loader=batches, model=m, optimizer=None, device="cpu")
    tr.logger = _CaptureLogger()
    for i in range(2):
        EPS.push(E(1, 10 + i)[0])
        for e in E(8, 20 + i):
            EPS.push(e)
    for e in E(16, 30):  # the logger branch draws 16 more samples (punet_trainer.py:95-96, :15-17)
        EPS.push(e)
    metric = tr._validate_impl(contextlib.nullcontext)
    (a, k), = tr.logger.calls
    out["punet_val"] = np.array([metric, a[2]], dtype=np.float64)  # metric_val, loss_val
    assert a[1] == metric and not EPS.q
    mo, lo_ = O.punet_validate(Po, cfg, batches, [E(1, 10 + i)[0] for i in range(2)], [E(8, 20 + i) for i in range(2)])
    assert abs(mo - metric) < 1e-6 and abs(lo_ - a[2]) <= 1e-5 * abs(a[2]), (mo, metric, lo_, a[2])

    # ---------------- MeanTeacherTrainer._validate_impl + sample_from_model ---------------------------------------
    Pt_np = O.synth_params(cfg.spec(), seeds[1] + 1)  # a teacher that differs from the student
    mbatches = [(x, x1, x2, gt), (xb, x1b, x2b, gtb)]
    for masking in (True, False):
        tag = "mask" if masking else "weight"
        tr = mt.MeanTeacherTrainer(name="g", train_loader=[], val_loader=mbatches, model=m, optimizer=None, device="cpu",
                                   do_consensus_masking=masking)
        tr.teacher.load_state_dict({k: torch.from_numpy(v) for k, v in Pt_np.items()})
        tr.n_samples = n_samples
        tr.logger = _CaptureLogger()
        for i in range(2):
            for e in E(n_samples, 40 + i):
                EPS.push(e)
            EPS.push(E(1, 50 + i)[0])
            for e in E(n_samples, 60 + i):
                EPS.push(e)
        for e in E(n_samples, 70):  # the logger branch samples once more (mean_teacher_trainer.py:186-187)
            EPS.push(e)
        metric = tr._validate_impl(contextlib.nullcontext)
        (a, k), = tr.logger.calls
        assert a[1] == metric and not EPS.q
        out[f"mt_val_{tag}"] = np.array([metric, a[2], k["gt_metric"]], dtype=np.float64)
        out[f"mt_val_{tag}_y"] = a[6].detach().numpy().copy()
        out[f"mt_val_{tag}_z"] = a[7].detach().numpy().copy()
        out[f"mt_val_{tag}_logged_samples"] = a[9].detach().numpy().copy()  # sample_from_model() on the last batch's state
        r = O.mean_teacher_validate(Po, O.to_torch(Pt_np), cfg, mbatches, [E(n_samples, 40 + i) for i in range(2)],
                                    [E(1, 50 + i)[0] for i in range(2)], [E(n_samples, 60 + i) for i in range(2)], masking)
        assert abs(r["metric"] - metric) < 1e-6 and abs(r["gt_metric"] - k["gt_metric"]) < 1e-6
        assert abs(r["loss"] - a[2]) <= 1e-5 * abs(a[2])
        assert (r["z"].numpy() == a[7].numpy()).all()
    # sample_from_model() on its own (mean_teacher_trainer.py:90-93)
    with torch.no_grad():
        tr.model.forward(x2, gt, training=True)
        for e in E(n_samples, 80):
            EPS.push(e)
        out["sample_from_model"] = tr.sample_from_model().numpy().copy()

    # ---------------- punet_pseudo_prediction / punet_prediction through their real bodies -------------------------
    rng = np.random.default_rng(seeds[0])
    tmp = tempfile.mkdtemp()
    img_a = (rng.normal(size=(40, 56)) * 30 + 100).astype("float32")
    img_b = (rng.normal(size=(80, 104)) * 30 + 100).astype("float32")  # 2 x 3 blocks of 40 with halo 8 below
    out["img_a"], out["img_b"] = img_a, img_b
    pa = os.path.join(tmp, "cellA_0.tif")
    open(pa, "w").close()
    IMAGES[pa] = img_a
    for e in eps_draws(5, 1, L, seeds[2] + 90):
        EPS.push(e)
    punet_pseudo_prediction(tmp + "/", tmp + "/out/", m, prior_samples=5, device="cpu", cellname_="cellA", split_name="val")
    assert not EPS.q
    out["pseudo_mean"] = WRITTEN[tmp + "/out/annotations/val/cellA/cellA_0.tif"]
    out["pseudo_consensus"] = WRITTEN[tmp + "/out/consensus/val/cellA/cellA_0.tif"]
    mo, co = O.pseudo_prediction(Po, cfg, img_a, eps_draws(5, 1, L, seeds[2] + 90))
    assert np.abs(mo - out["pseudo_mean"]).max() < 1e-6 and (co == out["pseudo_consensus"]).all()
    assert out["pseudo_consensus"].dtype == np.uint8 and 0 < out["pseudo_consensus"].mean() < 1
    # punet_prediction: block 40 / halo 8 via a patched call (the reference hard-codes 384 / 64 at :45-46; the stand-in
    # receives those and substitutes the small geometry so the fixture stays small)
    pb = os.path.join(tmp, "imgB.tif")
    open(pb, "w").close()
    IMAGES[pb] = img_b
    geo = {"block": (40, 40), "halo": (8, 8)}
    real = te.util.prediction.predict_with_halo
    import prob_utils.my_predictions.punet_predictions as pp
    pp.predict_with_halo = lambda **kw: real(**{**kw, "block_shape": geo["block"], "halo": geo["halo"]})
    n_tiles = len(list(O.halo_tiles(img_b.shape, geo["block"], geo["halo"])))
    eps_tiles = [eps_draws(3, 1, L, seeds[2] + 100 + t) for t in range(n_tiles)]
    for et in eps_tiles:
        for e in et:
            EPS.push(e)
    punet_prediction(pb, tmp + "/pred", m, prior_samples=3, device="cpu")
    assert not EPS.q
    out["tiled_pred"] = WRITTEN[os.path.join(tmp + "/pred", "imgB.tif")].astype("float32")
    out["tiled_geo"] = np.array(list(geo["block"]) + list(geo["halo"]) + [3, n_tiles], dtype=np.int64)
    to = O.tiled_prediction(Po, cfg, img_b, eps_tiles, geo["block"], geo["halo"])
    assert np.abs(to - out["tiled_pred"]).max() < 1e-6
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"[gen_golden] wrote {name}.npz ({os.path.getsize(os.path.join(HERE, name + '.npz')) / 1e6:.2f} MB)")


def gen_validate_case(name, nf, B, H, W, n_samples, seeds=(23, 24, 25)):
    """`_validate_impl` of the three remaining self-training trainers, run through the reference's own methods over two
    batches: FixMatchTrainer (fixmatch_trainer.py:111-160), AdaMTTrainer (adamt_trainer.py:139-189) and AdaMatchTrainer
    (adamatch_trainer.py:118-168) before and after `consensus_masking_is_activated` flips (:138)."""
    out = {}
    C, L = 1, 6
    cfg = O.PUNetConfig(C, 1, nf, L, 3, 1.0, consensus_masking=True, rl_swap=False)
    x, x1, x2, gt = O.synth_batch(B, C, H, W, seeds[0])
    xb, x1b, x2b, gtb = O.synth_batch(B, C, H, W, seeds[0] + 1)
    out["meta"] = np.array([B, C, H, W, seeds[0], seeds[1], seeds[2], n_samples] + list(nf), dtype=np.int64)
    m, Pnp = make_ref_model(cfg, seeds[1], True)
    Po = O.to_torch(Pnp)
    Pt_np = O.synth_params(cfg.spec(), seeds[1] + 1)
    mbatches = [(x, x1, x2, gt), (xb, x1b, x2b, gtb)]

    def E(n, k):
        return eps_draws(n, B, L, seeds[2] + k)

    def run(tr, tag, masking, teacher, use_z):
        tr.n_samples = n_samples
        tr.logger = _CaptureLogger()
        for i in range(2):
            for e in E(n_samples, 40 + i):
                EPS.push(e)
            EPS.push(E(1, 50 + i)[0])
            for e in E(n_samples, 60 + i):
                EPS.push(e)
        for e in E(n_samples, 70):  # the logger branch samples once more
            EPS.push(e)
        metric = tr._validate_impl(contextlib.nullcontext)
        (a, k), = tr.logger.calls
        assert a[1] == metric and not EPS.q
        out[f"{tag}_val"] = np.array([metric, a[2], k["gt_metric"]], dtype=np.float64)
        out[f"{tag}_y"] = a[6].detach().numpy().copy()
        out[f"{tag}_z"] = a[7].detach().numpy().copy()
        r = O.mean_teacher_validate(Po, O.to_torch(Pt_np) if teacher else Po, cfg, mbatches,
                                    [E(n_samples, 40 + i) for i in range(2)], [E(1, 50 + i)[0] for i in range(2)],
                                    [E(n_samples, 60 + i) for i in range(2)], masking, use_consensus=use_z)
        assert abs(r["metric"] - metric) < 1e-6 and abs(r["gt_metric"] - k["gt_metric"]) < 1e-6, (tag, r["metric"], metric)
        assert abs(r["loss"] - a[2]) <= 1e-5 * abs(a[2]), (tag, r["loss"], a[2])
        assert (r["z"].numpy() == a[7].numpy()).all()

    for masking in (True, False):
        mk = "mask" if masking else "weight"
        tr = mt.FixMatchTrainer(name="g", train_loader=[], val_loader=mbatches, model=m, optimizer=None, device="cpu",
                                do_consensus_masking=masking)
        run(tr, f"fixmatch_{mk}", masking, False, True)
        tr = mt.AdaMTTrainer(source_train_loader=[], target_train_loader=[0], name="g", val_loader=mbatches, model=m,
                             optimizer=None, device="cpu", do_consensus_masking=masking)
        tr.teacher.load_state_dict({k: torch.from_numpy(v) for k, v in Pt_np.items()})
        run(tr, f"adamt_{mk}", masking, True, True)
        assert tr.target_val_loader is tr.val_loader
        tr = mt.AdaMatchTrainer(source_train_loader=[], target_train_loader=[0], name="g", val_loader=mbatches, model=m,
                                optimizer=None, device="cpu", do_consensus_masking=masking)
        run(tr, f"adamatch_off_{mk}", masking, False, False)
        tr.consensus_masking_is_activated = True
        run(tr, f"adamatch_on_{mk}", masking, False, True)
    assert abs(out["adamatch_off_mask_val"][1] - out["adamatch_on_mask_val"][1]) > 1e-3 * abs(out["adamatch_on_mask_val"][1])
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"[gen_golden] wrote {name}.npz ({os.path.getsize(os.path.join(HERE, name + '.npz')) / 1e6:.2f} MB)")


def gen_plain_unet_case(name, nf, H, W, seed=1):
    """BASELINE.json configs[0]: plain Unet forward (prob_utils/my_models/unet.py, apply_last_layer=True) on one
    synthetic patch, CPU reference."""
    from prob_utils.my_models.unet import Unet
    m = Unet(1, 1, nf, {'w': 'he_normal', 'b': 'normal'}, apply_last_layer=True, padding=True)
    for i in range(1, len(nf)):  # the same in-memory channel fix (unet.py:27-28)
        c = nn.Conv2d(nf[i - 1], nf[i], 3, stride=1, padding=1)
        init_weights(c)
        m.contracting_path[i].layers[1] = c
    spec = O.plain_unet_spec(1, 1, nf, True)
    sd = m.state_dict()
    assert list(sd.keys()) == list(spec.keys())
    Pnp = O.synth_params(spec, seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in Pnp.items()})
    x, _, _, _ = O.synth_batch(1, 1, H, W, 0)
    with torch.no_grad():
        y = m(x, False)
        yo = O.plain_unet_forward(O.to_torch(Pnp), x, len(nf), True)
    assert (y - yo).abs().max().item() < 1e-5
    out = {"meta": np.array([1, 1, H, W, 0, seed] + list(nf), dtype=np.int64), "chk_x": csum(x), "logits": y.numpy()}
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"[gen_golden] wrote {name}.npz ({os.path.getsize(os.path.join(HERE, name + '.npz')) / 1e6:.2f} MB)")


if __name__ == "__main__":
    if "--only-api" in sys.argv:
        gen_api_case("api_small", [32, 64, 64, 96], B=2, H=32, W=32, n_samples=4)
        sys.exit(0)
    if "--only-validate" in sys.argv:
        gen_validate_case("validate_small", [32, 64, 64, 96], B=2, H=32, W=32, n_samples=4)
        sys.exit(0)
    if "--only-classes" in sys.argv:
        gen_model_case("two_classes", [32, 32, 64, 64], fixed=True, B=2, C=2, H=32, W=40, n_list=[4], full_grads=False, K=2)
        sys.exit(0)
    if "--only-unet" in sys.argv:
        gen_plain_unet_case("plain_unet_256", [64, 128, 256, 512], 256, 256)
        sys.exit(0)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    # (1) literal, unmodified reference (uniform filters run as committed)
    gen_model_case("uniform32_literal", [32, 32, 32, 32], fixed=False, B=2, C=1, H=32, W=32, n_list=[3, 8],
                   full_grads=True)
    # (2) intended wiring, non-uniform filters incl. a non-multiple-of-64 width
    gen_model_case("fixed_small", [32, 64, 96, 128], fixed=True, B=2, C=1, H=64, W=64, n_list=[3, 8, 16],
                   full_grads=False)
    # (3) 3-channel input, non-square, W not a multiple of 32 (config-4-like)
    gen_model_case("fixed_rgb", [32, 32, 64, 64], fixed=True, B=1, C=3, H=48, W=88, n_list=[4], full_grads=False)
    # (4) the scripts' filters (LIVECell/livecell_punet.py:50) at a small patch
    gen_model_case("full_nf", [64, 128, 256, 512], fixed=True, B=1, C=1, H=64, W=64, n_list=[4], full_grads=False)
    # (4b) num_classes = 2 (and two input channels): Fcomb's last layer, the posterior's input and the criterion per class
    gen_model_case("two_classes", [32, 32, 64, 64], fixed=True, B=2, C=2, H=32, W=40, n_list=[4], full_grads=False, K=2)
    # (5) one real iteration of every PUNet trainer
    gen_trainer_case("trainers_small", [32, 64, 64, 96], B=2, H=32, W=32, n_samples=4)
    # (5b) model-API corners, validation loops, prediction bodies
    gen_api_case("api_small", [32, 64, 64, 96], B=2, H=32, W=32, n_samples=4)
    # (5c) validation loops of FixMatch / AdaMT / AdaMatch
    gen_validate_case("validate_small", [32, 64, 64, 96], B=2, H=32, W=32, n_samples=4)
    # (6) BASELINE.json configs[0]: plain Unet forward, one 256x256x1 patch
    gen_plain_unet_case("plain_unet_256", [64, 128, 256, 512], 256, 256)
