"""CPU: the oracle (oracle/punet_oracle.py) against the committed golden fixtures, which hold outputs of the
REFERENCE itself (tests/golden/gen_golden.py).  Tolerances: fp32 CPU vs fp32 CPU, same library -> 1e-5 relative."""
import os

import numpy as np
import pytest
import torch

from oracle import punet_oracle as O

CASES = ["uniform32_literal", "fixed_small", "fixed_rgb", "full_nf", "two_classes"]  # two_classes: num_classes = 2, C = 2


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _setup(fx):
    meta = fx["meta"]
    B, C, H, W, s0, s1, s2, fixed = [int(v) for v in meta[:8]]
    nf = [int(v) for v in meta[8:]]
    K = int(fx["num_classes"][0])
    cfg = O.PUNetConfig(C, K, nf, 6, 3, 1.0, consensus_masking=True, rl_swap=False)
    Pnp = O.synth_params(cfg.spec(), s1)
    batch = O.synth_batch(B, C, H, W, s0)
    return cfg, Pnp, batch, (B, C, H, W, s0, s1, s2)


def _eps(n, b, l, seed):
    return torch.randn(n, b, l, generator=torch.Generator().manual_seed(seed))


def _csum(a):
    a = np.asarray(a, dtype=np.float64)
    return np.array([a.sum(), np.abs(a).sum()])


@pytest.mark.parametrize("name", CASES)
def test_synth_inputs_reproduce(golden_dir, name):
    fx = _load(golden_dir, name)
    cfg, Pnp, (x, x1, x2, y), _ = _setup(fx)
    np.testing.assert_allclose(_csum(x), fx["chk_x"], rtol=1e-6)
    np.testing.assert_allclose(_csum(x1), fx["chk_x1"], rtol=1e-6)
    np.testing.assert_allclose(_csum(x2), fx["chk_x2"], rtol=1e-6)
    np.testing.assert_allclose(_csum(y), fx["chk_y"], rtol=0)
    np.testing.assert_allclose(np.stack([_csum(v) for v in Pnp.values()]), fx["chk_params"], rtol=1e-9)


@pytest.mark.parametrize("name", CASES)
def test_inference_path(golden_dir, name):
    fx = _load(golden_dir, name)
    cfg, Pnp, (x, x1, x2, y), (B, C, H, W, s0, s1, s2) = _setup(fx)
    P = O.to_torch(Pnp)
    nmax = fx["logits"].shape[0]
    eps = _eps(nmax, B, 6, s2)
    with torch.no_grad():
        st = O.punet_forward(P, cfg, x1, None, False)
        np.testing.assert_allclose(st["prior"][0].numpy(), fx["prior_mu"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(st["prior"][1].numpy(), fx["prior_sigma"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(st["features"][:, :, :8, :8].numpy(), fx["features_crop"], rtol=1e-5, atol=1e-5)
        for i in range(nmax):
            lg = O.punet_sample(P, cfg, st, eps[i]).numpy()
            np.testing.assert_allclose(lg, fx["logits"][i], rtol=1e-5, atol=2e-5)
    for key in fx.files:
        if key.startswith("y_n"):
            n = int(key[3:])
            for masking, tag in ((False, "weight"), (True, "mask")):
                yo, zo = O.sample_consensus(P, cfg, x1, eps[:n], masking=masking)
                np.testing.assert_allclose(yo.numpy(), fx[key], rtol=1e-5, atol=1e-6)
                zr = fx[f"z_n{n}_{tag}"]
                assert zo.numpy().dtype == zr.dtype
                assert (zo.numpy() == zr).all()  # bit-exact consensus
            assert 0.005 < fx[f"z_n{n}_mask"].mean() < 0.995, "fixture mask is trivial"


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("mode", ["none", "weight", "mask"])
def test_training_path(golden_dir, name, mode):
    fx = _load(golden_dir, name)
    cfg, Pnp, (x, x1, x2, y), (B, C, H, W, s0, s1, s2) = _setup(fx)
    nlast = max(int(k[3:]) for k in fx.files if k.startswith("y_n"))
    y_soft = torch.from_numpy(fx[f"y_n{nlast}"])
    consm = None if mode == "none" else torch.from_numpy(fx[f"z_n{nlast}_{mode}"])
    eps_s = _eps(1, B, 6, s2 + 100)[0]
    P = O.to_torch(Pnp, requires_grad=True)
    loss, aux = O.training_loss(P, cfg, x2, y_soft, consm, eps_s)
    sc = fx[f"train_{mode}_scalars"]
    got = np.array([loss.item(), aux["reconstruction_loss"].item(), aux["mean_reconstruction_loss"].item(),
                    aux["kl"].item(), aux["reg"].item()])
    np.testing.assert_allclose(got, sc, rtol=2e-5)
    np.testing.assert_allclose(aux["state"]["posterior"][0].detach().numpy(), fx[f"train_{mode}_post_mu"], rtol=1e-5,
                               atol=1e-6)
    np.testing.assert_allclose(aux["reconstruction"].detach().numpy(), fx[f"train_{mode}_recon"], rtol=1e-5, atol=2e-5)
    grads = O.grads_of(loss, P)
    dig = fx[f"train_{mode}_grad_digest"]
    idx = np.random.default_rng(7)
    for row, (k, g) in zip(dig, grads.items()):
        a = g.numpy().ravel()
        ii = np.random.default_rng(7).integers(0, a.size, size=64)
        nrm = np.sqrt((a.astype(np.float64) ** 2).sum())
        assert abs(nrm - row[0]) <= 1e-4 * row[0] + 1e-9, k
        np.testing.assert_allclose(a[ii], row[1:], rtol=1e-3, atol=1e-5 * (np.abs(row[1:]).max() + 1e-9), err_msg=k)
    if mode == "mask" and "train_mask_grads_flat" in fx.files:
        flat = np.concatenate([g.numpy().ravel() for g in grads.values()])
        ref = fx["train_mask_grads_flat"]
        assert np.abs(flat - ref).max() <= 1e-4 * np.abs(ref).max()


def test_dice_score(golden_dir):
    fx = _load(golden_dir, "trainers_small")
    a, b = fx["dice_inputs_a"], fx["dice_inputs_b"]
    got = [O.dice_score(a, b), O.dice_score(a, b, threshold_seg=0.5), O.dice_score(a, b, 0.5, 0.5)]
    np.testing.assert_allclose(got, fx["dice_values"], rtol=1e-12)


def _digest(t):
    a = t.detach().numpy().astype(np.float32).ravel()
    ii = np.random.default_rng(7).integers(0, a.size, size=64)
    return np.concatenate([[np.sqrt((a.astype(np.float64) ** 2).sum())], a[ii].astype(np.float64)])


def _check_digest(P, ref, lr, what):
    for row, (k, v) in zip(ref, P.items()):
        d = _digest(v)
        # first-step Adam is sign-like: entries with |g| ~ eps are ill-conditioned -> tolerance in units of lr
        assert np.abs(d[1:] - row[1:]).max() <= 2e-2 * lr + 1e-6 * np.abs(row[1:]).max(), (what, k)
        assert abs(d[0] - row[0]) <= 1e-5 * row[0] + 1e-7, (what, k)


def test_trainer_steps(golden_dir):
    """One real iteration of each reference trainer vs the oracle's step functions."""
    fx = _load(golden_dir, "trainers_small")
    meta = fx["meta"]
    B, C, H, W, s0, s1, s2, n = [int(v) for v in meta[:8]]
    nf = [int(v) for v in meta[8:]]
    cfg = O.PUNetConfig(C, 1, nf, 6, 3, 1.0, consensus_masking=True, rl_swap=False)
    Pnp = O.synth_params(cfg.spec(), s1)
    x, x1, x2, gt = O.synth_batch(B, C, H, W, s0)
    xs, _, _, ys = O.synth_batch(B, C, H, W, s0 + 50)
    eps_t, eps_s, eps_sup = _eps(n, B, 6, s2), _eps(1, B, 6, s2 + 100)[0], _eps(1, B, 6, s2 + 200)[0]
    lr = 1e-3
    for masking, tag in ((True, "mt_mask"), (False, "mt_weight")):
        Ps, Pt = O.to_torch(Pnp, True), O.to_torch(Pnp)
        r = O.mean_teacher_step(Ps, Pt, cfg, O.AdamState(Ps, lr=lr), x1, x2, eps_t, eps_s, masking)
        np.testing.assert_allclose(r["loss"].item(), fx[f"{tag}_loss"][0], rtol=2e-5)
        _check_digest(Ps, fx[f"{tag}_student"], lr, tag)
        _check_digest(Pt, fx[f"{tag}_teacher"], lr, tag)
    P = O.to_torch(Pnp, True)
    r = O.fixmatch_step(P, cfg, O.AdamState(P, lr=lr), x1, x2, eps_t, eps_s, True, torch.tensor(fx["fm_src_dist"]).float())
    np.testing.assert_allclose(r["loss"].item(), fx["fm_loss"][0], rtol=2e-5)
    _check_digest(P, fx["fm_student"], lr, "fm")
    for it0 in (0, 5):
        Ps = O.to_torch(Pnp, True)
        Pt = O.to_torch({k: v * np.float32(0.5) for k, v in Pnp.items()})
        r = O.adamt_step(Ps, Pt, cfg, O.AdamState(Ps, lr=lr), xs, ys, x1, x2, eps_sup, eps_t, eps_s, True, it0)
        np.testing.assert_allclose(r["loss"].item(), fx[f"adamt_it{it0}_loss"][0], rtol=2e-5)
        _check_digest(Ps, fx[f"adamt_it{it0}_student"], lr, "adamt")
        _check_digest(Pt, fx[f"adamt_it{it0}_teacher"], lr, "adamt-teacher")
    P = O.to_torch(Pnp, True)
    r = O.punet_step(P, cfg, O.AdamState(P, lr=lr), xs, ys, eps_sup)
    np.testing.assert_allclose(r["loss"].item(), fx["punet_loss"][0], rtol=2e-5)
    _check_digest(P, fx["punet_student"], lr, "punet")
    P = O.to_torch(Pnp, True)
    zmask = (torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(9)) > 0.3).long()
    r = O.punet_step(P, cfg, O.AdamState(P, lr=lr), xs, ys, eps_sup, consm=zmask)
    np.testing.assert_allclose(r["loss"].item(), fx["pseudo_loss"][0], rtol=2e-5)
    _check_digest(P, fx["pseudo_student"], lr, "pseudo")


def test_api_corners_validation_and_prediction(golden_dir):
    """api_small.npz (gen_golden.py::gen_api_case): kl_divergence / reconstruct / elbo option space, the two validation
    loops, sample_from_model, punet_pseudo_prediction and punet_prediction as the reference's own code produced them."""
    fx = _load(golden_dir, "api_small")
    B, C, H, W, s0, s1, s2, n = [int(v) for v in fx["meta"][:8]]
    nf = [int(v) for v in fx["meta"][8:]]
    cfg = O.PUNetConfig(C, 1, nf, 6, 3, 1.0, consensus_masking=True, rl_swap=False)
    Pnp = O.synth_params(cfg.spec(), s1)
    P = O.to_torch(Pnp)
    E = lambda k, m: _eps(k, B, 6, s2 + m)
    a, b = O.synth_batch(B, C, H, W, s0), O.synth_batch(B, C, H, W, s0 + 1)
    x, x1, x2, gt = a
    with torch.no_grad():
        st = O.punet_forward(P, cfg, x2, gt, True)
        (mq, sq), (mp, sp) = st["posterior"], st["prior"]
        np.testing.assert_allclose(mq.numpy(), fx["post_mu"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(sq.numpy(), fx["post_sigma"], rtol=1e-5)
        np.testing.assert_allclose(O.kl_diag_gaussians(mq, sq, mp, sp).numpy(), fx["kl_analytic"], rtol=1e-5, atol=1e-6)
        zg = torch.from_numpy(fx["z_given"])
        np.testing.assert_allclose(O.kl_sampled(mq, sq, mp, sp, zg).numpy(), fx["kl_sampled_z"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(O.kl_sampled(mq, sq, mp, sp, mq + sq * E(1, 2)[0]).numpy(), fx["kl_sampled_calc"], rtol=1e-5,
                                   atol=1e-5)
        np.testing.assert_allclose(O.fcomb(P, st["features"], zg, 3).numpy(), fx["recon_z"], atol=1e-5)
        np.testing.assert_allclose(O.fcomb(P, st["features"], mq + sq * E(1, 3)[0], 3).numpy(), fx["recon_calc"], atol=1e-5)
    consm = (torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(21)) > 0.4).float()
    for tag, kw, tgt in (("sampledkl", dict(analytic_kl=False), gt), ("othertarget", dict(), torch.roll(gt, shifts=3, dims=3))):
        Pg = O.to_torch(Pnp, requires_grad=True)
        lo, aux = O.training_loss(Pg, cfg, x2, gt, consm, E(1, 4)[0], target=tgt, **kw)
        sc = fx[f"elbo_{tag}_scalars"]
        np.testing.assert_allclose([lo.item(), aux["reconstruction_loss"].item(), aux["mean_reconstruction_loss"].item(),
                                    aux["kl"].item(), aux["reg"].item()], sc, rtol=1e-5)
        for row, g in zip(fx[f"elbo_{tag}_grad_digest"], O.grads_of(lo, Pg).values()):
            assert abs(float(g.double().pow(2).sum().sqrt()) - row[0]) <= 1e-4 * row[0] + 1e-8
    mo, lo_ = O.punet_validate(P, cfg, [(x, gt), (b[0], b[3])], [E(1, 10)[0], E(1, 11)[0]], [E(8, 20), E(8, 21)])
    np.testing.assert_allclose([mo, lo_], fx["punet_val"], rtol=1e-5)
    Pt = O.to_torch(O.synth_params(cfg.spec(), s1 + 1))
    for masking, tag in ((True, "mask"), (False, "weight")):
        r = O.mean_teacher_validate(P, Pt, cfg, [a, b], [E(n, 40), E(n, 41)], [E(1, 50)[0], E(1, 51)[0]], [E(n, 60), E(n, 61)],
                                    masking)
        np.testing.assert_allclose([r["metric"], r["loss"], r["gt_metric"]], fx[f"mt_val_{tag}"], rtol=1e-5)
        assert (r["z"].numpy() == fx[f"mt_val_{tag}_z"]).all()
        np.testing.assert_allclose(r["y"].numpy(), fx[f"mt_val_{tag}_y"], atol=1e-6)
    g1 = lambda k, seed: torch.randn(k, 1, 6, generator=torch.Generator().manual_seed(seed))
    mean, cons = O.pseudo_prediction(P, cfg, fx["img_a"], g1(5, s2 + 90))
    np.testing.assert_allclose(mean, fx["pseudo_mean"], atol=1e-6)
    assert (cons == fx["pseudo_consensus"]).all()
    bh, bw, hh, hw, ns, nt = [int(v) for v in fx["tiled_geo"]]
    tiled = O.tiled_prediction(P, cfg, fx["img_b"], [g1(ns, s2 + 100 + t) for t in range(nt)], (bh, bw), (hh, hw))
    np.testing.assert_allclose(tiled, fx["tiled_pred"], atol=1e-6)


def test_validation_loops_of_fixmatch_adamt_adamatch(golden_dir):
    """validate_small.npz (gen_golden.py::gen_validate_case): `_validate_impl` of FixMatchTrainer, AdaMTTrainer and
    AdaMatchTrainer (before / after consensus_masking_is_activated) as the reference's own code ran them."""
    fx = _load(golden_dir, "validate_small")
    B, C, H, W, s0, s1, s2, n = [int(v) for v in fx["meta"][:8]]
    nf = [int(v) for v in fx["meta"][8:]]
    cfg = O.PUNetConfig(C, 1, nf, 6, 3, 1.0, consensus_masking=True, rl_swap=False)
    P = O.to_torch(O.synth_params(cfg.spec(), s1))
    Pt = O.to_torch(O.synth_params(cfg.spec(), s1 + 1))
    E = lambda k, m: _eps(k, B, 6, s2 + m)
    batches = [O.synth_batch(B, C, H, W, s0), O.synth_batch(B, C, H, W, s0 + 1)]
    for masking, mk in ((True, "mask"), (False, "weight")):
        for which, teacher, use_z in (("fixmatch", P, True), ("adamt", Pt, True), ("adamatch_off", P, False),
                                      ("adamatch_on", P, True)):
            r = O.mean_teacher_validate(P, teacher, cfg, batches, [E(n, 40), E(n, 41)], [E(1, 50)[0], E(1, 51)[0]],
                                        [E(n, 60), E(n, 61)], masking, use_consensus=use_z)
            np.testing.assert_allclose([r["metric"], r["loss"], r["gt_metric"]], fx[f"{which}_{mk}_val"], rtol=1e-5)
            assert (r["z"].numpy() == fx[f"{which}_{mk}_z"]).all()
            np.testing.assert_allclose(r["y"].numpy(), fx[f"{which}_{mk}_y"], atol=1e-6)
