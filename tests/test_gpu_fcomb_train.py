"""GPU: K7t, the fused Fcomb training kernel (forward + BCE + whole backward in one launch), against (a) the per-op path it
replaces and (b) the CPU oracle's autograd - logits, loss, d loss / d features, every Fcomb weight / bias gradient and the
latent gradient dz - for CH in {32, 64}, with and without a middle layer, with and without the consensus mask, a pixel count
that does not fill the last workgroup, gradient accumulation, and run-to-run bit-reproducibility."""
import numpy as np
import pytest
import torch

from oracle import punet_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _setup(dev, nf, no_convs_fcomb, B, H, W, fuse):
    from probabilistic_domain_adaptation_amd.engine import ParamLayout, PUNetEngine
    cfg = O.PUNetConfig(1, 1, nf, 6, no_convs_fcomb, 1.0, consensus_masking=True, rl_swap=False)
    Pnp = O.synth_params(cfg.spec(), 2)
    lay = ParamLayout(1, 1, nf, 6, no_convs_fcomb)
    eng = PUNetEngine(lay, dev)
    eng.fuse_fcomb = fuse
    flat = torch.zeros(lay.numel, device=dev)
    for k, v in lay.views(flat).items():
        v.copy_(torch.from_numpy(Pnp[k]))
    return cfg, Pnp, lay, eng, flat


def _run(dev, nf, ncf, B, H, W, fuse, masked, accumulate=False, loss_kind="bce"):
    cfg, Pnp, lay, eng, flat = _setup(dev, nf, ncf, B, H, W, fuse)
    cfg.rl_swap = loss_kind == "dice"
    x, x1, x2, y = O.synth_batch(B, 1, H, W, 5)
    g = torch.Generator().manual_seed(6)
    eps = torch.randn(B, 6, generator=g)
    consm = (torch.rand(B, 1, H, W, generator=g) > 0.3).float() if masked else None
    st = eng.elbo_forward(flat, eng.to_nhwc(x2.to(dev)), y.to(dev), consm.to(dev) if masked else None, eps.to(dev), loss_kind, 1.0, 0.5, True)
    assert (st.get("fused_fcomb") is not None) == fuse
    gflat = torch.full((lay.numel,), 0.25 if accumulate else 0.0, device=dev)  # (alignment pads keep the fill value)
    eng.elbo_backward(flat, gflat, st, accumulate=accumulate)
    torch.cuda.synchronize()
    return cfg, Pnp, lay, st, gflat, (x2, y, consm, eps)


@pytest.mark.parametrize("nf,ncf", [([64, 64, 64, 64], 3), ([32, 64, 64, 64], 3), ([64, 64, 64, 64], 2), ([32, 32, 64, 64], 2)],
                         ids=["C64-mid1", "C32-mid1", "C64-mid0", "C32-mid0"])
@pytest.mark.parametrize("masked", [True, False])
def test_fused_equals_per_op_path_and_oracle(dev, nf, ncf, masked):
    B, H, W = 2, 40, 24  # 1 920 pixels = 60 tiles: the last workgroup of four waves is only partly filled at some grid sizes
    cfg, Pnp, lay, st_f, g_f, (x2, y, consm, eps) = _run(dev, nf, ncf, B, H, W, True, masked)
    _, _, _, st_u, g_u, _ = _run(dev, nf, ncf, B, H, W, False, masked)
    assert (st_f["logits"] - st_u["logits"]).abs().max().item() < 2e-5
    np.testing.assert_allclose(st_f["rec"].item(), st_u["rec"].item(), rtol=1e-5)
    vf, vu = lay.views(g_f), lay.views(g_u)
    for k in vf:  # EVERY parameter gradient: the Fcomb ones directly, all others through d loss / d features and dz
        a, b = vf[k], vu[k]
        assert torch.isfinite(a).all(), k
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-9, k
    # and against the oracle's autograd (gscale 0.5, no regulariser)
    P = O.to_torch(Pnp, requires_grad=True)
    lo, aux = O.training_loss(P, cfg, x2, y, consm, eps, reg_weight=0.0)
    go = O.grads_of(0.5 * lo, P)
    np.testing.assert_allclose(st_f["rec"].item(), aux["reconstruction_loss"].item(), rtol=2e-4)
    for k, ref in go.items():
        if k.startswith("fcomb."):
            e = (vf[k].cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-8)
            assert e < 5e-4, (k, e)


def test_fused_accumulates_and_is_bit_reproducible(dev):
    nf = [64, 64, 64, 64]
    _, _, lay, st1, g1, _ = _run(dev, nf, 3, 2, 32, 32, True, True)
    _, _, _, st2, g2, _ = _run(dev, nf, 3, 2, 32, 32, True, True)
    assert torch.equal(g1, g2) and torch.equal(st1["logits"], st2["logits"]) and torch.equal(st1["rec"], st2["rec"])
    _, _, _, _, g3, _ = _run(dev, nf, 3, 2, 32, 32, True, True, accumulate=True)
    v1, v3 = lay.views(g1), lay.views(g3)
    for k in v1:
        assert (v3[k] - (v1[k] + 0.25)).abs().max().item() <= 1e-6 * v1[k].abs().max().item() + 1e-6, k


@pytest.mark.parametrize("nf,ncf", [([64, 64, 64, 64], 3), ([32, 32, 64, 64], 2)], ids=["C64-mid1", "C32-mid0"])
@pytest.mark.parametrize("masked", [True, False])
def test_fused_dice_equals_per_op_path_and_oracle(dev, nf, ncf, masked):
    """The Dice criterion (rl_swap=True: every LIVECell / MitoEM script; recalled torch_em definition, parity unpinned) through the
    fused kernel - two launches, the ratio is global - against the per-op path (dot, dice_sums, dice_bwd, ...) and the oracle's
    restatement of the same definition: logits, loss, every gradient."""
    B, H, W = 2, 40, 24
    cfg, Pnp, lay, st_f, g_f, (x2, y, consm, eps) = _run(dev, nf, ncf, B, H, W, True, masked, loss_kind="dice")
    _, _, _, st_u, g_u, _ = _run(dev, nf, ncf, B, H, W, False, masked, loss_kind="dice")
    assert (st_f["logits"] - st_u["logits"]).abs().max().item() < 2e-5
    np.testing.assert_allclose(st_f["rec"].item(), st_u["rec"].item(), rtol=1e-5)
    vf, vu = lay.views(g_f), lay.views(g_u)
    for k in vf:
        a, b = vf[k], vu[k]
        assert torch.isfinite(a).all(), k
        assert (a - b).abs().max().item() <= 5e-5 * b.abs().max().item() + 1e-9, k
    P = O.to_torch(Pnp, requires_grad=True)
    lo, aux = O.training_loss(P, cfg, x2, y, consm, eps, reg_weight=0.0)
    go = O.grads_of(0.5 * lo, P)
    np.testing.assert_allclose(st_f["rec"].item(), aux["reconstruction_loss"].item(), rtol=2e-4)
    for k, ref in go.items():
        if k.startswith("fcomb."):
            e = (vf[k].cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-8)
            assert e < 2e-3, (k, e)


def test_unsupported_configurations_use_the_per_op_path(dev):
    """Two middle layers (the reference's default no_convs_fcomb = 4) keep the per-op kernels."""
    cfg, Pnp, lay, eng, flat = _setup(dev, [32, 64, 64, 64], 4, 1, 32, 32, True)
    x, x1, x2, y = O.synth_batch(1, 1, 32, 32, 5)
    eps = torch.zeros(1, 6)
    st = eng.elbo_forward(flat, eng.to_nhwc(x2.to(dev)), y.to(dev), None, eps.to(dev), "bce", 1.0, 1.0, True)
    assert st.get("fused_fcomb") is None
    cfg, Pnp, lay, eng, flat = _setup(dev, [32, 64, 64, 64], 3, 1, 32, 32, True)
    st = eng.elbo_forward(flat, eng.to_nhwc(x2.to(dev)), y.to(dev), None, eps.to(dev), "dice", 1.0, 1.0, True)
    assert st.get("fused_fcomb") is not None  # (round 3: the Dice criterion runs through the fused kernel too)
