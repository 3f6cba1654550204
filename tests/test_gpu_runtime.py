"""GPU: the step-level native runtime (csrc/runtime.hip: forward / sample / elbo / backward as native host code behind the C ABI)
against the Python engine, which enqueues the same kernels in the same order: bit-identical pseudo-labels, logits, loss scalars and
gradients; a whole Mean-Teacher step driven through four library calls equals the fused Python step; an arena that is too small is
refused before anything is launched."""
import numpy as np
import os

import pytest
import torch

from oracle import punet_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _model(dev, nf, C=1, nfc=3, latent=6, seed=1, rl_swap=False):
    from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
    m = ProbabilisticUnet(C, 1, nf, latent, nfc, 1.0, consensus_masking=True, rl_swap=rl_swap, device=dev, seed=0)
    cfg = O.PUNetConfig(C, 1, nf, latent, nfc, 1.0, consensus_masking=True, rl_swap=False)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in O.synth_params(cfg.spec(), seed).items()})
    m.set_conv_mode("x6")  # (explicit: these tests are about the graph walkers, whatever $PUNET_CONV_MODE says)
    return m


def _random_configs(count, seed=2024):
    """Seeded draws from the configuration class the runtime covers (runtime.supported): depth, widths, input channels, Fcomb depth,
    latent size, batch, patch shape, arithmetic and criterion all vary together."""
    import random
    r = random.Random(seed)
    out = []
    for i in range(count):
        levels = r.choice([3, 4, 4, 5])
        nf = [r.choice([32, 64])] + [8 * r.randint(2, 12) for _ in range(levels - 1)]
        q = 2 ** (levels - 1)
        H, W = q * r.randint(2, 5), q * r.randint(2, 6)
        out.append(pytest.param(nf, r.randint(1, 4), r.choice([2, 3]), r.choice([2, 4, 6]), r.randint(1, 3), H, W,
                                r.choice(["x6", "x6", "f16x3", "f16"]), r.choice(["bce", "dice"]), id=f"random{i}"))
    return out


@pytest.mark.parametrize("nf,C,nfc,latent,B,H,W,mode,crit",
                         [pytest.param([32, 64, 64, 96], 1, 3, 6, 2, 32, 64, "x6", "bce", id="nf4"),
                          pytest.param([64, 40, 72], 3, 2, 2, 1, 24, 40, "x6", "bce", id="depth3_rgb_fcomb2_latent2"),
                          pytest.param([32, 32, 64, 64, 96], 2, 3, 6, 2, 48, 80, "x6", "bce", id="depth5_two_channels")]
                         + _random_configs(*[int(v) for v in os.environ.get("PUNET_FUZZ_RT", "24:2024").split(":")]))
def test_native_runtime_equals_python_engine_bit_for_bit(dev, nf, C, nfc, latent, B, H, W, mode, crit):
    from probabilistic_domain_adaptation_amd.runtime import NativeRuntime
    m = _model(dev, nf, C, nfc, latent, rl_swap=crit == "dice")
    m.set_conv_mode(mode)
    eng, flat = m.engine, m.flat_params
    rt = NativeRuntime(m.layout, dev, conv_mode=mode, dice=crit == "dice")
    assert [e[:3] for e in rt.entries()] == [e[:3] for e in m.layout.entries] and rt.unet_numel == m.layout.unet_numel
    x, x1, x2, y = [t.to(dev) for t in O.synth_batch(B, C, H, W, 7)]
    g = torch.Generator().manual_seed(3)
    n = 4
    eps_n, eps_s = torch.randn(n, B, latent, generator=g).to(dev), torch.randn(B, latent, generator=g).to(dev)
    consm = (torch.rand(B, 1, H, W, generator=g) > 0.3).float().to(dev)
    # ---- teacher / inference path
    with torch.no_grad():
        out = eng.forward_features(flat, eng.to_nhwc(x1))
        ye, ze, le = eng.sample(flat, out["features"], eng.reparam(out["prior_muls"], eps_n), 0.9, 0.1, True, want_logits=True)
    yr, zr, lr = rt.pseudo_labels(flat, eng.to_nhwc(x1), eps_n, 0.9, 0.1, True, want_logits=True)
    torch.cuda.synchronize()
    assert torch.equal(ye, yr) and torch.equal(ze, zr) and torch.equal(le, lr)
    if (H * W) % 32:  # the gradient pass needs whole 32-pixel tiles per image: refused loudly, the trainers take the Python engine
        with pytest.raises(RuntimeError, match="multiple of 32"):
            rt.elbo_fwd_bwd(flat, torch.zeros_like(flat), eng.to_nhwc(x2), y, consm, eps_s, 1.0, 0.5, False)
        return
    # ---- student path: loss scalars, logits, every gradient; then the accumulate form on top
    st = eng.elbo_forward(flat, eng.to_nhwc(x2), y, consm, eps_s, crit, 1.0, 0.5, True)
    SENT = 12345.678  # (the flat buffer has alignment pads nobody writes: a NaN fill would defeat torch.equal)
    ge = torch.full_like(flat, SENT)
    eng.elbo_backward(flat, ge, st, False)
    gr = torch.full_like(flat, SENT)
    hits = []
    rec, kl, lg = rt.elbo_fwd_bwd(flat, gr, eng.to_nhwc(x2), y, consm, eps_s, 1.0, 0.5, False, after_unet=lambda: hits.append(1),
                                  want_logits=True)
    torch.cuda.synchronize()
    assert hits == [1]  # the data-parallel hook fired once (and not in the sizing walk)
    assert torch.equal(rec, st["rec"]) and torch.equal(kl, st["kl"]) and torch.equal(lg, st["logits"])
    assert torch.isfinite(gr).all() and torch.equal(ge, gr)
    for k, v in m.layout.views(gr).items():
        assert not (v == SENT).any(), k  # every gradient tensor was written
    eng.elbo_backward(flat, ge, eng.elbo_forward(flat, eng.to_nhwc(x1), y, None, eps_s, crit, 1.0, 0.5, True), True)
    rt.elbo_fwd_bwd(flat, gr, eng.to_nhwc(x1), y, None, eps_s, 1.0, 0.5, True)
    torch.cuda.synchronize()
    assert torch.equal(ge, gr)
    # ---- regulariser
    reg_e, norms = eng.l2_reg(flat)
    eng.l2_reg_add_grad(flat, ge, norms, 1e-5)
    reg_r = rt.l2_reg_grad(flat, gr, 1e-5)
    torch.cuda.synchronize()
    assert torch.equal(reg_e, reg_r) and torch.equal(ge, gr)
    # ---- the arena bound is exact, and a smaller arena is refused before anything runs
    need = rt.arena_bytes(B, H, W, 1, True)
    from probabilistic_domain_adaptation_amd import _lib
    assert _lib.call("punet_rt_last_arena_used", rt._h) > 0
    small = torch.empty(need // 2, dtype=torch.uint8, device=dev)
    before = gr.clone()
    with pytest.raises(RuntimeError, match="arena"):
        rt.elbo_fwd_bwd(flat, gr, eng.to_nhwc(x2), y, consm, eps_s, arena=small)
    torch.cuda.synchronize()
    assert torch.equal(before, gr)
    # ---- so is an arena whose base is not 256-byte aligned (the kernels read what is carved from it with 16-byte vector loads)
    skew = torch.empty(need + 512, dtype=torch.uint8, device=dev)[16:]
    with pytest.raises(RuntimeError, match="256-byte aligned"):
        rt.elbo_fwd_bwd(flat, gr, eng.to_nhwc(x2), y, consm, eps_s, arena=skew)
    with pytest.raises(RuntimeError, match="256-byte aligned"):
        rt.pseudo_labels(flat, eng.to_nhwc(x1), eps_n, 0.9, 0.1, True, arena=skew)
    torch.cuda.synchronize()
    assert torch.equal(before, gr)


def test_mean_teacher_step_through_four_library_calls(dev, monkeypatch):
    """pseudo-labels, ELBO forward / backward, regulariser, Adam + EMA: the fused Python step (my_trainer/fused.py) and the same
    step driven through the native runtime leave identical weights, teacher and Adam moments."""
    import copy
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    from probabilistic_domain_adaptation_amd.runtime import NativeRuntime
    nf, B, H, W, n = [64, 64, 96, 128], 2, 64, 96, 4
    x, x1, x2, y = [t.to(dev) for t in O.synth_batch(B, 1, H, W, 0)]
    g = torch.Generator().manual_seed(5)
    eps_n, eps_s = torch.randn(n, B, 6, generator=g).to(dev), torch.randn(1, B, 6, generator=g).to(dev)
    res = []
    for native in (False, True):
        monkeypatch.setenv("PUNET_RUNTIME", "native" if native else "python")  # the fused steps dispatch on it (FU.native_ok)
        m = _model(dev, nf)
        assert FU.native_ok(m) == native
        t = copy.deepcopy(m)
        opt = FusedAdam(m, lr=1e-3)
        for it in range(2):
            loss, ys, zs, _ = FU.fused_self_training_step(m, t, opt, x1, x2, n, True, eps_pl=eps_n, eps_s=eps_s)
        torch.cuda.synchronize()
        res.append((float(loss), m.flat_params.clone(), t.flat_params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), ys.clone(),
                    zs.clone()))
    assert np.isfinite(res[0][0]) and res[0][0] == res[1][0]
    for a, b in zip(res[0][1:], res[1][1:]):
        assert torch.equal(a, b)
    assert not torch.equal(res[0][1], res[0][2])  # the EMA teacher really differs from the student


def test_a_latent_size_the_runtime_refuses_takes_the_python_engine(dev, monkeypatch):
    """latent_dim = 65 is inside the per-op kernels' constructor space and outside punet_rt_create's; the fused step,
    the MC pseudo-labels and the oracle must agree instead of the dispatch raising."""
    import copy
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    from probabilistic_domain_adaptation_amd.runtime import supported
    monkeypatch.setenv("PUNET_RUNTIME", "native")
    nf, B, H, W, n, latent = [32, 32, 64, 64], 2, 32, 32, 3, 65
    m = _model(dev, nf, latent=latent)
    assert not supported(m.layout) and not FU.native_ok(m)
    t = copy.deepcopy(m)
    x, x1, x2, y = [v.to(dev) for v in O.synth_batch(B, 1, H, W, 0)]
    g = torch.Generator().manual_seed(5)
    eps_n, eps_s = torch.randn(n, B, latent, generator=g), torch.randn(1, B, latent, generator=g)
    yp, zp = m.pseudo_labels(x1, n, masking=True, eps=eps_n.to(dev))
    cfg = O.PUNetConfig(1, 1, nf, latent, 3, 1.0, consensus_masking=True, rl_swap=False)
    Po = O.to_torch(O.synth_params(cfg.spec(), 1))
    yo, zo = O.sample_consensus(Po, cfg, x1.cpu(), eps_n, 0.9, 0.1, True)
    torch.cuda.synchronize()
    assert float((yp.cpu() - yo).abs().max()) < 2e-5 and torch.equal(zp.cpu(), zo.float())
    opt = FusedAdam(m, lr=1e-3)
    before = m.flat_params.clone()
    loss, ys, zs, _ = FU.fused_self_training_step(m, t, opt, x1, x2, n, True, eps_pl=eps_n.to(dev), eps_s=eps_s.to(dev))
    torch.cuda.synchronize()
    assert np.isfinite(float(loss)) and not torch.equal(before, m.flat_params)
    assert torch.equal(ys, yp) and torch.equal(zs, zp)


def test_joint_and_supervised_steps_native_equal_python(dev, monkeypatch):
    """AdaMT-style joint step (two accumulating gradient passes, warm-up EMA momentum from the device counter) and the supervised
    step: native runtime == Python engine, bit for bit, over two iterations."""
    import copy
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    nf, B, H, W, n = [32, 64, 64, 96], 2, 32, 64, 4
    xs, _, _, ys = [t.to(dev) for t in O.synth_batch(B, 1, H, W, 11)]
    x, x1, x2, y = [t.to(dev) for t in O.synth_batch(B, 1, H, W, 12)]
    g = torch.Generator().manual_seed(6)
    e_sup, e_pl, e_s = torch.randn(1, B, 6, generator=g).to(dev), torch.randn(n, B, 6, generator=g).to(dev), torch.randn(1, B, 6, generator=g).to(dev)
    res = {}
    for rtm in ("python", "native"):
        monkeypatch.setenv("PUNET_RUNTIME", rtm)
        m = _model(dev, nf)
        t = copy.deepcopy(m)
        opt = FusedAdam(m, lr=1e-3)
        for it in range(2):
            loss, yy, zz = FU.fused_joint_step(m, t, opt, xs, ys, x1, x2, n, True, e_sup, e_pl, e_s, 0.999, True, 1e-5, it)
        m2 = _model(dev, nf, seed=2)
        opt2 = FusedAdam(m2, lr=1e-3)
        l2 = FU.fused_supervised_step(m2, opt2, x2, y, None, e_s, 1e-5)
        torch.cuda.synchronize()
        res[rtm] = (float(loss), m.flat_params.clone(), t.flat_params.clone(), yy.clone(), zz.clone(), float(l2), m2.flat_params.clone())
    assert res["python"][0] == res["native"][0] and res["python"][5] == res["native"][5]
    for a, b in zip(res["python"][1:5] + res["python"][6:], res["native"][1:5] + res["native"][6:]):
        assert torch.equal(a, b)


def test_one_call_self_training_step(dev, monkeypatch):
    """punet_rt_self_training_step - the whole Mean-Teacher iteration as ONE library call (what a non-Python host would issue) -
    against the fused Python step: same weights, teacher, moments, pseudo-labels and loss scalars after two iterations."""
    import copy
    from probabilistic_domain_adaptation_amd import _lib
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    nf, B, H, W, n = [32, 64, 64, 96], 2, 32, 64, 4
    x, x1, x2, y = [t.to(dev) for t in O.synth_batch(B, 1, H, W, 21)]
    g = torch.Generator().manual_seed(8)
    eps_n, eps_s = torch.randn(n, B, 6, generator=g).to(dev), torch.randn(1, B, 6, generator=g).to(dev)
    monkeypatch.setenv("PUNET_RUNTIME", "python")
    ma = _model(dev, nf)
    ta = copy.deepcopy(ma)
    oa = FusedAdam(ma, lr=1e-3)
    for it in range(2):
        la, ya, za, _ = FU.fused_self_training_step(ma, ta, oa, x1, x2, n, True, eps_pl=eps_n, eps_s=eps_s)
    mb = _model(dev, nf)
    tb = copy.deepcopy(mb)
    ob = FusedAdam(mb, lr=1e-3)
    rt = mb.native_runtime()
    ar = rt.arena(max(rt.arena_bytes(B, H, W, n, True), rt.arena_bytes(B, H, W, n, False)))
    yb, zb = torch.empty(B, 1, H, W, device=dev), torch.empty(B, 1, H, W, device=dev)
    sc = torch.zeros(3, device=dev)
    eng = mb.engine
    for it in range(2):
        want = ob.sync_device_state(0.999, 1.0, None)  # the host's part of the optimizer: the device state block
        _lib.call("punet_rt_self_training_step", rt._h, mb.flat_params.data_ptr(), tb.flat_params.data_ptr(), ob.grad_flat.data_ptr(),
                  ob.exp_avg.data_ptr(), ob.exp_avg_sq.data_ptr(), ob.hstate.data_ptr(), ob.hyper.data_ptr(), 1, eng.to_nhwc(x1).data_ptr(),
                  eng.to_nhwc(x2).data_ptr(), B, H, W, eps_n.data_ptr(), n, eps_s.data_ptr(), 0.9, 0.1, 1, 1, 1.0, 1e-5, 0.9, 0.999, 1e-8,
                  yb.data_ptr(), zb.data_ptr(), sc.data_ptr(), None, ar.data_ptr(), ar.numel(), torch.cuda.current_stream().cuda_stream)
        ob._advance_mirrors(want)
    torch.cuda.synchronize()
    assert torch.equal(ma.flat_params, mb.flat_params) and torch.equal(ta.flat_params, tb.flat_params)
    assert torch.equal(oa.exp_avg, ob.exp_avg) and torch.equal(oa.exp_avg_sq, ob.exp_avg_sq)
    assert torch.equal(ya, yb) and torch.equal(za, zb)
    assert float(la) == float(sc[0] + 1.0 * sc[1] + 1e-5 * sc[2])


@pytest.mark.parametrize("mode", ["f16x3", "f16"])
def test_native_runtime_in_the_fp16_operand_modes(dev, monkeypatch, mode):
    """The runtime carves the amax slots out of its arena and handles them itself (first-layer publication, inherited slots through
    fused pooling and up-sampling, bounds through their backward, standalone passes behind the Fcomb / latent-head backward): a
    Mean-Teacher step in f16x3 / f16 equals the Python engine's bit for bit."""
    import copy
    from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    nf, B, H, W, n = [64, 64, 96, 128], 2, 64, 96, 4
    x, x1, x2, y = [t.to(dev) for t in O.synth_batch(B, 1, H, W, 0)]
    g = torch.Generator().manual_seed(5)
    eps_n, eps_s = torch.randn(n, B, 6, generator=g).to(dev), torch.randn(1, B, 6, generator=g).to(dev)
    res = {}
    for rtm in ("python", "native"):
        monkeypatch.setenv("PUNET_RUNTIME", rtm)
        m = ProbabilisticUnet(1, 1, nf, 6, 3, 1.0, consensus_masking=True, device=dev, seed=1, conv_mode=mode)
        t = copy.deepcopy(m)
        assert FU.native_ok(m, t) == (rtm == "native")
        opt = FusedAdam(m, lr=1e-3)
        for it in range(2):
            loss, ys, zs, _ = FU.fused_self_training_step(m, t, opt, x1, x2, n, False, eps_pl=eps_n, eps_s=eps_s)
        torch.cuda.synchronize()
        res[rtm] = (float(loss), m.flat_params.clone(), t.flat_params.clone(), ys.clone(), zs.clone())
    assert np.isfinite(res["native"][0]) and res["python"][0] == res["native"][0]
    for a, b in zip(res["python"][1:], res["native"][1:]):
        assert torch.equal(a, b)


def test_native_runtime_with_the_dice_criterion(dev, monkeypatch):
    """rl_swap=True - what every LIVECell / MitoEM script of the reference builds: the Mean-Teacher step through the native runtime
    (punet_rt_set_criterion) equals the Python engine's, bit for bit."""
    import copy
    from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    nf, B, H, W, n = [64, 64, 96, 128], 2, 64, 96, 4
    x, x1, x2, y = [t.to(dev) for t in O.synth_batch(B, 1, H, W, 0)]
    g = torch.Generator().manual_seed(5)
    eps_n, eps_s = torch.randn(n, B, 6, generator=g).to(dev), torch.randn(1, B, 6, generator=g).to(dev)
    res = {}
    for rtm in ("python", "native"):
        monkeypatch.setenv("PUNET_RUNTIME", rtm)
        m = ProbabilisticUnet(1, 1, nf, 6, 3, 1.0, consensus_masking=True, rl_swap=True, device=dev, seed=1)
        m.set_conv_mode("x6")
        t = copy.deepcopy(m)
        assert FU.native_ok(m, t) == (rtm == "native")
        opt = FusedAdam(m, lr=1e-3)
        for it in range(2):
            loss, ys, zs, _ = FU.fused_self_training_step(m, t, opt, x1, x2, n, True, eps_pl=eps_n, eps_s=eps_s)
        torch.cuda.synchronize()
        res[rtm] = (float(loss), m.flat_params.clone(), t.flat_params.clone(), ys.clone(), zs.clone())
    assert np.isfinite(res["native"][0]) and 0.0 < res["native"][0] and res["python"][0] == res["native"][0]
    for a, b in zip(res["python"][1:], res["native"][1:]):
        assert torch.equal(a, b)


def test_one_call_supervised_and_joint_steps(dev, monkeypatch):
    """punet_rt_supervised_step / punet_rt_joint_step (AdaMT with its warm-up EMA momentum from the device counter) as single
    library calls against the fused Python steps."""
    import copy
    from probabilistic_domain_adaptation_amd import _lib
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    nf, B, H, W, n = [32, 64, 64, 96], 2, 32, 64, 4
    xs, _, _, ys = [t.to(dev) for t in O.synth_batch(B, 1, H, W, 11)]
    x, x1, x2, y = [t.to(dev) for t in O.synth_batch(B, 1, H, W, 12)]
    g = torch.Generator().manual_seed(6)
    e_sup, e_pl, e_s = torch.randn(1, B, 6, generator=g).to(dev), torch.randn(n, B, 6, generator=g).to(dev), torch.randn(1, B, 6, generator=g).to(dev)
    S = torch.cuda.current_stream().cuda_stream
    monkeypatch.setenv("PUNET_RUNTIME", "python")
    ma = _model(dev, nf); ta = copy.deepcopy(ma); oa = FusedAdam(ma, lr=1e-3)
    for it in range(2):
        la, ya, za = FU.fused_joint_step(ma, ta, oa, xs, ys, x1, x2, n, True, e_sup, e_pl, e_s, 0.999, True, 1e-5, it)
    ms = _model(dev, nf, seed=2); os_ = FusedAdam(ms, lr=1e-3)
    ls = FU.fused_supervised_step(ms, os_, x2, y, None, e_s, 1e-5)
    # ---- the same through single library calls
    mb = _model(dev, nf); tb = copy.deepcopy(mb); ob = FusedAdam(mb, lr=1e-3)
    rt = mb.native_runtime()
    ar = rt.arena(max(rt.arena_bytes(B, H, W, n, True), rt.arena_bytes(B, H, W, n, False)))
    yb, zb, sc = torch.empty(B, 1, H, W, device=dev), torch.empty(B, 1, H, W, device=dev), torch.zeros(5, device=dev)
    eng = mb.engine
    for it in range(2):
        want = ob.sync_device_state(0.999, 1.0, it)  # AdaMT: momentum min(1 - 1/(it + 1), 0.999), evaluated on the device
        _lib.call("punet_rt_joint_step", rt._h, mb.flat_params.data_ptr(), tb.flat_params.data_ptr(), ob.grad_flat.data_ptr(),
                  ob.exp_avg.data_ptr(), ob.exp_avg_sq.data_ptr(), ob.hstate.data_ptr(), ob.hyper.data_ptr(), 1, eng.to_nhwc(xs).data_ptr(),
                  ys.data_ptr(), B, eng.to_nhwc(x1).data_ptr(), eng.to_nhwc(x2).data_ptr(), B, H, W, e_sup.data_ptr(), e_pl.data_ptr(), n,
                  e_s.data_ptr(), 0.9, 0.1, 1, 1, 1.0, 1e-5, 0.9, 0.999, 1e-8, yb.data_ptr(), zb.data_ptr(), sc.data_ptr(), None,
                  ar.data_ptr(), ar.numel(), S)
        ob._advance_mirrors(want)
    mt = _model(dev, nf, seed=2); ot = FusedAdam(mt, lr=1e-3)
    rt2 = mt.native_runtime()
    ar2 = rt2.arena(rt2.arena_bytes(B, H, W, 1, True))
    sc2 = torch.zeros(3, device=dev)
    want = ot.sync_device_state(0.0, 1.0, None)
    _lib.call("punet_rt_supervised_step", rt2._h, mt.flat_params.data_ptr(), ot.grad_flat.data_ptr(), ot.exp_avg.data_ptr(),
              ot.exp_avg_sq.data_ptr(), ot.hstate.data_ptr(), ot.hyper.data_ptr(), mt.engine.to_nhwc(x2).data_ptr(), y.data_ptr(), None,
              B, H, W, e_s.data_ptr(), 1.0, 1e-5, 0.9, 0.999, 1e-8, sc2.data_ptr(), None, ar2.data_ptr(), ar2.numel(), S)
    ot._advance_mirrors(want)
    torch.cuda.synchronize()
    assert torch.equal(ma.flat_params, mb.flat_params) and torch.equal(ta.flat_params, tb.flat_params)
    assert torch.equal(ya, yb) and torch.equal(za, zb)
    assert abs(float(la) - float(0.5 * (sc[0] + sc[1] + sc[2] + sc[3]) + 1e-5 * sc[4])) <= 1e-6 * abs(float(la))
    assert torch.equal(ms.flat_params, mt.flat_params)
    assert abs(float(ls) - float(sc2[0] + sc2[1] + 1e-5 * sc2[2])) <= 1e-6 * abs(float(ls))


@pytest.mark.parametrize("dice,products,rccl", [(0, 6, 0), (1, 3, 0), (0, 6, 1)])
def test_c_host_runs_the_training_loop(dev, tmp_path, dice, products, rccl):
    """tools/c_host/mean_teacher_host.c - plain C99, no Python, no torch - drives three Mean-Teacher iterations through
    punet_rt_self_training_step; students, teachers, pseudo-labels, masks and loss terms equal the Python trainers' fused step
    bit for bit (BCE in the parity arithmetic; Dice in f16x3).  rccl = 1: the same loop as a DATA-PARALLEL host - an RCCL
    communicator built in C, the flat gradient all-reduced through PunetRtHooks (U-Net slice on a communication stream under the
    encoders' backward, remainder before the update): with the one rank a one-GPU box can hold the collectives are the identity
    and the result must not change by a bit."""
    import copy
    import struct
    import subprocess
    import numpy as np
    import __graft_entry__ as ge
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    exe = ge.build_c_host()
    nf, B, H, W, n, iters, lr = [32, 64, 64, 96], 2, 32, 64, 4, 3, 1e-3
    m = _model(dev, nf, rl_swap=bool(dice))
    m.set_conv_mode("f16x3" if products == 3 else "x6")
    t = copy.deepcopy(m)
    x, x1, x2, y = [v.to(dev) for v in O.synth_batch(B, 1, H, W, 21)]
    g = torch.Generator().manual_seed(8)
    e_pl, e_s = torch.randn(n, B, 6, generator=g), torch.randn(1, B, 6, generator=g)
    case, out = tmp_path / "case.bin", tmp_path / "out.bin"
    with open(case, "wb") as f:
        f.write(struct.pack(f"<4i{len(nf)}i7id", 1, len(nf), 6, 3, *nf, B, H, W, n, iters, dice, products, lr))
        for a in (m.flat_params, m.engine.to_nhwc(x1), m.engine.to_nhwc(x2), e_pl, e_s):
            f.write(a.detach().cpu().contiguous().numpy().astype(np.float32).tobytes())
    env = dict(os.environ, PUNET_C_HOST_RCCL=str(rccl), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, str(case), str(out)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout
    if rccl:
        if "data parallel: RCCL communicator" not in r.stdout:
            pytest.skip("the C host was built without RCCL (no rccl.h / librccl.so in this image)")
    opt = FusedAdam(m, lr=lr)
    for _ in range(iters):
        loss, yy, zz, _ = FU.fused_self_training_step(m, t, opt, x1, x2, n, True, e_pl.to(dev), e_s.to(dev), 0.999, 1e-5)
    torch.cuda.synchronize()
    got = np.fromfile(out, dtype=np.float32)
    P, px = m.flat_params.numel(), B * H * W
    assert got.size == 2 * P + 2 * px + 3
    assert np.array_equal(got[:P], m.flat_params.cpu().numpy()), r.stdout
    assert np.array_equal(got[P:2 * P], t.flat_params.cpu().numpy())
    assert np.array_equal(got[2 * P:2 * P + px], yy.cpu().numpy().ravel()) and np.array_equal(got[2 * P + px:2 * P + 2 * px], zz.cpu().numpy().ravel())
    rec, kl, reg = got[-3:]
    assert abs(float(loss) - (float(rec) + float(kl) + 1e-5 * float(reg))) <= 1e-6 * abs(float(loss))
