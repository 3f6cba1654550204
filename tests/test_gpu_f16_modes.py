"""GPU: the fp16-operand fast modes ("f16x3": two fp16 planes, hh + hl + lh; "f16": one plane = what the reference's fp16
autocast computes with, LIVECell/livecell_mt.py:107) with per-tensor power-of-two scaling from amax slots (csrc/common.h).
They are NOT parity modes; these tests pin their error bands against PyTorch CPU fp32 / the reference's logits, the scaling
machinery itself (tensors far outside fp16's range), and what VERDICT r2 #1 asks of them: f16x3 logits within 1e-4 of the
reference's, f16 at most the bf16 figure / 8."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import punet_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from probabilistic_domain_adaptation_amd import _lib
    _lib.load()
    return _lib


def rel(a, b):
    return (a - b).abs().max().item() / b.abs().max().item()  # (no absolute epsilon: some cases are deliberately tiny)


def _amax_slot(L, t, S):
    s = torch.zeros(1, device="cuda")
    L.call("punet_amax", t.data_ptr(), t.numel(), s.data_ptr(), 0, S)
    return s


# (planes, magnitude of x, of w, of dz): tensors in fp16's comfort zone and far outside it in both directions - the amax
# scaling must make the result independent of the magnitude (power-of-two scaling commutes with the split)
@pytest.mark.parametrize("planes,sx,sw,sdz", [(2, 1.0, 1.0, 1.0), (2, 3e4, 1e-6, 1e-9), (2, 1e-7, 5e3, 4e5), (1, 1.0, 1.0, 1.0),
                                              (1, 2e-8, 1.0, 1e-10)],
                         ids=["f16x3", "f16x3-huge-x-tiny-w-dz", "f16x3-tiny-x-huge-w-dz", "f16", "f16-tiny"])
@pytest.mark.parametrize("cin,cout", [(96, 64), (40, 72)], ids=["16x16x32-kernel", "32x32x16-kernel"])
def test_kernels_error_bands_and_scaling(L, planes, sx, sw, sdz, cin, cout):
    S = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 24, 40
    x = (sx * torch.randn(B, cin, H, W, generator=g)).requires_grad_(True)
    w = (sw * torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)).requires_grad_(True)
    b = (sx * sw * torch.randn(cout, generator=g)).requires_grad_(True)
    dz = sdz * torch.randn(B, cout, H, W, generator=g)
    y = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    y.backward(dz.double())
    Wp = w.detach().permute(2, 3, 1, 0).contiguous().cuda()
    nw = 9 * cin * cout
    pf = torch.zeros(3 * nw, dtype=torch.bfloat16, device="cuda")
    pd = torch.zeros(3 * nw, dtype=torch.bfloat16, device="cuda")
    aw, aw2 = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    L.call("punet_split_weights_f16", Wp.data_ptr(), pf.data_ptr(), nw, cin, cout, 0, aw.data_ptr(), S)
    L.call("punet_split_weights_f16", Wp.data_ptr(), pd.data_ptr(), nw, cin, cout, 1, aw2.data_ptr(), S)
    X = x.detach().permute(0, 2, 3, 1).contiguous().cuda()
    DZ = dz.permute(0, 2, 3, 1).contiguous().cuda()
    ax, adz = _amax_slot(L, X, S), _amax_slot(L, DZ, S)
    Y = torch.empty(B, H, W, cout, device="cuda")
    DX = torch.zeros(B, H, W, cin, device="cuda")
    ay, adx = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    dw = torch.zeros(3, 3, cin, cout, device="cuda")
    db = torch.zeros(cout, device="cuda")
    ns = L.call("punet_wgrad_x6_slices", cin, cout, B, H, W)
    ws = torch.empty(L.call("punet_wgrad_workspace_floats", cin, cout, 3, ns), device="cuda")
    L.call("punet_conv3x3_f16", X.data_ptr(), cin, None, 0, pf.data_ptr(), nw, cout, 0, b.detach().cuda().data_ptr(), 0, Y.data_ptr(),
           cout, None, None, None, None, 0, 0, B, H, W, planes, ax.data_ptr(), None, aw.data_ptr(), ay.data_ptr(), S)
    L.call("punet_conv3x3_f16", DZ.data_ptr(), cout, None, 0, pd.data_ptr(), nw, cin, 0, None, 0, DX.data_ptr(), cin, None, None, None,
           None, 0, 0, B, H, W, planes, adz.data_ptr(), None, aw2.data_ptr(), adx.data_ptr(), S)
    L.call("punet_conv_wgrad_f16", X.data_ptr(), cin, DZ.data_ptr(), cout, dw.data_ptr(), cin, 0, db.data_ptr(), ws.data_ptr(), ns,
           B, H, W, 0, 1.0, planes, 0, ax.data_ptr(), adz.data_ptr(), S)
    torch.cuda.synchronize()
    assert float(aw) == float(w.detach().abs().max()) and float(ax) == float(x.detach().abs().max())
    # the published output amax is exact: the epilogue's own values
    assert float(ay) == float(Y.abs().max()) and float(adx) == float(DX.abs().max())
    errs = {"fwd": rel(Y.permute(0, 3, 1, 2).cpu().double(), y.detach()), "dgrad": rel(DX.permute(0, 3, 1, 2).cpu().double(), x.grad),
            "wgrad": rel(dw.permute(3, 2, 0, 1).cpu().double(), w.grad)}
    hi, lo = {2: (2e-6, 0.0), 1: (4e-3, 2e-5)}[planes]  # f16x3: fp32-class (x3 / bf16 planes: 1e-5); f16: 8x finer than bf16's 3e-3..
    print(f"\nplanes {planes} magnitudes ({sx:g}, {sw:g}, {sdz:g}) cin {cin}: {errs}")
    for k, e in errs.items():
        assert lo <= e < hi, (planes, errs)
    assert rel(db.cpu().double(), b.grad) < 5e-6  # the bias gradient is an exact fp32 column sum in every mode


def test_two_sources_share_one_scale_and_accumulate_keeps_a_bound(L):
    """Concat convolution (unet_blocks.py:56) with sources of very different magnitude: one common scale from the larger amax;
    an accumulating input gradient leaves a slot that bounds the final tensor."""
    S = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(5)
    B, H, W, c0, c1, cout = 1, 16, 32, 32, 64, 64
    x0 = 300.0 * torch.randn(B, c0, H, W, generator=g)
    x1 = 1e-3 * torch.randn(B, c1, H, W, generator=g)
    w = torch.randn(cout, c0 + c1, 3, 3, generator=g) / 30.0
    ref = F.conv2d(torch.cat([x0, x1], 1).double(), w.double(), None, padding=1)
    Wp = w.permute(2, 3, 1, 0).contiguous().cuda()
    nw = 9 * (c0 + c1) * cout
    pf = torch.zeros(3 * nw, dtype=torch.bfloat16, device="cuda")
    aw = torch.zeros(1, device="cuda")
    L.call("punet_split_weights_f16", Wp.data_ptr(), pf.data_ptr(), nw, c0 + c1, cout, 0, aw.data_ptr(), S)
    X0, X1 = x0.permute(0, 2, 3, 1).contiguous().cuda(), x1.permute(0, 2, 3, 1).contiguous().cuda()
    a0, a1 = _amax_slot(L, X0, S), _amax_slot(L, X1, S)
    Y = torch.empty(B, H, W, cout, device="cuda")
    ay = torch.zeros(1, device="cuda")
    L.call("punet_conv3x3_f16", X0.data_ptr(), c0, X1.data_ptr(), c1, pf.data_ptr(), nw, cout, 0, None, 0, Y.data_ptr(), cout, None,
           None, None, None, 0, 0, B, H, W, 2, a0.data_ptr(), a1.data_ptr(), aw.data_ptr(), ay.data_ptr(), S)
    torch.cuda.synchronize()
    assert rel(Y.permute(0, 3, 1, 2).cpu().double(), ref) < 2e-6
    # accumulate: second launch adds onto Y; the slot then holds max(first, total) >= the final amax
    L.call("punet_conv3x3_f16", X0.data_ptr(), c0, X1.data_ptr(), c1, pf.data_ptr(), nw, cout, 0, None, 0, Y.data_ptr(), cout, None,
           None, None, None, 1, 0, B, H, W, 2, a0.data_ptr(), a1.data_ptr(), aw.data_ptr(), ay.data_ptr(), S)
    torch.cuda.synchronize()
    assert rel(Y.permute(0, 3, 1, 2).cpu().double(), 2 * ref) < 2e-6 and float(ay) >= float(Y.abs().max())
    b = torch.zeros(1, device="cuda")
    L.call("punet_amax_bound", ay.data_ptr(), 0.25, a1.data_ptr(), 1.0, b.data_ptr(), S)
    torch.cuda.synchronize()
    assert abs(float(b) - (0.25 * float(ay) + float(a1))) <= 1e-6 * float(b)


def _fixed_small(golden_dir, dev):
    from probabilistic_domain_adaptation_amd.engine import ParamLayout
    fx = np.load(os.path.join(golden_dir, "fixed_small.npz"))
    meta = fx["meta"]
    B, C, H, W, s0, s1, s2, fixed = [int(v) for v in meta[:8]]
    nf = [int(v) for v in meta[8:]]
    cfg = O.PUNetConfig(C, 1, nf, 6, 3, 1.0, consensus_masking=True, rl_swap=False)
    Pnp = O.synth_params(cfg.spec(), s1)
    lay = ParamLayout(C, 1, nf, 6, 3)
    flat = torch.zeros(lay.numel, device=dev)
    for k, v in lay.views(flat).items():
        v.copy_(torch.from_numpy(Pnp[k]))
    return fx, cfg, Pnp, lay, flat, O.synth_batch(B, C, H, W, s0), (B, s2)


def test_model_logits_against_the_reference(golden_dir, L):
    """VERDICT r2 #1 'done': f16x3 logits error <= 1e-4 and f16 <= the bf16 figure / 8 against the REFERENCE's logits
    (fixture fixed_small)."""
    from probabilistic_domain_adaptation_amd.engine import PUNetEngine
    dev = torch.device("cuda:0")
    fx, cfg, Pnp, lay, flat, (x, x1, x2, y), (B, s2) = _fixed_small(golden_dir, dev)
    out = {}
    for m in ("x6", "x3", "bf16", "f16x3", "f16"):
        eng = PUNetEngine(lay, dev, conv_mode=m)
        st = eng.forward_features(flat, eng.to_nhwc(x1))
        nmax = fx["logits"].shape[0]
        eps = torch.randn(nmax, B, 6, generator=torch.Generator().manual_seed(s2)).to(dev)
        z = eng.reparam(st["prior_muls"], eps)
        _, _, logits = eng.sample(flat, st["features"], z, want_logits=True)
        torch.cuda.synchronize()
        out[m] = float(np.abs(logits.cpu().numpy() - fx["logits"]).max())
    print("\nlogits max abs err vs the reference: " + ", ".join(f"{k} {v:.3e}" for k, v in out.items()))
    assert out["x6"] < 1e-3
    assert out["f16x3"] <= 1e-4, out
    assert 1e-5 < out["f16"] <= out["bf16"] / 8.0, out


@pytest.mark.parametrize("mode", ["f16x3", "f16"])
def test_training_step_gradients_against_the_oracle(golden_dir, mode):
    """One ELBO forward / backward in a fast mode: loss scalars and all gradients against the CPU oracle's autograd (ReLUs gated
    by the GPU's forward, as in the parity tests).  f16x3 meets the PARITY tests' gradient bar (5e-4 of each tensor's max); f16
    stays within the 11-bit arithmetic's band.  The gradients of the deep encoder layers are O(1e-6) of the head's - far below
    fp16's normal range without the per-tensor scaling."""
    from probabilistic_domain_adaptation_amd.engine import PUNetEngine
    dev = torch.device("cuda:0")
    fx, cfg, Pnp, lay, flat, (x, x1, x2, y), (B, s2) = _fixed_small(golden_dir, dev)
    H, W = x.shape[2], x.shape[3]
    eng = PUNetEngine(lay, dev, conv_mode=mode)
    eps = torch.randn(B, 6, generator=torch.Generator().manual_seed(9))
    consm = (torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(3)) > 0.3).float()
    taps = {}
    st = eng.elbo_forward(flat, eng.to_nhwc(x2.to(dev)), y.to(dev), consm.to(dev), eps.to(dev), "bce", 1.0, 1.0, True, taps)
    g = torch.full((lay.numel,), float("nan"), device=dev)
    eng.elbo_backward(flat, g, st, False)
    torch.cuda.synchronize()
    gates = {k[5:]: (v.t > 0).permute(0, 3, 1, 2).cpu() for k, v in taps.items() if k.startswith("conv:")}
    P = O.to_torch(Pnp, requires_grad=True)
    O.RELU_GATES = gates
    try:
        lo, aux = O.training_loss(P, cfg, x2, y, consm, eps, reg_weight=0.0)
        go = O.grads_of(lo, P)
    finally:
        O.RELU_GATES = None
    # how many imposed gates differ from the oracle's own ReLU (bounded by what the arithmetic's own error explains)
    print(O.relu_gate_report(*{"f16x3": (3e-4, 1e-4), "f16": (3e-2, 2e-2)}[mode]))
    tol_l, tol_g = {"f16x3": (2e-4, 5e-4), "f16": (2e-2, 5e-2)}[mode]
    np.testing.assert_allclose([st["rec"].item(), st["kl"].item()], [aux["reconstruction_loss"].item(), aux["kl"].item()], rtol=tol_l)
    gv = lay.views(g)
    worst = 0.0
    for k, ref in go.items():
        assert torch.isfinite(gv[k]).all(), k
        e = (gv[k].cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-8)
        worst = max(worst, e)
        assert e < tol_g, (mode, k, e)
    print(f"\n{mode}: worst gradient error {worst:.2e} of the tensor max over {len(go)} tensors")


@pytest.mark.parametrize("mode", ["f16x3", "f16"])
def test_mean_teacher_step_runs_is_reproducible_and_close_to_x6(mode):
    """A fused Mean-Teacher step in a fast mode: finite, bit-reproducible run to run (the amax slots use an order-independent
    atomic max), the loss within the mode's band of the parity arithmetic's."""
    import copy
    from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    dev = torch.device("cuda:0")
    nf = [64, 64, 96, 128]
    x, x1, x2, y = [t.to(dev) for t in O.synth_batch(2, 1, 64, 96, 0)]
    eps = torch.randn(4, 2, 6, generator=torch.Generator().manual_seed(5)).to(dev)
    res = {}
    for m_ in ("x6", mode, mode):
        m = ProbabilisticUnet(1, 1, nf, 6, 3, 1.0, consensus_masking=True, device=dev, seed=1, conv_mode=m_)
        t = copy.deepcopy(m)
        assert t.engine.conv_mode == m_
        opt = FusedAdam(m, lr=1e-4)
        loss, ys, zs, _ = FU.fused_self_training_step(m, t, opt, x1, x2, 4, False, eps_pl=eps, eps_s=eps[:1])
        torch.cuda.synchronize()
        res.setdefault(m_, []).append((float(loss), m.flat_params.clone(), t.flat_params.clone(), ys.clone()))
    (l0, p0, t0, y0), (l1, p1, t1, y1) = res[mode]
    assert np.isfinite(l0) and l0 == l1 and torch.equal(p0, p1) and torch.equal(t0, t1) and torch.equal(y0, y1)
    lx = res["x6"][0][0]
    assert abs(l0 - lx) <= {"f16x3": 1e-5, "f16": 5e-3}[mode] * abs(lx), (l0, lx)
    assert (y0 - res["x6"][0][3]).abs().max().item() <= {"f16x3": 1e-5, "f16": 2e-2}[mode]


def test_mixed_precision_flag_selects_the_documented_mode(monkeypatch):
    """`mixed_precision=True` (what every reference script passes, LIVECell/livecell_mt.py:107) is no longer ignored: it switches
    the student - and the teacher copied from it - to $PUNET_MIXED_PRECISION_MODE (default "f16x3"); without the argument a
    trainer runs the parity arithmetic.  One Mean-Teacher iteration in that mode lands within 1e-5 of the parity iteration."""
    import contextlib
    from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam, MeanTeacherTrainer
    dev = torch.device("cuda:0")
    batch = O.synth_batch(2, 1, 32, 32, 60)
    eps = {"teacher": torch.randn(4, 2, 6, generator=torch.Generator().manual_seed(1)),
           "student": torch.randn(1, 2, 6, generator=torch.Generator().manual_seed(2))}

    class _P:
        def update(self, n):
            pass

    def run(**kw):
        m = ProbabilisticUnet(1, 1, [32, 64, 64, 96], 6, 3, 1.0, consensus_masking=True, rl_swap=False, device=dev, seed=3)
        tr = MeanTeacherTrainer(name=None, train_loader=[batch], val_loader=None, model=m, optimizer=FusedAdam(m, lr=1e-3), device=dev,
                                do_consensus_masking=False, **kw)
        tr.n_samples = 4
        tr.eps_source = lambda kind, n, b: eps[kind]
        tr._train_epoch_impl(_P(), contextlib.nullcontext, None)
        torch.cuda.synchronize()
        return tr

    monkeypatch.delenv("PUNET_MIXED_PRECISION_MODE", raising=False)
    monkeypatch.delenv("PUNET_CONV_MODE", raising=False)
    a = run()
    assert a.model.engine.conv_mode == "x6" and a.teacher.engine.conv_mode == "x6" and a.mixed_precision is False
    b = run(mixed_precision=True)
    assert b.model.engine.conv_mode == "f16x3" and b.teacher.engine.conv_mode == "f16x3"
    assert abs(float(b.last_loss) - float(a.last_loss)) <= 1e-5 * abs(float(a.last_loss))
    assert not torch.equal(a.model.flat_params, b.model.flat_params)  # (it really was a different arithmetic)
    # the switch is made on the caller's model in place - restore_precision() undoes it for student and teacher ...
    b.restore_precision()
    assert b.model.engine.conv_mode == "x6" and b.teacher.engine.conv_mode == "x6"
    # ... fit() applies it on entry (also to a teacher assigned since) and restores on exit, with graph replay across the switches
    c = run(mixed_precision=True, graph=True)
    c.restore_precision()
    seen = []
    step = c._train_epoch_impl
    c._train_epoch_impl = lambda *a, **k: (seen.append((c.model.engine.conv_mode, c.teacher.engine.conv_mode)), step(*a, **k))[1]
    c.fit(iterations=2)
    torch.cuda.synchronize()
    assert seen and all(s == ("f16x3", "f16x3") for s in seen)
    assert c.model.engine.conv_mode == "x6" and c.teacher.engine.conv_mode == "x6"
    assert torch.isfinite(c.model.flat_params).all() and len(c._graphs) == 1  # (graphs of a replaced engine are dropped, not replayed)
    # a student and a teacher in different arithmetics without mixed_precision: reported, not silently mixed
    d = run()
    d.teacher.set_conv_mode("f16")
    with pytest.warns(UserWarning, match="different conv arithmetics"):
        d._apply_mixed_precision()
    monkeypatch.setenv("PUNET_MIXED_PRECISION_MODE", "f16")
    assert run(mixed_precision=True).model.engine.conv_mode == "f16"
    monkeypatch.setenv("PUNET_MIXED_PRECISION_MODE", "off")
    assert run(mixed_precision=True).model.engine.conv_mode == "x6"


def test_f16_storage_mode_runs_on_fp16_tensors_and_stays_close_to_f16_arithmetic():
    """"f16s": every activation of the forward passes IS an fp16 tensor (unscaled, as under the reference's autocast), the
    logits stay within the one-plane arithmetic's own error of the fp32-storage "f16" mode, the student pass's gradients (fp32
    in memory, scaled per tensor in the loaders) within a few per cent of "f16"'s - fused Fcomb kernel and per-op path (Dice)."""
    import copy
    from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    dev = torch.device("cuda:0")
    nf = [64, 64, 96, 128]
    x, x1, x2, y = [t.to(dev) for t in O.synth_batch(2, 1, 64, 96, 0)]
    eps = torch.randn(4, 2, 6, generator=torch.Generator().manual_seed(5)).to(dev)
    outs = {}
    for mode in ("x6", "f16", "f16s"):
        m = ProbabilisticUnet(1, 1, nf, 6, 3, 1.0, consensus_masking=True, device=dev, seed=1, conv_mode=mode)
        taps = {}
        with torch.no_grad():
            st = m.engine.forward_features(m.flat_params, m.engine.to_nhwc(x1), taps)
            z = m.engine.reparam(st["prior_muls"], eps)
            ym, cz, lg = m.engine.sample(m.flat_params, st["features"], z, 0.9, 0.1, False, want_logits=True)
        torch.cuda.synchronize()
        dt = {v.t.dtype for k, v in taps.items()}
        assert dt == ({torch.float16} if mode == "f16s" else {torch.float32}), (mode, dt)
        outs[mode] = lg.float().cpu()
    e_f16 = (outs["f16"] - outs["x6"]).abs().max().item()
    e_s = (outs["f16s"] - outs["x6"]).abs().max().item()
    print(f"\nlogits max abs err vs x6: f16 {e_f16:.3e}, f16s {e_s:.3e}")
    assert 1e-6 < e_s < 3.0 * e_f16 + 1e-3 and not torch.equal(outs["f16s"], outs["f16"])
    grads = {}
    for mode in ("f16", "f16s"):
        for loss_kind in ("bce", "dice"):
            m = ProbabilisticUnet(1, 1, nf, 6, 3, 1.0, consensus_masking=True, device=dev, seed=1, conv_mode=mode)
            eng = m.engine
            consm = (torch.rand(2, 1, 64, 96, generator=torch.Generator().manual_seed(3)) > 0.3).float().to(dev)
            st = eng.elbo_forward(m.flat_params, eng.to_nhwc(x2), y, consm, eps[0], loss_kind, 1.0, 1.0, True)
            if mode == "f16s":
                assert st["feat"].t.dtype == torch.float16 and st["enc_q"].t.dtype == torch.float16
                assert (st.get("fused_fcomb") is not None) == (loss_kind == "bce")
            g = torch.zeros_like(m.flat_params)
            eng.elbo_backward(m.flat_params, g, st, False)
            torch.cuda.synchronize()
            grads[(mode, loss_kind)] = (g.clone(), st["rec"].item())
    for loss_kind in ("bce", "dice"):
        (ga, ra), (gb, rb) = grads[("f16", loss_kind)], grads[("f16s", loss_kind)]
        assert torch.isfinite(gb).all() and abs(ra - rb) <= 5e-3 * abs(ra) + 1e-3, (loss_kind, ra, rb)
        rel_ = (ga - gb).norm().item() / ga.norm().item()
        print(f"{loss_kind}: relative gradient difference f16s vs f16 {rel_:.3e}")
        assert 1e-7 < rel_ < 0.02, (loss_kind, rel_)
    m = ProbabilisticUnet(1, 1, nf, 6, 3, 1.0, consensus_masking=True, device=dev, seed=1, conv_mode="f16s")
    t = copy.deepcopy(m)
    assert t.engine.conv_mode == "f16s"
    opt = FusedAdam(m, lr=1e-4)
    loss, ys, zs, _ = FU.fused_self_training_step(m, t, opt, x1, x2, 4, True, eps_pl=eps, eps_s=eps[:1])
    torch.cuda.synchronize()
    assert np.isfinite(float(loss)) and ys.dtype == torch.float32 and 0.0 <= float(ys.min()) and float(ys.max()) <= 1.0


def test_fast_modes_at_512_against_the_oracle():
    """VERDICT r2 #1 'done' at production size: one 512 x 512 patch, nf = [64, 128, 256, 512], N = 8 MC samples - logits of the
    fp16-operand modes against the CPU oracle (which the golden fixtures pin to the reference).  f16x3 stays in the parity
    arithmetic's class (x6 itself sits at ~8e-5 here: fp32 summation order); its consensus mask differs from the oracle's only
    where a sigmoid lies within 1e-5 of a threshold, as for x6; f16 is the 11-bit band."""
    from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
    dev = torch.device("cuda:0")
    NF, n = [64, 128, 256, 512], 8
    cfg = O.PUNetConfig(1, 1, NF, 6, 3, 1.0, consensus_masking=True, rl_swap=False)
    Pnp = O.synth_params(cfg.spec(), 1)
    P = O.to_torch(Pnp)
    x, x1, x2, y = O.synth_batch(1, 1, 512, 512, 0)
    eps = torch.randn(n, 1, 6, generator=torch.Generator().manual_seed(2))
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        st = O.punet_forward(P, cfg, x1, None, False)
        logits = torch.stack([O.punet_sample(P, cfg, st, eps[i]) for i in range(n)])
    probs = torch.sigmoid(logits)
    ref = (((probs >= 0.9) | (probs <= 0.1)).float().sum(0) == n).float()
    safe = ((probs - 0.9).abs().min(0).values > 1e-5) & ((probs - 0.1).abs().min(0).values > 1e-5)
    err = {}
    for mode in ("x6", "f16x3", "f16", "f16s"):
        m = ProbabilisticUnet(1, 1, NF, 6, 3, 1.0, consensus_masking=True, rl_swap=False, device=dev, seed=0, conv_mode=mode)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in Pnp.items()})
        eng = m.engine
        with torch.no_grad():
            out = eng.forward_features(m.flat_params, eng.to_nhwc(x1))
            z = eng.reparam(out["prior_muls"], eps.to(dev))
            ym, cz, lg = eng.sample(m.flat_params, out["features"], z, 0.9, 0.1, True, want_logits=True)
        torch.cuda.synchronize()
        err[mode] = (lg.cpu() - logits).abs().max().item()
        mism = int((cz.cpu() != ref).sum())
        print(f"\n512x512 N={n} {mode}: logits max abs err vs oracle {err[mode]:.3e}; consensus mask differs on {mism} of {ref.numel()} "
              f"pixels ({int((~safe).sum())} within 1e-5 of a threshold)")
        if mode in ("x6", "f16x3"):
            assert torch.equal(cz.cpu()[safe], ref[safe])
        del m
    assert err["x6"] < 1e-3 and err["f16x3"] < 2e-4 and err["f16x3"] < 2.0 * err["x6"] + 2e-5, err
    assert 1e-4 < err["f16"] < 0.2 and 1e-4 < err["f16s"] < 0.3, err


@pytest.mark.parametrize("planes", [2, 1], ids=["f16x3", "f16"])
def test_many_tiles_with_three_channel_blocks(L, planes):
    """More tiles than the persistent grid and an output-channel-block count (3) that does not divide it: a workgroup's successive
    tiles differ in their channel block (round 4: the 16x16x32 kernel contracted the next tile's weight columns in a tile's last
    chunk there - all arithmetics share that kernel)."""
    S = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(13)
    B, H, W, cin, cout = 4, 128, 128, 64, 192
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    ref = F.conv2d(x, w, None, padding=1)
    Wp = w.permute(2, 3, 1, 0).contiguous().cuda()
    nw = 9 * cin * cout
    pf = torch.zeros(3 * nw, dtype=torch.bfloat16, device="cuda")
    aw, ay = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    L.call("punet_split_weights_f16", Wp.data_ptr(), pf.data_ptr(), nw, cin, cout, 0, aw.data_ptr(), S)
    X = x.permute(0, 2, 3, 1).contiguous().cuda()
    ax = _amax_slot(L, X, S)
    Y = torch.empty(B, H, W, cout, device="cuda")
    L.call("punet_conv3x3_f16", X.data_ptr(), cin, None, 0, pf.data_ptr(), nw, cout, 0, None, 0, Y.data_ptr(), cout, None, None, None, None,
           0, 0, B, H, W, planes, ax.data_ptr(), None, aw.data_ptr(), ay.data_ptr(), S)
    torch.cuda.synchronize()
    assert rel(Y.permute(0, 3, 1, 2).cpu(), ref) < (2e-6 if planes == 2 else 4e-3)
