"""GPU: the opt-in reduced-precision conv modes ("x3": two bf16 planes / three products, "bf16": one plane), the
counterpart of the reference scripts' mixed_precision=True.  They are NOT parity modes: these tests pin their measured
error bands (so a silent precision regression or a mode that does not switch shows up) and that the default stays x6."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import punet_oracle as O

pytestmark = pytest.mark.gpu

# relative-to-max error bands per mode: (must be below, must be above) - the lower bound proves the mode really switched
BANDS = {6: (3e-6, 0.0), 3: (1e-4, 2e-7), 1: (3e-2, 2e-4)}


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from probabilistic_domain_adaptation_amd import _lib
    _lib.load()
    return _lib


def rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


@pytest.mark.parametrize("products", [6, 3, 1])
def test_kernels_error_bands(L, products):
    S = torch.cuda.current_stream().cuda_stream
    OPTS = lambda: L.x6_opts(products)  # the arithmetic travels with each call (no process state in the library)
    g = torch.Generator().manual_seed(3)
    B, H, W, cin, cout = 2, 24, 40, 96, 64
    x = torch.randn(B, cin, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)).requires_grad_(True)
    b = torch.randn(cout, generator=g).requires_grad_(True)
    dz = torch.randn(B, cout, H, W, generator=g)
    y = F.conv2d(x, w, b, padding=1)
    y.backward(dz)
    Wp = w.detach().permute(2, 3, 1, 0).contiguous().cuda()
    nw = 9 * cin * cout
    pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    pd = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    L.call("punet_split_weights_x6", Wp.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, S)
    L.call("punet_split_weights_x6", Wp.data_ptr(), pd.data_ptr(), nw, cin, cout, 3, 1, S)
    X = x.detach().permute(0, 2, 3, 1).contiguous().cuda()
    DZ = dz.permute(0, 2, 3, 1).contiguous().cuda()
    Y = torch.empty(B, H, W, cout, device="cuda")
    DX = torch.zeros(B, H, W, cin, device="cuda")
    dw = torch.zeros(3, 3, cin, cout, device="cuda")
    db = torch.zeros(cout, device="cuda")
    ns = L.call("punet_wgrad_x6_slices", cin, cout, B, H, W)
    ws = torch.empty(L.call("punet_wgrad_workspace_floats", cin, cout, 3, ns), device="cuda")
    L.call("punet_conv3x3_x6", X.data_ptr(), cin, None, 0, pf.data_ptr(), nw, cout, 0, b.detach().cuda().data_ptr(), 0, Y.data_ptr(),
           cout, None, 0, 0, B, H, W, OPTS(), S)
    L.call("punet_conv3x3_x6", DZ.data_ptr(), cout, None, 0, pd.data_ptr(), nw, cin, 0, None, 0, DX.data_ptr(), cin, None, 0, 0,
           B, H, W, OPTS(), S)
    L.call("punet_conv_wgrad_x6", X.data_ptr(), cin, DZ.data_ptr(), cout, dw.data_ptr(), cin, 0, db.data_ptr(), ws.data_ptr(), ns,
           B, H, W, 0, 1.0, OPTS(), S)
    torch.cuda.synchronize()
    hi, lo = BANDS[products]
    errs = {"fwd": rel(Y.permute(0, 3, 1, 2).cpu(), y.detach()), "dgrad": rel(DX.permute(0, 3, 1, 2).cpu(), x.grad),
            "wgrad": rel(dw.permute(3, 2, 0, 1).cpu(), w.grad)}
    for k, e in errs.items():
        assert lo <= e < hi, (products, errs)
    assert rel(db.cpu(), b.grad) < 5e-6  # the bias gradient is an exact fp32 column sum in every mode


def test_options_word_rejects_other_values(L):
    """products must be 6, 3 or 1 and the MFMA shape 16 or 32; anything else is an argument error, not a silent default."""
    S = torch.cuda.current_stream().cuda_stream
    x = torch.zeros(1, 8, 32, 8, device="cuda")
    pf = torch.zeros(3 * 9 * 8 * 8, dtype=torch.bfloat16, device="cuda")
    for bad in (L.x6_opts(2, 16), L.x6_opts(6, 8)):
        with pytest.raises(RuntimeError):
            L.call("punet_conv3x3_x6", x.data_ptr(), 8, None, 0, pf.data_ptr(), 9 * 8 * 8, 8, 0, None, 0, x.clone().data_ptr(), 8, None,
                   0, 0, 1, 8, 32, bad, S)


def test_concurrent_threads_keep_their_own_arithmetic(L):
    """Two host threads issue x6 and x3 convolutions at the same time, each on its own stream: with the options word per
    call (instead of the process-wide setter of round 1) each gets its own arithmetic, identical to a serial run."""
    import threading
    g = torch.Generator().manual_seed(8)
    B, H, W, cin, cout = 1, 32, 64, 64, 64
    x = torch.randn(B, H, W, cin, generator=g).cuda()
    w = (torch.randn(3, 3, cin, cout, generator=g) / 24.0).cuda()
    nw = 9 * cin * cout
    pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    L.call("punet_split_weights_x6", w.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()

    def run(products, out, stream, reps):
        for _ in range(reps):
            L.call("punet_conv3x3_x6", x.data_ptr(), cin, None, 0, pf.data_ptr(), nw, cout, 0, None, 0, out.data_ptr(), cout, None, 0, 0,
                   B, H, W, L.x6_opts(products), stream.cuda_stream)

    serial = {}
    for products in (6, 3):
        serial[products] = torch.empty(B, H, W, cout, device="cuda")
        run(products, serial[products], torch.cuda.current_stream(), 1)
    torch.cuda.synchronize()
    assert not torch.equal(serial[6], serial[3])  # the two arithmetics really differ on this input
    outs = {6: torch.empty_like(serial[6]), 3: torch.empty_like(serial[3])}
    streams = {6: torch.cuda.Stream(), 3: torch.cuda.Stream()}
    th = [threading.Thread(target=run, args=(p_, outs[p_], streams[p_], 200)) for p_ in (6, 3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    assert torch.equal(outs[6], serial[6]) and torch.equal(outs[3], serial[3])


@pytest.mark.parametrize("mode,logit_band", [("x3", (2e-2, 1e-6)), ("bf16", (2.0, 1e-4)), ("bf16s", (2.0, 1e-4))])
def test_model_fast_mode_against_reference_logits(golden_dir, mode, logit_band):
    """Whole inference path in a fast mode against the reference's logits (fixture 'fixed_small'); the measured error is
    printed for DESIGN.md.  x3 stays well inside what fp16 autocast gives the reference on CUDA; bf16 is coarse."""
    from probabilistic_domain_adaptation_amd.engine import ParamLayout, PUNetEngine
    dev = torch.device("cuda:0")
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
    x, x1, x2, y = O.synth_batch(B, C, H, W, s0)
    out = {}
    for m in ("x6", mode):
        eng = PUNetEngine(lay, dev, conv_mode=m)
        st = eng.forward_features(flat, eng.to_nhwc(x1))
        nmax = fx["logits"].shape[0]
        eps = torch.randn(nmax, B, 6, generator=torch.Generator().manual_seed(s2)).to(dev)
        z = eng.reparam(st["prior_muls"], eps)
        _, _, logits = eng.sample(flat, st["features"], z, want_logits=True)
        torch.cuda.synchronize()
        out[m] = float(np.abs(logits.cpu().numpy() - fx["logits"]).max())
    print(f"\nlogits max abs err vs reference: x6 {out['x6']:.3e}, {mode} {out[mode]:.3e}")
    assert out["x6"] < 1e-3
    hi, lo = logit_band
    assert lo < out[mode] < hi, out


def test_fast_mode_training_step_runs_and_default_is_x6(golden_dir):
    from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    dev = torch.device("cuda:0")
    assert ProbabilisticUnet(1, 1, [32, 32, 32, 32], 6, 3, device=dev, seed=0).engine.conv_mode == os.environ.get("PUNET_CONV_MODE", "x6")
    x, x1, x2, y = [t.to(dev) for t in O.synth_batch(2, 1, 64, 64, 0)]
    losses = {}
    for m in ("x6", "x3"):
        model = ProbabilisticUnet(1, 1, [32, 64, 64, 64], 6, 3, 1.0, device=dev, seed=1, conv_mode=m)
        opt = FusedAdam(model, lr=1e-4)
        eps = torch.randn(2, 6, generator=torch.Generator().manual_seed(5)).to(dev)
        loss = FU.fused_supervised_step(model, opt, x, y, eps=eps)
        torch.cuda.synchronize()
        losses[m] = float(loss)
    assert np.isfinite(losses["x3"]) and abs(losses["x3"] - losses["x6"]) < 1e-3 * abs(losses["x6"]), losses
    with pytest.raises(ValueError):
        ProbabilisticUnet(1, 1, [32, 32, 32, 32], 6, 3, device=dev, conv_mode="fp8")


def test_bf16_storage_mode_runs_on_bf16_tensors_and_stays_close_to_bf16_arithmetic(golden_dir):
    """"bf16s": the no-gradient forward path keeps its activations in bf16 (every block output IS a bf16 tensor), the result
    stays within the bf16 arithmetic's own error of the fp32-storage bf16 mode, and a Mean-Teacher step in that mode runs
    (teacher pass on bf16 tensors, student pass on fp32 tensors as in "bf16")."""
    import copy
    from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    dev = torch.device("cuda:0")
    nf = [64, 64, 96, 128]
    x, x1, x2, y = [t.to(dev) for t in O.synth_batch(2, 1, 64, 96, 0)]
    eps = torch.randn(4, 2, 6, generator=torch.Generator().manual_seed(5)).to(dev)
    outs = {}
    for mode in ("x6", "bf16", "bf16s"):
        m = ProbabilisticUnet(1, 1, nf, 6, 3, 1.0, consensus_masking=True, device=dev, seed=1, conv_mode=mode)
        taps = {}
        with torch.no_grad():
            st = m.engine.forward_features(m.flat_params, m.engine.to_nhwc(x1), taps)
            z = m.engine.reparam(st["prior_muls"], eps)
            ym, cz, lg = m.engine.sample(m.flat_params, st["features"], z, 0.9, 0.1, False, want_logits=True)
        torch.cuda.synchronize()
        dt = {v.t.dtype for k, v in taps.items()}
        assert dt == ({torch.bfloat16} if mode == "bf16s" else {torch.float32}), (mode, dt)
        outs[mode] = (lg.float().cpu(), ym.cpu())
    e_bf16 = (outs["bf16"][0] - outs["x6"][0]).abs().max().item()
    e_s = (outs["bf16s"][0] - outs["x6"][0]).abs().max().item()
    print(f"\nlogits max abs err vs x6: bf16 {e_bf16:.3e}, bf16s {e_s:.3e}")
    assert 1e-4 < e_s < 3.0 * e_bf16 + 1e-2
    assert not torch.equal(outs["bf16s"][0], outs["bf16"][0])
    # training: the student pass stores bf16 activations too (fp32 gradients); its gradients must stay within the one-plane
    # arithmetic's error of the fp32-storage bf16 mode - fused Fcomb kernel and per-op Fcomb path (Dice) alike
    grads = {}
    for mode in ("bf16", "bf16s"):
        for loss_kind in ("bce", "dice"):
            m = ProbabilisticUnet(1, 1, nf, 6, 3, 1.0, consensus_masking=True, device=dev, seed=1, conv_mode=mode)
            eng = m.engine
            consm = (torch.rand(2, 1, 64, 96, generator=torch.Generator().manual_seed(3)) > 0.3).float().to(dev)
            st = eng.elbo_forward(m.flat_params, eng.to_nhwc(x2), y, consm, eps[0], loss_kind, 1.0, 1.0, True)
            if mode == "bf16s":
                assert st["feat"].t.dtype == torch.bfloat16 and st["enc_q"].t.dtype == torch.bfloat16
                assert (st.get("fused_fcomb") is not None) == (loss_kind == "bce")
            g = torch.zeros_like(m.flat_params)
            eng.elbo_backward(m.flat_params, g, st, False)
            torch.cuda.synchronize()
            grads[(mode, loss_kind)] = (g.clone(), st["rec"].item())
    for loss_kind in ("bce", "dice"):
        (ga, ra), (gb, rb) = grads[("bf16", loss_kind)], grads[("bf16s", loss_kind)]
        assert torch.isfinite(gb).all() and abs(ra - rb) <= 2e-2 * abs(ra) + 1e-3, (loss_kind, ra, rb)
        rel = (ga - gb).norm().item() / ga.norm().item()
        print(f"{loss_kind}: relative gradient difference bf16s vs bf16 {rel:.3e}")
        assert 1e-6 < rel < 0.1, (loss_kind, rel)
    m = ProbabilisticUnet(1, 1, nf, 6, 3, 1.0, consensus_masking=True, device=dev, seed=1, conv_mode="bf16s")
    t = copy.deepcopy(m)
    assert t.engine.conv_mode == "bf16s"
    opt = FusedAdam(m, lr=1e-4)
    loss, ys, zs, _ = FU.fused_self_training_step(m, t, opt, x1, x2, 4, True, eps_pl=eps, eps_s=eps[:1])
    torch.cuda.synchronize()
    assert np.isfinite(float(loss)) and ys.dtype == torch.float32 and 0.0 <= float(ys.min()) and float(ys.max()) <= 1.0
