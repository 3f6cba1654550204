"""GPU: the FOLD instantiations of the 3x3 split-plane kernels (round 5, tile WIDTH quantization).  When the image width is not a
whole number of 32-pixel tiles but B * W is (88 or 176 pixels at batch 4: the deep levels of 704 x 704 patches), the tiles are
laid over the virtual row of all B images side by side; a tile may hold the last columns of one image and the first ones of the
next, with two zero columns between them in its LDS halo image.  Per pixel the contraction runs in the same order as in the
unfolded launch, so everything must be BIT-identical to it: outputs, ReLU gate bits, the fused 2x2 average, the gated and
accumulating input gradient, two sources, the fp16-plane arithmetic, and the combination with the stream-K tail."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
NOFOLD = 1 << 24


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from probabilistic_domain_adaptation_amd import _lib
    _lib.load()
    return _lib


def S():
    return torch.cuda.current_stream().cuda_stream


def call(L, x0, x1, planes, nw, n_row, n_off, bias, y, n_out, gate_bits, gate_out, pool, accumulate, relu, B, H, W, extra_opts, scratch,
         f16=0, amax=(None, None, None, None)):
    L.call("punet_conv3x3_splitk", x0.data_ptr(), x0.shape[-1], x1.data_ptr() if x1 is not None else None,
           x1.shape[-1] if x1 is not None else 0, planes.data_ptr(), nw, n_row, n_off, bias.data_ptr() if bias is not None else None, 0,
           y.data_ptr(), n_out, None, gate_bits.data_ptr() if gate_bits is not None else None,
           gate_out.data_ptr() if gate_out is not None else None, pool.data_ptr() if pool is not None else None, accumulate, relu,
           B, H, W, L.x6_opts(6, 16) | extra_opts, f16, *[t.data_ptr() if t is not None else None for t in amax],
           scratch.data_ptr() if scratch is not None else None, 4 * scratch.numel() if scratch is not None else 0, S())


# (B, H, W, c0, c1, cout): W % 32 != 0, W >= 32, fewer tiles over the virtual row than per image; heights ragged against the 8-row
# tiles as well; the last four end the virtual row INSIDE a tile (B * W % 32 != 0)
CASES = [(4, 88, 88, 64, 0, 64), (4, 22, 88, 128, 0, 192), (2, 16, 48, 64, 64, 64), (4, 9, 40, 96, 0, 96), (8, 6, 36, 32, 0, 32),
         (4, 44, 176, 64, 0, 128), (2, 10, 80, 160, 32, 200), (4, 130, 72, 32, 0, 64),
         (3, 12, 88, 64, 0, 64), (2, 20, 88, 96, 0, 128), (3, 7, 50, 64, 32, 96), (5, 10, 34, 32, 0, 64)]


@pytest.mark.parametrize("B,H,W,c0,c1,cout", CASES)
@pytest.mark.parametrize("splitk", [1, 0, 2], ids=["whole", "planned-tail", "forced-tail"])
def test_folded_launch_is_bit_identical_to_the_unfolded_one(L, B, H, W, c0, c1, cout, splitk):
    g = torch.Generator().manual_seed(B * 100 + W)
    cin = c0 + c1
    X0 = torch.randn(B, H, W, c0, generator=g).cuda()
    X1 = torch.randn(B, H, W, c1, generator=g).cuda() if c1 else None
    w = (torch.randn(3, 3, cin, cout, generator=g) / np.sqrt(cin * 9)).cuda()
    b = torch.randn(cout, generator=g).cuda()
    nw = 9 * cin * cout
    pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    pd = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    L.call("punet_split_weights_x6", w.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, S())
    L.call("punet_split_weights_x6", w.data_ptr(), pd.data_ptr(), nw, cin, cout, 3, 1, S())
    scratch = torch.zeros(L.call("punet_conv_splitk_scratch_bytes") // 4, dtype=torch.int32, device="cuda")
    sk = splitk << 16
    even = H % 2 == 0 and W % 2 == 0
    res = {}
    for tag, extra in (("fold", sk), ("per-image", sk | NOFOLD)):
        Y = torch.full((B, H, W, cout), float("nan"), device="cuda")
        gate = torch.zeros(L.call("punet_gate_dwords", B, H, W, cout), dtype=torch.int32, device="cuda")
        pool = torch.full((B, H // 2, W // 2, cout), float("nan"), device="cuda") if even else None
        call(L, X0, X1, pf, nw, cout, 0, b, Y, cout, None, gate, pool, 0, 1, B, H, W, extra, scratch)
        # input gradient w.r.t. the first source, gated by the bits the forward of a layer with c0 outputs would have published,
        # accumulating into an existing tensor
        gx = torch.randint(-2 ** 31, 2 ** 31 - 1, (L.call("punet_gate_dwords", B, H, W, c0),), generator=g, dtype=torch.int64).to(torch.int32).cuda() \
            if tag == "fold" else res["gx"]
        res["gx"] = gx
        DZ = torch.randn(B, H, W, cout, generator=torch.Generator().manual_seed(5)).cuda()
        DX = torch.randn(B, H, W, c0, generator=torch.Generator().manual_seed(6)).cuda()
        call(L, DZ, None, pd, nw, cin, 0, None, DX, c0, gx, None, None, 1, 0, B, H, W, extra, scratch)
        torch.cuda.synchronize()
        res[tag] = (Y, gate, pool, DX)
    for i, name in enumerate(("output", "gate bits", "pooled output", "input gradient")):
        a_, b_ = res["fold"][i], res["per-image"][i]
        if a_ is None:
            continue
        assert torch.isfinite(a_.float()).all() if a_.dtype.is_floating_point else True, name
        assert torch.equal(a_, b_), f"{name}: folded and per-image launches differ"
    assert int(scratch[:1024].abs().sum().item()) == 0


def test_folded_launch_against_torch_and_in_fp16_planes(L):
    """One folded shape against PyTorch's convolution (the unfolded kernel is held to it elsewhere; this guards against both being
    wrong the same way), and the two-plane fp16 arithmetic folded vs per image, amax publication included."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    B, H, W, cin, cout = 4, 24, 88, 64, 96
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.relu(F.conv2d(x, w, b, padding=1))
    X = x.permute(0, 2, 3, 1).contiguous().cuda()
    Wp = w.permute(2, 3, 1, 0).contiguous().cuda()
    nw = 9 * cin * cout
    pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    L.call("punet_split_weights_x6", Wp.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, S())
    Y = torch.empty(B, H, W, cout, device="cuda")
    call(L, X, None, pf, nw, cout, 0, b.cuda(), Y, cout, None, None, None, 0, 1, B, H, W, 1 << 16, None)
    torch.cuda.synchronize()
    got = Y.permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max().item() / ref.abs().max().item() < 2e-6
    # fp16 planes
    ph = torch.zeros(3 * nw, dtype=torch.bfloat16, device="cuda")
    aw = torch.zeros(1, device="cuda")
    L.call("punet_split_weights_f16", Wp.data_ptr(), ph.data_ptr(), nw, cin, cout, 0, aw.data_ptr(), S())
    ax = X.abs().max().reshape(1).contiguous()
    outs = []
    for extra in (1 << 16, (1 << 16) | NOFOLD):
        Yh = torch.empty(B, H, W, cout, device="cuda")
        ay = torch.zeros(1, device="cuda")
        call(L, X, None, ph, nw, cout, 0, b.cuda(), Yh, cout, None, None, None, 0, 1, B, H, W, extra, None, f16=2, amax=(ax, None, aw, ay))
        torch.cuda.synchronize()
        outs.append((Yh, ay))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert (outs[0][0].permute(0, 3, 1, 2).cpu() - ref).abs().max().item() / ref.abs().max().item() < 3e-6


def test_mean_teacher_iteration_with_folded_levels_against_the_oracle(L):
    """End to end: batch 4 of 88 x 176 patches, nf = [64, 64, 128, 128] - the first level is 176 wide (5.5 tiles), the second 88 (2.75):
    both run folded (B * W = 704 / 352), the deeper ones (44, 22 wide) per image.  Pseudo-labels through both graph walkers are
    bit-identical and match the oracle, one Mean-Teacher iteration (forward, backward through folded input gradients with gate
    bits and accumulation, Adam, EMA) matches the oracle's."""
    import copy
    from oracle import punet_oracle as O
    from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    dev = torch.device("cuda:0")
    nf, B, H, W, n = [64, 64, 128, 128], 4, 88, 176, 4
    cfg = O.PUNetConfig(1, 1, nf, 6, 3, 1.0, consensus_masking=True, rl_swap=False)
    Pnp = O.synth_params(cfg.spec(), 21)
    m = ProbabilisticUnet(1, 1, nf, 6, 3, 1.0, consensus_masking=True, rl_swap=False, device=dev, seed=0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in Pnp.items()})
    _, x1, x2, _ = O.synth_batch(B, 1, H, W, 31)
    g = torch.Generator().manual_seed(8)
    e_t, e_s = torch.randn(n, B, 6, generator=g), torch.randn(1, B, 6, generator=g)
    Po = O.to_torch(Pnp)
    with torch.no_grad():
        yo, zo = O.sample_consensus(Po, cfg, x1, e_t, masking=True)
        y_rt, z_rt = m.pseudo_labels(x1.to(dev), n, 0.9, 0.1, True, e_t.to(dev))          # the native runtime
        m.forward(x1.to(dev), None, training=False)
        y_en, z_en = m.sample_consensus(n, 0.9, 0.1, True, e_t.to(dev))                    # the Python engine
    torch.cuda.synchronize()
    assert torch.equal(y_rt, y_en) and torch.equal(z_rt, z_en)
    assert (y_rt.cpu() - yo).abs().max().item() < 1e-4
    with torch.no_grad():  # the consensus mask may differ where a sample's sigmoid lies within 1e-5 of a threshold (fp32 summation order)
        st = O.punet_forward(Po, cfg, x1, None, False)
        probs = torch.sigmoid(torch.stack([O.punet_sample(Po, cfg, st, e_t[i]) for i in range(n)]))
    safe = ((probs - 0.9).abs().min(0).values > 1e-5) & ((probs - 0.1).abs().min(0).values > 1e-5)
    assert torch.equal(z_rt.cpu()[safe], zo[safe])
    t = copy.deepcopy(m)
    opt = FusedAdam(m, lr=1e-3)
    loss, y, z, _ = FU.fused_self_training_step(m, t, opt, x1.to(dev), x2.to(dev), n, True, e_t.to(dev), e_s.to(dev), 0.99, 1e-5)
    torch.cuda.synchronize()
    Ps, Pt = O.to_torch(Pnp, True), O.to_torch(Pnp)
    ref = O.mean_teacher_step(Ps, Pt, cfg, O.AdamState(Ps, lr=1e-3), x1, x2, e_t, e_s[0], True, momentum=0.99)
    np.testing.assert_allclose(float(loss), ref["loss"].item(), rtol=2e-4)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    for k, v in Ps.items():
        assert (sd[k] - v.detach()).abs().max().item() <= 2.5e-3, k  # (Adam's first step: sign(g) * lr; 2 lr where g is ~0)


@pytest.mark.parametrize("B,H,W,cin,cout", [(4, 44, 44, 128, 576), (4, 11, 88, 64, 64), (2, 8, 50, 32, 96)])
def test_pure_1x1_launches_walk_the_flat_pixel_list(L, B, H, W, cin, cout):
    """A 1x1 convolution without gate bits, mask or per-image bias does not see the image geometry: the launcher runs it over the
    flat list of B * H * W pixels as one image of 32-pixel rows (K5x's GEMMs at 704 x 704: 44 / 88 / 176-pixel-wide tensors).  The
    result equals PyTorch's and, bit for bit, the same call on the explicitly flattened tensor."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) / np.sqrt(cin)
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, w, b)
    X = x.permute(0, 2, 3, 1).contiguous().cuda()
    Wp = w.permute(2, 3, 1, 0).contiguous().cuda()
    nw = cin * cout
    pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    L.call("punet_split_weights_x6", Wp.data_ptr(), pf.data_ptr(), nw, cin, cout, 1, 0, S())
    bd = b.cuda()
    Y = torch.full((B, H, W, cout), float("nan"), device="cuda")
    L.call("punet_conv1x1_x6_gated", X.data_ptr(), cin, None, 0, pf.data_ptr(), nw, cout, 0, bd.data_ptr(), 0, Y.data_ptr(), cout, None, None,
           None, 0, 0, B, H, W, S())
    Y2 = torch.full((B, H, W, cout), float("nan"), device="cuda")
    n = B * H * W
    L.call("punet_conv1x1_x6_gated", X.data_ptr(), cin, None, 0, pf.data_ptr(), nw, cout, 0, bd.data_ptr(), 0, Y2.data_ptr(), cout, None, None,
           None, 0, 0, 1, n // 32, 32, S())
    torch.cuda.synchronize()
    assert torch.equal(Y, Y2)
    got = Y.permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max().item() / ref.abs().max().item() < 2e-6


# ---- the weight-gradient kernel's FOLD instantiation: 32-pixel column strips over the virtual row of all images (widths that are
# multiples of 8).  The sums run over the same pixels in another ORDER than the per-image launch: equal to rounding, not bit for bit.
WG_CASES = [(4, 88, 88, 64, 64), (4, 44, 176, 64, 128), (2, 16, 48, 64, 64), (3, 12, 88, 64, 64), (4, 9, 40, 96, 96),
            (2, 10, 80, 160, 32), (4, 130, 72, 32, 64), (5, 7, 56, 32, 64), (2, 3, 104, 8, 16), (7, 5, 40, 24, 40)]


@pytest.mark.parametrize("B,H,W,cin,cout", WG_CASES)
def test_folded_weight_gradient_against_autograd_and_the_unfolded_launch(L, B, H, W, cin, cout):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B * 1000 + W)
    x = (torch.randn(B, cin, H, W, generator=g) * torch.exp2(torch.randint(-6, 6, (B, cin, H, W), generator=g).float())).double()
    dz = (torch.randn(B, cout, H, W, generator=g) * torch.exp2(torch.randint(-6, 6, (B, cout, H, W), generator=g).float())).double()
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w, b, padding=1).backward(dz)
    X = x.float().permute(0, 2, 3, 1).contiguous().cuda()
    DZ = dz.float().permute(0, 2, 3, 1).contiguous().cuda()
    ns = L.call("punet_wgrad_x6_slices", cin, cout, B, H, W)
    ws = torch.empty(L.call("punet_wgrad_workspace_floats", cin, cout, 3, ns), device="cuda")
    out = {}
    for name, opts in (("fold", 0), ("nofold", NOFOLD)):
        dw = torch.full((3, 3, cin, cout), float("nan"), device="cuda")
        db = torch.full((cout,), float("nan"), device="cuda")
        L.call("punet_conv_wgrad_x6", X.data_ptr(), cin, DZ.data_ptr(), cout, dw.data_ptr(), cin, 0, db.data_ptr(), ws.data_ptr(), ns,
               B, H, W, 0, 1.0, opts, S())
        torch.cuda.synchronize()
        out[name] = (dw.permute(3, 2, 0, 1).double().cpu(), db.double().cpu())
    rel = lambda a_, b_: (a_ - b_).abs().max().item() / (b_.abs().max().item() + 1e-300)
    for name in out:
        assert rel(out[name][0], w.grad) < 5e-6, name
        assert rel(out[name][1], b.grad) < 5e-6, name
    assert rel(out["fold"][0], out["nofold"][0]) < 2e-6
    # f16x3 planes (no options word at this entry point: folded whenever the shape is; unit scales, operands inside fp16's range)
    xs, ds = x.float() * 2.0 ** -4, dz.float() * 2.0 ** -4
    xs = torch.where(xs.abs() < 2.0 ** -10, torch.zeros_like(xs), xs)  # (nothing under fp16's two-plane resolution)
    ds = torch.where(ds.abs() < 2.0 ** -10, torch.zeros_like(ds), ds)
    w2 = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xs.double(), w2, None, padding=1).backward(ds.double())
    X2, D2 = xs.permute(0, 2, 3, 1).contiguous().cuda(), ds.permute(0, 2, 3, 1).contiguous().cuda()
    dw = torch.full((3, 3, cin, cout), float("nan"), device="cuda")
    L.call("punet_conv_wgrad_f16", X2.data_ptr(), cin, D2.data_ptr(), cout, dw.data_ptr(), cin, 0, None, ws.data_ptr(), ns, B, H, W, 0, 1.0,
           2, 0, None, None, S())
    torch.cuda.synchronize()
    assert rel(dw.permute(3, 2, 0, 1).double().cpu(), w2.grad) < 2e-5
