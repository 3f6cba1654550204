"""GPU: every HIP kernel of libpunet_hip.so against a plain fp32 PyTorch CPU evaluation of the same op
(the ops the reference dispatches).  Tolerances are fp32 summation-order tolerances, written per test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from probabilistic_domain_adaptation_amd import _lib
    _lib.load()
    assert _lib.call("punet_device_ok") == 1, _lib.last_error()
    return torch.device("cuda:0")


def _lib():
    from probabilistic_domain_adaptation_amd import _lib as L
    return L


def S():
    return torch.cuda.current_stream().cuda_stream


def nhwc(t):  # [B,C,H,W] cpu -> NHWC cuda
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):  # NHWC cuda -> [B,C,H,W] cpu
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def wpack(w):  # [cout,cin,k,k] -> [k,k,cin,cout] cuda
    return w.permute(2, 3, 1, 0).contiguous().cuda()


def wunpack(wp):  # [k,k,cin,cout] cuda -> [cout,cin,k,k] cpu
    return wp.permute(3, 2, 0, 1).contiguous().cpu()


def rel_err(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


CONV_CASES = [
    # B, H, W, c0, c1, cout, k
    (1, 8, 32, 8, 0, 32, 3),
    (2, 16, 16, 64, 0, 64, 3),
    (1, 24, 40, 32, 0, 96, 3),     # W not a multiple of 32, cout % 64 == 32
    (2, 8, 8, 128, 0, 128, 3),     # tile mostly outside the image
    (1, 16, 32, 64, 32, 32, 3),    # two-source (concat) K loop
    (1, 64, 64, 96, 64, 64, 3),
    (2, 16, 32, 64, 0, 64, 1),     # 1x1
    (1, 4, 4, 192, 0, 160, 3),
    (4, 128, 128, 64, 0, 192, 3),  # more tiles than a persistent grid, three output-channel blocks (not a divisor of the grid)
    (3, 96, 160, 96, 64, 160, 3),
]


@pytest.mark.parametrize("B,H,W,c0,c1,cout,k", CONV_CASES)
def test_conv_fwd(dev, B, H, W, c0, c1, cout, k):
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(B, c0, H, W, generator=g)
    x1 = torch.randn(B, c1, H, W, generator=g) if c1 else None
    w = torch.randn(cout, c0 + c1, k, k, generator=g) / np.sqrt((c0 + c1) * k * k)
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(torch.cat([x0, x1], 1) if c1 else x0, w, b, padding=k // 2))
    X0, X1, Wp, Bd = nhwc(x0), (nhwc(x1) if c1 else None), wpack(w), b.cuda()
    y = torch.full((B, H, W, cout), float("nan"), device=dev)
    _lib().call("punet_conv_fwd", X0.data_ptr(), c0, X1.data_ptr() if c1 else None, c1, Wp.data_ptr(), Bd.data_ptr(), 0,
                y.data_ptr(), cout, B, H, W, k, 1, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(y), ref) < 2e-6


def test_conv_fwd_per_image_bias_no_relu(dev):
    B, H, W, c, cout = 3, 8, 32, 64, 64
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, c, H, W, generator=g)
    w = torch.randn(cout, c, 1, 1, generator=g) / 8
    bi = torch.randn(B, cout, generator=g)
    ref = F.conv2d(x, w) + bi[:, :, None, None]
    y = torch.empty(B, H, W, cout, device=dev)
    X, Wp, Bd = nhwc(x), wpack(w), bi.cuda()
    _lib().call("punet_conv_fwd", X.data_ptr(), c, None, 0, Wp.data_ptr(), Bd.data_ptr(), cout, y.data_ptr(), cout, B, H, W,
                1, 0, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(y), ref) < 2e-6


@pytest.mark.parametrize("B,H,W,c0,c1,cout,k", CONV_CASES)
def test_conv_dgrad_wgrad(dev, B, H, W, c0, c1, cout, k):
    """dgrad (both sources, ReLU mask epilogue, accumulate) and wgrad (+bias) against autograd."""
    g = torch.Generator().manual_seed(3)
    cin = c0 + c1
    x = torch.randn(B, cin, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)).requires_grad_(True)
    b = torch.randn(cout, generator=g).requires_grad_(True)
    dz = torch.randn(B, cout, H, W, generator=g)
    y = F.conv2d(x, w, b, padding=k // 2)
    y.backward(dz)
    L = _lib()
    DZ, Wp = nhwc(dz), wpack(w.detach())
    # dgrad per source, with a mask and with accumulate
    off = 0
    for cs in [c0] + ([c1] if c1 else []):
        mask = torch.randn(B, cs, H, W, generator=g)
        prev = torch.randn(B, cs, H, W, generator=g)
        M, DX = nhwc(mask), nhwc(prev)
        L.call("punet_conv_dgrad", DZ.data_ptr(), cout, Wp.data_ptr(), cin, off, cs, DX.data_ptr(), M.data_ptr(), 1, B, H, W, k,
               S())
        torch.cuda.synchronize()
        ref = prev + x.grad[:, off:off + cs] * (mask > 0)
        assert rel_err(nchw(DX), ref) < 3e-6, f"dgrad source at {off}"
        off += cs
    # wgrad
    X = nhwc(x.detach())
    dw = torch.full((k, k, cin, cout), float("nan"), device=dev)
    db = torch.full((cout,), float("nan"), device=dev)
    off = 0
    for si, cs in enumerate([c0] + ([c1] if c1 else [])):
        xs = X[..., off:off + cs].contiguous()
        ns = L.call("punet_wgrad_slices", cs, cout, B, H, W)
        ws = torch.empty(L.call("punet_wgrad_workspace_floats", cs, cout, k, ns), device=dev)
        L.call("punet_conv_wgrad", xs.data_ptr(), cs, DZ.data_ptr(), cout, dw.data_ptr(), cin, off,
               db.data_ptr() if si == 0 else None, ws.data_ptr(), ns, B, H, W, k, 0, 1.0, S())
        off += cs
    torch.cuda.synchronize()
    assert rel_err(wunpack(dw), w.grad) < 5e-6
    assert rel_err(db.cpu(), b.grad) < 5e-6
    # accumulate + scale
    dw2 = dw.clone()
    xs = X[..., :c0].contiguous()
    ns = L.call("punet_wgrad_slices", c0, cout, B, H, W)
    ws = torch.empty(L.call("punet_wgrad_workspace_floats", c0, cout, k, ns), device=dev)
    L.call("punet_conv_wgrad", xs.data_ptr(), c0, DZ.data_ptr(), cout, dw2.data_ptr(), cin, 0, None, ws.data_ptr(), ns, B, H,
           W, k, 1, 0.5, S())
    torch.cuda.synchronize()
    exp = w.grad.clone()
    exp[:, :c0] *= 1.5
    assert rel_err(wunpack(dw2), exp) < 5e-6


@pytest.mark.parametrize("cin,cout,B,H,W", [(1, 64, 2, 16, 24), (2, 64, 1, 32, 32), (3, 32, 2, 8, 40), (1, 32, 1, 8, 8),
                                             # round 5: 2..4 channels run the first layer on the matrix cores (conv3x3_first_mfma_kernel): ragged
                                             # 16-pixel blocks, a single row / column, four channels (two K-steps), wide first layers
                                             (3, 64, 2, 13, 37), (4, 64, 1, 9, 50), (4, 128, 2, 5, 17), (2, 256, 1, 3, 16), (3, 64, 1, 1, 1),
                                             (3, 16, 1, 40, 3), (4, 8, 1, 6, 6)])
def test_conv_small(dev, cin, cout, B, H, W):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, cin, H, W, generator=g)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / 3).requires_grad_(True)
    b = torch.randn(cout, generator=g).requires_grad_(True)
    dz = torch.randn(B, cout, H, W, generator=g)
    pre = F.conv2d(x, w, b, padding=1)
    pre.backward(dz)
    L = _lib()
    X, Wp, Bd, DZ = nhwc(x), wpack(w.detach()), b.detach().cuda(), nhwc(dz)
    y = torch.empty(B, H, W, cout, device=dev)
    L.call("punet_conv3x3_small_fwd", X.data_ptr(), cin, Wp.data_ptr(), Bd.data_ptr(), y.data_ptr(), cout, B, H, W, 1, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(y), F.relu(pre.detach())) < 2e-6
    dw = torch.empty(3, 3, cin, cout, device=dev)
    db = torch.empty(cout, device=dev)
    ws = torch.empty(L.call("punet_small_wgrad_workspace_floats", cin, cout), device=dev)
    L.call("punet_conv3x3_small_wgrad", X.data_ptr(), cin, DZ.data_ptr(), cout, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, H,
           W, 0, 1.0, S())
    torch.cuda.synchronize()
    assert rel_err(wunpack(dw), w.grad) < 5e-6
    assert rel_err(db.cpu(), b.grad) < 5e-6


@pytest.mark.parametrize("B,H,W,C", [(2, 16, 24, 32), (1, 7, 9, 8), (1, 64, 64, 64)])
def test_pool(dev, B, H, W, C):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, H, W, generator=g).requires_grad_(True)
    y = F.avg_pool2d(x, 2, 2, 0, ceil_mode=True)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    L = _lib()
    X = nhwc(x.detach())
    Y = torch.empty(B, y.shape[2], y.shape[3], C, device=dev)
    L.call("punet_pool2x2_fwd", X.data_ptr(), Y.data_ptr(), B, H, W, C, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(Y), y.detach()) < 1e-6
    prev = torch.randn(B, C, H, W, generator=g)
    DX = nhwc(prev)
    DY = nhwc(dy)
    L.call("punet_pool2x2_bwd", DY.data_ptr(), X.data_ptr(), DX.data_ptr(), B, H, W, C, 1, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(DX), prev + x.grad * (x.detach() > 0)) < 1e-6


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 8, 32), (1, 5, 11, 8), (1, 32, 32, 64), (1, 1, 4, 8)])
def test_upsample(dev, B, H, W, C):
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, C, H, W, generator=g).requires_grad_(True)
    y = F.interpolate(x, mode="bilinear", scale_factor=2, align_corners=True)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    L = _lib()
    X = nhwc(x.detach())
    Y = torch.empty(B, 2 * H, 2 * W, C, device=dev)
    L.call("punet_upsample2x_fwd", X.data_ptr(), Y.data_ptr(), B, H, W, C, S())
    torch.cuda.synchronize()
    assert (nchw(Y) - y.detach()).abs().max().item() < 2e-6
    DX = torch.empty(B, H, W, C, device=dev)
    DY = nhwc(dy)
    L.call("punet_upsample2x_bwd", DY.data_ptr(), X.data_ptr(), DX.data_ptr(), B, H, W, C, S())
    torch.cuda.synchronize()
    assert (nchw(DX) - x.grad * (x.detach() > 0)).abs().max().item() < 1e-5


def test_latent_head_and_kl(dev):
    B, C, H, W, Ld = 3, 96, 8, 8, 6
    g = torch.Generator().manual_seed(7)
    enc = F.relu(torch.randn(B, C, H, W, generator=g)).requires_grad_(True)
    w = (torch.randn(2 * Ld, C, 1, 1, generator=g) / 10).requires_grad_(True)
    b = (torch.randn(2 * Ld, generator=g) / 10).requires_grad_(True)
    mp = (torch.randn(B, 2 * Ld, generator=g) / 3).requires_grad_(True)
    eps = torch.randn(B, Ld, generator=g)
    m = enc.mean(2, keepdim=True).mean(3, keepdim=True)
    muls = F.conv2d(m, w, b)[:, :, 0, 0]
    mu_q, sig_q = muls[:, :Ld], muls[:, Ld:].exp()
    mu_p, sig_p = mp[:, :Ld], mp[:, Ld:].exp()
    z = mu_q + sig_q * eps
    q = torch.distributions.Independent(torch.distributions.Normal(mu_q, sig_q), 1)
    p = torch.distributions.Independent(torch.distributions.Normal(mu_p, sig_p), 1)
    kl = torch.distributions.kl.kl_divergence(q, p).mean()
    dz = torch.randn(B, Ld, generator=g)
    beta, gscale = 1.7, 0.5
    (gscale * beta * kl + (z * dz).sum()).backward()
    L = _lib()
    E, Wp, Bd = nhwc(enc.detach()), wpack(w.detach()), b.detach().cuda()
    mean = torch.empty(B, C, device=dev)
    mu = torch.empty(B, 2 * Ld, device=dev)
    ws = torch.empty(L.call("punet_chan_sum_workspace_floats", B, H * W, C), device=dev)
    L.call("punet_latent_head_fwd", E.data_ptr(), Wp.data_ptr(), Bd.data_ptr(), mean.data_ptr(), mu.data_ptr(), ws.data_ptr(), B,
           H * W, C, 2 * Ld, S())
    MP, EPS = mp.detach().cuda(), eps.cuda()
    zz = torch.empty(1, B, Ld, device=dev)
    L.call("punet_reparam", mu.data_ptr(), EPS.data_ptr(), zz.data_ptr(), 1, B, Ld, S())
    klo = torch.empty(1, device=dev)
    L.call("punet_latent_kl", mu.data_ptr(), MP.data_ptr(), klo.data_ptr(), B, Ld, S())
    torch.cuda.synchronize()
    assert (mu.cpu() - muls.detach()).abs().max().item() < 2e-6
    assert (zz[0].cpu() - z.detach()).abs().max().item() < 2e-6
    assert abs(klo.item() - kl.item()) < 1e-5 * abs(kl.item()) + 1e-6
    dq = torch.empty(B, 2 * Ld, device=dev)
    dp = torch.empty(B, 2 * Ld, device=dev)
    DZ = dz.cuda()
    L.call("punet_latent_kl_bwd", mu.data_ptr(), MP.data_ptr(), EPS.data_ptr(), DZ.data_ptr(), gscale * beta / B,
           dq.data_ptr(), dp.data_ptr(), B, Ld, S())
    dw = torch.empty(1, 1, C, 2 * Ld, device=dev)
    db = torch.empty(2 * Ld, device=dev)
    dmean = torch.empty(B, C, device=dev)
    denc = torch.empty(B, H, W, C, device=dev)
    L.call("punet_latent_head_bwd", dq.data_ptr(), mean.data_ptr(), Wp.data_ptr(), E.data_ptr(), dw.data_ptr(), db.data_ptr(),
           dmean.data_ptr(), denc.data_ptr(), B, H * W, C, 2 * Ld, 0, S())
    torch.cuda.synchronize()
    assert rel_err(dp.cpu(), mp.grad) < 1e-5
    assert rel_err(wunpack(dw), w.grad) < 1e-5
    assert rel_err(db.cpu(), b.grad) < 1e-5
    assert rel_err(nchw(denc), enc.grad * (enc.detach() > 0)) < 1e-5


@pytest.mark.parametrize("CH,nmid,N,B,H,W", [(64, 1, 8, 2, 16, 16), (32, 1, 3, 1, 8, 8), (64, 2, 4, 1, 8, 16), (32, 0, 2, 2, 8, 8)])
def test_fcomb_sampler(dev, CH, nmid, N, B, H, W):
    """K7 against the reference's Fcomb algebra: tile z, cat, 1x1 convs (probabilistic_unet.py:200-214)."""
    Ld = 6
    g = torch.Generator().manual_seed(8)
    feat = F.relu(torch.randn(B, CH, H, W, generator=g))
    w1 = torch.randn(CH, CH + Ld, 1, 1, generator=g) / 8
    b1 = torch.randn(CH, generator=g) / 10
    wm = [torch.randn(CH, CH, 1, 1, generator=g) / 6 for _ in range(nmid)]
    bm = [torch.randn(CH, generator=g) / 10 for _ in range(nmid)]
    w3 = torch.randn(1, CH, 1, 1, generator=g) * 0.6
    b3 = torch.randn(1, generator=g)
    z = torch.randn(N, B, Ld, generator=g)
    logits = []
    for n in range(N):
        zt = z[n][:, :, None, None].expand(B, Ld, H, W)
        h = F.relu(F.conv2d(torch.cat([feat, zt], 1), w1, b1))
        for wl, bl in zip(wm, bm):
            h = F.relu(F.conv2d(h, wl, bl))
        logits.append(F.conv2d(h, w3, b3))
    logits = torch.stack(logits)
    s = torch.sigmoid(logits)
    yref = s.sum(0) / N
    conf = ((s >= 0.9) + (s <= 0.1)).float().sum(0) / N
    L = _lib()
    # flat weight block laid out like the engine's parameter buffer: [w_mid0, b_mid0, w_mid1, b_mid1, ...]
    mid = torch.cat([torch.cat([wpack(wl).reshape(-1), bl.cuda()]) for wl, bl in zip(wm, bm)]) if nmid else None
    stride = CH * CH + CH
    Fd, W1, B1, W3, B3, Z = nhwc(feat), wpack(w1), b1.cuda(), wpack(w3), b3.cuda(), z.cuda()
    zb = torch.empty(N, B, CH, device=dev)
    L.call("punet_fcomb_zbias", Z.data_ptr(), W1.data_ptr(), B1.data_ptr(), zb.data_ptr(), N * B, Ld, CH, S())
    for masking in (0, 1):
        y = torch.empty(B, 1, H, W, device=dev)
        c = torch.empty(B, 1, H, W, device=dev)
        lg = torch.empty(N, B, 1, H, W, device=dev)
        L.call("punet_fcomb_sample", Fd.data_ptr(), W1.data_ptr(), zb.data_ptr(), mid.data_ptr() if nmid else None, stride,
               (mid.data_ptr() + 4 * CH * CH) if nmid else None, stride, W3.data_ptr(), B3.data_ptr(), y.data_ptr(),
               c.data_ptr(), lg.data_ptr(), N, B, H * W, CH, nmid, 0.9, 0.1, masking, S())
        torch.cuda.synchronize()
        assert (lg.cpu() - logits).abs().max().item() < 2e-5 * (logits.abs().max().item() + 1)
        assert (y.cpu() - yref).abs().max().item() < 1e-5
        cref = (conf == 1).float() if masking else conf
        # a flip can only happen where a sigmoid is within rounding of a threshold
        mism = (c.cpu() != cref)
        near = (((s - 0.9).abs() < 1e-5) | ((s - 0.1).abs() < 1e-5)).any(0)
        assert not (mism & ~near).any()
    # K8 on the reference logits is bit-exact w.r.t. the same sigmoid rule
    y8 = torch.empty(B, 1, H, W, device=dev)
    c8 = torch.empty(B, 1, H, W, device=dev)
    LG = logits.cuda()
    L.call("punet_consensus", LG.data_ptr(), N, B * H * W, 0.9, 0.1, 0, y8.data_ptr(), c8.data_ptr(), S())
    torch.cuda.synchronize()
    mism = (c8.cpu() != conf)
    near = (((s - 0.9).abs() < 1e-6) | ((s - 0.1).abs() < 1e-6)).any(0)
    assert not (mism & ~near).any()
    assert (y8.cpu() - yref).abs().max().item() < 1e-6


def test_dot_and_loss(dev):
    B, H, W, CH = 2, 16, 16, 64
    g = torch.Generator().manual_seed(9)
    h = F.relu(torch.randn(B, CH, H, W, generator=g)).requires_grad_(True)
    w3 = (torch.randn(1, CH, 1, 1, generator=g) / 4).requires_grad_(True)
    b3 = torch.randn(1, generator=g).requires_grad_(True)
    tgt = torch.rand(B, 1, H, W, generator=g)
    cm = (torch.rand(B, 1, H, W, generator=g) > 0.4).float()
    logit = F.conv2d(h, w3, b3)
    loss = F.binary_cross_entropy_with_logits(logit * cm, tgt * cm, reduction="none").sum()
    gscale = 0.5
    (gscale * loss).backward()
    L = _lib()
    Hd, W3, B3 = nhwc(h.detach()), wpack(w3.detach()), b3.detach().cuda()
    P = B * H * W
    lg = torch.empty(B, 1, H, W, device=dev)
    L.call("punet_dot_bias", Hd.data_ptr(), W3.data_ptr(), B3.data_ptr(), lg.data_ptr(), P, CH, S())
    dl = torch.empty(B, 1, H, W, device=dev)
    ls = torch.empty(1, device=dev)
    ws = torch.empty(L.call("punet_loss_workspace_floats"), device=dev)
    TG, CM = tgt.cuda(), cm.cuda()  # keep the device copies alive across the asynchronous launches
    L.call("punet_bce_fwd_bwd", lg.data_ptr(), TG.data_ptr(), CM.data_ptr(), gscale, dl.data_ptr(), ls.data_ptr(),
           ws.data_ptr(), P, S())
    dz = torch.empty(B, H, W, CH, device=dev)
    dw3 = torch.empty(CH, device=dev)
    db3 = torch.empty(1, device=dev)
    ws2 = torch.empty(L.call("punet_dot_bwd_workspace_floats", CH), device=dev)
    L.call("punet_dot_bwd", Hd.data_ptr(), dl.data_ptr(), W3.data_ptr(), dz.data_ptr(), dw3.data_ptr(), db3.data_ptr(),
           ws2.data_ptr(), P, CH, 0, S())
    torch.cuda.synchronize()
    assert (lg.cpu() - logit.detach()).abs().max().item() < 1e-5
    assert abs(ls.item() - loss.item()) < 1e-5 * loss.item()
    assert rel_err(nchw(dz), h.grad * (h.detach() > 0)) < 1e-5
    assert rel_err(dw3.cpu(), w3.grad.reshape(-1)) < 1e-5
    assert rel_err(db3.cpu(), b3.grad) < 1e-5
    # no mask
    L.call("punet_bce_fwd_bwd", lg.data_ptr(), TG.data_ptr(), None, 1.0, None, ls.data_ptr(), ws.data_ptr(), P, S())
    torch.cuda.synchronize()
    ref = F.binary_cross_entropy_with_logits(logit.detach(), tgt, reduction="sum").item()
    assert abs(ls.item() - ref) < 1e-5 * ref


def test_adam_ema_and_l2(dev):
    from probabilistic_domain_adaptation_amd.engine import ParamLayout, PUNetEngine
    lay = ParamLayout(1, 1, [32, 32, 32, 32], 6, 3)
    eng = PUNetEngine(lay, dev)
    g = torch.Generator().manual_seed(10)
    n = lay.numel
    p0 = torch.randn(n, generator=g) * 0.1
    grads = [torch.randn(n, generator=g) * 0.01 for _ in range(3)]
    teacher0 = torch.randn(n, generator=g) * 0.1
    # torch reference
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3)
    tr = teacher0.clone()
    P, M, V, T = p0.cuda(), torch.zeros(n, device=dev), torch.zeros(n, device=dev), teacher0.cuda()
    for step, gr in enumerate(grads, 1):
        pr.grad = gr.clone() * 0.5
        opt.step()
        tr = tr * 0.99 + pr.detach() * (1.0 - 0.99)
        eng.adam_ema(P, gr.cuda(), M, V, T, step, 1e-3, gscale=0.5, ema_momentum=0.99)
    torch.cuda.synchronize()
    assert (P.cpu() - pr.detach()).abs().max().item() < 2e-6
    assert (T.cpu() - tr).abs().max().item() < 2e-6
    # l2 reg + gradient
    views = lay.views(p0.clone().requires_grad_(True))
    names = [k for k in views if k.startswith(("posterior.", "prior.", "fcomb.layers."))]
    flat = p0.clone().requires_grad_(True)
    vs = lay.views(flat)
    reg = sum(vs[k].norm(2) for k in names)
    reg.backward()
    Pd = p0.cuda()
    r, norms = eng.l2_reg(Pd)
    G = torch.zeros(n, device=dev)
    eng.l2_reg_add_grad(Pd, G, norms, 1.0)
    torch.cuda.synchronize()
    assert abs(r.item() - reg.item()) < 1e-5 * reg.item()
    assert (G.cpu() - flat.grad).abs().max().item() < 1e-6


def test_dice_score(dev, golden_dir):
    import os
    from probabilistic_domain_adaptation_amd.engine import ParamLayout, PUNetEngine
    fx = np.load(os.path.join(golden_dir, "trainers_small.npz"))
    eng = PUNetEngine(ParamLayout(1, 1, [32, 32, 32, 32], 6, 3), dev)
    a, b = torch.from_numpy(fx["dice_inputs_a"]).cuda(), torch.from_numpy(fx["dice_inputs_b"]).cuda()
    got = [eng.dice_score(a, b), eng.dice_score(a, b, 0.5), eng.dice_score(a, b, 0.5, 0.5)]
    np.testing.assert_allclose(got, fx["dice_values"], rtol=1e-5)
