"""GPU: the bf16x6 (three-plane split, fp32-accurate) 3x3 convolution kernels against PyTorch CPU fp32 at the SAME
tolerance as the fp32-MFMA kernels (it is a parity mode, not a reduced-precision mode)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import os
import random

CASES = [(1, 8, 32, 8, 0, 32), (2, 16, 16, 64, 0, 64), (1, 24, 40, 32, 0, 96), (2, 8, 8, 128, 0, 128),
         (1, 16, 32, 64, 32, 32), (1, 64, 64, 96, 64, 64), (1, 4, 4, 192, 0, 160),
         # more tiles than the persistent grid (2 x 256 workgroups) AND a channel-block count (3) that does not divide it: a
         # workgroup's successive tiles have different output-channel blocks (the reference's default filters [32, 64, 128, 192])
         (4, 128, 128, 64, 0, 192), (4, 128, 128, 192, 0, 64), (3, 96, 160, 96, 64, 160)]


def _random_cases(count, seed):
    """Seeded draws over the shapes the C ABI accepts: any image size from one pixel up, channel counts that are multiples of 8
    (one or two sources), batch 1-3 (PUNET_FUZZ_X6=count:seed widens them)."""
    r = random.Random(seed)
    return [(r.randint(1, 3), r.randint(1, 50), r.randint(1, 70), 8 * r.randint(1, 20), r.choice([0, 0, 8 * r.randint(1, 12)]),
             8 * r.randint(1, 24)) for _ in range(count)]


CASES += _random_cases(*[int(v) for v in os.environ.get("PUNET_FUZZ_X6", "24:5").split(":")])


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from probabilistic_domain_adaptation_amd import _lib
    _lib.load()
    return _lib


@pytest.fixture(params=[32, 16], ids=["mfma32x32x16", "mfma16x16x32"], autouse=True)
def mfma_shape(request, L):
    """Every test of this module runs with both MFMA shapes of the 3x3 kernels (16 = conv3x3_x6b_kernel, taken by layers
    whose contracted channel count is a multiple of 32); the shape travels in each call's options word (OPTS())."""
    _SHAPE[0] = request.param
    yield request.param
    _SHAPE[0] = 16


_SHAPE = [16]


def OPTS():
    from probabilistic_domain_adaptation_amd import _lib
    return _lib.x6_opts(6, _SHAPE[0])


def S():
    return torch.cuda.current_stream().cuda_stream


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def rel_err(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


@pytest.mark.parametrize("B,H,W,c0,c1,cout", CASES)
def test_conv_x6_fwd_and_dgrad(L, B, H, W, c0, c1, cout):
    g = torch.Generator().manual_seed(11)
    cin = c0 + c1
    # wide dynamic range on purpose: magnitudes spread over ~2^20
    scale = torch.exp2(torch.randint(-10, 10, (B, cin, H, W), generator=g).float())
    x = (torch.randn(B, cin, H, W, generator=g) * scale).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)).requires_grad_(True)
    b = torch.randn(cout, generator=g)
    y = F.conv2d(x, w, b, padding=1)
    dz = torch.randn(B, cout, H, W, generator=g)
    y.backward(dz)
    Wp = w.detach().permute(2, 3, 1, 0).contiguous().cuda()  # [3][3][cin][cout]
    nw = 9 * cin * cout
    pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    pd = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    L.call("punet_split_weights_x6", Wp.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, S())
    L.call("punet_split_weights_x6", Wp.data_ptr(), pd.data_ptr(), nw, cin, cout, 3, 1, S())
    X = nhwc(x.detach())
    X0 = X[..., :c0].contiguous()
    X1 = X[..., c0:].contiguous() if c1 else None
    Y = torch.full((B, H, W, cout), float("nan"), device="cuda")
    Bd = b.cuda()
    L.call("punet_conv3x3_x6", X0.data_ptr(), c0, X1.data_ptr() if c1 else None, c1, pf.data_ptr(), nw, cout, 0, Bd.data_ptr(), 0,
           Y.data_ptr(), cout, None, 0, 1, B, H, W, OPTS(), S())
    torch.cuda.synchronize()
    assert rel_err(nchw(Y), F.relu(y.detach())) < 2e-6
    DZ = nhwc(dz)
    off = 0
    for cs in [c0] + ([c1] if c1 else []):
        mask = torch.randn(B, cs, H, W, generator=g)
        prev = torch.randn(B, cs, H, W, generator=g)
        M, DX = nhwc(mask), nhwc(prev)
        L.call("punet_conv3x3_x6", DZ.data_ptr(), cout, None, 0, pd.data_ptr(), nw, cin, off, None, 0, DX.data_ptr(), cs,
               M.data_ptr(), 1, 0, B, H, W, OPTS(), S())
        torch.cuda.synchronize()
        ref = prev + x.grad[:, off:off + cs] * (mask > 0)
        assert rel_err(nchw(DX), ref) < 3e-6
        off += cs


@pytest.mark.parametrize("mode", [0, 1])
def test_split_weights_batch_matches_per_slab(L, mode):
    """One table-driven launch over a flat buffer == one launch per slab (bit-identical planes)."""
    g = torch.Generator().manual_seed(5)
    shapes = [(8, 32), (32, 32), (64, 24), (40, 128)]  # (cin, cout); a gap between slabs stands in for biases
    offs, off = [], 0
    for cin, cout in shapes:
        offs.append(off)
        off += 9 * cin * cout + cout + 8 - (cout % 8 or 8) + 8
    flat = torch.randn(off, generator=g).cuda()
    one = torch.zeros(3 * off, dtype=torch.bfloat16, device="cuda")
    bat = torch.zeros(3 * off, dtype=torch.bfloat16, device="cuda")
    rows, units = [], 0
    for (cin, cout), o in zip(shapes, offs):
        L.call("punet_split_weights_x6", flat.data_ptr() + 4 * o, one.data_ptr() + 2 * 3 * o, 9 * cin * cout, cin, cout, 3, mode, S())
        rows.append([o, cin, cout, units])
        units += 9 * cin * cout // 8
    rows.append([0, 0, 0, units])
    table = torch.tensor(rows, dtype=torch.int64, device="cuda")
    L.call("punet_split_weights_x6_batch", flat.data_ptr(), bat.data_ptr(), table.data_ptr(), len(shapes), units, mode, S())
    torch.cuda.synchronize()
    assert torch.equal(one.view(torch.int16), bat.view(torch.int16))


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 16, 64, 64, 64), (1, 24, 40, 32, 96), (1, 8, 8, 128, 32), (2, 40, 72, 64, 160)])
def test_gate_bits_equal_fp32_gate(L, B, H, W, cin, cout):
    """The forward publishes (y > 0) as bits; the input gradient gated by those bits is bit-identical to the one gated by
    the fp32 tensor itself (ragged widths, channel counts that are not multiples of 64 included)."""
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, H, W, cin, generator=g).cuda()
    w1 = (torch.randn(3, 3, cin, cout, generator=g) / np.sqrt(9 * cin)).cuda()
    b1 = torch.randn(cout, generator=g).cuda()
    nw = 9 * cin * cout
    pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    pd = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    L.call("punet_split_weights_x6", w1.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, S())
    L.call("punet_split_weights_x6", w1.data_ptr(), pd.data_ptr(), nw, cin, cout, 3, 1, S())
    # y = relu(conv(x)) with its gate bits
    y = torch.empty(B, H, W, cout, device="cuda")
    nb = L.call("punet_gate_dwords", B, H, W, cout)
    bits = torch.full((nb,), -1, dtype=torch.int32, device="cuda")
    L.call("punet_conv3x3_x6_gated", x.data_ptr(), cin, None, 0, pf.data_ptr(), nw, cout, 0, b1.data_ptr(), 0, y.data_ptr(), cout,
           None, None, bits.data_ptr(), 0, 1, B, H, W, OPTS(), S())
    y_plain = torch.empty_like(y)
    L.call("punet_conv3x3_x6", x.data_ptr(), cin, None, 0, pf.data_ptr(), nw, cout, 0, b1.data_ptr(), 0, y_plain.data_ptr(), cout,
           None, 0, 1, B, H, W, OPTS(), S())
    torch.cuda.synchronize()
    assert torch.equal(y, y_plain)
    # decode the documented layout on the host and compare with (y > 0)
    tiles, cbs = (W + 31) // 32, (cout + 63) // 64
    d = bits.cpu().numpy().astype(np.uint32).reshape(B, H, tiles, cbs, 64)
    yy = (y.cpu().numpy() > 0)
    for (bb, hh, xx, cc) in [(0, 0, 0, 0), (B - 1, H - 1, W - 1, cout - 1), (0, H // 2, W // 3, cout // 2), (B - 1, 1, 5, 7)]:
        lane = (xx % 32) + 32 * ((cc % 8) // 4)
        bit = 16 * ((cc % 64) // 32) + 4 * ((cc % 32) // 8) + (cc % 4)
        assert bool((d[bb, hh, xx // 32, cc // 64, lane] >> bit) & 1) == bool(yy[bb, hh, xx, cc])
    # now a second conv's input gradient w.r.t. y, gated either way (here: treat y as the source, its own grad as dz)
    dz = torch.randn(B, H, W, cin, generator=g).cuda()  # gradient of a conv cout -> cin applied to y
    w2 = (torch.randn(3, 3, cout, cin, generator=g) / np.sqrt(9 * cout)).cuda()
    p2 = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    L.call("punet_split_weights_x6", w2.data_ptr(), p2.data_ptr(), nw, cout, cin, 3, 1, S())
    prev = torch.randn(B, H, W, cout, generator=g).cuda()
    for acc in (0, 1):
        d_mask, d_bits = prev.clone(), prev.clone()
        L.call("punet_conv3x3_x6_gated", dz.data_ptr(), cin, None, 0, p2.data_ptr(), nw, cout, 0, None, 0, d_mask.data_ptr(), cout,
               y.data_ptr(), None, None, acc, 0, B, H, W, OPTS(), S())
        L.call("punet_conv3x3_x6_gated", dz.data_ptr(), cin, None, 0, p2.data_ptr(), nw, cout, 0, None, 0, d_bits.data_ptr(), cout,
               None, bits.data_ptr(), None, acc, 0, B, H, W, OPTS(), S())
        torch.cuda.synchronize()
        assert torch.equal(d_mask, d_bits)
    with pytest.raises(RuntimeError):
        L.call("punet_conv3x3_x6_gated", dz.data_ptr(), cin, None, 0, p2.data_ptr(), nw, cout, 0, None, 0, d_bits.data_ptr(), cout,
               y.data_ptr(), bits.data_ptr(), None, 0, 0, B, H, W, OPTS(), S())


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 16, 64, 32, 64), (1, 24, 40, 64, 96)])
def test_pool_and_upsample_backward_take_the_gate_bits(L, B, H, W, cin, cout):
    """K4/K5 backward gated by the bits the forward conv published == gated by the fp32 activation (bit-identical)."""
    g = torch.Generator().manual_seed(23)
    x = torch.randn(B, H, W, cin, generator=g).cuda()
    w1 = (torch.randn(3, 3, cin, cout, generator=g) / np.sqrt(9 * cin)).cuda()
    b1 = torch.randn(cout, generator=g).cuda()
    nw = 9 * cin * cout
    pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    L.call("punet_split_weights_x6", w1.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, S())
    y = torch.empty(B, H, W, cout, device="cuda")
    bits = torch.zeros(L.call("punet_gate_dwords", B, H, W, cout), dtype=torch.int32, device="cuda")
    L.call("punet_conv3x3_x6_gated", x.data_ptr(), cin, None, 0, pf.data_ptr(), nw, cout, 0, b1.data_ptr(), 0, y.data_ptr(), cout,
           None, None, bits.data_ptr(), 0, 1, B, H, W, OPTS(), S())
    dyp = torch.randn(B, (H + 1) // 2, (W + 1) // 2, cout, generator=g).cuda()
    dyu = torch.randn(B, 2 * H, 2 * W, cout, generator=g).cuda()
    prev = torch.randn(B, H, W, cout, generator=g).cuda()
    for acc in (0, 1):
        a_, b_ = prev.clone(), prev.clone()
        L.call("punet_pool2x2_bwd_gated", dyp.data_ptr(), y.data_ptr(), None, a_.data_ptr(), B, H, W, cout, acc, S())
        L.call("punet_pool2x2_bwd_gated", dyp.data_ptr(), None, bits.data_ptr(), b_.data_ptr(), B, H, W, cout, acc, S())
        torch.cuda.synchronize()
        assert torch.equal(a_, b_)
    a_, b_ = torch.empty_like(prev), torch.empty_like(prev)
    L.call("punet_upsample2x_bwd_gated", dyu.data_ptr(), y.data_ptr(), None, a_.data_ptr(), B, H, W, cout, S())
    L.call("punet_upsample2x_bwd_gated", dyu.data_ptr(), None, bits.data_ptr(), b_.data_ptr(), B, H, W, cout, S())
    torch.cuda.synchronize()
    assert torch.equal(a_, b_) and (a_ == 0).float().mean().item() > 0.2


@pytest.mark.parametrize("B,H,W,cin,cout,rows", [(2, 16, 64, 64, 64, 70), (1, 24, 40, 32, 96, 32), (1, 8, 8, 72, 64, 72)])
def test_conv1x1_x6_fwd_and_dgrad(L, B, H, W, cin, cout, rows):
    """The Fcomb 1x1 layers through the split-plane kernel: forward with a per-image bias (the folded z term) and the
    input gradient w.r.t. a channel window of a wider weight (Fcomb's first layer has C + latent_dim input rows)."""
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, cin, H, W, generator=g).requires_grad_(True)
    wfull = (torch.randn(cout, rows, 1, 1, generator=g) / np.sqrt(rows))
    w = wfull[:, :cin].clone().requires_grad_(True)
    bimg = torch.randn(B, cout, generator=g)
    y = F.conv2d(x, w) + bimg[:, :, None, None]
    dz = torch.randn(B, cout, H, W, generator=g)
    F.relu(y).backward(dz)
    Wp = wfull.permute(2, 3, 1, 0).contiguous().cuda()  # [1][1][rows][cout]
    nw = cin * cout
    pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    pd = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    L.call("punet_split_weights_x6", Wp.data_ptr(), pf.data_ptr(), nw, cin, cout, 1, 0, S())
    L.call("punet_split_weights_x6", Wp.data_ptr(), pd.data_ptr(), nw, cin, cout, 1, 1, S())
    X = nhwc(x.detach())
    Y = torch.full((B, H, W, cout), float("nan"), device="cuda")
    bits = torch.zeros(L.call("punet_gate_dwords", B, H, W, cout), dtype=torch.int32, device="cuda")
    Bd = bimg.cuda()
    L.call("punet_conv1x1_x6_gated", X.data_ptr(), cin, None, 0, pf.data_ptr(), nw, cout, 0, Bd.data_ptr(), cout, Y.data_ptr(), cout,
           None, None, bits.data_ptr(), 0, 1, B, H, W, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(Y), F.relu(y.detach())) < 2e-6
    # input gradient of relu(y) w.r.t. x: dz gated by (y > 0) first, then contracted with W^T; gate of x itself: none
    DZ = nhwc(dz * (y.detach() > 0))
    DX = torch.full((B, H, W, cin), float("nan"), device="cuda")
    L.call("punet_conv1x1_x6_gated", DZ.data_ptr(), cout, None, 0, pd.data_ptr(), nw, cin, 0, None, 0, DX.data_ptr(), cin,
           None, None, None, 0, 0, B, H, W, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(DX), x.grad) < 3e-6


@pytest.mark.parametrize("B,H,W,rows,cout,off,cs", [(2, 16, 64, 72, 64, 0, 64), (2, 16, 64, 72, 64, 64, 8), (4, 128, 128, 40, 32, 0, 32),
                                                    (4, 128, 128, 96, 64, 32, 64)])
def test_conv1x1_x6_dgrad_channel_window_and_many_tiles(L, B, H, W, rows, cout, off, cs):
    """the wide 1x1 kernel (conv1x1_x6w_kernel) also serves the Fcomb layers - contractions of 32 / 64 channels, output
    windows (n_off > 0) of a wider weight, more tiles than the persistent grid: the input gradient w.r.t. channels [off, off + cs)
    of a layer with `rows` input channels (Fcomb's first layer: features + latent rows), accumulated into an existing tensor."""
    g = torch.Generator().manual_seed(33)
    x = torch.randn(B, rows, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(cout, rows, 1, 1, generator=g) / np.sqrt(rows)).requires_grad_(True)
    dz = torch.randn(B, cout, H, W, generator=g)
    F.conv2d(x, w).backward(dz)
    Wp = w.detach().permute(2, 3, 1, 0).contiguous().cuda()  # [1][1][rows][cout]
    nw = rows * cout
    pd = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    L.call("punet_split_weights_x6", Wp.data_ptr(), pd.data_ptr(), nw, rows, cout, 1, 1, S())
    prev = torch.randn(B, cs, H, W, generator=g)
    DX = nhwc(prev)
    L.call("punet_conv1x1_x6_gated", nhwc(dz).data_ptr(), cout, None, 0, pd.data_ptr(), nw, rows, off, None, 0, DX.data_ptr(), cs,
           None, None, None, 1, 0, B, H, W, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(DX), prev + x.grad[:, off:off + cs]) < 3e-6


@pytest.mark.parametrize("B,H,W,c0,c1,cout", [(2, 16, 32, 64, 0, 64), (1, 24, 40, 32, 0, 96), (1, 8, 8, 128, 64, 128),
                                              (3, 2, 2, 16, 0, 8), (1, 64, 96, 64, 0, 200)])
def test_conv_x6_pooled_epilogue_is_bit_identical(L, B, H, W, c0, c1, cout):
    """punet_conv3x3_x6_pooled = punet_conv3x3_x6_gated + punet_pool2x2_fwd of its output, bit for bit (output, gate bits
    and the 2x2 average), including partial tiles in x, y and the channel block."""
    g = torch.Generator().manual_seed(21)
    cin = c0 + c1
    X = torch.randn(B, H, W, cin, generator=g).cuda()
    X0 = X[..., :c0].contiguous()
    X1 = X[..., c0:].contiguous() if c1 else None
    Wp = (torch.randn(3, 3, cin, cout, generator=g) / np.sqrt(cin * 9)).cuda()
    b = torch.randn(cout, generator=g).cuda()
    nw = 9 * cin * cout
    pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    L.call("punet_split_weights_x6", Wp.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, S())
    nd = L.call("punet_gate_dwords", B, H, W, cout)
    y1, y2 = torch.empty(B, H, W, cout, device="cuda"), torch.empty(B, H, W, cout, device="cuda")
    g1 = torch.zeros(nd, dtype=torch.int32, device="cuda")
    g2 = torch.zeros(nd, dtype=torch.int32, device="cuda")
    q1 = torch.empty(B, H // 2, W // 2, cout, device="cuda")
    q2 = torch.full((B, H // 2, W // 2, cout), float("nan"), device="cuda")
    L.call("punet_conv3x3_x6_gated", X0.data_ptr(), c0, X1.data_ptr() if c1 else None, c1, pf.data_ptr(), nw, cout, 0, b.data_ptr(), 0,
           y1.data_ptr(), cout, None, None, g1.data_ptr(), 0, 1, B, H, W, OPTS(), S())
    L.call("punet_pool2x2_fwd", y1.data_ptr(), q1.data_ptr(), B, H, W, cout, S())
    L.call("punet_conv3x3_x6_pooled", X0.data_ptr(), c0, X1.data_ptr() if c1 else None, c1, pf.data_ptr(), nw, cout, 0, b.data_ptr(), 0,
           y2.data_ptr(), cout, g2.data_ptr(), q2.data_ptr(), 1, B, H, W, OPTS(), S())
    torch.cuda.synchronize()
    assert torch.equal(y1, y2) and torch.equal(g1, g2)
    assert torch.equal(q1, q2)
    assert (q1 > 0).any()


def test_conv_x6_pooled_rejects_odd_sizes(L):
    X = torch.zeros(1, 5, 8, 8, device="cuda")
    pf = torch.zeros(3 * 9 * 8 * 8, dtype=torch.bfloat16, device="cuda")
    y, q = torch.empty(1, 5, 8, 8, device="cuda"), torch.empty(1, 2, 4, 8, device="cuda")
    with pytest.raises(RuntimeError, match="even H and W"):
        L.call("punet_conv3x3_x6_pooled", X.data_ptr(), 8, None, 0, pf.data_ptr(), 9 * 8 * 8, 8, 0, None, 0, y.data_ptr(), 8, None,
               q.data_ptr(), 1, 1, 5, 8, OPTS(), S())


def test_whole_model_agrees_between_mfma_shapes(L):
    """U-Net + prior + MC logits of a small PUNet with both kernels: the same function to fp32 rounding."""
    from oracle import punet_oracle as O
    from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
    nf = [32, 64, 96, 128]
    m = ProbabilisticUnet(1, 1, nf, 6, 3, 1.0, consensus_masking=True, rl_swap=False, device="cuda", seed=3)
    x, x1, x2, y = O.synth_batch(2, 1, 64, 96, 5)
    eps = torch.randn(4, 2, 6, generator=torch.Generator().manual_seed(1)).cuda()
    outs = []
    for shape in (32, 16):
        eng = m.engine
        eng.x6_opts = L.x6_opts(6, shape)
        with torch.no_grad():
            out = eng.forward_features(m.flat_params, eng.to_nhwc(x1))
            z = eng.reparam(out["prior_muls"], eps)
            ym, cz, lg = eng.sample(m.flat_params, out["features"], z, 0.9, 0.1, True, want_logits=True)
        torch.cuda.synchronize()
        outs.append((out["features"].t.clone(), lg.clone()))
    (f32, l32), (f16, l16) = outs
    assert (f32 - f16).abs().max().item() < 1e-5 * max(1.0, f32.abs().max().item())
    assert (l32 - l16).abs().max().item() < 1e-4 * max(1.0, l32.abs().max().item())
