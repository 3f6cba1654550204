"""GPU: K5x - the first convolution of an UpConvBlock (unet_blocks.py:51-57) with its `up` source contracted at LOW resolution
(csrc/upconv.hip): conv3x3(up(x))[p] = sum_t up(W_t . x)[p + t - 1].  The pieces (weight re-layout, interpolating gather and its
adjoint) and the composite forward / input gradient against PyTorch's CPU fp32 evaluation of the layer AS THE REFERENCE WRITES IT
(F.interpolate + torch.cat + F.conv2d + ReLU and its autograd)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from probabilistic_domain_adaptation_amd import _lib
    _lib.load()
    return _lib


def S():
    return torch.cuda.current_stream().cuda_stream


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def rel_err(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


def up(t):
    return F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True)


def gather_reference(z, cout):
    """sum_t shift_t(up(z_t)) with zero padding at the high resolution; z: [B, 9 * cout, h, w] (tap-major channels)."""
    B, _, h, w = z.shape
    out = torch.zeros(B, cout, 2 * h, 2 * w, dtype=z.dtype)
    for t in range(9):
        ky, kx = t // 3, t % 3
        u = F.pad(up(z[:, t * cout:(t + 1) * cout]), (1, 1, 1, 1))  # u[q + 1] = up(z_t)[q]
        out += u[:, :, ky:ky + 2 * h, kx:kx + 2 * w]               # out[p] += up(z_t)[p + t - 1]
    return out


def test_relayout(L):
    g = torch.Generator().manual_seed(1)
    cup, cb, cout = 24, 16, 8
    w = torch.randn(3, 3, cup + cb, cout, generator=g)
    W = w.cuda()
    w9 = torch.full((cup, 9 * cout), float("nan"), device="cuda")
    wb = torch.full((3, 3, cb, cout), float("nan"), device="cuda")
    L.call("punet_upconv_relayout", W.data_ptr(), cup, cb, cout, w9.data_ptr(), wb.data_ptr(), S())
    torch.cuda.synchronize()
    assert torch.equal(w9.cpu(), w[:, :, :cup].reshape(9, cup, cout).permute(1, 0, 2).reshape(cup, 9 * cout))
    assert torch.equal(wb.cpu(), w[:, :, cup:])


@pytest.mark.parametrize("B,h,w,cout,bits", [(2, 5, 7, 64, True), (1, 8, 16, 128, True), (3, 1, 1, 64, True), (2, 6, 20, 24, False),
                                            (1, 33, 18, 8, False),
                                            (2, 16400, 1, 8, False)])  # more than 65,535 output rows: the grid's z dimension
def test_gather_forward_and_gate_bits(L, B, h, w, cout, bits):
    g = torch.Generator().manual_seed(2)
    z = torch.randn(B, 9 * cout, h, w, generator=g)
    ypre = torch.randn(B, cout, 2 * h, 2 * w, generator=g)
    ref = F.relu(ypre + gather_reference(z, cout))
    Z, Y = nhwc(z), nhwc(ypre)  # z NHWC = [B][h][w][9][cout]
    H, W = 2 * h, 2 * w
    gb = torch.full((L.call("punet_gate_dwords", B, H, W, cout),), -1, dtype=torch.int32, device="cuda") if bits else None
    L.call("punet_upconv_gather_fwd", Z.data_ptr(), Y.data_ptr(), gb.data_ptr() if bits else None, 1, B, h, w, cout, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(Y), ref) < 2e-6
    if bits:  # every bit of the documented layout against (y > 0)
        tiles, cbs = (W + 31) // 32, (cout + 63) // 64
        d = gb.cpu().numpy().astype(np.uint32).reshape(B, H, tiles, cbs, 64)
        yy = Y.cpu().numpy() > 0
        xs, cs = np.arange(W), np.arange(cout)
        lane = (xs[:, None] % 32) + 32 * ((cs[None, :] % 8) // 4)
        bit = 16 * ((cs % 64) // 32) + 4 * ((cs % 32) // 8) + (cs % 4)
        got = (d[:, :, xs[:, None] // 32, cs[None, :] // 64, lane] >> bit[None, :]) & 1
        assert np.array_equal(got.astype(bool), yy)
    # without the ReLU (and in place on a fresh copy)
    Y2 = nhwc(ypre)
    L.call("punet_upconv_gather_fwd", Z.data_ptr(), Y2.data_ptr(), None, 0, B, h, w, cout, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(Y2), ypre + gather_reference(z, cout)) < 2e-6


@pytest.mark.parametrize("B,h,w,cout", [(2, 5, 7, 16), (1, 8, 16, 64), (2, 1, 1, 8), (1, 2, 33, 24), (2100, 32, 1, 8)])  # the last one: more than 65,535 rows (the row loop); h stays small - the fp64 reference has exact weights, fp32 src = scale * o does not at h ~ 3e4
def test_gather_backward_is_the_adjoint(L, B, h, w, cout):
    g = torch.Generator().manual_seed(3)
    z = torch.randn(B, 9 * cout, h, w, generator=g, dtype=torch.float64).requires_grad_(True)
    dz = torch.randn(B, cout, 2 * h, 2 * w, generator=g)
    gather_reference(z, cout).backward(dz.double())
    DZ = nhwc(dz)
    DZT = torch.full((B, h, w, 9 * cout), float("nan"), device="cuda")
    L.call("punet_upconv_gather_bwd", DZ.data_ptr(), DZT.data_ptr(), B, h, w, cout, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(DZT).double(), z.grad) < 2e-6


@pytest.mark.parametrize("B,h,w,cup,cb,cout", [(2, 8, 16, 64, 32, 64), (1, 16, 8, 128, 64, 64), (1, 4, 4, 256, 128, 128),
                                               (2, 6, 10, 40, 24, 64), (3, 40, 56, 192, 96, 128)])
def test_composite_forward_and_input_gradient_against_the_layer_as_written(L, B, h, w, cup, cb, cout):
    """y = ReLU(conv3x3(cat[up(x), bridge]) + b): 1x1 x6 convolution of x with w9, 3x3 x6 convolution of the bridge, gather - and the
    gradient w.r.t. x: adjoint gather of the gated dz, one 1x1 x6 input gradient."""
    from probabilistic_domain_adaptation_amd import _lib
    OPTS = _lib.x6_opts(6, 16)
    g = torch.Generator().manual_seed(4)
    x = torch.relu(torch.randn(B, cup, h, w, generator=g)).requires_grad_(True)
    bridge = torch.relu(torch.randn(B, cb, 2 * h, 2 * w, generator=g))
    wt = (torch.randn(cout, cup + cb, 3, 3, generator=g) * np.sqrt(2.0 / (9 * (cup + cb)))).requires_grad_(True)
    bias = 0.1 * torch.randn(cout, generator=g)
    y = F.relu(F.conv2d(torch.cat([up(x), bridge], 1), wt, bias, padding=1))
    dy = torch.randn(B, cout, 2 * h, 2 * w, generator=g)
    y.backward(dy)
    Wp = wt.detach().permute(2, 3, 1, 0).contiguous().cuda()  # [3][3][cin][cout]
    w9 = torch.empty(cup, 9 * cout, device="cuda")
    wb = torch.empty(3, 3, cb, cout, device="cuda")
    L.call("punet_upconv_relayout", Wp.data_ptr(), cup, cb, cout, w9.data_ptr(), wb.data_ptr(), S())
    n9, nb = cup * 9 * cout, 9 * cb * cout
    p9f = torch.empty(3 * n9, dtype=torch.bfloat16, device="cuda")
    p9d = torch.empty(3 * n9, dtype=torch.bfloat16, device="cuda")
    pbf = torch.empty(3 * nb, dtype=torch.bfloat16, device="cuda")
    L.call("punet_split_weights_x6", w9.data_ptr(), p9f.data_ptr(), n9, cup, 9 * cout, 1, 0, S())
    L.call("punet_split_weights_x6", w9.data_ptr(), p9d.data_ptr(), n9, cup, 9 * cout, 1, 1, S())
    L.call("punet_split_weights_x6", wb.data_ptr(), pbf.data_ptr(), nb, cb, cout, 3, 0, S())
    X, Bg, Bd = nhwc(x.detach()), nhwc(bridge), bias.cuda()
    H, W = 2 * h, 2 * w
    Z = torch.empty(B, h, w, 9 * cout, device="cuda")
    Y = torch.empty(B, H, W, cout, device="cuda")
    bits = torch.zeros(L.call("punet_gate_dwords", B, H, W, cout), dtype=torch.int32, device="cuda")
    L.call("punet_conv1x1_x6_gated", X.data_ptr(), cup, None, 0, p9f.data_ptr(), n9, 9 * cout, 0, None, 0, Z.data_ptr(), 9 * cout,
           None, None, None, 0, 0, B, h, w, S())
    L.call("punet_conv3x3_x6_gated", Bg.data_ptr(), cb, None, 0, pbf.data_ptr(), nb, cout, 0, Bd.data_ptr(), 0, Y.data_ptr(), cout,
           None, None, None, 0, 0, B, H, W, OPTS, S())
    L.call("punet_upconv_gather_fwd", Z.data_ptr(), Y.data_ptr(), bits.data_ptr(), 1, B, h, w, cout, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(Y), y.detach()) < 3e-6
    # input gradient w.r.t. x (gated by x's own ReLU): dz = dy * (y > 0)
    DZ = nhwc(dy * (y.detach() > 0))
    DZT = torch.empty(B, h, w, 9 * cout, device="cuda")
    L.call("punet_upconv_gather_bwd", DZ.data_ptr(), DZT.data_ptr(), B, h, w, cout, S())
    DX = torch.full((B, h, w, cup), float("nan"), device="cuda")
    L.call("punet_conv1x1_x6_gated", DZT.data_ptr(), 9 * cout, None, 0, p9d.data_ptr(), n9, cup, 0, None, 0, DX.data_ptr(), cup,
           X.data_ptr(), None, None, 0, 0, B, h, w, S())
    torch.cuda.synchronize()
    ref = x.grad * (x.detach() > 0)  # x is the leaf (a post-ReLU tensor): the kernel applies ITS gate, as the producing layer's backward needs
    assert rel_err(nchw(DX), ref) < 4e-6
    # weight gradient of the `up` rows in the same form: dW[t][ci][co] = sum_j x[j][ci] dzt[j][t][co]; the bridge rows stay untouched
    ctot = cup + cb
    dw = torch.full((3, 3, ctot, cout), float("nan"), device="cuda")
    dw[:, :, :cup] = 0.25  # (written, not accumulated)
    ns = L.call("punet_upconv_wgrad_slices", cup, cout, B, h, w)
    ws = torch.empty(L.call("punet_upconv_wgrad_workspace_floats", cup, cout, ns), device="cuda")
    L.call("punet_upconv_wgrad_x6", X.data_ptr(), cup, DZT.data_ptr(), cout, dw.data_ptr(), ctot, ws.data_ptr(), ns, B, h, w, 0, 1.0, S())
    torch.cuda.synchronize()
    wref = wt.grad.permute(2, 3, 1, 0)  # [3][3][cin][cout]
    assert rel_err(dw[:, :, :cup].cpu(), wref[:, :, :cup]) < 5e-6
    assert torch.isnan(dw[:, :, cup:]).all()
    L.call("punet_upconv_wgrad_x6", X.data_ptr(), cup, DZT.data_ptr(), cout, dw.data_ptr(), ctot, ws.data_ptr(), ns, B, h, w, 1, 0.5, S())
    torch.cuda.synchronize()
    assert rel_err(dw[:, :, :cup].cpu(), 1.5 * wref[:, :, :cup]) < 5e-6


@pytest.mark.parametrize("wide", ["2", "1", "0"], ids=["128-channel-tiles", "64-channel-tiles", "8-channel-chunks"])
def test_wide_1x1_kernel_over_many_tiles(L, wide):
    """The 1x1 x6 kernel for wide contractions (conv1x1_x6w_kernel) on a problem with more tiles than the persistent grid and a
    channel-block count (9) that does not divide it - a workgroup's successive tiles differ in their output-channel block - forward
    (bias, ReLU, gate bits; and plain, as K5x's GEMM runs it) and input gradient (gate by bits, accumulate), against PyTorch CPU fp32.  $PUNET_CONV1X1_WIDE is read once
    per process, so each setting runs in a child process."""
    import os
    import subprocess
    import sys
    code = r"""
import numpy as np, torch, torch.nn.functional as F
from probabilistic_domain_adaptation_amd import _lib as L
L.load()
S = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(9)
B, H, W, cin, cout = 4, 64, 128, 128, 576
x = torch.relu(torch.randn(B, cin, H, W, generator=g)).requires_grad_(True)
w = (torch.randn(cout, cin, 1, 1, generator=g) / np.sqrt(cin)).requires_grad_(True)
b = torch.randn(cout, generator=g)
y = F.relu(F.conv2d(x, w, b))
dz = torch.randn(B, cout, H, W, generator=g)
y.backward(dz)
Wp = w.detach().permute(2, 3, 1, 0).contiguous().cuda()
nw = cin * cout
pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda"); pd = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
L.call("punet_split_weights_x6", Wp.data_ptr(), pf.data_ptr(), nw, cin, cout, 1, 0, S)
L.call("punet_split_weights_x6", Wp.data_ptr(), pd.data_ptr(), nw, cin, cout, 1, 1, S)
X = x.detach().permute(0, 2, 3, 1).contiguous().cuda()
Y = torch.full((B, H, W, cout), float("nan"), device="cuda")
bits = torch.zeros(L.call("punet_gate_dwords", B, H, W, cout), dtype=torch.int32, device="cuda")
L.call("punet_conv1x1_x6_gated", X.data_ptr(), cin, None, 0, pf.data_ptr(), nw, cout, 0, b.cuda().data_ptr(), 0, Y.data_ptr(), cout,
       None, None, bits.data_ptr(), 0, 1, B, H, W, S)
torch.cuda.synchronize()
rel = lambda a, r: (a - r).abs().max().item() / r.abs().max().item()
e1 = rel(Y.permute(0, 3, 1, 2).cpu(), y.detach())
# input gradient w.r.t. x: dz gated by (y > 0), contracted with W^T, gated by x's own ReLU (as bits of a tensor shaped like x), accumulated
DZ = (dz * (y.detach() > 0)).permute(0, 2, 3, 1).contiguous().cuda()
xb = torch.zeros(L.call("punet_gate_dwords", B, H, W, cin), dtype=torch.int32, device="cuda")
L.call("punet_gate_bits_f16", X.half().data_ptr(), xb.data_ptr(), B, H, W, cin, S)
prev = torch.randn(B, H, W, cin, generator=g)
DX = prev.cuda()
L.call("punet_conv1x1_x6_gated", DZ.data_ptr(), cout, None, 0, pd.data_ptr(), nw, cin, 0, None, 0, DX.data_ptr(), cin,
       None, xb.data_ptr(), None, 1, 0, B, H, W, S)
torch.cuda.synchronize()
ref = prev.permute(0, 3, 1, 2) + x.grad * (x.detach().half() > 0)
e2 = rel(DX.permute(0, 3, 1, 2).cpu(), ref)
# the plain forward (no ReLU, no gate bits: K5x's z, which the 128-channel-tile form computes as the transposed tile and stores in
# whole lines), with and without a bias, on an image whose width is not a multiple of the tile's and rows beyond the last tile row
e3 = 0.0
for (B2, H2, W2), bias2 in (((4, 64, 128), True), ((3, 13, 45), False), ((2, 9, 32), True)):
    x2 = torch.relu(torch.randn(B2, cin, H2, W2, generator=g))
    y2 = F.conv2d(x2, w.detach(), b if bias2 else None)
    X2 = x2.permute(0, 2, 3, 1).contiguous().cuda()
    Y2 = torch.full((B2, H2, W2, cout), float("nan"), device="cuda")
    L.call("punet_conv1x1_x6_gated", X2.data_ptr(), cin, None, 0, pf.data_ptr(), nw, cout, 0, b.cuda().data_ptr() if bias2 else None, 0,
           Y2.data_ptr(), cout, None, None, None, 0, 0, B2, H2, W2, S)
    torch.cuda.synchronize()
    e3 = max(e3, rel(Y2.permute(0, 3, 1, 2).cpu(), y2))
print("ERR", e1, e2, e3)
assert e1 < 2e-6 and e2 < 3e-6 and e3 < 2e-6, (e1, e2, e3)
"""
    env = dict(os.environ, PUNET_CONV1X1_WIDE=wide)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
