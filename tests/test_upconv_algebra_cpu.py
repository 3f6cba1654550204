"""CPU: the algebra behind K5x (csrc/upconv.hip) in fp64, independent of any kernel - bilinear x2 interpolation (per channel) and a
convolution's channel contraction (per pixel) commute, so the first convolution of an UpConvBlock (unet_blocks.py:51-57)

    y = conv3x3(cat[up(x), bridge])

equals, EXACTLY (same real-number function; in fp64 to ~1e-13), the low-resolution form the HIP path computes:

    y = conv3x3_bridge(bridge) + sum_t shift_t(up(W_t . x)),        z_t = W_t . x = a 1x1 convolution of x with the [cup][9 cout] matrix w9,

with zero padding at the HIGH resolution; and autograd of that form gives the input gradient (nine adjoint-interpolated maps of dz, one
1x1 input gradient) and the weight gradient (x^T . those maps) the backward kernels compute."""
import pytest
import torch
import torch.nn.functional as F


def up(t):
    return F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True)


def lowres_form(x, bridge, w, b):
    """The K5x formulation: w [cout, cup + cb, 3, 3]; returns the pre-activation [B, cout, 2h, 2w]."""
    cout, cup = w.shape[0], x.shape[1]
    B, _, h, wd = x.shape
    # w9[ci][t * cout + co] = w[co][ci][ky][kx]  (punet_upconv_relayout), as the weight of ONE 1x1 convolution with 9 cout outputs
    w9 = w[:, :cup].permute(2, 3, 0, 1).reshape(9 * cout, cup, 1, 1)
    z = F.conv2d(x, w9)                                   # [B, 9 cout, h, w]: the nine z_t, tap-major
    y = F.conv2d(bridge, w[:, cup:], b, padding=1)        # the bridge half (with the bias)
    for t in range(9):                                    # the interpolating gather (upconv_gather_fwd_kernel)
        ky, kx = t // 3, t % 3
        u = F.pad(up(z[:, t * cout:(t + 1) * cout]), (1, 1, 1, 1))   # zero padding at the high resolution
        y = y + u[:, :, ky:ky + 2 * h, kx:kx + 2 * wd]
    return y


@pytest.mark.parametrize("B,h,w,cup,cb,cout", [(2, 5, 7, 6, 4, 5), (1, 1, 1, 3, 2, 4), (1, 8, 3, 4, 4, 2), (2, 16, 16, 8, 8, 8)])
def test_lowres_form_equals_the_layer_as_written(B, h, w, cup, cb, cout):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, cup, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    bridge = torch.randn(B, cb, 2 * h, 2 * w, generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(cout, cup + cb, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(cout, generator=g, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(B, cout, 2 * h, 2 * w, generator=g, dtype=torch.float64)

    ref = F.conv2d(torch.cat([up(x), bridge], 1), wt, b, padding=1)           # unet_blocks.py:51-57 (before the ReLU)
    gref = torch.autograd.grad(ref, (x, bridge, wt, b), dy)
    new = lowres_form(x, bridge, wt, b)
    gnew = torch.autograd.grad(new, (x, bridge, wt, b), dy)

    scale = ref.abs().max().item()
    assert (new - ref).abs().max().item() <= 1e-12 * max(scale, 1.0)
    for a_, b_ in zip(gnew, gref):
        assert (a_ - b_).abs().max().item() <= 1e-11 * max(b_.abs().max().item(), 1.0)


def test_a_quarter_of_the_multiply_adds():
    """The `up` half as written contracts cup channels at every HIGH-resolution pixel and tap; the low-resolution form contracts them
    once per LOW-resolution pixel and tap: exactly a quarter (what is left at the high resolution is a gather, no contraction)."""
    B, h, w, cup, cout = 4, 64, 64, 512, 256
    as_written = 2 * B * (2 * h) * (2 * w) * cup * cout * 9
    lowres = 2 * B * h * w * cup * (9 * cout)
    assert as_written == 4 * lowres
