"""GPU: the stream-K tail of the persistent 3x3 kernels (punet_conv3x3_splitk; round 5, tile quantization).  The tiles of the
last partial round of the persistent grid are dealt as one range of K-quads in equal runs over the workgroups; the workgroup
that completes a tile sums its parts in worker order and runs the epilogue.  Checked: against PyTorch CPU fp32 at the unsplit kernels' own tolerance, every epilogue
flavour (bias + ReLU + gate bits + fused pooling; input gradient gated by bits with accumulation; two sources; fp16 planes with
amax publication), bit-reproducibility over back-to-back launches of changing shapes (the hand-off between workgroups on
different XCDs), arrival counters left at zero, and the 88 x 88 level of a 704 x 704 patch (more tiles than the grid)."""
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


class Scratch:
    def __init__(self, L):
        self.n = L.call("punet_conv_splitk_scratch_bytes")
        self.t = torch.zeros(self.n // 4, dtype=torch.int32, device="cuda")

    def counters_zero(self):
        return int(self.t[:1024].abs().sum().item()) == 0


NOFOLD = 1 << 24  # PUNET_X6_NOFOLD: this module is about the rounds of per-image tiling (tests/test_gpu_fold.py covers the folded launches)


def sk_opts(L, run):
    """run = quads (32 contracted channels) per workgroup run of the tail: PUNET_X6_SPLITK(run + 1); 0 = the library's plan, -1 = never"""
    return L.x6_opts(6, 16) | ((run + 1 if run > 0 else (1 if run < 0 else 0)) << 16) | NOFOLD


def conv(L, sc, x0, x1, planes, nw, n_row, n_off, bias, y, n_out, gate_bits, gate_out, pool, accumulate, relu, B, H, W, parts,
         f16=0, amax=(None, None, None, None)):
    L.call("punet_conv3x3_splitk", x0.data_ptr(), x0.shape[-1], x1.data_ptr() if x1 is not None else None,
           x1.shape[-1] if x1 is not None else 0, planes.data_ptr(), nw, n_row, n_off, bias.data_ptr() if bias is not None else None, 0,
           y.data_ptr(), n_out, None, gate_bits.data_ptr() if gate_bits is not None else None,
           gate_out.data_ptr() if gate_out is not None else None, pool.data_ptr() if pool is not None else None, accumulate, relu,
           B, H, W, sk_opts(L, parts), f16, *[t.data_ptr() if t is not None else None for t in amax],
           sc.t.data_ptr() if sc is not None else None, sc.n if sc is not None else 0, S())


# (B, H, W, c0, c1, cout, run): run = quads per workgroup run; runs that do not divide K / 32 make tiles of 2 and 3 parts and
# runs that span two tiles
CASES = [(1, 48, 96, 64, 0, 64, 1), (2, 40, 72, 128, 0, 192, 1), (2, 40, 72, 128, 0, 192, 3), (1, 16, 32, 64, 64, 64, 2),
         (1, 30, 50, 96, 160, 128, 3), (3, 8, 8, 512, 0, 64, 5), (4, 88, 88, 256, 0, 256, 2), (1, 6, 10, 192, 0, 96, 4),
         (4, 88, 88, 256, 0, 256, 7), (2, 65, 88, 128, 0, 128, 3)]


@pytest.mark.parametrize("B,H,W,c0,c1,cout,run", CASES)
def test_split_tail_forward_and_input_gradient_against_torch(L, B, H, W, c0, c1, cout, run):
    g = torch.Generator().manual_seed(3)
    cin = c0 + c1
    x = torch.randn(B, cin, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)).requires_grad_(True)
    b = torch.randn(cout, generator=g) * 0.1
    y = F.relu(F.conv2d(x, w, b, padding=1))
    dz = torch.randn(B, cout, H, W, generator=g)
    y.backward(dz * (y > 0))
    Wp = w.detach().permute(2, 3, 1, 0).contiguous().cuda()
    nw = 9 * cin * cout
    pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    pd = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    L.call("punet_split_weights_x6", Wp.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, S())
    L.call("punet_split_weights_x6", Wp.data_ptr(), pd.data_ptr(), nw, cin, cout, 3, 1, S())
    X = nhwc(x.detach())
    X0 = X[..., :c0].contiguous()
    X1 = X[..., c0:].contiguous() if c1 else None
    sc = Scratch(L)
    even = H % 2 == 0 and W % 2 == 0
    out = {}
    for tag, s_, p_ in (("split", sc, run), ("whole", None, -1)):
        Y = torch.full((B, H, W, cout), float("nan"), device="cuda")
        gate = torch.zeros(L.call("punet_gate_dwords", B, H, W, cout), dtype=torch.int32, device="cuda")
        pool = torch.full((B, H // 2, W // 2, cout), float("nan"), device="cuda") if even else None
        conv(L, s_, X0, X1, pf, nw, cout, 0, b.cuda(), Y, cout, None, gate, pool, 0, 1, B, H, W, p_)
        torch.cuda.synchronize()
        out[tag] = (Y, gate, pool)
    Y, gate, pool = out["split"]
    assert rel_err(nchw(Y), y.detach()) < 2e-6
    assert rel_err(Y, out["whole"][0]) < 4e-6  # same products, another summation order
    if even:
        assert torch.equal(pool, F.avg_pool2d(Y.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1))  # bit-identical to pooling its own output
    assert sc.counters_zero()
    # the gate bits the split launch wrote gate the input gradient exactly like (its own) y > 0
    DZ = nhwc(dz * (y > 0).float())
    off = 0
    for cs in [c0] + ([c1] if c1 else []):
        gx = torch.ones(L.call("punet_gate_dwords", B, H, W, cs), dtype=torch.int32, device="cuda") * -1  # all gates open
        prev = torch.randn(B, cs, H, W, generator=g)
        DX = nhwc(prev)
        # input gradient: contraction over cout
        conv(L, sc, DZ, None, pd, nw, cin, off, None, DX, cs, gx, None, None, 1, 0, B, H, W, run)
        torch.cuda.synchronize()
        ref = prev + x.grad[:, off:off + cs]
        assert rel_err(nchw(DX), ref) < 3e-6
        off += cs
    assert sc.counters_zero()


def test_split_tail_is_bit_reproducible_over_back_to_back_launches(L):
    """The hand-off crosses XCDs (per-XCD L2s are not coherent): 40 launches back to back, alternating two layers that share ONE
    scratch, each compared bit for bit with its first result - a stale or torn partial anywhere shows up here."""
    g = torch.Generator().manual_seed(9)
    sc = Scratch(L)
    layers = []
    for (B, H, W, cin, cout, parts) in [(4, 88, 88, 256, 256, 3), (4, 44, 44, 512, 512, 5), (2, 40, 72, 128, 192, 1), (4, 65, 88, 512, 512, 0)]:
        x = torch.relu(torch.randn(B, H, W, cin, generator=g)).cuda()
        w = (torch.randn(3, 3, cin, cout, generator=g) / np.sqrt(cin * 9)).cuda()
        b = torch.randn(cout, generator=g).cuda()
        nw = 9 * cin * cout
        pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
        L.call("punet_split_weights_x6", w.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, S())
        layers.append((B, H, W, cin, cout, parts, x, pf, nw, b))
    first = {}
    for rep in range(14):
        for i, (B, H, W, cin, cout, parts, x, pf, nw, b) in enumerate(layers):
            Y = torch.empty(B, H, W, cout, device="cuda")
            conv(L, sc, x, None, pf, nw, cout, 0, b, Y, cout, None, None, None, 0, 1, B, H, W, parts)
            if i not in first:
                first[i] = Y
            else:
                assert torch.equal(Y, first[i]), f"launch {rep} of layer {i} differs from its first result"
    torch.cuda.synchronize()
    assert sc.counters_zero()


def test_cost_model_splits_the_704_levels_and_leaves_whole_rounds_alone(L):
    """opts = default (no override): the library's own plan.  At the 88 x 88 x 512 level of a 704 x 704 patch (1,056 tiles on 512
    resident workgroups) it must split - seen as a result that differs in the last bits from the unsplit launch - and at the
    64 x 64 x 512 level of a 512 x 512 patch (512 tiles: one whole round) it must not (bit-identical)."""
    import os
    if os.environ.get("PUNET_SPLITK"):
        pytest.skip("$PUNET_SPLITK overrides the library's plan")
    cap = 2 * L.call("punet_compute_units")
    g = torch.Generator().manual_seed(4)
    sc = Scratch(L)
    res = {}
    for hw in (88, 64):
        B, cin, cout = 4, 512, 512
        x = torch.relu(torch.randn(B, hw, hw, cin, generator=g)).cuda()
        w = (torch.randn(3, 3, cin, cout, generator=g) / np.sqrt(cin * 9)).cuda()
        b = torch.randn(cout, generator=g).cuda()
        nw = 9 * cin * cout
        pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
        L.call("punet_split_weights_x6", w.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, S())
        ys = []
        for s_ in (sc, None):
            Y = torch.empty(B, hw, hw, cout, device="cuda")
            conv(L, s_, x, None, pf, nw, cout, 0, b, Y, cout, None, None, None, 0, 1, B, hw, hw, 0)
            ys.append(Y)
        torch.cuda.synchronize()
        res[hw] = (torch.equal(ys[0], ys[1]), rel_err(ys[0], ys[1]))
    ntiles88 = 4 * 11 * 3 * 8
    if ntiles88 % cap:  # (on a part with another CU count the rounds fall differently)
        assert not res[88][0] and res[88][1] < 4e-6
    if (4 * 8 * 2 * 8) % cap == 0:
        assert res[64][0]


def test_split_tail_with_fp16_planes_publishes_the_same_amax(L):
    """f16x3 (two fp16 planes, per-tensor power-of-two scaling): the combining workgroup rescales and publishes the output's amax
    exactly as the unsplit launch does."""
    g = torch.Generator().manual_seed(6)
    B, H, W, cin, cout, parts = 2, 40, 72, 128, 192, 3
    x = torch.relu(torch.randn(B, H, W, cin, generator=g)).cuda()
    w = (torch.randn(3, 3, cin, cout, generator=g) / np.sqrt(cin * 9)).cuda()
    b = torch.randn(cout, generator=g).cuda()
    nw = 9 * cin * cout
    pf = torch.zeros(3 * nw, dtype=torch.bfloat16, device="cuda")
    aw = torch.zeros(1, device="cuda")
    L.call("punet_split_weights_f16", w.data_ptr(), pf.data_ptr(), nw, cin, cout, 0, aw.data_ptr(), S())
    ax = x.abs().max().reshape(1).contiguous()
    sc = Scratch(L)
    outs = []
    for s_, p_ in ((sc, parts), (None, -1)):
        Y = torch.empty(B, H, W, cout, device="cuda")
        ay = torch.zeros(1, device="cuda")
        conv(L, s_, x, None, pf, nw, cout, 0, b, Y, cout, None, None, None, 0, 1, B, H, W, p_, f16=2, amax=(ax, None, aw, ay))
        torch.cuda.synchronize()
        outs.append((Y, ay))
    ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2).cpu().double(), w.permute(3, 2, 0, 1).cpu().double(), b.cpu().double(), padding=1))
    assert rel_err(nchw(outs[0][0]).double(), ref) < 3e-6
    assert rel_err(outs[0][0], outs[1][0]) < 4e-6
    assert outs[0][1].item() == outs[0][0].abs().max().item() and abs(outs[0][1].item() - outs[1][1].item()) <= 2e-6 * outs[1][1].item()
    assert sc.counters_zero()


def _random_split_cases(count, seed):
    import random
    r = random.Random(seed)
    out = []
    for _ in range(count):
        c0 = 32 * r.randint(1, 8)
        c1 = r.choice([0, 0, 32 * r.randint(1, 4)])
        nq = (c0 + c1) // 32
        out.append((r.randint(1, 4), r.randint(3, 70), r.randint(3, 100), c0, c1, 8 * r.randint(1, 40), r.randint(1, max(1, nq - 1)) if nq > 1 else 1))
    return out


@pytest.mark.parametrize("B,H,W,c0,c1,cout,run", _random_split_cases(16, 77))
def test_split_tail_seeded_draws_against_the_whole_launch(L, B, H, W, c0, c1, cout, run):
    """Seeded draws over batch, image size (any, ragged tiles included), one or two sources, output channels that are not multiples
    of the 64-channel tile, and the run length: the split launch equals the whole launch up to summation order, three launches in
    a row are bit-identical, the counters end at zero."""
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W)
    cin = c0 + c1
    X0 = torch.relu(torch.randn(B, H, W, c0, generator=g)).cuda()
    X1 = torch.relu(torch.randn(B, H, W, c1, generator=g)).cuda() if c1 else None
    w = (torch.randn(3, 3, cin, cout, generator=g) / np.sqrt(cin * 9)).cuda()
    b = torch.randn(cout, generator=g).cuda()
    nw = 9 * cin * cout
    pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
    L.call("punet_split_weights_x6", w.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, S())
    sc = Scratch(L)
    outs = []
    for s_, r_ in ((sc, run), (sc, run), (sc, run), (None, -1)):
        Y = torch.full((B, H, W, cout), float("nan"), device="cuda")
        conv(L, s_, X0, X1, pf, nw, cout, 0, b, Y, cout, None, None, None, 0, 1, B, H, W, r_)
        outs.append(Y)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.isfinite(outs[0]).all() and rel_err(outs[0], outs[3]) < 4e-6
    assert sc.counters_zero()


def test_split_tail_under_load_from_another_stream(L):
    """MI355X_MICROARCH.md: test every inter-workgroup hand-off under UNEVEN load.  A side stream keeps the chip busy with large
    GEMMs of its own (its workgroups take CUs and L2 away at arbitrary moments) while the main stream runs split launches back to
    back: every result must be bit-identical to the one computed on an idle chip."""
    g = torch.Generator().manual_seed(12)
    sc = Scratch(L)
    layers = []
    for (B, H, W, cin, cout, run) in [(4, 88, 88, 512, 512, 0), (4, 96, 96, 256, 256, 1), (4, 65, 88, 512, 512, 0), (2, 40, 72, 128, 192, 1)]:
        x = torch.relu(torch.randn(B, H, W, cin, generator=g)).cuda()
        w = (torch.randn(3, 3, cin, cout, generator=g) / np.sqrt(cin * 9)).cuda()
        b = torch.randn(cout, generator=g).cuda()
        nw = 9 * cin * cout
        pf = torch.empty(3 * nw, dtype=torch.bfloat16, device="cuda")
        L.call("punet_split_weights_x6", w.data_ptr(), pf.data_ptr(), nw, cin, cout, 3, 0, S())
        Y = torch.empty(B, H, W, cout, device="cuda")
        conv(L, sc, x, None, pf, nw, cout, 0, b, Y, cout, None, None, None, 0, 1, B, H, W, run)
        layers.append((B, H, W, cin, cout, run, x, pf, nw, b, Y))
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    stop = torch.cuda.Event()
    with torch.cuda.stream(side):
        for _ in range(60):  # ~60 x 0.1-0.3 ms of foreign work spread over the test
            a = torch.tanh(a @ a * 1e-3)
        stop.record()
    for rep in range(10):
        for i, (B, H, W, cin, cout, run, x, pf, nw, b, ref) in enumerate(layers):
            Y = torch.empty(B, H, W, cout, device="cuda")
            conv(L, sc, x, None, pf, nw, cout, 0, b, Y, cout, None, None, None, 0, 1, B, H, W, run)
            assert torch.equal(Y, ref), f"repetition {rep} of layer {i} differs from the idle-chip result"
    torch.cuda.synchronize()
    assert sc.counters_zero()
