"""Host-side engine: walks the Probabilistic U-Net graph and enqueues the HIP kernels of libpunet_hip.so.

This is the host logic above the C ABI (include/punet_hip.h).  It mirrors the module structure of the reference
(prob_utils/my_models/{unet,unet_blocks,probabilistic_unet}.py) but owns no arithmetic: every FLOP runs in a
hand-written gfx950 kernel.  PyTorch-ROCm is used for device memory (torch.empty), streams and nothing else.

Data layout in HBM
  * parameters: ONE flat fp32 buffer; every conv weight stored [ky][kx][cin][cout] (K-major slabs the kernels read
    with unit stride along cout), biases [cout].  Tensor order = the reference's state_dict() order.
    The same layout is used for gradients, Adam moments and the EMA teacher, so optimizer / EMA / RCCL all-reduce
    are single flat-buffer operations.
  * activations: NHWC fp32, one tensor per conv output (kept for the backward when a tape is recorded).
"""
from __future__ import annotations

import math
import os
import weakref
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib


# --------------------------------------------------------------------------------------------
# parameter layout
# --------------------------------------------------------------------------------------------
def encoder_conv_indices(n_blocks: int, convs_per_block: int = 3) -> List[List[int]]:
    """nn.Sequential indices of the convs inside the reference's Encoder (probabilistic_unet.py:44-63)."""
    idx, out = 0, []
    for i in range(n_blocks):
        if i != 0:
            idx += 1
        blk = []
        for _ in range(convs_per_block):
            blk.append(idx)
            idx += 2
        out.append(blk)
    return out


class ConvP:
    """One conv layer's slot in the flat parameter buffer."""
    __slots__ = ("name", "cin", "cout", "k", "w_off", "b_off")

    def __init__(self, name, cin, cout, k):
        self.name, self.cin, self.cout, self.k = name, cin, cout, k
        self.w_off = self.b_off = -1

    @property
    def w_numel(self):
        return self.k * self.k * self.cin * self.cout


SMALL_CIN = 7  # first layers (image, or [image, mask] of the posterior) with up to 7 channels run through the K1s kernels


class ParamLayout:
    """Flat layout of all parameters; names and logical shapes follow the reference state-dict (M0)."""

    def __init__(self, input_channels=1, num_classes=1, num_filters=(64, 128, 256, 512), latent_dim=6,
                 no_convs_fcomb=3, kind="punet", apply_last_layer=True):
        """kind="punet": the full ProbabilisticUnet; kind="unet": the plain Unet of unet.py:6-69 (names without the
        "unet." prefix, optional 1x1 `last_layer`)."""
        nf = list(num_filters)
        self.kind = kind
        pre = "unet." if kind == "punet" else ""
        if num_classes < 1:
            raise ValueError("num_classes must be >= 1")
        if any(f % 8 or f < 8 for f in nf):
            raise ValueError("num_filters must be multiples of 8 (the MFMA kernels contract 8 channels per K block)")
        if nf[0] not in (8, 16, 32, 64, 128, 256):
            raise ValueError("num_filters[0] must be a power of two in 8..256 (the first-layer kernels give every thread four "
                             "output channels of a pixel: num_filters[0] / 4 has to divide the block)")
        # num_filters[0] in {32, 64} (the reference's default and every script): the fused Fcomb kernels K7 / K7x / K7t, which
        # keep one pixel's channels in registers; any other multiple of 8 runs Fcomb layer by layer (correct, not tuned)
        if not (input_channels >= 1 and input_channels + num_classes <= SMALL_CIN):
            raise ValueError(f"input_channels + num_classes must be <= {SMALL_CIN} (the first-layer kernels; the posterior sees both)")
        if no_convs_fcomb < 2:
            raise ValueError("no_convs_fcomb must be >= 2")
        self.input_channels, self.num_classes, self.nf = input_channels, num_classes, nf
        self.latent_dim, self.no_convs_fcomb = latent_dim, no_convs_fcomb
        self.convs: "OrderedDict[str, ConvP]" = OrderedDict()

        def add(name, cin, cout, k):
            self.convs[name] = ConvP(name, cin, cout, k)

        n = len(nf)
        self.unet_down: List[List[str]] = []
        for i, co in enumerate(nf):
            ci = input_channels if i == 0 else nf[i - 1]
            off = 0 if i == 0 else 1
            names = [f"{pre}contracting_path.{i}.layers.{off + k}" for k in (0, 2, 4)]
            add(names[0], ci, co, 3)
            add(names[1], co, co, 3)
            add(names[2], co, co, 3)
            self.unet_down.append(names)
        self.unet_up: List[List[str]] = []
        out = nf[-1]
        for j, i in enumerate(range(n - 2, -1, -1)):
            ci = out + nf[i]
            out = nf[i]
            names = [f"{pre}upsampling_path.{j}.conv_block.layers.{k}" for k in (0, 2, 4)]
            add(names[0], ci, out, 3)
            add(names[1], out, out, 3)
            add(names[2], out, out, 3)
            self.unet_up.append(names)
        self.enc: Dict[str, List[List[str]]] = {}
        self.fcomb_mid = []
        if kind == "unet" and apply_last_layer:
            add("last_layer", nf[0], num_classes, 1)  # unet.py:45-46
        for net, extra in (("prior", 0), ("posterior", num_classes)) if kind == "punet" else ():
            blocks = []
            for i, blk in enumerate(encoder_conv_indices(n)):
                ci = (input_channels + extra) if i == 0 else nf[i - 1]
                names = [f"{net}.encoder.layers.{li}" for li in blk]
                for k, nm in enumerate(names):
                    add(nm, ci if k == 0 else nf[i], nf[i], 3)
                blocks.append(names)
            self.enc[net] = blocks
            add(f"{net}.conv_layer", nf[-1], 2 * latent_dim, 1)
        if kind == "punet":
            add("fcomb.layers.0", nf[0] + latent_dim, nf[0], 1)
            for k in range(no_convs_fcomb - 2):
                nm = f"fcomb.layers.{2 * (k + 1)}"
                add(nm, nf[0], nf[0], 1)
                self.fcomb_mid.append(nm)
            add("fcomb.last_layer", nf[0], num_classes, 1)

        off = 0
        self.entries: List[Tuple[str, int, int, Tuple[int, ...]]] = []  # (state-dict name, offset, numel, logical shape)
        for c in self.convs.values():
            c.w_off = off
            self.entries.append((c.name + ".weight", off, c.w_numel, (c.cout, c.cin, c.k, c.k)))
            off += c.w_numel
            c.b_off = off
            self.entries.append((c.name + ".bias", off, c.cout, (c.cout,)))
            off += c.cout
            off = (off + 3) // 4 * 4  # keep every tensor 16-byte aligned (float4 loads)
        self.numel = off
        self.n_params = sum(e[2] for e in self.entries)
        # the U-Net's tensors come first (state-dict order): [0, unet_numel) is its slice of every flat buffer
        self.unet_numel = next((o for nm, o, _, _ in self.entries if not nm.startswith(pre)), off) if kind == "punet" else off

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        """Logical (reference-shaped) view of one tensor inside a flat buffer (no copy)."""
        for nm, off, n, shape in self.entries:
            if nm == name:
                return self._view(flat, off, n, shape)
        raise KeyError(name)

    @staticmethod
    def _view(flat, off, n, shape):
        t = flat[off:off + n]
        if len(shape) == 4:
            co, ci, k, _ = shape
            return t.view(k, k, ci, co).permute(3, 2, 0, 1)  # storage [ky][kx][cin][cout] seen as [cout,cin,ky,kx]
        return t

    def views(self, flat: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((nm, self._view(flat, off, n, shape)) for nm, off, n, shape in self.entries)

    def l2_chunk_table(self, chunk: int = 16384) -> Tuple[torch.Tensor, int]:
        """{offset, length, tensor index} triples for posterior, prior and fcomb.layers (mean_teacher_trainer.py:115-116)."""
        rows, t = [], 0
        for nm, off, n, _ in self.entries:
            if nm.startswith("posterior.") or nm.startswith("prior.") or nm.startswith("fcomb.layers."):
                for s in range(0, n, chunk):
                    rows.append((off + s, min(chunk, n - s), t))
                t += 1
        return torch.tensor(rows, dtype=torch.int64), t


# --------------------------------------------------------------------------------------------
# activations + tape
# --------------------------------------------------------------------------------------------
class Act:
    """An NHWC activation tensor with its geometry; relu_out marks tensors whose gradient must be gated by (t > 0)."""
    __slots__ = ("t", "B", "H", "W", "C", "relu_out", "needs_grad", "bits", "amax")

    def __init__(self, t, B, H, W, C, relu_out=False, needs_grad=True):
        self.t, self.B, self.H, self.W, self.C = t, B, H, W, C
        self.relu_out, self.needs_grad = relu_out, needs_grad
        self.bits = None  # (t > 0) as bits (layout: punet_gate_dwords) when the producing kernel published them
        self.amax = None  # fp16-operand modes: one-element device tensor holding (a bound of) max |t|, see csrc/common.h

    @property
    def ptr(self):
        return self.t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# bf16 products per fp32 product in the split-plane kernels; handed to the library with every call (options word)
_X6_PRODUCTS = {"x6": 6, "x3": 3, "bf16": 1, "bf16s": 1, "f16x3": 3, "f16": 1, "f16s": 1}
# fp16 planes per operand of the fp16-operand modes (csrc/common.h): "f16x3" = h + l, products hh + hl + lh (22-23 significand
# bits at x3's cost); "f16" = one plane, the analogue of the reference scripts' fp16 autocast (LIVECell/livecell_mt.py:107)
# ("f16s" = "f16" + fp16 STORAGE of the activations, unscaled as under autocast; gradients stay fp32 and are scaled per tensor)
_F16_PLANES = {"f16x3": 2, "f16": 1, "f16s": 1}
_ACT16 = {"bf16s": torch.bfloat16, "f16s": torch.float16}  # activation storage type of the 16-bit-storage modes


def _aptr(t):
    """Pointer of an amax slot (one-element tensor) or None."""
    return t.data_ptr() if t is not None else None


class Workspace:
    """Grow-only scratch buffers keyed by purpose (split-K slabs, reduction partials).

    A captured hipGraph bakes the raw pointers of these buffers into its kernel nodes (my_trainer/fused.py::GraphedStep).
    While a capture is in progress every buffer handed out is also appended to `CAPTURE_KEEP`, and the GraphedStep holds that
    list for as long as its graph lives: a buffer that has to grow later (an eager call with a larger shape: validation or
    prediction with a bigger batch, a second capture) is simply replaced here - the old tensor stays alive exactly as long as a
    graph still writes through it and is freed with that graph."""

    CAPTURE_KEEP: Optional[list] = None

    def __init__(self, device):
        self.device = device
        self.bufs: Dict[str, torch.Tensor] = {}

    def get(self, key: str, nfloats: int) -> torch.Tensor:
        b = self.bufs.get(key)
        if b is None or b.numel() < nfloats:
            b = torch.empty(max(int(nfloats), 1), dtype=torch.float32, device=self.device)
            self.bufs[key] = b
        keep = Workspace.CAPTURE_KEEP
        if keep is not None and not any(b is k for k in keep):
            keep.append(b)
        return b


class PUNetEngine:
    """Enqueues the kernels for one model configuration on one GPU.  Stateless w.r.t. weights: every call takes the
    flat parameter tensor to use (student or teacher)."""

    def __init__(self, layout: ParamLayout, device, conv_mode: Optional[str] = None):
        self.L = layout
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("PUNetEngine needs a ROCm device: the product path has no CPU fallback")
        _lib.load()
        self.ws = Workspace(self.device)
        self.prof = None  # when a list: (kind, algorithmic FLOPs, start event, end event) per 3x3 MFMA launch
        self.prof_kinds = None  # optional set of kinds to bracket (every event pair costs ~1 us of GPU idle time)
        self.wT = None  # flipped + transposed weight slabs for the dgrad pass (prepare_dgrad_weights)
        # 3x3 conv arithmetic.  Parity modes (same tolerances): "x6" = three-plane bf16 split on the BF16 matrix cores
        # (fp32-accurate, default), "f32" = fp32-input MFMA.  Opt-in fast modes (the analogue of the reference scripts'
        # mixed_precision=True, e.g. livecell_mt.py:107; NOT parity modes): "x3" = two planes, hh+hm+mh (16 significand
        # bits, more than the fp16 autocast of the reference), "bf16" = one plane.  Storage stays fp32 in every mode.
        self.conv_mode = conv_mode or os.environ.get("PUNET_CONV_MODE", "x6")
        if self.conv_mode not in _X6_PRODUCTS and self.conv_mode != "f32":
            raise ValueError("PUNET_CONV_MODE must be one of 'x6', 'f32', 'x3', 'bf16', 'bf16s', 'f16x3', 'f16', 'f16s'")
        self.x6 = self.conv_mode in _X6_PRODUCTS  # the split-plane kernel family
        self.f16 = _F16_PLANES.get(self.conv_mode, 0)  # fp16 planes per operand (0: the bf16 planes of the x6 family)
        self._wamax = {}  # flat.data_ptr() -> device float [nconv]: amax slot per 3x3 weight slab (fp16-operand modes)
        self._x6_index = {}  # conv name -> row of the split table (= index of its slot in _wamax)
        self._amax_pool, self._amax_next = None, 0
        # "bf16s" = "bf16" + bf16 STORAGE of the activations wherever no backward follows (forward_features: inference, the
        # teacher / weak pass): half the activation traffic, no fp32 -> bf16 conversion in the conv loaders
        self.act16 = _ACT16.get(self.conv_mode)  # None: fp32 storage
        self._sfx16 = "bf16s" if self.act16 is torch.bfloat16 else "f16s"  # suffix of the 16-bit-storage entry points
        self._infer16 = False  # set while forward_features runs in that mode
        # per-call options word of the split-plane kernels: products of this mode, MFMA shape 16 ($PUNET_X6_MFMA_SHAPE=32
        # selects the older 32x32x16 kernels, an A/B switch)
        shape = 32 if os.environ.get("PUNET_X6_MFMA_SHAPE", "16") == "32" else 16
        self.x6_opts = _lib.x6_opts(_X6_PRODUCTS.get(self.conv_mode, 6), shape)
        # the last conv of a contracting block also writes the next block's 2x2 average (PUNET_FUSE_POOL=0: separate kernel)
        self.fuse_pool = os.environ.get("PUNET_FUSE_POOL", "1") != "0"
        # training path: Fcomb forward + BCE + backward in one kernel (K7t) where it applies (PUNET_FUSE_FCOMB=0: per-op path)
        self.fuse_fcomb = os.environ.get("PUNET_FUSE_FCOMB", "1") != "0"
        self.sampler_x6 = os.environ.get("PUNET_SAMPLER_X6", "1") != "0"  # K7 on the BF16 matrix cores (0: fp32-MFMA kernel)
        self._x6 = {}  # (flat.data_ptr(), mode) -> bf16 planes buffer [3 * numel]
        self._x6_table = None  # (device int64 [nconv+1, 4], nconv, total_units) for punet_split_weights_x6_batch
        self._x6_1x1 = {}  # (flat.data_ptr(), mode) -> {conv name: bf16 planes} for the Fcomb 1x1 layers
        # K5x (csrc/upconv.hip): the first conv of an up-block with its `up` source contracted at LOW resolution - the parity
        # arithmetic only ($PUNET_UPCONV=0: the layer as written, on the materialised up-sampled tensor; an A/B switch)
        self.upconv = self.conv_mode == "x6" and os.environ.get("PUNET_UPCONV", "1") != "0"
        # stream-K tail of the persistent 3x3 kernels (round 5; $PUNET_SPLITK=0: every launch whole, an A/B switch)
        self.splitk = os.environ.get("PUNET_SPLITK", "") != "0"
        self._sk = None
        self._up = {}  # (flat.data_ptr(), conv name) -> dict(w9, wb fp32 re-layouts; p9, pb forward planes; p9d input-gradient planes)
        tab, self.l2_ntensors = layout.l2_chunk_table()
        self.l2_chunks = tab.to(self.device)
        self.l2_nchunks = tab.shape[0]

    def _splitk_scratch(self):
        """The stream-K scratch of this engine's 3x3 launches (punet_conv3x3_splitk): arrival counters (zero once, left zero by
        every launch) + partial-accumulator slots; one buffer for the engine's life - launches are stream-ordered."""
        if self._sk is None:
            self._sk = torch.zeros(_lib.call("punet_conv_splitk_scratch_bytes") // 4, dtype=torch.int32, device=self.device)
        return self._sk

    def _timed(self, kind, flops, name, *args):
        """Launch through the C ABI; with self.prof set, bracket the launch with HIP events on the launch stream.
        The 3x3 split-plane entry points are re-targeted at their superset punet_conv3x3_splitk with this engine's stream-K
        scratch (tile quantization of the persistent grid; $PUNET_SPLITK=0 leaves the launches whole)."""
        if self.splitk and name in ("punet_conv3x3_x6_gated", "punet_conv3x3_x6_pooled", "punet_conv3x3_f16"):
            sk = self._splitk_scratch()
            tail = (sk.data_ptr(), 4 * sk.numel(), args[-1])
            if name == "punet_conv3x3_x6_gated":    # ..., y, n_out, mask, gate_bits, gate_out, accumulate, relu, B, H, W, opts, stream
                a = args[:15] + (None,) + args[15:21] + (0, None, None, None, None)
            elif name == "punet_conv3x3_x6_pooled":  # ..., y, n_out, gate_out, pool_out, relu, B, H, W, opts, stream
                a = args[:12] + (None, None, args[12], args[13], 0) + args[14:19] + (0, None, None, None, None)
            else:                                    # ..., pool_out, accumulate, relu, B, H, W, planes, amax x0 / x1 / w / y, stream
                a = args[:21] + (0,) + args[21:26]
            name, args = "punet_conv3x3_splitk", a + tail
        if self.prof is None or (self.prof_kinds is not None and kind not in self.prof_kinds):
            return _lib.call(name, *args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = _lib.call(name, *args)
        e1.record()
        self.prof.append((kind, flops, e0, e1))
        return r

    # ---------------------------------------------------------------- helpers
    def _empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def _slot(self):
        """A zeroed amax slot (one-element view of a pool tensor that stays alive as long as a slot of it is referenced)."""
        if self._amax_pool is None or self._amax_next >= self._amax_pool.numel():
            self._amax_pool, self._amax_next = torch.zeros(256, dtype=torch.float32, device=self.device), 0
        v = self._amax_pool[self._amax_next:self._amax_next + 1]
        self._amax_next += 1
        return v

    def _amax_of(self, t: torch.Tensor):
        """Standalone amax pass over a tensor whose producer does not publish it (first-layer conv, Fcomb / latent-head backward)."""
        sl = self._slot()
        _lib.call("punet_amax", t.data_ptr(), t.numel(), sl.data_ptr(), 0, _stream())
        return sl

    def _wamax_ptr(self, flat, conv: "ConvP"):
        return self._wamax[flat.data_ptr()].data_ptr() + 4 * self._x6_index[conv.name]

    def _wptr(self, flat, conv: ConvP):
        return flat.data_ptr() + 4 * conv.w_off

    def _bptr(self, flat, conv: ConvP):
        return flat.data_ptr() + 4 * conv.b_off

    def to_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        """[B,C,H,W] (any strides) -> contiguous NHWC fp32 on the device (the same bytes when C == 1)."""
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        B, C, H, W = x.shape
        if C == 1:
            return x.view(B, H, W, 1)
        y = self._empty(B, H, W, C)
        _lib.call("punet_nchw_to_nhwc", x.data_ptr(), y.data_ptr(), B, C, H * W, _stream())
        return y

    def to_nchw(self, a: "Act") -> torch.Tensor:
        """NHWC activation -> contiguous [B,C,H,W] tensor (what the reference's attributes such as unet_features hold)."""
        if a.t.dtype != torch.float32:  # bf16 storage (fast mode): inspection path only
            return a.t.float().permute(0, 3, 1, 2).contiguous()
        if a.C == 1:
            return a.t.view(a.B, 1, a.H, a.W)
        y = self._empty(a.B, a.C, a.H, a.W)
        _lib.call("punet_nhwc_to_nchw", a.ptr, y.data_ptr(), a.B, a.C, a.H * a.W, _stream())
        return y

    # ---------------------------------------------------------------- forward ops
    def conv(self, flat, name: str, x0: Act, x1: Optional[Act] = None, relu=True, tape=None, taps=None, tape_bits=False,
             pooled: Optional[list] = None) -> Act:
        """tape_bits: publish the ReLU gate bits of the output even without a tape (Fcomb layers, whose backward is
        written out by hand in elbo_backward).  pooled: a list that receives the 2x2 average of the output when the kernel
        can produce it in its epilogue (split-plane 3x3 kernel, even H and W); left empty otherwise."""
        c = self.L.convs[name]
        B, H, W = x0.B, x0.H, x0.W
        cin = x0.C + (x1.C if x1 is not None else 0)
        assert cin == c.cin, (name, cin, c.cin)
        if self._infer16 and c.k == 3:  # 16-bit activation storage ("bf16s" / "f16s"): the gradients, if any follow, stay fp32
            y = Act(torch.empty(B, H, W, c.cout, dtype=self.act16, device=self.device), B, H, W, c.cout, relu_out=relu)
            if cin <= SMALL_CIN:
                _lib.call("punet_conv3x3_small_fwd_" + self._sfx16, x0.ptr, cin, self._wptr(flat, c), self._bptr(flat, c), y.ptr, c.cout,
                          B, H, W, int(relu), _stream())
                if relu and tape is not None:  # the first-layer kernel publishes no gate bits: one small pass makes them
                    y.bits = torch.empty(_lib.call("punet_gate_dwords", B, H, W, c.cout), dtype=torch.int32, device=self.device)
                    _lib.call("punet_gate_bits_bf16" if self.act16 is torch.bfloat16 else "punet_gate_bits_f16", y.ptr,
                              y.bits.data_ptr(), B, H, W, c.cout, _stream())
            else:
                q = None
                if pooled is not None and H % 2 == 0 and W % 2 == 0:
                    q = Act(torch.empty(B, H // 2, W // 2, c.cout, dtype=self.act16, device=self.device), B, H // 2, W // 2, c.cout)
                    pooled.append(q)
                if relu and (tape is not None or tape_bits):
                    y.bits = torch.empty(_lib.call("punet_gate_dwords", B, H, W, c.cout), dtype=torch.int32, device=self.device)
                planes = self._x6[(flat.data_ptr(), 0)]
                args = (x0.ptr, x0.C, x1.ptr if x1 is not None else None, x1.C if x1 is not None else 0,
                        planes.data_ptr() + 2 * 3 * c.w_off, c.w_numel, c.cout, 0, self._bptr(flat, c), y.ptr, c.cout,
                        y.bits.data_ptr() if y.bits is not None else None, q.ptr if q is not None else None, int(relu), B, H, W)
                if self.f16:  # fp16 storage is unscaled (as under autocast); the weights carry their slab's power-of-two scale
                    self._timed("conv3x3_fwd", 2.0 * B * H * W * cin * c.cout * 9, "punet_conv3x3_f16s", *args,
                                self._wamax_ptr(flat, c), _stream())
                else:
                    self._timed("conv3x3_fwd", 2.0 * B * H * W * cin * c.cout * 9, "punet_conv3x3_bf16s", *args, _stream())
            if tape is not None:
                tape.append(("conv", c, x0, x1, y))
            if taps is not None:
                taps["conv:" + name] = y
            return y
        y = Act(self._empty(B, H, W, c.cout), B, H, W, c.cout, relu_out=relu)
        if cin <= SMALL_CIN:
            assert x1 is None and c.k == 3
            if self.f16:  # the next layer scales its operand by this tensor's amax: published by the kernel itself
                y.amax = self._slot()
                _lib.call("punet_conv3x3_small_fwd_amax", x0.ptr, cin, self._wptr(flat, c), self._bptr(flat, c), y.ptr, c.cout,
                          B, H, W, int(relu), y.amax.data_ptr(), _stream())
            else:
                _lib.call("punet_conv3x3_small_fwd", x0.ptr, cin, self._wptr(flat, c), self._bptr(flat, c), y.ptr, c.cout,
                          B, H, W, int(relu), _stream())
        elif self.f16 and c.k == 3:  # fp16 planes scaled by powers of two from the operands' amax slots; publishes y's amax
            planes = self._x6[(flat.data_ptr(), 0)]
            if relu and (tape is not None or tape_bits):
                y.bits = torch.empty(_lib.call("punet_gate_dwords", B, H, W, c.cout), dtype=torch.int32, device=self.device)
            q = None
            if pooled is not None and H % 2 == 0 and W % 2 == 0:
                q = Act(self._empty(B, H // 2, W // 2, c.cout), B, H // 2, W // 2, c.cout)
                pooled.append(q)
            y.amax = self._slot()
            if q is not None:
                q.amax = y.amax  # an average never exceeds the maximum
            self._timed("conv3x3_fwd", 2.0 * B * H * W * cin * c.cout * 9, "punet_conv3x3_f16", x0.ptr, x0.C,
                        x1.ptr if x1 is not None else None, x1.C if x1 is not None else 0, planes.data_ptr() + 2 * 3 * c.w_off,
                        c.w_numel, c.cout, 0, self._bptr(flat, c), 0, y.ptr, c.cout, None, None,
                        y.bits.data_ptr() if y.bits is not None else None, q.ptr if q is not None else None, 0, int(relu), B, H, W,
                        self.f16, _aptr(x0.amax), _aptr(x1.amax) if x1 is not None else None, self._wamax_ptr(flat, c),
                        y.amax.data_ptr(), _stream())
        elif self.x6 and c.k == 3:
            planes = self._x6[(flat.data_ptr(), 0)]
            if relu and (tape is not None or tape_bits):  # a backward will follow: publish the ReLU gate of y as bits
                y.bits = torch.empty(_lib.call("punet_gate_dwords", B, H, W, c.cout), dtype=torch.int32, device=self.device)
            if pooled is not None and H % 2 == 0 and W % 2 == 0:
                q = Act(self._empty(B, H // 2, W // 2, c.cout), B, H // 2, W // 2, c.cout)
                pooled.append(q)
                self._timed("conv3x3_fwd", 2.0 * B * H * W * cin * c.cout * 9, "punet_conv3x3_x6_pooled", x0.ptr, x0.C,
                            x1.ptr if x1 is not None else None, x1.C if x1 is not None else 0,
                            planes.data_ptr() + 2 * 3 * c.w_off, c.w_numel, c.cout, 0, self._bptr(flat, c), 0, y.ptr, c.cout,
                            y.bits.data_ptr() if y.bits is not None else None, q.ptr, int(relu), B, H, W, self.x6_opts, _stream())
            else:
                self._timed("conv3x3_fwd", 2.0 * B * H * W * cin * c.cout * 9, "punet_conv3x3_x6_gated", x0.ptr, x0.C,
                            x1.ptr if x1 is not None else None, x1.C if x1 is not None else 0,
                            planes.data_ptr() + 2 * 3 * c.w_off, c.w_numel, c.cout, 0, self._bptr(flat, c), 0, y.ptr, c.cout,
                            None, None, y.bits.data_ptr() if y.bits is not None else None, 0, int(relu), B, H, W, self.x6_opts,
                            _stream())
        elif self.x6 and c.k == 1 and x1 is None and name in self._x6_1x1.get((flat.data_ptr(), 0), {}):
            planes = self._x6_1x1[(flat.data_ptr(), 0)][name]
            if relu and tape_bits:
                y.bits = torch.empty(_lib.call("punet_gate_dwords", B, H, W, c.cout), dtype=torch.int32, device=self.device)
            self._timed("conv1x1_fwd", 2.0 * B * H * W * cin * c.cout, "punet_conv1x1_x6_gated", x0.ptr, cin, None, 0,
                        planes.data_ptr(), cin * c.cout, c.cout, 0, self._bptr(flat, c), 0, y.ptr, c.cout, None, None,
                        y.bits.data_ptr() if y.bits is not None else None, 0, int(relu), B, H, W, _stream())
        else:
            self._timed(f"conv{c.k}x{c.k}_fwd", 2.0 * B * H * W * cin * c.cout * c.k * c.k, "punet_conv_fwd", x0.ptr, x0.C,
                        x1.ptr if x1 is not None else None, x1.C if x1 is not None else 0, self._wptr(flat, c),
                        self._bptr(flat, c), 0, y.ptr, c.cout, B, H, W, c.k, int(relu), _stream())
        if tape is not None:
            tape.append(("conv", c, x0, x1, y))
        if taps is not None:
            taps["conv:" + name] = y
        return y

    # ---------------------------------------------------------------- K5x: up-block entry conv at low resolution
    @property
    def UPCONV_MIN_CUP_GRAD(self) -> int:
        """Forward passes a backward follows keep the layer as written below this many `up` channels (the library's figure, shared
        with csrc/runtime.hip)."""
        return _lib.call("punet_upconv_min_cup_grad")

    def _upconv_ok(self, c: ConvP, x: Act, bridge: Act) -> bool:
        """Shapes the low-resolution form serves: the library's rule (punet_upconv_eligible, also used by csrc/runtime.hip: the two
        graph walkers agree bit for bit) on fp32 tensors in the parity arithmetic."""
        return bool(self.upconv and x.t.dtype == torch.float32 and bridge.t.dtype == torch.float32
                    and _lib.call("punet_upconv_eligible", x.C, bridge.C, c.cout, c.k))

    def _up_entry(self, flat, name: str) -> dict:
        d = self._up.get((flat.data_ptr(), name))
        if d is None:
            raise RuntimeError(f"K5x planes of {name} are missing for this parameter buffer: call prepare_forward_weights / "
                               "prepare_dgrad_weights on it before a pass (they re-lay out and split the up-blocks' entry slabs)")
        return d

    def _up_weights(self, flat, c: ConvP, cup: int, mode: int):
        """Re-layout + plane split of one up-block entry slab: w9 [cup][9 cout] (the 1x1 form of the `up` rows) and the bridge rows
        as a 3x3 slab; mode 0: forward planes of both, mode 1: input-gradient planes of w9 (the bridge's input gradient takes the
        ordinary planes of the whole slab)."""
        key = (flat.data_ptr(), c.name)
        d = self._up.get(key)
        cb, n9, nb = c.cin - cup, cup * 9 * c.cout, 9 * (c.cin - cup) * c.cout
        if d is None:
            d = {"w9": self._empty(n9), "wb": self._empty(nb),
                 "p9": torch.empty(3 * n9, dtype=torch.bfloat16, device=self.device),
                 "pb": torch.empty(3 * nb, dtype=torch.bfloat16, device=self.device),
                 "p9d": torch.empty(3 * n9, dtype=torch.bfloat16, device=self.device)}
            self._up[key] = d
        _lib.call("punet_upconv_relayout", self._wptr(flat, c), cup, cb, c.cout, d["w9"].data_ptr(), d["wb"].data_ptr(), _stream())
        if mode == 0:
            _lib.call("punet_split_weights_x6", d["w9"].data_ptr(), d["p9"].data_ptr(), n9, cup, 9 * c.cout, 1, 0, _stream())
            _lib.call("punet_split_weights_x6", d["wb"].data_ptr(), d["pb"].data_ptr(), nb, cb, c.cout, 3, 0, _stream())
        else:
            _lib.call("punet_split_weights_x6", d["w9"].data_ptr(), d["p9d"].data_ptr(), n9, cup, 9 * c.cout, 1, 1, _stream())
        return d

    def _split_up(self, flat, mode: int):
        if not self.upconv or self.L.kind not in ("punet", "unet"):
            return
        nf = self.L.nf
        for j, names in enumerate(self.L.unet_up):
            c = self.L.convs[names[0]]
            cup = c.cin - nf[len(nf) - 2 - j]
            if _lib.call("punet_upconv_eligible", cup, c.cin - cup, c.cout, c.k):
                self._up_weights(flat, c, cup, mode)

    def upconv_fwd(self, flat, name: str, x: Act, bridge: Act, tape=None, taps=None) -> Act:
        """ReLU(conv3x3(cat[up(x), bridge]) + b) (unet_blocks.py:51-57) as  1x1 conv of x with w9 at the low resolution, 3x3 conv of
        the bridge, interpolating gather (csrc/upconv.hip).  The up-sampled tensor is not materialised."""
        c = self.L.convs[name]
        B, h, w, cup, cb = x.B, x.H, x.W, x.C, bridge.C
        H, W = 2 * h, 2 * w
        d = self._up_entry(flat, name)
        z = self._empty(B, h, w, 9 * c.cout)
        self._timed("conv1x1_fwd", 2.0 * B * h * w * cup * 9 * c.cout, "punet_conv1x1_x6_gated", x.ptr, cup, None, 0, d["p9"].data_ptr(),
                    cup * 9 * c.cout, 9 * c.cout, 0, None, 0, z.data_ptr(), 9 * c.cout, None, None, None, 0, 0, B, h, w, _stream())
        y = Act(self._empty(B, H, W, c.cout), B, H, W, c.cout, relu_out=True)
        self._timed("conv3x3_fwd", 2.0 * B * H * W * cb * c.cout * 9, "punet_conv3x3_x6_gated", bridge.ptr, cb, None, 0,
                    d["pb"].data_ptr(), 9 * cb * c.cout, c.cout, 0, self._bptr(flat, c), 0, y.ptr, c.cout, None, None, None, 0, 0,
                    B, H, W, self.x6_opts, _stream())
        if tape is not None:
            y.bits = torch.empty(_lib.call("punet_gate_dwords", B, H, W, c.cout), dtype=torch.int32, device=self.device)
        _lib.call("punet_upconv_gather_fwd", z.data_ptr(), y.ptr, y.bits.data_ptr() if y.bits is not None else None, 1, B, h, w,
                  c.cout, _stream())
        if tape is not None:
            tape.append(("upconv", c, x, bridge, y))
        if taps is not None:
            taps["conv:" + name] = y
        return y

    def pool(self, x: Act, tape=None) -> Act:
        OH, OW = (x.H + 1) // 2, (x.W + 1) // 2
        y = Act(self._empty(x.B, OH, OW, x.C), x.B, OH, OW, x.C)
        _lib.call("punet_pool2x2_fwd", x.ptr, y.ptr, x.B, x.H, x.W, x.C, _stream())
        y.amax = x.amax  # a bound: an average never exceeds the maximum
        if tape is not None:
            tape.append(("pool", x, y))
        return y

    def upsample(self, x: Act, tape=None) -> Act:
        if x.t.dtype != torch.float32:
            y = Act(torch.empty(x.B, 2 * x.H, 2 * x.W, x.C, dtype=x.t.dtype, device=self.device), x.B, 2 * x.H, 2 * x.W, x.C)
            _lib.call("punet_upsample2x_fwd_bf16s" if x.t.dtype == torch.bfloat16 else "punet_upsample2x_fwd_f16s", x.ptr, y.ptr, x.B,
                      x.H, x.W, x.C, _stream())
            if tape is not None:
                tape.append(("up", x, y))
            return y
        y = Act(self._empty(x.B, 2 * x.H, 2 * x.W, x.C), x.B, 2 * x.H, 2 * x.W, x.C)
        y.amax = x.amax  # a bound: bilinear interpolation is a convex combination
        _lib.call("punet_upsample2x_fwd", x.ptr, y.ptr, x.B, x.H, x.W, x.C, _stream())
        if tape is not None:
            tape.append(("up", x, y))
        return y

    def encoder_stack(self, flat, blocks: Sequence[Sequence[str]], x: Act, tape=None, taps=None, tag="") -> Tuple[Act, List[Act]]:
        """The 4-block contracting stack shared by Unet (unet.py:50-55) and Encoder (probabilistic_unet.py:67-69)."""
        outs = []
        pooled: list = []
        for i, names in enumerate(blocks):
            if i != 0:
                if pooled:  # the previous block's last conv has written its 2x2 average already
                    y = pooled.pop()
                    if tape is not None:
                        tape.append(("pool", x, y))
                    x = y
                else:
                    x = self.pool(x, tape)
            for k, nm in enumerate(names):
                last = k == len(names) - 1 and i + 1 < len(blocks) and self.fuse_pool
                x = self.conv(flat, nm, x, None, True, tape, taps, pooled=pooled if last else None)
            outs.append(x)
            if taps is not None:
                taps[f"{tag}{i}"] = x
        return x, outs

    def unet(self, flat, x: Act, tape=None, taps=None) -> Act:
        """Unet.forward with apply_last_layer=False (unet.py:50-69; UpConvBlock.forward unet_blocks.py:49-59)."""
        if x.H % (1 << (len(self.L.nf) - 1)) or x.W % (1 << (len(self.L.nf) - 1)):
            raise ValueError(f"H and W must be divisible by {1 << (len(self.L.nf) - 1)} (got {x.H}x{x.W})")
        x, outs = self.encoder_stack(flat, self.L.unet_down, x, tape, taps, "unet.down")
        for j, names in enumerate(self.L.unet_up):
            bridge = outs[-j - 2]
            c0 = self.L.convs[names[0]]
            if self._upconv_ok(c0, x, bridge) and (tape is None or x.C >= self.UPCONV_MIN_CUP_GRAD):
                assert 2 * x.W == bridge.W and 2 * x.H == bridge.H  # unet_blocks.py:55
                x = self.upconv_fwd(flat, names[0], x, bridge, tape, taps)
            else:
                up = self.upsample(x, tape)
                assert up.W == bridge.W and up.H == bridge.H  # unet_blocks.py:55
                x = self.conv(flat, names[0], up, bridge, True, tape, taps)  # concat order [up, bridge] (unet_blocks.py:56)
            x = self.conv(flat, names[1], x, None, True, tape, taps)
            x = self.conv(flat, names[2], x, None, True, tape, taps)
            if taps is not None:
                taps[f"unet.up{j}"] = x
        return x

    def latent(self, flat, net: str, x: Act, tape=None, taps=None):
        """AxisAlignedConvGaussian.forward (probabilistic_unet.py:113-142) -> (muls [B, 2L], mean [B, C], enc Act)."""
        enc, _ = self.encoder_stack(flat, self.L.enc[net], x, tape, taps, f"{net}.enc")
        c = self.L.convs[f"{net}.conv_layer"]
        P = enc.H * enc.W
        mean = self._empty(enc.B, enc.C)
        muls = self._empty(enc.B, c.cout)
        ws = self.ws.get("chan", _lib.call("punet_chan_sum_workspace_floats", enc.B, P, enc.C))
        _lib.call({torch.bfloat16: "punet_latent_head_fwd_bf16s", torch.float16: "punet_latent_head_fwd_f16s",
                   torch.float32: "punet_latent_head_fwd"}[enc.t.dtype], enc.ptr,
                  self._wptr(flat, c), self._bptr(flat, c), mean.data_ptr(), muls.data_ptr(), ws.data_ptr(), enc.B, P, enc.C, c.cout,
                  _stream())
        return muls, mean, enc

    # ---------------------------------------------------------------- inference / teacher path
    def forward_features(self, flat, x_nhwc: torch.Tensor, taps=None):
        """ProbabilisticUnet.forward(patch, None, training=False) (probabilistic_unet.py:292-293)."""
        B, H, W, C = x_nhwc.shape
        self.prepare_forward_weights(flat)
        xin = Act(x_nhwc, B, H, W, C, needs_grad=False)
        self._infer16 = self.act16 is not None
        try:
            muls_p, _, _ = self.latent(flat, "prior", xin, None, taps)
            feat = self.unet(flat, xin, None, taps)
        finally:
            self._infer16 = False
        return {"features": feat, "prior_muls": muls_p}

    def _fcomb_ptrs(self, flat):
        L = self.L
        c1, c3 = L.convs["fcomb.layers.0"], L.convs["fcomb.last_layer"]
        nmid = len(L.fcomb_mid)
        if nmid:
            m0 = L.convs[L.fcomb_mid[0]]
            wmid, bmid = self._wptr(flat, m0), self._bptr(flat, m0)
            stride = (L.convs[L.fcomb_mid[1]].w_off - m0.w_off) if nmid > 1 else 0
        else:
            wmid = bmid = None
            stride = 0
        return c1, c3, nmid, wmid, bmid, stride

    def _fcomb_hidden(self, flat, feat: Act, zb_ptr: int, record=False, taps=None):
        """The hidden layers of Fcomb for ONE latent sample, layer by layer (probabilistic_unet.py:200-214): layers.0 on the
        features with the z term as a per-image bias (zb_ptr: [B][CH]), then the middle layers.  Returns their activations."""
        L = self.L
        B, H, W, CH = feat.B, feat.H, feat.W, L.nf[0]
        c1 = L.convs["fcomb.layers.0"]
        h = Act(self._empty(B, H, W, CH), B, H, W, CH, relu_out=True)
        p1 = self._x6_1x1.get((flat.data_ptr(), 0), {}).get(c1.name) if self.x6 else None
        if p1 is not None:
            if record:
                h.bits = torch.empty(_lib.call("punet_gate_dwords", B, H, W, CH), dtype=torch.int32, device=self.device)
            self._timed("conv1x1_fwd", 2.0 * B * H * W * CH * CH, "punet_conv1x1_x6_gated", feat.ptr, CH, None, 0, p1.data_ptr(),
                        CH * CH, CH, 0, zb_ptr, CH, h.ptr, CH, None, None,
                        h.bits.data_ptr() if h.bits is not None else None, 0, 1, B, H, W, _stream())
        else:
            _lib.call("punet_conv_fwd", feat.ptr, CH, None, 0, self._wptr(flat, c1), zb_ptr, CH, h.ptr, CH, B, H, W, 1, 1,
                      _stream())
        hs = [h]
        if taps is not None:
            taps["conv:fcomb.layers.0"] = h
        for nm in L.fcomb_mid:
            h = self.conv(flat, nm, h, None, True, None, taps, tape_bits=record)
            hs.append(h)
        return hs

    def _last_layer_rows(self, flat):
        """fcomb.last_layer for num_classes = K > 1 (no reference script uses it): the kernels take ONE class's CH weights as
        a contiguous row, the parameter is stored [CH][K] - per-class rows [K, CH] and biases [K] as small device copies."""
        c3 = self.L.convs["fcomb.last_layer"]
        K, CH = c3.cout, c3.cin
        w = flat[c3.w_off:c3.w_off + CH * K].view(CH, K).t().contiguous()
        b = flat[c3.b_off:c3.b_off + K].contiguous()
        return w, b

    def sample(self, flat, feat: Act, z: torch.Tensor, upper=0.9, lower=0.1, masking=False, want_logits=False,
               want_consensus=True):
        """N Monte-Carlo samples through Fcomb + consensus (K7).  z: [N, B, L] latent draws.
        Returns (y_mean [B,1,H,W], consensus [B,1,H,W] float, logits [N,B,1,H,W] or None)."""
        N, B, Ld = z.shape
        CH = self.L.nf[0]
        c1, c3, nmid, wmid, bmid, stride = self._fcomb_ptrs(flat)
        zb = self._empty(N, B, CH)
        _lib.call("punet_fcomb_zbias", z.data_ptr(), self._wptr(flat, c1), self._bptr(flat, c1), zb.data_ptr(), N * B, Ld, CH,
                  _stream())
        HW = feat.H * feat.W
        # split-plane arithmetic on the BF16 matrix cores in every x6-family mode (always six products: parity), fp32 MFMA in "f32"
        fn = "punet_fcomb_sample_x6_bf16s" if feat.t.dtype == torch.bfloat16 else \
            "punet_fcomb_sample_x6_f16s" if feat.t.dtype == torch.float16 else \
            ("punet_fcomb_sample_x6" if self.x6 and self.sampler_x6 else "punet_fcomb_sample")
        K = c3.cout
        if CH not in (32, 64):  # outside the fused sampler's domain: Fcomb layer by layer per sample, then the consensus kernel K8
            if feat.t.dtype != torch.float32:
                feat = Act(feat.t.float(), B, feat.H, feat.W, CH, relu_out=True)
            P = B * HW
            if K == 1:
                w3s, b3s = [self._wptr(flat, c3)], [self._bptr(flat, c3)]
            else:
                w3k, b3k = self._last_layer_rows(flat)
                w3s, b3s = [w3k[k].data_ptr() for k in range(K)], [b3k.data_ptr() + 4 * k for k in range(K)]
            lg = self._empty(K, N, B, feat.H, feat.W)
            for n in range(N):
                h = self._fcomb_hidden(flat, feat, zb[n].data_ptr())[-1]
                for k in range(K):
                    _lib.call("punet_dot_bias", h.ptr, w3s[k], b3s[k], lg[k, n].data_ptr(), P, CH, _stream())
            y = cz = None
            if want_consensus:
                y, cz = self._empty(K, B, feat.H, feat.W), self._empty(K, B, feat.H, feat.W)
                for k in range(K):
                    _lib.call("punet_consensus", lg[k].data_ptr(), N, P, float(upper), float(lower), int(masking), y[k].data_ptr(),
                              cz[k].data_ptr(), _stream())
                y, cz = y.permute(1, 0, 2, 3).contiguous(), cz.permute(1, 0, 2, 3).contiguous()
            return y, cz, lg.permute(1, 2, 0, 3, 4).contiguous() if want_logits else None
        if K == 1:
            w3s, b3s = [self._wptr(flat, c3)], [self._bptr(flat, c3)]
        else:  # one launch per class (the hidden layers are recomputed: a rarely used configuration, kept simple)
            w3k, b3k = self._last_layer_rows(flat)
            w3s, b3s = [w3k[k].data_ptr() for k in range(K)], [b3k.data_ptr() + 4 * k for k in range(K)]
        ys, czs, lgs = [], [], []
        for k in range(K):
            y = self._empty(B, 1, feat.H, feat.W) if want_consensus else None
            cz = self._empty(B, 1, feat.H, feat.W) if want_consensus else None
            logits = self._empty(N, B, 1, feat.H, feat.W) if want_logits else None
            _lib.call(fn, feat.ptr, self._wptr(flat, c1),
                      zb.data_ptr(), wmid, stride, bmid, stride,
                      w3s[k], b3s[k], y.data_ptr() if y is not None else None,
                      cz.data_ptr() if cz is not None else None, logits.data_ptr() if logits is not None else None, N, B, HW,
                      CH, nmid, float(upper), float(lower), int(masking), _stream())
            ys.append(y), czs.append(cz), lgs.append(logits)
        if K == 1:
            return ys[0], czs[0], lgs[0]
        return (torch.cat(ys, 1) if want_consensus else None, torch.cat(czs, 1) if want_consensus else None,
                torch.cat(lgs, 2) if want_logits else None)

    def reparam(self, muls: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
        """z = mu + exp(log_sigma) * eps for eps [N, B, L] (Normal.rsample; probabilistic_unet.py:302,349)."""
        N, B, Ld = eps.shape
        z = self._empty(N, B, Ld)
        _lib.call("punet_reparam", muls.data_ptr(), eps.data_ptr(), z.data_ptr(), N, B, Ld, _stream())
        return z

    def latent_split(self, muls: torch.Tensor):
        """(mu, sigma) [B, L] each of the Normal the reference stores (probabilistic_unet.py:136-141)."""
        B, Ld = muls.shape[0], muls.shape[1] // 2
        loc, scale = self._empty(B, Ld), self._empty(B, Ld)
        _lib.call("punet_latent_split", muls.data_ptr(), loc.data_ptr(), scale.data_ptr(), B, Ld, _stream())
        return loc, scale

    def kl_rows(self, muls_q: torch.Tensor, muls_p: torch.Tensor, z: Optional[torch.Tensor] = None) -> torch.Tensor:
        """kl_divergence(analytic = z is None) (probabilistic_unet.py:324-339) -> [B]."""
        B, Ld = muls_q.shape[0], muls_q.shape[1] // 2
        rows, mean = self._empty(B), self._empty(1)
        _lib.call("punet_latent_kl_rows", muls_q.data_ptr(), muls_p.data_ptr(), z.data_ptr() if z is not None else None,
                  rows.data_ptr(), mean.data_ptr(), B, Ld, _stream())
        return rows

    def align_distribution(self, y: torch.Tensor, source_distribution: torch.Tensor):
        """FixMatch distribution alignment (fixmatch_trainer.py:77-84) -> (aligned y, ratio [2]); no host sync."""
        y = y.contiguous()
        out, ratio = torch.empty_like(y), self._empty(2)
        cnt = self.ws.get("count", 2)
        _lib.call("punet_align_distribution", y.data_ptr(), source_distribution.data_ptr(), out.data_ptr(), ratio.data_ptr(),
                  cnt.data_ptr(), y.numel(), _stream())
        return out, ratio

    def consensus(self, logits: torch.Tensor, upper=0.9, lower=0.1, masking=False):
        """K8 on given logits [N, ...] (unit parity of mean_teacher_trainer.py:74-86)."""
        N = logits.shape[0]
        P = logits[0].numel()
        y = self._empty(*logits.shape[1:])
        z = self._empty(*logits.shape[1:])
        _lib.call("punet_consensus", logits.data_ptr(), N, P, float(upper), float(lower), int(masking), y.data_ptr(),
                  z.data_ptr(), _stream())
        return y, z

    # ---------------------------------------------------------------- training path
    def train_forward(self, flat, x_nhwc, segm, record=True, taps=None):
        """forward(patch, segm, training=True) (probabilistic_unet.py:285-293): posterior on [patch, segm], prior and U-Net
        features.  segm: [B,1,H,W] float.  With record the tapes for elbo_backward are kept."""
        B, H, W, C = x_nhwc.shape
        st = {"B": B, "H": H, "W": W, "recorded": bool(record)}
        self.prepare_forward_weights(flat)
        xin = Act(x_nhwc, B, H, W, C, needs_grad=False)
        K = self.L.num_classes
        seg_nhwc = segm if K == 1 else self.to_nhwc(segm)  # [B,1,H,W] is its own NHWC form
        xq_t = self._empty(B, H, W, C + K)  # torch.cat([patch, segm], 1) (probabilistic_unet.py:118), NHWC
        _lib.call("punet_concat_nhwc", x_nhwc.data_ptr(), C, seg_nhwc.data_ptr(), K, xq_t.data_ptr(), B * H * W, _stream())
        xq = Act(xq_t, B, H, W, C + K, needs_grad=False)
        tq = [] if record else None
        tp = [] if record else None
        tu = [] if record else None
        self._infer16 = self.act16 is not None and record  # the student pass stores its activations in 16 bits too (gradients fp32)
        try:
            muls_q, mean_q, enc_q = self.latent(flat, "posterior", xq, tq, taps)
            muls_p, mean_p, enc_p = self.latent(flat, "prior", xin, tp, taps)
            feat = self.unet(flat, xin, tu, taps)
        finally:
            self._infer16 = False
        st.update(muls_q=muls_q, muls_p=muls_p, mean_q=mean_q, mean_p=mean_p, enc_q=enc_q, enc_p=enc_p, feat=feat,
                  tapes=(tq, tp, tu))
        return st

    def elbo_head(self, flat, st, y_tgt, consm, eps, loss_kind="bce", beta=1.0, gscale=1.0, record=True, taps=None,
                  analytic_kl=True, posterior_mean=False):
        """elbo(segm, consm, analytic_kl, reconstruct_posterior_mean) (probabilistic_unet.py:341-371) on the state of
        train_forward: z = posterior.rsample() (:349), KL (:351-353), Fcomb on z - or on mu_q when posterior_mean (:356-358;
        the reference reads `.loc` of an Independent there, which raises AttributeError - the intended mean is used),
        criterion (:360-369).  y_tgt, consm: [B,1,H,W] float (consm may be None); eps: [B, L].  With record the
        d loss / d logit (scaled by gscale) is kept for elbo_backward."""
        L = self.L
        B, H, W = st["B"], st["H"], st["W"]
        Ld, CH = L.latent_dim, L.nf[0]
        record = bool(record and st["recorded"])
        st = dict(st)
        st.update(gscale=gscale, beta=beta)
        muls_q, muls_p, feat = st["muls_q"], st["muls_p"], st["feat"]
        eps3 = eps.reshape(1, B, Ld).contiguous()
        z = self.reparam(muls_q, eps3)  # [1, B, L]
        kl = self._empty(1)
        kl_rows = self._empty(B)
        _lib.call("punet_latent_kl_rows", muls_q.data_ptr(), muls_p.data_ptr(), None if analytic_kl else z.data_ptr(),
                  kl_rows.data_ptr(), kl.data_ptr(), B, Ld, _stream())
        zf = self.reparam(muls_q, torch.zeros_like(eps3)) if posterior_mean else z  # what Fcomb sees
        c1, c3 = L.convs["fcomb.layers.0"], L.convs["fcomb.last_layer"]
        zb = self._empty(B, CH)
        _lib.call("punet_fcomb_zbias", zf.data_ptr(), self._wptr(flat, c1), self._bptr(flat, c1), zb.data_ptr(), B, Ld, CH,
                  _stream())
        nmid = len(L.fcomb_mid)
        K = c3.cout
        if (record and loss_kind in ("bce", "dice") and taps is None and self.fuse_fcomb and K == 1
                and (loss_kind == "bce" or feat.t.dtype == torch.float32)
                and _lib.call("punet_fcomb_train_supported", CH, nmid, H * W)):
            # K7t: forward, criterion and the whole Fcomb backward in one kernel; the weight-gradient slabs wait in `ws` for
            # elbo_backward (a fresh buffer per pass: the joint trainers hold two passes' worth between forward and backward)
            cm_ = L.convs[L.fcomb_mid[0]] if nmid else None
            ws = self._empty(_lib.call("punet_fcomb_train_workspace_floats", B, H * W, CH, nmid))
            logits, dfeat, rec = self._empty(B, 1, H, W), self._empty(B, H, W, CH), self._empty(1)
            self._timed("fcomb_train", 2.0 * 3 * B * H * W * CH * (CH * (1 + nmid) + 1),
                        "punet_fcomb_train_fwd_bwd_dice" if loss_kind == "dice" else
                        {torch.bfloat16: "punet_fcomb_train_fwd_bwd_bf16s", torch.float16: "punet_fcomb_train_fwd_bwd_f16s",
                         torch.float32: "punet_fcomb_train_fwd_bwd"}[feat.t.dtype], feat.ptr,
                        self._wptr(flat, c1), zb.data_ptr(), self._wptr(flat, cm_) if nmid else None,
                        self._bptr(flat, cm_) if nmid else None, self._wptr(flat, c3), self._bptr(flat, c3), y_tgt.data_ptr(),
                        consm.data_ptr() if consm is not None else None, float(gscale), logits.data_ptr(), dfeat.data_ptr(),
                        rec.data_ptr(), ws.data_ptr(), B, H * W, CH, nmid, _stream())
            st.update(z=zf, z_posterior=z, eps=eps3, kl=kl, kl_rows=kl_rows, rec=rec, logits=logits, dlogit=None, hs=None,
                      loss_kind=loss_kind, kl_sampled=not analytic_kl, z_is_mean=bool(posterior_mean), recorded=record,
                      fused_fcomb=ws, dfeat=dfeat)
            return st
        if feat.t.dtype != torch.float32:  # 16-bit storage outside the fused kernel's domain: the per-op kernels read fp32
            f32 = Act(feat.t.float(), B, H, W, CH, relu_out=True)
            f32.bits = feat.bits
            st["feat_f32"] = f32
            feat = f32
        # Fcomb, single sample, activations materialised for the backward
        hs = self._fcomb_hidden(flat, feat, zb.data_ptr(), record, taps)
        h = hs[-1]
        P = B * H * W
        # num_classes = K > 1: the last layer, the criterion and their backward run in a class-major [K, B, H, W] layout (one
        # dot-product launch per class, the element-wise criterion over K * P elements); the facade permutes at the boundary
        if K == 1:
            logits = self._empty(B, 1, H, W)
            _lib.call("punet_dot_bias", h.ptr, self._wptr(flat, c3), self._bptr(flat, c3), logits.data_ptr(), P, CH, _stream())
            tgt_k, cm_k, lg_k = y_tgt, consm, logits
        else:
            w3k, b3k = self._last_layer_rows(flat)
            lg_k = self._empty(K, B, H, W)
            for k in range(K):
                _lib.call("punet_dot_bias", h.ptr, w3k[k].data_ptr(), b3k.data_ptr() + 4 * k, lg_k[k].data_ptr(), P, CH, _stream())
            tgt_k = y_tgt.permute(1, 0, 2, 3).contiguous()
            cm_k = consm.permute(1, 0, 2, 3).contiguous() if consm is not None else None
            logits = lg_k.permute(1, 0, 2, 3).contiguous()
        dlogit = torch.empty_like(lg_k) if record else None
        rec = self._empty(1)
        wsl = self.ws.get("loss", _lib.call("punet_loss_workspace_floats"))
        cm = cm_k.data_ptr() if cm_k is not None else None
        if loss_kind == "bce":
            _lib.call("punet_bce_fwd_bwd", lg_k.data_ptr(), tgt_k.data_ptr(), cm, float(gscale),
                      dlogit.data_ptr() if record else None, rec.data_ptr(), wsl.data_ptr(), K * P, _stream())
        elif loss_kind == "dice":
            # [recalled torch_em definition, parity unpinned] per channel 1 - 2 sum(p t) / max(sum(p^2) + sum(t^2), eps) over
            # batch and pixels, summed over the channels: one class = one contiguous block of the class-major layout
            if dlogit is None:
                dlogit = torch.empty_like(lg_k)
            if K == 1:
                _lib.call("punet_dice_fwd_bwd", logits.data_ptr(), y_tgt.data_ptr(), cm, float(gscale), dlogit.data_ptr(),
                          rec.data_ptr(), wsl.data_ptr(), P, _stream())
            else:
                per_class = self._empty(K)
                for k in range(K):
                    _lib.call("punet_dice_fwd_bwd", lg_k[k].data_ptr(), tgt_k[k].data_ptr(),
                              cm_k[k].data_ptr() if cm_k is not None else None, float(gscale), dlogit[k].data_ptr(),
                              per_class.data_ptr() + 4 * k, wsl.data_ptr(), P, _stream())
                rec = per_class.sum().reshape(1)  # K values (no reference script uses num_classes > 1: kept simple)
        else:
            raise ValueError(loss_kind)
        st.update(z=zf, z_posterior=z, eps=eps3, kl=kl, kl_rows=kl_rows, rec=rec, logits=logits, dlogit=dlogit, hs=hs,
                  loss_kind=loss_kind, kl_sampled=not analytic_kl, z_is_mean=bool(posterior_mean), recorded=record)
        return st

    def elbo_forward(self, flat, x_nhwc, y_tgt, consm, eps, loss_kind="bce", beta=1.0, gscale=1.0, record=True, taps=None):
        """forward(patch, segm, training=True) + elbo(segm, consm) with segm = y_tgt for both (every reference trainer)."""
        st = self.train_forward(flat, x_nhwc, y_tgt, record, taps)
        return self.elbo_head(flat, st, y_tgt, consm, eps, loss_kind, beta, gscale, record, taps)

    def _wgrad(self, gflat, c: ConvP, x: Act, ci_off: int, dz: torch.Tensor, with_bias: bool, accumulate: bool):
        B, H, W = x.B, x.H, x.W
        dw = gflat.data_ptr() + 4 * c.w_off
        db = gflat.data_ptr() + 4 * c.b_off
        if c.cin <= SMALL_CIN:
            ws = self.ws.get("wgrad", _lib.call("punet_small_wgrad_workspace_floats", c.cin, c.cout))
            _lib.call("punet_conv3x3_small_wgrad", x.ptr, c.cin, dz.data_ptr(), c.cout, dw, db, ws.data_ptr(), B, H, W,
                      int(accumulate), 1.0, _stream())
            return
        x6 = self.x6 and c.k == 3
        ns = _lib.call("punet_wgrad_x6_slices" if x6 else "punet_wgrad_slices", x.C, c.cout, B, H, W)
        ws = self.ws.get("wgrad", _lib.call("punet_wgrad_workspace_floats", x.C, c.cout, c.k, ns))
        if x6 and self.f16 and x.t.dtype == torch.float32:
            self._timed("conv3x3_wgrad", 2.0 * B * H * W * x.C * c.cout * 9, "punet_conv_wgrad_f16", x.ptr, x.C, dz.data_ptr(),
                        c.cout, dw, c.cin, ci_off, db if with_bias else None, ws.data_ptr(), ns, B, H, W, int(accumulate), 1.0,
                        self.f16, 0, _aptr(x.amax), _aptr(getattr(dz, "punet_amax", None)), _stream())
            return
        if x6 and x.t.dtype == torch.float16:  # "f16s": the stored operand is unscaled, the gradient scaled by its amax
            self._timed("conv3x3_wgrad", 2.0 * B * H * W * x.C * c.cout * 9, "punet_conv_wgrad_f16", x.ptr, x.C, dz.data_ptr(),
                        c.cout, dw, c.cin, ci_off, db if with_bias else None, ws.data_ptr(), ns, B, H, W, int(accumulate), 1.0,
                        1, 1, None, _aptr(getattr(dz, "punet_amax", None)), _stream())
            return
        if x6 and x.t.dtype == torch.bfloat16:
            self._timed("conv3x3_wgrad", 2.0 * B * H * W * x.C * c.cout * 9, "punet_conv_wgrad_bf16s", x.ptr, x.C, dz.data_ptr(),
                        c.cout, dw, c.cin, ci_off, db if with_bias else None, ws.data_ptr(), ns, B, H, W, int(accumulate), 1.0,
                        _stream())
            return
        if x6:
            self._timed("conv3x3_wgrad", 2.0 * B * H * W * x.C * c.cout * 9, "punet_conv_wgrad_x6", x.ptr, x.C, dz.data_ptr(),
                        c.cout, dw, c.cin, ci_off, db if with_bias else None, ws.data_ptr(), ns, B, H, W, int(accumulate), 1.0,
                        self.x6_opts, _stream())
            return
        self._timed(f"conv{c.k}x{c.k}_wgrad", 2.0 * B * H * W * x.C * c.cout * c.k * c.k, "punet_conv_wgrad", x.ptr, x.C,
                    dz.data_ptr(), c.cout, dw, c.cin, ci_off, db if with_bias else None, ws.data_ptr(), ns, B, H, W, c.k,
                    int(accumulate), 1.0, _stream())

    def _split_x6(self, flat, mode: int):
        """(Re)build the three-plane bf16 split of every 3x3 conv weight slab of `flat` (mode 0: forward orientation,
        mode 1: input-gradient orientation).  Weights change every optimizer step, so callers refresh per pass."""
        key = (flat.data_ptr(), mode)
        planes = self._x6.get(key)
        if planes is None:
            planes = torch.empty(3 * flat.numel(), dtype=torch.bfloat16, device=self.device)
            self._x6[key] = planes
        if self._x6_table is None:
            rows, units = [], 0
            for c in self.L.convs.values():
                if c.k == 3 and c.cin > SMALL_CIN:
                    self._x6_index[c.name] = len(rows)
                    rows.append([c.w_off, c.cin, c.cout, units])
                    units += c.w_numel // 8
            rows.append([0, 0, 0, units])
            self._x6_table = (torch.tensor(rows, dtype=torch.int64, device=self.device), len(rows) - 1, units)
        table, nconv, units = self._x6_table
        if self.f16:  # fp16 planes h (+ l) of w * 2^k, k from the slab's amax (one slot per slab, recomputed with the planes)
            wam = self._wamax.get(flat.data_ptr())
            fresh = wam is None
            if fresh:
                wam = self._wamax[flat.data_ptr()] = torch.zeros(nconv, dtype=torch.float32, device=self.device)
            # (the input-gradient orientation is split from the SAME weights as the forward planes of this step: their amax stands)
            _lib.call("punet_split_weights_f16_batch", flat.data_ptr(), planes.data_ptr(), table.data_ptr(), nconv, units, mode,
                      wam.data_ptr(), 1 if (mode == 0 or fresh) else 0, _stream())
            return planes
        _lib.call("punet_split_weights_x6_batch", flat.data_ptr(), planes.data_ptr(), table.data_ptr(), nconv, units, mode,
                  _stream())
        return planes

    def _fcomb_1x1(self):
        """(conv, rows) of the Fcomb 1x1 layers the split-plane kernel serves: the feature rows of layers.0 (its last
        latent_dim rows are the z term, folded into a per-image bias) and the middle layers."""
        L = self.L
        if L.kind != "punet" or L.nf[0] % 8:
            return []
        return [(L.convs["fcomb.layers.0"], L.nf[0])] + [(L.convs[nm], L.nf[0]) for nm in L.fcomb_mid]

    def _split_x6_1x1(self, flat, mode: int):
        key = (flat.data_ptr(), mode)
        d = self._x6_1x1.get(key)
        if d is None:
            d = {c.name: torch.empty(3 * rows * c.cout, dtype=torch.bfloat16, device=self.device) for c, rows in self._fcomb_1x1()}
            self._x6_1x1[key] = d
        for c, rows in self._fcomb_1x1():
            _lib.call("punet_split_weights_x6", self._wptr(flat, c), d[c.name].data_ptr(), rows * c.cout, rows, c.cout, 1, mode,
                      _stream())
        return d

    def prepare_forward_weights(self, flat):
        # every pass starts here: its amax slots come from a pool allocated INSIDE the pass (kept alive by the views handed out).
        # A pool carried over from an earlier pass would be baked into a hipGraph captured now - not re-zeroed by the replay (slots
        # accumulate with atomic max) and freed as soon as later eager passes have used it up.
        self._amax_pool = None
        if self.x6:
            self._split_x6(flat, 0)
            self._split_x6_1x1(flat, 0)
            self._split_up(flat, 0)

    def prepare_dgrad_weights(self, flat):
        """Bring every conv weight slab into forward-kernel orientation for the input-gradient pass
        (wt[k*k-1-t][cout][cin] = w[t][cin][cout]); ~0.1 ms per step, after which dgrad IS the forward kernel."""
        if self.x6:
            self._split_x6(flat, 1)
            self._split_x6_1x1(flat, 1)
            self._split_up(flat, 1)
        if self.wT is None or self.wT.numel() != flat.numel():
            self.wT = torch.empty_like(flat)
        served = {c.name for c, _ in self._fcomb_1x1()} if self.x6 else set()
        for c in self.L.convs.values():
            if (self.x6 and c.k == 3) or c.name in served:
                continue  # served by the x6 planes
            if c.cin > SMALL_CIN and c.cin % 4 == 0 and c.cout % 8 == 0:
                _lib.call("punet_flip_transpose_weights", self._wptr(flat, c), self.wT.data_ptr() + 4 * c.w_off, c.cin, c.cout,
                          c.k, _stream())

    def _dgrad(self, flat, c: ConvP, dz, cin_off: int, cin: int, dx, mask_ptr, acc: int, B: int, H: int, W: int, bits=None):
        """dx[..., cin_off : cin_off + cin] of conv c given dz (epilogue: ReLU gate by `mask` - or by its bit form `bits`
        where the kernel takes it -, accumulate)."""
        flops = 2.0 * B * H * W * cin * c.cout * c.k * c.k
        if self.f16 and c.k == 3:
            planes = self._x6[(flat.data_ptr(), 1)]
            use_bits = bits is not None and mask_ptr is not None
            if getattr(dx, "punet_amax", None) is None:
                dx.punet_amax = self._slot()  # (accumulating into dx: the slot keeps max(previous, new total) - a bound)
            self._timed("conv3x3_dgrad", flops, "punet_conv3x3_f16", dz.data_ptr(), c.cout, None, 0,
                        planes.data_ptr() + 2 * 3 * c.w_off, c.w_numel, c.cin, cin_off, None, 0, dx.data_ptr(), cin,
                        None if use_bits else mask_ptr, bits.data_ptr() if use_bits else None, None, None, acc, 0, B, H, W, self.f16,
                        _aptr(getattr(dz, "punet_amax", None)), None, self._wamax_ptr(flat, c), dx.punet_amax.data_ptr(), _stream())
        elif self.x6 and c.k == 3:
            planes = self._x6[(flat.data_ptr(), 1)]
            use_bits = bits is not None and mask_ptr is not None
            self._timed("conv3x3_dgrad", flops, "punet_conv3x3_x6_gated", dz.data_ptr(), c.cout, None, 0,
                        planes.data_ptr() + 2 * 3 * c.w_off, c.w_numel, c.cin, cin_off, None, 0, dx.data_ptr(), cin,
                        None if use_bits else mask_ptr, bits.data_ptr() if use_bits else None, None, acc, 0, B, H, W, self.x6_opts,
                        _stream())
        elif self.x6 and c.k == 1 and c.name in self._x6_1x1.get((flat.data_ptr(), 1), {}) and cin_off == 0:
            planes = self._x6_1x1[(flat.data_ptr(), 1)][c.name]  # built for the first `cin` rows of the slab
            use_bits = bits is not None and mask_ptr is not None
            self._timed("conv1x1_dgrad", flops, "punet_conv1x1_x6_gated", dz.data_ptr(), c.cout, None, 0, planes.data_ptr(),
                        cin * c.cout, cin, 0, None, 0, dx.data_ptr(), cin, None if use_bits else mask_ptr,
                        bits.data_ptr() if use_bits else None, None, acc, 0, B, H, W, _stream())
        elif c.cin % 4 == 0:
            self._timed(f"conv{c.k}x{c.k}_dgrad", flops, "punet_conv_dgrad_t", dz.data_ptr(), c.cout,
                        self.wT.data_ptr() + 4 * c.w_off, c.cin, cin_off, cin, dx.data_ptr(), mask_ptr, acc, B, H, W, c.k,
                        _stream())
        else:  # Fcomb's first layer (cin = C + latent_dim is not a multiple of 4): transposed-read variant
            self._timed(f"conv{c.k}x{c.k}_dgrad", flops, "punet_conv_dgrad", dz.data_ptr(), c.cout, self._wptr(flat, c), c.cin,
                        cin_off, cin, dx.data_ptr(), mask_ptr, acc, B, H, W, c.k, _stream())

    def _grad_into(self, gm: dict, a: Act):
        """(buffer, accumulate flag) for the gradient of activation a."""
        g = gm.get(id(a))
        if g is None:
            g = torch.empty(a.t.shape, dtype=torch.float32, device=self.device)  # gradients are fp32 whatever the storage of a
            gm[id(a)] = g
            return g, 0
        return g, 1

    def _upconv_backward(self, flat, gflat, c: ConvP, x: Act, dz: torch.Tensor, gm: dict, accumulate: bool):
        """Everything the `up` source of y = conv3x3(cat[up(x), bridge]) receives from dz = d loss / d (pre-activation of y), in the
        low-resolution form (csrc/upconv.hip): the nine adjoint-interpolated maps of dz ONCE, then the weight gradient of the `up`
        rows (x^T . maps, rows [0, cup) of every tap) and d loss / d x (one 1x1 input gradient with w9, gated by x's ReLU)."""
        B, h, w, cup = x.B, x.H, x.W, x.C
        d = self._up_entry(flat, c.name)
        dzt = self._empty(B, h, w, 9 * c.cout)
        _lib.call("punet_upconv_gather_bwd", dz.data_ptr(), dzt.data_ptr(), B, h, w, c.cout, _stream())
        ns = _lib.call("punet_upconv_wgrad_slices", cup, c.cout, B, h, w)
        ws = self.ws.get("wgrad", _lib.call("punet_upconv_wgrad_workspace_floats", cup, c.cout, ns))
        self._timed("conv1x1_wgrad", 2.0 * B * h * w * cup * 9 * c.cout, "punet_upconv_wgrad_x6", x.ptr, cup, dzt.data_ptr(), c.cout,
                    gflat.data_ptr() + 4 * c.w_off, c.cin, ws.data_ptr(), ns, B, h, w, int(accumulate), 1.0, _stream())
        if not x.needs_grad:
            return
        g, acc = self._grad_into(gm, x)
        use_bits = x.bits is not None and x.relu_out
        self._timed("conv1x1_dgrad", 2.0 * B * h * w * cup * 9 * c.cout, "punet_conv1x1_x6_gated", dzt.data_ptr(), 9 * c.cout, None, 0,
                    d["p9d"].data_ptr(), cup * 9 * c.cout, cup, 0, None, 0, g.data_ptr(), cup,
                    x.ptr if (x.relu_out and not use_bits) else None, x.bits.data_ptr() if use_bits else None, None, acc, 0, B, h, w,
                    _stream())

    def _backward_tape(self, flat, gflat, tape, gm: dict, accumulate: bool):
        # as-written up-block convs whose `up` source qualifies for K5x's input gradient: up-sampled tensor -> its low-res source
        up_src, skip_up = {}, set()
        if self.upconv:
            for op in tape:
                if op[0] == "up":
                    up_src[id(op[2])] = op[1]
            for op in tape:  # keep only those consumed by ONE eligible concat conv
                if op[0] == "conv" and op[3] is not None and id(op[2]) in up_src and not self._upconv_ok(op[1], up_src[id(op[2])], op[3]):
                    del up_src[id(op[2])]
        for op in reversed(tape):
            if op[0] == "conv":
                _, c, x0, x1, y = op
                dz = gm.pop(id(y))
                lowres = up_src.pop(id(x0), None) if x1 is not None else None  # x0 = up(lowres): K5x's backward of the `up` rows
                if lowres is not None:
                    self._upconv_backward(flat, gflat, c, lowres, dz, gm, accumulate)
                    self._wgrad(gflat, c, x1, x0.C, dz, True, accumulate)
                    skip_up.add(id(x0))
                else:
                    self._wgrad(gflat, c, x0, 0, dz, True, accumulate)
                    if x1 is not None:
                        self._wgrad(gflat, c, x1, x0.C, dz, False, accumulate)
                off = 0
                for src in (x0, x1):
                    if src is None:
                        continue
                    if src is x0 and lowres is not None:
                        pass  # (done above: gradient of the low-resolution source, the `up` op's backward is skipped)
                    elif src.needs_grad:
                        g, acc = self._grad_into(gm, src)
                        self._dgrad(flat, c, dz, off, src.C, g, src.ptr if src.relu_out else None, acc, src.B, src.H, src.W,
                                    bits=src.bits)
                    off += src.C
            elif op[0] == "upconv":  # K5x forward: everything that concerns the `up` source in the low-resolution form (the up-sampled
                _, c, x, bridge, y = op  # tensor never exists), the bridge's gradients by the ordinary kernels
                dz = gm.pop(id(y))
                self._upconv_backward(flat, gflat, c, x, dz, gm, accumulate)
                self._wgrad(gflat, c, bridge, x.C, dz, True, accumulate)
                if bridge.needs_grad:
                    g, acc = self._grad_into(gm, bridge)
                    self._dgrad(flat, c, dz, x.C, bridge.C, g, bridge.ptr if bridge.relu_out else None, acc, bridge.B, bridge.H, bridge.W,
                                bits=bridge.bits)
            elif op[0] == "pool":
                _, x, y = op
                dy = gm.pop(id(y))
                g, acc = self._grad_into(gm, x)
                bits = x.bits.data_ptr() if (x.relu_out and x.bits is not None) else None
                _lib.call("punet_pool2x2_bwd_gated", dy.data_ptr(), x.ptr if (x.relu_out and bits is None) else None, bits,
                          g.data_ptr(), x.B, x.H, x.W, x.C, acc, _stream())
                if self.f16:  # |dx| <= |dy| / 4 (+ what was there): a bound from the slots, no pass over the tensor
                    prev = getattr(g, "punet_amax", None) if acc else None
                    sl = prev if prev is not None else self._slot()
                    _lib.call("punet_amax_bound", _aptr(getattr(dy, "punet_amax", None)), 0.25, _aptr(prev), 1.0, sl.data_ptr(), _stream())
                    g.punet_amax = sl
            else:
                _, x, y = op
                if id(y) in skip_up:  # its consumer's backward has produced x's gradient in the low-resolution form already
                    continue
                dy = gm.pop(id(y))
                assert id(x) not in gm
                g, _ = self._grad_into(gm, x)
                bits = x.bits.data_ptr() if (x.relu_out and x.bits is not None) else None
                _lib.call("punet_upsample2x_bwd_gated", dy.data_ptr(), x.ptr if (x.relu_out and bits is None) else None, bits,
                          g.data_ptr(), x.B, x.H, x.W, x.C, _stream())
                if self.f16:  # the transposed x2 interpolation sums weights of at most 2 per dimension: |dx| <= 4 max |dy|
                    g.punet_amax = self._slot()
                    _lib.call("punet_amax_bound", _aptr(getattr(dy, "punet_amax", None)), 4.0, None, 0.0, g.punet_amax.data_ptr(), _stream())

    def elbo_backward(self, flat, gflat, st, accumulate=False, after_unet=None):
        """Gradient of gscale * (reconstruction_loss + beta * kl) w.r.t. every parameter into gflat
        (= d(-elbo)/d theta, probabilistic_unet.py:371), written or accumulated.  `after_unet()` is called once the
        U-Net's slice of gflat (the first L.unet_numel floats) is final - the data-parallel step starts that slice's
        all-reduce there, under the encoders' backward."""
        L = self.L
        if not st.get("recorded", True):
            raise RuntimeError("elbo_backward needs a state recorded with grad mode on (forward() ran under no_grad)")
        B, H, W = st["B"], st["H"], st["W"]
        Ld, CH = L.latent_dim, L.nf[0]
        acc = int(accumulate)
        self.prepare_dgrad_weights(flat)
        c1, c3 = L.convs["fcomb.layers.0"], L.convs["fcomb.last_layer"]
        feat: Act = st["feat"]
        if st.get("fused_fcomb") is not None:  # K7t ran in elbo_head: only the slab reduction and the z term are left
            nmid = len(L.fcomb_mid)
            cm_ = L.convs[L.fcomb_mid[0]] if nmid else None
            g = gflat.data_ptr()
            S = self._empty(B, CH)
            wsc = self.ws.get("chan", _lib.call("punet_chan_sum_workspace_floats", B, H * W // 32, CH))
            _lib.call("punet_fcomb_train_reduce", st["fused_fcomb"].data_ptr(), g + 4 * c1.w_off, c1.cin, g + 4 * c1.b_off,
                      g + 4 * cm_.w_off if nmid else None, g + 4 * cm_.b_off if nmid else None, g + 4 * c3.w_off,
                      g + 4 * c3.b_off, S.data_ptr(), wsc.data_ptr(), B, H * W, CH, nmid, acc, _stream())
            dz = self._empty(B, Ld)
            _lib.call("punet_fcomb_z_bwd", S.data_ptr(), st["z"].data_ptr(), self._wptr(flat, c1), g + 4 * c1.w_off,
                      dz.data_ptr(), B, Ld, CH, acc, _stream())
            gm_u = {id(feat): st["dfeat"]}
        else:
            dz, gm_u = self._fcomb_backward_ops(flat, gflat, st, accumulate)
        if self.f16:  # d loss / d features comes from the Fcomb kernels (fp32 / six-product arithmetic): they publish no amax
            dfeat_t = gm_u[id(feat)]
            dfeat_t.punet_amax = self._amax_of(dfeat_t)
        self._backward_tape(flat, gflat, st["tapes"][2], gm_u, accumulate)
        if after_unet is not None:
            after_unet()
        self._latent_backward(flat, gflat, st, dz, accumulate)

    def _fcomb_backward_ops(self, flat, gflat, st, accumulate):
        """The per-op Fcomb backward (Dice criterion, more than one middle layer, taps requested): returns (dz [B, L], the
        gradient map holding d loss / d features)."""
        L = self.L
        B, H, W = st["B"], st["H"], st["W"]
        Ld, CH = L.latent_dim, L.nf[0]
        acc = int(accumulate)
        P = B * H * W
        c1, c3 = L.convs["fcomb.layers.0"], L.convs["fcomb.last_layer"]
        hs: List[Act] = st["hs"]
        feat: Act = st["feat"]
        fread: Act = st.get("feat_f32", feat)  # what the per-op kernels read (an fp32 copy when the features are stored as bf16)
        # last layer C -> 1
        dz_h = torch.empty_like(hs[-1].t)
        ws = self.ws.get("dot", _lib.call("punet_dot_bwd_workspace_floats", CH))
        K = c3.cout
        if K == 1:
            _lib.call("punet_dot_bwd", hs[-1].ptr, st["dlogit"].data_ptr(), self._wptr(flat, c3), dz_h.data_ptr(),
                      gflat.data_ptr() + 4 * c3.w_off, gflat.data_ptr() + 4 * c3.b_off, ws.data_ptr(), P, CH, acc, _stream())
        else:  # per class (dlogit is class-major [K, B, H, W]); the K contributions to d h and the [CH][K] weight layout by ATen
            w3k, _ = self._last_layer_rows(flat)
            dw3k, db3k, part = self._empty(K, CH), self._empty(K), torch.empty_like(dz_h)
            for k in range(K):
                _lib.call("punet_dot_bwd", hs[-1].ptr, st["dlogit"][k].data_ptr(), w3k[k].data_ptr(),
                          (dz_h if k == 0 else part).data_ptr(), dw3k[k].data_ptr(), db3k.data_ptr() + 4 * k, ws.data_ptr(), P, CH,
                          0, _stream())
                if k:
                    dz_h += part
            gw = gflat[c3.w_off:c3.w_off + CH * K].view(CH, K)
            gb = gflat[c3.b_off:c3.b_off + K]
            if accumulate:
                gw += dw3k.t()
                gb += db3k
            else:
                gw.copy_(dw3k.t())
                gb.copy_(db3k)
        # middle 1x1 layers, last to first
        for k in range(len(L.fcomb_mid) - 1, -1, -1):
            c = L.convs[L.fcomb_mid[k]]
            xin = hs[k]
            self._wgrad(gflat, c, xin, 0, dz_h, True, accumulate)
            dprev = torch.empty_like(xin.t)
            self._dgrad(flat, c, dz_h, 0, xin.C, dprev, xin.ptr, 0, B, H, W, bits=xin.bits)
            dz_h = dprev
        # first layer: feature rows by wgrad (+ bias), z rows and dz from the per-image column sums
        self._wgrad(gflat, c1, fread, 0, dz_h, True, accumulate)
        S = self._empty(B, CH)
        wsc = self.ws.get("chan", _lib.call("punet_chan_sum_workspace_floats", B, H * W, CH))
        _lib.call("punet_chan_sum", dz_h.data_ptr(), S.data_ptr(), wsc.data_ptr(), B, H * W, CH, 1.0, _stream())
        dz = self._empty(B, Ld)
        _lib.call("punet_fcomb_z_bwd", S.data_ptr(), st["z"].data_ptr(), self._wptr(flat, c1), gflat.data_ptr() + 4 * c1.w_off,
                  dz.data_ptr(), B, Ld, CH, acc, _stream())
        gm_u = {}
        dfeat, _ = self._grad_into(gm_u, feat)
        self._dgrad(flat, c1, dz_h, 0, CH, dfeat, fread.ptr, 0, B, H, W, bits=feat.bits)
        return dz, gm_u

    def _latent_backward(self, flat, gflat, st, dz, accumulate):
        """KL + reparameterisation gradients into the two latent heads, then the encoders' backward."""
        L = self.L
        B = st["B"]
        Ld = L.latent_dim
        acc = int(accumulate)
        tq, tp = st["tapes"][0], st["tapes"][1]
        # latent heads: KL + reparameterisation
        dq = self._empty(B, 2 * Ld)
        dp = self._empty(B, 2 * Ld)
        coef = float(st["gscale"]) * float(st["beta"]) / B
        _lib.call("punet_latent_kl_bwd_ex", st["muls_q"].data_ptr(), st["muls_p"].data_ptr(), st["eps"].data_ptr(), dz.data_ptr(),
                  coef, int(st.get("kl_sampled", False)), int(st.get("z_is_mean", False)), dq.data_ptr(), dp.data_ptr(), B, Ld,
                  _stream())
        for net, d, mean, enc, tape in (("posterior", dq, st["mean_q"], st["enc_q"], tq),
                                       ("prior", dp, st["mean_p"], st["enc_p"], tp)):
            c = L.convs[f"{net}.conv_layer"]
            gm = {}
            denc, _ = self._grad_into(gm, enc)
            dmean = self._empty(B, enc.C)
            if enc.t.dtype != torch.float32:
                _lib.call("punet_latent_head_bwd_bits", d.data_ptr(), mean.data_ptr(), self._wptr(flat, c), enc.bits.data_ptr(),
                          gflat.data_ptr() + 4 * c.w_off, gflat.data_ptr() + 4 * c.b_off, dmean.data_ptr(), denc.data_ptr(), B, enc.H,
                          enc.W, enc.C, c.cout, acc, _stream())
            else:
                _lib.call("punet_latent_head_bwd", d.data_ptr(), mean.data_ptr(), self._wptr(flat, c), enc.ptr,
                          gflat.data_ptr() + 4 * c.w_off, gflat.data_ptr() + 4 * c.b_off, dmean.data_ptr(), denc.data_ptr(), B,
                          enc.H * enc.W, enc.C, c.cout, acc, _stream())
            if self.f16:
                denc.punet_amax = self._amax_of(denc)
            self._backward_tape(flat, gflat, tape, gm, accumulate)

    # ---------------------------------------------------------------- regulariser / optimiser
    def l2_reg(self, flat) -> Tuple[torch.Tensor, torch.Tensor]:
        """sum of un-squared 2-norms over posterior, prior, fcomb.layers (utils.py:32-40) -> (reg [1], norms [T])."""
        partial = self.ws.get("l2p", self.l2_nchunks)
        norms = self._empty(self.l2_ntensors)
        reg = self._empty(1)
        _lib.call("punet_l2_reg", flat.data_ptr(), self.l2_chunks.data_ptr(), self.l2_nchunks, self.l2_ntensors,
                  partial.data_ptr(), norms.data_ptr(), reg.data_ptr(), _stream())
        return reg, norms

    def l2_reg_add_grad(self, flat, gflat, norms, coef: float):
        _lib.call("punet_l2_reg_add_grad", flat.data_ptr(), gflat.data_ptr(), self.l2_chunks.data_ptr(), self.l2_nchunks,
                  norms.data_ptr(), float(coef), _stream())

    def adam_ema(self, flat, gflat, m, v, teacher, step: int, lr: float, betas=(0.9, 0.999), eps=1e-8, gscale=1.0,
                 ema_momentum=0.999, do_adam=True):
        _lib.call("punet_adam_ema", flat.data_ptr(), gflat.data_ptr() if gflat is not None else None,
                  m.data_ptr() if m is not None else None, v.data_ptr() if v is not None else None,
                  teacher.data_ptr() if teacher is not None else None, flat.numel(), float(gscale), float(lr),
                  float(betas[0]), float(betas[1]), float(eps), int(step), float(ema_momentum), int(do_adam), _stream())

    def adam_ema_dev(self, flat, gflat, m, v, teacher, hstate, hyper, betas=(0.9, 0.999), eps=1e-8):
        """Adam (+ EMA) with the step-dependent scalars computed on the device from `hstate` (graph-capturable)."""
        _lib.call("punet_adam_hyper", hstate.data_ptr(), hyper.data_ptr(), _stream())
        _lib.call("punet_adam_ema_dev", flat.data_ptr(), gflat.data_ptr(), m.data_ptr(), v.data_ptr(),
                  teacher.data_ptr() if teacher is not None else None, flat.numel(), hyper.data_ptr(), float(betas[0]),
                  float(betas[1]), float(eps), _stream())

    def dice_score(self, seg: torch.Tensor, gt: torch.Tensor, threshold_seg=None, threshold_gt=None) -> float:
        """dice_score (my_utils/util.py:17-44) on device tensors; returns a host float (one sync)."""
        sums = self._empty(3)
        ws = self.ws.get("loss", _lib.call("punet_loss_workspace_floats"))
        seg = seg.contiguous()
        gt = gt.contiguous()
        assert seg.numel() == gt.numel()
        _lib.call("punet_dice_score_sums", seg.data_ptr(), gt.data_ptr(), float(threshold_seg or 0.0),
                  int(threshold_seg is not None), float(threshold_gt or 0.0), int(threshold_gt is not None),
                  sums.data_ptr(), ws.data_ptr(), seg.numel(), _stream())
        a, b, c = [float(v) for v in sums.tolist()]
        return (2.0 * a) / (b + c + 1e-7)
