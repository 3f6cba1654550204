"""Weak / strong view generation on the GPU.

Counterpart of the transform chains the reference runs on DataLoader workers (`my_weak_augmentations` /
`my_strong_augmentations`, LIVECell/livecell_fm.py:43-67 and MitoEM/common.py:50-68): `my_standardize_torch`
(prob_utils/my_utils/util.py:9-14) followed by randomly applied `GaussianBlur`, `AdditiveGaussianNoise`, `RandomContrast`.
The random DECISIONS are drawn here on the host, per item, exactly where the reference draws them (apply-or-not with
probability p, kernel size, sigma, noise scale, alpha); the arithmetic runs in `csrc/augment.hip` (K13).  The N(0,1)
noise field itself comes from torch's device generator (plumbing).

The three transforms live in the un-vendored `torch_em` / `torchvision`: their definitions are recalled, not pinned
(DESIGN.md section 4); the parameter ranges are the reference's own call sites.
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

from .. import _lib

MAX_KSIZE = 63


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check(x):
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        raise RuntimeError("view generation needs a float32 [B,C,H,W] tensor on the ROCm device (no CPU fallback)")
    return x.contiguous()


def standardize_stats(x: torch.Tensor) -> torch.Tensor:
    """[B,2] = per item (mean, Bessel-corrected std) over C*H*W."""
    x = _check(x)
    B = x.shape[0]
    stats = torch.empty(B, 2, device=x.device)
    ws = torch.empty(_lib.call("punet_standardize_workspace_doubles", B), dtype=torch.float64, device=x.device)
    _lib.call("punet_standardize_stats", x.data_ptr(), stats.data_ptr(), ws.data_ptr(), B, x[0].numel(), _stream())
    return stats


def apply_view(x: torch.Tensor, params, noise: Optional[torch.Tensor] = None, contrast_mean: float = 0.0,
               standardize: bool = True, eps: float = 1e-7) -> torch.Tensor:
    """Deterministic part: params[b] = (kernel size, sigma, noise scale, alpha) -> the view of every item."""
    x = _check(x)
    B, C, H, W = x.shape
    p = np.asarray(params, dtype=np.float32).reshape(B, 4)
    ks = p[:, 0].astype(np.int64)
    if np.any(ks < 1) or np.any(ks % 2 == 0) or np.any(ks > MAX_KSIZE):
        raise ValueError(f"kernel sizes must be odd and in [1, {MAX_KSIZE}]: {ks.tolist()}")
    if np.any((ks > 1) & ~(p[:, 1] > 0)):
        raise ValueError("sigma must be > 0 where a blur is applied")
    if int(ks.max()) // 2 >= min(H, W):
        raise ValueError("reflect padding needs blur radius < image size")
    if noise is not None:
        noise = _check(noise)
        if noise.shape != x.shape:
            raise ValueError(f"noise shape {tuple(noise.shape)} != {tuple(x.shape)}")
    stats = standardize_stats(x) if standardize else None
    # through PINNED memory, asynchronously: a copy from pageable memory blocks the host until the stream has reached it, i.e. until
    # the previous training step has finished - the loader then builds every batch's views behind the step instead of ahead of it
    # (bench.py --workload fit: 0.98 -> of the bare step's rate); the caching host allocator keeps the block alive until the copy ran
    pd = torch.from_numpy(np.ascontiguousarray(p)).pin_memory().to(x.device, non_blocking=True)
    tmp = torch.empty_like(x)
    out = torch.empty_like(x)
    _lib.call("punet_augment_view", x.data_ptr(), stats.data_ptr() if stats is not None else None,
              noise.data_ptr() if noise is not None else None, pd.data_ptr(), int(ks.max()), float(contrast_mean), float(eps),
              tmp.data_ptr(), out.data_ptr(), B, C, H, W, _stream())
    return out


def my_standardize_torch(tensor: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """util.py:9-14 for every item of a [B,C,H,W] batch (the reference applies it per sample in the loader)."""
    B = tensor.shape[0]
    ident = np.tile(np.array([1.0, 1.0, 0.0, 1.0], np.float32), (B, 1))
    return apply_view(tensor, ident, None, 0.0, True, eps)


@dataclass
class ViewAugmenter:
    """One transform chain: standardise, then blur / noise / contrast, each applied with its own probability."""
    p_blur: float
    p_noise: float
    p_contrast: float
    ksize: Tuple[int, int] = (2, 24)            # torch_em GaussianBlur default: size = 2 * (randint(lo, hi) // 2) + 1
    sigma: Tuple[float, float] = (0.0, 5.0)     # torch_em GaussianBlur default
    noise_scale: Tuple[float, float] = (0.0, 0.3)
    alpha: Tuple[float, float] = (0.5, 2.0)
    contrast_mean: float = 0.0
    seed: Optional[int] = None

    def __post_init__(self):
        self.rng = np.random.default_rng(self.seed)
        self.gen = None

    def draw(self, B: int) -> np.ndarray:
        """[B,4] host-side decisions (kernel size, sigma, noise scale, alpha); identity entries where not applied."""
        p = np.tile(np.array([1.0, 1.0, 0.0, 1.0], np.float32), (B, 1))
        for b in range(B):
            if self.rng.random() < self.p_blur:
                p[b, 0] = 2 * (int(self.rng.integers(self.ksize[0], self.ksize[1])) // 2) + 1
                p[b, 1] = max(self.rng.uniform(self.sigma[0], self.sigma[1]), 1e-3)
            if self.rng.random() < self.p_noise:
                p[b, 2] = self.rng.uniform(self.noise_scale[0], self.noise_scale[1])
            if self.rng.random() < self.p_contrast:
                p[b, 3] = self.rng.uniform(self.alpha[0], self.alpha[1])
        return p

    def __call__(self, raw: torch.Tensor) -> torch.Tensor:
        raw = _check(raw)
        p = self.draw(raw.shape[0])
        noise = None
        if np.any(p[:, 2] != 0):
            if self.gen is None:
                self.gen = torch.Generator(device=raw.device)
                self.gen.manual_seed(int(self.rng.integers(0, 2 ** 31 - 1)))
            noise = torch.randn(raw.shape, device=raw.device, generator=self.gen)
        return apply_view(raw, p, noise, self.contrast_mean, True)


def my_weak_augmentations(p: float = 0.25, seed=None) -> ViewAugmenter:
    """livecell_fm.py:43-54 / MitoEM/common.py:50-57 (identical)."""
    return ViewAugmenter(p_blur=p, p_noise=p, p_contrast=0.0, noise_scale=(0.0, 0.15), seed=seed)


def my_strong_augmentations(p: Optional[float] = None, dataset: str = "livecell", seed=None) -> ViewAugmenter:
    """livecell_fm.py:57-67 (p = 0.9) or MitoEM/common.py:60-68 (p = 0.5, noise with p/2)."""
    if dataset == "livecell":
        p = 0.9 if p is None else p
        return ViewAugmenter(p_blur=p, p_noise=p, p_contrast=p, sigma=(1.0, 4.0), noise_scale=(0.1, 0.35), alpha=(0.33, 3.0),
                             contrast_mean=0.0, seed=seed)
    if dataset == "mitoem":
        p = 0.5 if p is None else p
        return ViewAugmenter(p_blur=p, p_noise=p / 2, p_contrast=p, sigma=(0.6, 3.0), noise_scale=(0.05, 0.25),
                             alpha=(0.33, 3.0), contrast_mean=0.0, seed=seed)
    raise ValueError(f"unknown dataset {dataset!r}")


class DualViewLoader:
    """Wraps a loader of (raw, y) into the (x, x1, x2, y) batches the self-training trainers consume
    (mean_teacher_trainer.py:101), building both views on the device."""

    def __init__(self, loader, weak: ViewAugmenter, strong: ViewAugmenter, device="cuda"):
        self.loader, self.weak, self.strong, self.device = loader, weak, strong, torch.device(device)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for raw, y in self.loader:
            raw = raw.to(self.device, torch.float32, non_blocking=True)
            yield my_standardize_torch(raw), self.weak(raw), self.strong(raw), y.to(self.device, non_blocking=True)
