"""CPU: host-side decision drawing of the view augmenters and the oracle's view restatement (no GPU calls)."""
import numpy as np
import torch

from oracle import punet_oracle as O
from probabilistic_domain_adaptation_amd.my_utils import augment


def test_draw_is_seeded_and_in_range():
    a = augment.my_strong_augmentations(seed=7).draw(256)
    b = augment.my_strong_augmentations(seed=7).draw(256)
    assert np.array_equal(a, b)
    ks = a[:, 0].astype(int)
    assert np.all(ks % 2 == 1) and ks.min() >= 1 and ks.max() <= 23
    blurred = ks > 1
    assert 0.8 < blurred.mean() <= 1.0                       # p = 0.9
    assert np.all((a[blurred, 1] >= 1.0) & (a[blurred, 1] <= 4.0))
    noisy = a[:, 2] != 0
    assert np.all((a[noisy, 2] >= 0.1) & (a[noisy, 2] <= 0.35))
    con = a[:, 3] != 1
    assert np.all((a[con, 3] >= 0.33) & (a[con, 3] <= 3.0))
    w = augment.my_weak_augmentations(seed=1).draw(512)
    assert np.all(w[:, 3] == 1.0) and 0.15 < (w[:, 0] > 1).mean() < 0.35
    m = augment.my_strong_augmentations(dataset="mitoem", seed=2).draw(512)
    assert 0.15 < (m[:, 2] != 0).mean() < 0.35               # noise applied with p/2 = 0.25


def test_oracle_view_properties():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 1, 40, 48, generator=g) * 9 + 4
    ident = [[1, 1.0, 0.0, 1.0]] * 2
    v = O.augment_view(x, ident, None)
    for b in range(2):
        assert abs(v[b].mean().item()) < 1e-5 and abs(v[b].std().item() - 1) < 1e-5
    taps = O.gaussian_taps(23, 4.0)
    assert abs(taps.sum().item() - 1) < 1e-6 and torch.equal(taps, taps.flip(0))
    c = torch.full((1, 1, 32, 32), 2.5)
    assert (O.augment_view(c, [[9, 2.0, 0.0, 1.0]], None, standardize=False) - 2.5).abs().max() < 1e-6


def test_no_cpu_fallback():
    import pytest
    with pytest.raises(RuntimeError):
        augment.my_standardize_torch(torch.zeros(1, 1, 8, 8))
