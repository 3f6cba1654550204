"""CPU: the flat parameter layout (host logic, no kernel call) over the constructor space - names, order and logical shapes equal
the reference's state dict as restated by the oracle's param_spec (pinned against the live reference by gen_golden.py), and the
configurations the HIP path does not take are refused with a ValueError instead of failing later."""
import pytest
import torch

from oracle import punet_oracle as O
from probabilistic_domain_adaptation_amd.engine import ParamLayout


@pytest.mark.parametrize("C,K,nf,latent,nfc", [
    (1, 1, [64, 128, 256, 512], 6, 3),      # every script
    (1, 1, [32, 64, 128, 192], 6, 4),       # the reference's defaults
    (3, 1, [32, 32, 64, 64, 96], 6, 2),     # RGB, five levels
    (2, 2, [32, 32, 64, 64], 6, 3),         # two classes
    (5, 2, [16, 32, 48], 2, 3),             # seven channels into the posterior, three levels, 16 first filters
    (1, 1, [128, 64, 64, 72], 6, 5),
])
def test_layout_equals_reference_state_dict(C, K, nf, latent, nfc):
    spec = O.param_spec(C, K, nf, latent, nfc)
    lay = ParamLayout(C, K, nf, latent, nfc)
    flat = torch.zeros(lay.numel)
    views = lay.views(flat)
    assert list(views.keys()) == list(spec.keys())
    for k, v in views.items():
        assert tuple(v.shape) == tuple(spec[k]), k
    assert sum(v.numel() for v in views.values()) <= lay.numel  # (alignment pads between the tensors)
    # every view aliases the flat buffer: writing through the view is visible in it
    last = list(views.values())[-1]
    last.fill_(1.0)
    assert flat.sum().item() == last.numel()


@pytest.mark.parametrize("C,K,nf,msg", [
    (1, 1, [48, 64, 96, 128], "power of two"),   # first-layer kernels: num_filters[0] / 4 must divide the block
    (1, 1, [32, 60, 64, 64], "multiples of 8"),
    (6, 2, [32, 32, 64, 64], "<= 7"),            # the posterior would see 8 channels
    (1, 0, [32, 32, 64, 64], "num_classes"),
])
def test_unsupported_configurations_raise(C, K, nf, msg):
    with pytest.raises(ValueError, match=msg):
        ParamLayout(C, K, nf, 6, 3)
