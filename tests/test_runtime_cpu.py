"""CPU: the host-only half of the step-level native runtime (csrc/runtime.hip) - its flat parameter layout must be the
ParamLayout of engine.py (= the reference's state-dict order) for every configuration it accepts, its arena
sizing is a dry walk of the graph (no device needed), and configurations outside its domain are refused."""
import ctypes as C

import pytest

from probabilistic_domain_adaptation_amd import _lib
from probabilistic_domain_adaptation_amd.engine import ParamLayout


def _create(ic, k, nf, latent, nfc):
    _lib.load()
    cfg = _lib.PunetRtConfig(ic, k, len(nf), latent, nfc, (C.c_int * 8)(*(list(nf) + [0] * (8 - len(nf)))))
    h = C.c_void_p()
    _lib.call("punet_rt_create", C.byref(cfg), C.byref(h))
    return h


@pytest.mark.parametrize("ic,nf,latent,nfc", [(1, [64, 128, 256, 512], 6, 3), (3, [32, 64, 128, 192], 6, 3),
                                               (1, [32, 32, 64, 64, 96], 2, 2), (2, [64, 40, 72], 6, 3), (6, [32, 48], 4, 3)])
def test_runtime_layout_is_the_param_layout(ic, nf, latent, nfc):
    lay = ParamLayout(ic, 1, nf, latent, nfc)
    h = _create(ic, 1, nf, latent, nfc)
    try:
        assert _lib.call("punet_rt_param_floats", h) == lay.numel
        assert _lib.call("punet_rt_unet_floats", h) == lay.unet_numel
        assert _lib.call("punet_rt_num_entries", h) == len(lay.entries)
        for i, (nm, off, num, shape) in enumerate(lay.entries):
            name = C.create_string_buffer(128)
            o, k, sh = C.c_long(), C.c_long(), (C.c_int * 4)()
            _lib.call("punet_rt_param_entry", h, i, name, 128, C.byref(o), C.byref(k), C.byref(sh))
            got = (name.value.decode(), o.value, k.value, tuple(sh) if sh[1] else (sh[0],))
            assert got == (nm, off, num, tuple(shape))
        # arena sizing = a dry walk of the call's graph: grows with the batch, training needs more than inference
        a1, a2 = _lib.call("punet_rt_arena_bytes", h, 1, 64, 64, 4, 1), _lib.call("punet_rt_arena_bytes", h, 2, 64, 64, 4, 1)
        i1 = _lib.call("punet_rt_arena_bytes", h, 1, 64, 64, 4, 0)
        assert 0 < i1 < a1 < a2
        assert _lib.call("punet_rt_tables_bytes", h) > 0
    finally:
        _lib.call("punet_rt_destroy", h)


@pytest.mark.parametrize("ic,k,nf,latent,nfc", [(1, 2, [32, 64], 6, 3), (1, 1, [16, 32], 6, 3), (1, 1, [32, 64], 6, 4), (7, 1, [32, 64], 6, 3),
                                                 (1, 1, [32, 60], 6, 3)])
def test_runtime_refuses_what_it_does_not_cover(ic, k, nf, latent, nfc):
    with pytest.raises(RuntimeError):
        _create(ic, k, nf, latent, nfc)


def test_supported_mirrors_every_precondition_of_punet_rt_create():
    """runtime.supported() decides whether the fused steps hand the graph walk to the native runtime; a layout it
    accepts but punet_rt_create refuses would raise instead of taking the Python engine (latent_dim = 65 did)."""
    from probabilistic_domain_adaptation_amd.runtime import supported
    cases = []
    for ic in (1, 3, 6):
        for nf in ([64, 128, 256, 512], [32, 64], [32, 40, 72], [16, 32], [128, 256], [64, 128, 256, 512, 512, 512, 512, 512]):
            for latent in (1, 6, 64, 65, 200):
                for nfc in (2, 3, 4):
                    cases.append((ic, nf, latent, nfc))
    assert any(lat == 65 for _, _, lat, _ in cases)
    n_ok = 0
    for ic, nf, latent, nfc in cases:
        try:
            lay = ParamLayout(ic, 1, nf, latent, nfc)
        except ValueError:
            continue  # outside the whole HIP path's constructor space
        try:
            h = _create(ic, 1, nf, latent, nfc)
            _lib.call("punet_rt_destroy", h)
            created = True
        except RuntimeError:
            created = False
        assert supported(lay) == created, (ic, nf, latent, nfc, supported(lay), created)
        n_ok += created
    assert n_ok > 10
