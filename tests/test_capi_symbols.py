"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/punet_hip.h declares
(no compute calls without a GPU)."""
import os
import re

import pytest


def _header_symbols():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "include", "punet_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(punet_[a-z0-9_]+)\s*\(", txt)))


def test_build_and_exports():
    import __graft_entry__ as g
    g.build()
    from probabilistic_domain_adaptation_amd import _lib
    lib = _lib.load()
    hdr = _header_symbols()
    assert len(hdr) >= 30
    for s in hdr:
        assert hasattr(lib, s), f"{s} declared in include/punet_hip.h but not exported"
    assert set(hdr) == set(_lib.declared_symbols()), set(hdr) ^ set(_lib.declared_symbols())
    assert b"gfx950" in _lib.call("punet_build_info")


def test_bad_arguments_are_reported_not_thrown():
    from probabilistic_domain_adaptation_amd import _lib
    _lib.load()
    with pytest.raises(RuntimeError, match="ksize"):
        _lib.call("punet_conv_fwd", 1, 8, None, 0, 1, None, 0, 1, 8, 1, 8, 8, 5, 1, None)
    with pytest.raises(RuntimeError, match="multiples of 8"):
        _lib.call("punet_conv_fwd", 1, 6, None, 0, 1, None, 0, 1, 8, 1, 8, 8, 3, 1, None)


def test_product_path_refuses_cpu():
    import torch
    from probabilistic_domain_adaptation_amd.engine import ParamLayout, PUNetEngine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        PUNetEngine(ParamLayout(1, 1, [32, 32, 32, 32]), torch.device("cpu"))


def test_layout_matches_reference_state_dict():
    from oracle import punet_oracle as O
    from probabilistic_domain_adaptation_amd.engine import ParamLayout
    for nf, c in (([64, 128, 256, 512], 1), ([32, 64, 96, 128], 3)):
        lay = ParamLayout(c, 1, nf, 6, 3)
        spec = O.param_spec(c, 1, nf, 6, 3)
        assert [(n, s) for n, _, _, s in lay.entries] == [(n, tuple(s)) for n, s in spec.items()]
    assert ParamLayout(1, 1, [64, 128, 256, 512]).n_params == 27349593


def test_package_exports_cover_the_reference_packages():
    """Every name the reference's hot-path packages export (prob_utils/my_models/__init__.py:1-2, my_trainer/__init__.py:1-6,
    my_predictions/__init__.py:2, my_utils/util.py) that is in scope imports from this package under the same name."""
    import importlib
    want = {"my_models": ["ProbabilisticUnet", "l2_regularisation", "clean_folder"],
            "my_trainer": ["MeanTeacherTrainer", "MeanTeacherLogger", "AdaMatchLogger", "AdaMatchTrainer", "AdaMTTrainer", "AdaMTLogger",
                           "FixMatchLogger", "FixMatchTrainer", "PUNetLogger", "PUNetTrainer", "PseudoLogger", "PseudoTrainerPUNet"],
            "my_predictions": ["punet_prediction", "punet_pseudo_prediction"],
            "my_utils": ["dice_score", "my_standardize_torch", "DummyLoss"]}
    for mod, names in want.items():
        m = importlib.import_module("probabilistic_domain_adaptation_amd." + mod)
        for n in names:
            assert hasattr(m, n), (mod, n)


def test_every_entry_point_survives_all_zero_arguments():
    """The ABI's error behaviour (include/punet_hip.h: "nothing throws across the ABI"): every exported function called with null
    pointers and zero sizes returns - the int-returning ones with an error code and a message in punet_last_error(), the size
    helpers with a number - instead of dereferencing, dividing by zero or launching.  One child process so that a crash is a test
    failure with the culprit's name, not the end of the test run.  No GPU is needed: argument validation precedes every HIP call."""
    import subprocess
    import sys
    code = r"""
import ctypes as C, sys
from probabilistic_domain_adaptation_amd import _lib
_lib.load()
for n in sorted(_lib._SIGS):
    res, args = _lib._SIGS[n]
    vals = []
    for a in args:
        if a in (C.c_float, C.c_double):
            vals.append(0.0)
        elif a in (C.c_int, C.c_long, C.c_size_t, C.c_uint, C.c_longlong):
            vals.append(0)
        elif isinstance(a, type) and issubclass(a, C._CFuncPtr):
            vals.append(a())   # a null callback
        else:
            vals.append(None)  # pointers, handles
    print("calling", n, flush=True)
    r = _lib._fn[n](*vals)
    if res is C.c_int and n not in _lib._NO_WRAP:
        assert r != 0 and _lib.last_error(), (n, r)
print("all returned")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    last = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("calling")][-1:]
    assert p.returncode == 0 and "all returned" in p.stdout, (p.returncode, last, p.stdout[-600:])
