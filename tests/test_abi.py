"""CPU: the C-ABI shared library loads and exports every symbol include/rgm.h declares (no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "rgm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rgm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from rgm import native
    lib = ctypes.CDLL(native.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/rgm.h but not exported"
    assert native.lib.rgm_version() >= 100


def test_product_fails_loudly_without_a_gpu_tensor():
    import pytest
    import torch
    from rgm.native import RgmError
    from guided_diffusion.dit import DiT_models
    m = DiT_models["DiTRotary-XS/8-cls"](input_size=[128, 16], in_channels=4, num_classes=16)
    with pytest.raises(RgmError):
        m(torch.zeros(1, 4, 128, 16), torch.zeros(1, dtype=torch.long))
