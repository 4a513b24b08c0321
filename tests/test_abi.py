"""The C-ABI library loads without a GPU and exports every symbol include/vello_hip.h declares;
create() fails loudly (no CPU fallback).  CPU only."""
import ctypes
import os
import re

import pytest

import vello_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vello_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vello_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(built):
    lib = ctypes.CDLL(vello_amd.library_path())
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/vello_hip.h but not exported"


def test_struct_sizes_match_reference_layouts(built):
    from vello_amd._lib import Bump, Capacities, LayoutStruct, RenderParamsStruct

    assert ctypes.sizeof(LayoutStruct) == 40 and ctypes.sizeof(Bump) == 32
    assert ctypes.sizeof(RenderParamsStruct) == 16 and ctypes.sizeof(Capacities) == 28


def test_no_cpu_fallback_without_gpu(built):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(vello_amd.VelloHipError, match="no CPU fallback"):
        vello_amd.Renderer()
    with pytest.raises(vello_amd.VelloHipError):
        vello_amd.Engine()


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vello_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "libvello_oracle" not in text and "import oracle" not in text and "from oracle" not in text, f
                assert "simt_emu" not in text or f == "_lib.py", f


def test_mask_luts_match_oracle(built):
    import numpy as np

    from oracle import oracle as O

    lib = vello_amd.load_library()
    l8 = np.zeros(1024, np.uint8)
    l16 = np.zeros(8192, np.uint8)
    lib.vello_hip_make_mask_lut(l8.ctypes.data)
    lib.vello_hip_make_mask_lut_16(l16.ctypes.data)
    assert np.array_equal(l8, O.make_mask_lut()) and np.array_equal(l16, O.make_mask_lut_16())
