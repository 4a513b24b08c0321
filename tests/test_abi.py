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


def test_cpp_example_builds_and_fails_loudly_without_a_gpu(built, tmp_path):
    # examples/hello_gradient.cpp: the host C++ mirror end to end (Scene -> Renderer::render_to_texture).  Here (no GPU)
    # it must build, link against the product library and refuse to run; linked against the emulated kernels (test
    # infrastructure) it must draw the scene.
    import subprocess

    import numpy as np
    import torch

    exe = str(tmp_path / "hello")
    src = os.path.join(ROOT, "examples", "hello_gradient.cpp")
    lib = os.path.join(ROOT, "vello_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O1", src, "-I", ROOT, "-L", lib, "-lvello_hip", f"-Wl,-rpath,{lib}", "-o", exe], check=True)
    if not torch.cuda.is_available():
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "no CPU fallback" in r.stderr
    emu = os.path.join(ROOT, "tests", "simt_emu")
    subprocess.run(["g++", "-std=c++17", "-O1", src, "-I", ROOT, "-L", emu, "-lvello_emu", f"-Wl,-rpath,{emu}", "-o", exe + "_emu"], check=True)
    out = str(tmp_path / "out.ppm")
    r = subprocess.run([exe + "_emu", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    data = open(out, "rb").read()
    header = b"P6\n512 512\n255\n"
    img = np.frombuffer(data[len(header):], dtype=np.uint8).reshape(512, 512, 3)
    assert tuple(img[10, 10]) == (26, 26, 31)            # base colour
    assert tuple(img[256, 300]) != tuple(img[256, 200])  # the sweep gradient varies around the centre
    assert tuple(img[256, 66]) == tuple(img[256, 446])   # the stroke-clipped ring, symmetric
