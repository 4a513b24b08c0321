"""The product's own fp64 sin/cos (vello_amd/csrc/engine/fp64_math.h) against libm, on the host.

The header is plain IEEE fp64 arithmetic with explicit fma, so a g++ build of it (through the SIMT emulator's
hip_runtime.h shim, -ffp-contract=off) computes the same bits as the gfx950 build.  The flattener's contract is
"fp64 value rounded once to f32" (the oracle calls libm for it): the two may differ only when the exact value lies
within ~2^-55 of an f32 rounding boundary, i.e. for ~2^-28 of arguments, so a 20 M sweep expects 0 and tolerates 2.
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = tmp_path / "fp64_math_check"
    subprocess.run(
        ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mfma",
         "-I", os.path.join(ROOT, "tests", "simt_emu"), "-I", os.path.join(ROOT, "vello_amd", "csrc", "engine"),
         os.path.join(ROOT, "tests", "fp64_math_check.cpp"), "-o", str(exe), "-lm"], check=True)
    return exe


def test_sincos_matches_libm_after_f32_rounding(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([str(exe), "20000000"], check=True, capture_output=True, text=True, timeout=300).stdout.split()
    n, mis_s, mis_c, worst_s, worst_c = int(out[0]), int(out[1]), int(out[2]), float(out[3]), float(out[4])
    assert n == 20000000
    assert mis_s <= 2 and mis_c <= 2, out
    assert worst_s < 2.0 and worst_c < 2.0, out   # fp64 ulps; measured 1.41


def test_pow_matches_libm_after_f32_rounding(tmp_path):
    # f64::pow_pos (|u|^(2/3) of the inverse integral, and other exponents up to 8): log-uniform 1e-6 ... 1e3, around 1,
    # every positive f32 incl. denormals.  A 600 M sweep measured 0 mismatches and 1.0 fp64 ulp.
    exe = _build(tmp_path)
    out = subprocess.run([str(exe), "40000000", "pow"], check=True, capture_output=True, text=True, timeout=300).stdout.split()
    n, mis, worst = int(out[0]), int(out[1]), float(out[2])
    assert n == 40000000
    assert mis <= 2, out
    assert worst < 2.0, out
