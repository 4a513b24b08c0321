"""Device check of common.h's GPU-only primitive bodies.  f2u / f2i are bare v_cvt_u32_f32 / v_cvt_i32_f32 on the
GPU and guarded C in the emulator; row_shr / row_shr0 are DPP row shifts (bound_ctrl), bcast_byte3 is v_perm, lane_value /
wave_read are v_readlane, wave_shfl is ds_bpermute, mask_rank_below is v_mbcnt, the wave scans are DPP ladders -- the
emulator parity suite runs the portable twins, so only a device run sees a toolchain or ISA difference in what ships."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["f2u", "f2i", "row_shr", "row_shr0", "bcast_byte3", "lane_value", "wave_shfl", "wave_read", "mask_rank_below",
         "wave_incl_scan_u32", "wave_incl_scan_max_u32"]


def test_gpu_primitive_bodies_equal_their_portable_forms(built):
    path = os.path.join(ROOT, "tests", "device_checks", "libvello_devcheck.so")
    assert os.path.exists(path), "tests/device_checks/libvello_devcheck.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(path)
    lib.vello_devcheck_primitives.argtypes = [ctypes.c_uint32, ctypes.c_void_p]
    lib.vello_devcheck_primitives.restype = ctypes.c_int
    bad = np.zeros(len(NAMES), dtype=np.uint64)
    # stride 1: all 2^32 f32 bit patterns (a fraction of a second on the device) + NaN / inf / +-2^31 / 2^32 edge values
    rc = lib.vello_devcheck_primitives(1, bad.ctypes.data)
    assert rc == 0, f"hip error {rc}"
    wrong = {n: int(b) for n, b in zip(NAMES, bad) if b}
    assert not wrong, f"GPU bodies differ from the portable forms: {wrong}"
