import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests.emu_lib import emu_library_path  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the emulator's heaviest cases (-m 'not gpu and not slow' is the quick CPU suite; -m 'not gpu' runs everything)")


# The emulator cases that take more than ~5 s each (pytest --durations, round 6): together two thirds of the CPU suite's ten minutes.
SLOW = ("test_emu_path_count_long_lines", "test_emu_fuzz_target_sizes_and_long_scenes", "test_emu_persistent_resolver_over_many_frames",
        "test_emu_flatten_kernel_sets", "test_emu_fuzz_extreme_values", "test_emu_front_fusion[5]", "test_emu_front_fusion[6]",
        "test_emu_fine_slices[thousands_of_segments]", "test_emu_fine_slices[fuzz]", "test_emu_fine_slices[blend_grid]",
        "test_emu_fuzz_auto_grow_from_tiny_pools", "test_emu_thousands_of_segments_in_one_tile", "test_emu_fuzz_whole_api",
        "test_emu_reference_catalogue_second_batch", "test_emu_stroked_line_kernel", "test_emu_clip_stage_partitioned",
        "test_emu_path_count_both_forms", "test_emu_reference_catalogue", "test_emu_pipeline", "test_emu_tiger", "test_emu_stroke_styles",
        "test_emu_many_draw_objects", "test_emu_pools_exactly_full")


def pytest_collection_modifyitems(config, items):
    for it in items:
        name = it.name
        if any(name == s or name.startswith(s + "[") or name.startswith(s) and "[" not in s for s in SLOW):
            it.add_marker(pytest.mark.slow)


@pytest.fixture(scope="session")
def built():
    """Builds the in-tree libraries once (no-op when they are up to date)."""
    import __graft_entry__

    libs = [os.path.join(ROOT, "vello_amd", "lib", "libvello_hip.so"), os.path.join(ROOT, "oracle", "libvello_oracle.so"),
            os.path.join(ROOT, "tests", "simt_emu", "libvello_emu.so"), os.path.join(ROOT, "tests", "device_checks", "libvello_devcheck.so")]
    if not all(os.path.exists(p) for p in libs):
        __graft_entry__.build()
    return True


@pytest.fixture()
def emu_engine(built):
    """Engine bound to the SIMT-emulated build of the kernel sources (CPU-only CI of kernel logic)."""
    import vello_amd
    import vello_amd._lib as L

    L._use_library(emu_library_path())
    try:
        yield vello_amd.Engine()
    finally:
        L._use_library(None)


@pytest.fixture()
def gpu_engine(built):
    import vello_amd
    import vello_amd._lib as L

    L._use_library(None)
    try:
        return vello_amd.Engine(device=0)
    except RuntimeError as e:  # vello_hip_create: VELLO_HIP_E_NO_DEVICE (-3): no MI355X here -- the -m gpu tests do not apply
        # exactly VELLO_HIP_E_NO_DEVICE, as Engine words it ("vello_hip_create failed (-3): no usable HIP device (...)"): any
        # other failure of create() is a real one and must fail the test, not skip it
        if str(e).startswith("vello_hip_create failed (-3):") and "no usable HIP device" in str(e):
            pytest.skip("no HIP device: " + str(e))
        raise
