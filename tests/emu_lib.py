"""Where the SIMT-emulated build of the kernel sources lives.  scripts/emu_variant_check.sh points the emulator suites at a scratch
build with other compile-time constants through VELLO_EMU_LIBRARY instead of overwriting the tree's copy."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def emu_library_path():
    return os.environ.get("VELLO_EMU_LIBRARY") or os.path.join(ROOT, "tests", "simt_emu", "libvello_emu.so")
