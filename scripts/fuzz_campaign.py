"""Long differential fuzz runs on the CPU: the kernel sources (SIMT-emulator build) against the oracle, beyond the slices the
test suites carry.  Prints the failing seeds only.

    python scripts/fuzz_campaign.py api      LO HI     whole scene API, 128 x 128 (tests: test_emu_fuzz_whole_api)
    python scripts/fuzz_campaign.py sizes    LO HI     awkward target sizes, up to 700 operations per scene
    python scripts/fuzz_campaign.py pools    LO HI     every pool started at 64 elements, auto-grow
    python scripts/fuzz_campaign.py extreme  LO HI     12 % of the points from {+-1e6 ... +-3e38, +-inf, NaN}
    FUZZ_GPU=1 ...             the product library on the GPU instead of the emulator build
    FUZZ_STROKE_KERNEL=1 ...   with flatten's stroked-line kernel forced on (VELLO_HIP_DEBUG_STROKE_KERNEL)
    FUZZ_FINE_SLICES=1 ...   every tile through the sliced path of fine (VELLO_HIP_DEBUG_FINE_SLICES)
    FUZZ_FLATTEN=coop|alone ...  flatten's heavy list by the wave-cooperative walk / by every lane on its own, whatever the engine would pick
    FUZZ_IN_FLIGHT=n ...     vello_hip_set_frames_in_flight(n): from 2 on flatten runs its stroke workgroups as a kernel of their own

Round 1 ran api 0-43500, sizes 0-6000, pools 0-4000, extreme 0-358 (some extreme seeds emit tens of millions of lines and take
minutes each on the emulator) (see DESIGN.md section 4 for what they found).
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vello_amd  # noqa: E402
import vello_amd._lib as L  # noqa: E402

ON_GPU = os.environ.get("FUZZ_GPU") == "1"              # the product library on a real MI355X instead of the emulator build
STROKE_KERNEL = os.environ.get("FUZZ_STROKE_KERNEL") == "1"  # VELLO_HIP_DEBUG_STROKE_KERNEL: flatten's stroked-line kernel for every scene
IN_FLIGHT = int(os.environ.get("FUZZ_IN_FLIGHT", "1"))
FINE_SLICES = os.environ.get("FUZZ_FINE_SLICES") == "1"  # VELLO_HIP_DEBUG_FINE_SLICES: slices of 4 fills for every tile
# round 5: which kernels take flatten's heavy list -- "coop" (VELLO_HIP_DEBUG_FLATTEN_COOP: the wave-cooperative walk), "alone"
# (VELLO_HIP_DEBUG_FLATTEN_ALONE: every lane on its own); unset: the engine's own choice (small scenes: the cooperative walk)
FLATTEN = os.environ.get("FUZZ_FLATTEN", "")
if not ON_GPU:
    L._use_library(os.path.join(ROOT, "tests", "simt_emu", "libvello_emu.so"))
elif os.environ.get("VELLO_AB_LIB"):  # another build of the product library: ab_tmp/libvello_hip_<X>.so (is a finding this round's?)
    L._use_library(os.path.join(ROOT, "ab_tmp", "libvello_hip_%s.so" % os.environ["VELLO_AB_LIB"]))
from oracle.oracle import Oracle  # noqa: E402
from tests.parity import compare_frame  # noqa: E402
from vello_amd import AaConfig  # noqa: E402
from workloads.fuzz import fuzz_scene  # noqa: E402

AAS = [AaConfig.Area, AaConfig.Msaa8, AaConfig.Msaa16]
BASES = [0xFF000000, 0xFFFFFFFF, 0x00000000, 0x80FF8040]
TINY = {"lines": 64, "binning": 64, "tile": 64, "seg_counts": 64, "segments": 64, "blend": 16, "ptcl": 64 * 4 + 64}


def one(mode, seed, eng):
    if mode == "api":
        w = h = 128
        scene, aa, base, kw = fuzz_scene(seed), AAS[seed % 3], BASES[seed % 4], {}
    elif mode == "sizes":
        w, h = [(333, 205), (97, 530), (512, 512), (16, 16), (1, 1), (700, 40)][seed % 6]
        scene = fuzz_scene(seed, size=max(w, h, 8), n_ops=[40, 700, 300][seed % 3])
        aa, base, kw = AAS[(seed // 2) % 3], BASES[seed % 4], {}
    elif mode == "pools":
        eng = vello_amd.Engine(capacities=TINY)
        eng.set_auto_grow(True)
        eng.set_frames_in_flight(IN_FLIGHT)
        eng.set_debug_flags(stroke_kernel=STROKE_KERNEL, fine_slices=FINE_SLICES, flatten_coop=FLATTEN == "coop", flatten_alone=FLATTEN == "alone")
        w, h = [(128, 128), (300, 200), (64, 64)][seed % 3]
        scene = fuzz_scene(seed, size=max(w, h), n_ops=[40, 300][seed % 2])
        aa, base, kw = AAS[(seed // 2) % 3], 0xFF203040, {}
    elif mode == "extreme":
        w = h = 128
        scene, aa, base = fuzz_scene(seed, n_ops=14, extreme=True), AAS[seed % 3], 0xFF000000
        # (scenes that emit > 8 M lines overflow this oracle; crossing indices beyond 16 bits collide in the slot diff of the back half)
        kw = {"min_agree": None, "oracle": Oracle(capacity_scale=4, auto_grow=True), "back_half": False}
    else:
        raise SystemExit(__doc__)
    r = vello_amd.Resolver().resolve(scene)
    compare_frame(eng, r.packed, r.layout, w, h, base, aa, f"fuzz_{mode}_{seed}", tol=1 if aa == AaConfig.Area else 0, resolved=r,
                  order_sensitive=True, **kw)


def main():
    if len(sys.argv) != 4:
        raise SystemExit(__doc__)
    mode, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    eng = vello_amd.Engine()
    eng.set_auto_grow(True)
    eng.set_frames_in_flight(IN_FLIGHT)
    eng.set_debug_flags(stroke_kernel=STROKE_KERNEL, fine_slices=FINE_SLICES, flatten_coop=FLATTEN == "coop", flatten_alone=FLATTEN == "alone")
    bad, t0 = [], time.time()
    t_said = t0
    for seed in range(lo, hi):
        try:
            one(mode, seed, eng)
        except Exception as e:  # noqa: BLE001 -- a campaign reports and carries on
            bad.append(seed)
            print("SEED", seed, type(e).__name__, str(e)[:300], flush=True)
        if (seed - lo) % 500 == 499 or time.time() - t_said > 30.0:  # (a run cut off by a timeout has said how far it came)
            t_said = time.time()
            print("at", seed + 1, "bad", bad, round(time.time() - t0, 1), "s", flush=True)
    print("done", mode, hi - lo, "bad", bad, round(time.time() - t0, 1), "s")


if __name__ == "__main__":
    main()
