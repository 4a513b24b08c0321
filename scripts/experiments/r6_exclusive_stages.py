"""Round 6: stages that run for ONE frame at a time while frames are in flight (VELLO_HIP_DEBUG_EXCLUSIVE_SHIFT: the stage's launches wait
for the same stage of the frame enqueued before).  Two kernels of the same kind share the chip worst (both issue-bound, or both
bandwidth-bound: DESIGN 6.3's batching estimate); does keeping k_fine / k_path_count / ... exclusive spread the frames' phases for good?
d2 (WORKLOAD=...), frames/s with four in flight per set of exclusive stages, alternating with none.  (profiles/r06_exclusive_stages.txt)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bench  # noqa: E402
import vello_amd  # noqa: E402
from vello_amd.renderer import STAGES  # noqa: E402


def main():
    wl = bench.Workload(os.environ.get("WORKLOAD", "d2"), 0)
    W, H = wl.width, wl.height
    torch.zeros(1, device="cuda").add_(1)
    torch.cuda.synchronize()
    engine = vello_amd.Engine(device=0, capacities=wl.caps)
    engine.upload_scene(wl.packed, wl.layout)
    lanes = int(os.environ.get("LANES", "4"))
    engine.set_frames_in_flight(lanes)
    ring = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(lanes)]
    torch.cuda.synchronize()

    def fps(names, n):
        flags = 0
        for nm in names:
            flags |= 1 << (8 + STAGES.index(nm))
        engine._check(engine._lib.vello_hip_set_debug_flags(engine._h, flags), "flags")
        for i in range(3 * lanes):
            engine.render_resident(W, H, bench.BASE_COLOR, wl.aa, out=ring[i % lanes])
        engine.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            engine.render_resident(W, H, bench.BASE_COLOR, wl.aa, out=ring[i % lanes])
            if i >= lanes - 1:
                engine.sync_frame(lanes - 1)
        rc = engine.sync()
        torch.cuda.synchronize()
        assert rc == 0
        return n / (time.perf_counter() - t0)

    fps([], 40)
    sets = [[], ["fine"], ["path_count"], ["fine", "path_count"], ["fine", "path_count", "flatten"], ["fine", "path_count", "path_tiling", "coarse"],
            ["fine", "path_count", "flatten", "coarse", "path_tiling", "backdrop"], ["flatten"], ["coarse"]]
    n = int(os.environ.get("N", "400"))
    for rep in range(3):
        for names in sets:
            print(f"{'+'.join(names) or 'none':60s} {fps(names, n):8.1f} frames/s  (20 steps: {fps(names, 20):8.1f})", flush=True)
    engine._check(engine._lib.vello_hip_set_debug_flags(engine._h, 0), "flags")


if __name__ == "__main__":
    main()
