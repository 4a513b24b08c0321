"""Round 6: what does each stage cost WITH FRAMES IN FLIGHT?  (profiles/r06_stage_marginal.txt)

A kernel's duration under overlap (rocprofv3's time line) says how long it shared the chip, not what it took from the other frames.
Here: the frame period with four frames in flight when vello_hip_render_resident stops after stage k (the measurement seam
VELLO_HIP_DEBUG_LAST_STAGE_SHIFT), k = every stage; the difference between consecutive rows is what stage k adds to the period.  The
same one frame at a time beside it.  WORKLOAD=d2|mmark|r1mix|tiger.
"""
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
    ring = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()

    def period(lanes, last, n):
        engine.set_frames_in_flight(lanes)
        engine._check(engine._lib.vello_hip_set_debug_flags(engine._h, ((last + 1) << 24) if last is not None else 0), "flags")
        for i in range(8):
            engine.render_resident(W, H, bench.BASE_COLOR, wl.aa, out=ring[i % lanes])
        engine.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            engine.render_resident(W, H, bench.BASE_COLOR, wl.aa, out=ring[i % lanes])
            if i >= lanes - 1:
                engine.sync_frame(lanes - 1)
        engine.sync()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    # full frames first (the scene's launch heuristics are learnt from finished frames)
    period(1, None, 10)
    period(4, None, 20)
    n = int(os.environ.get("N", "300"))
    print(f"# {wl.key}: frame period in us when render_resident stops after stage k; delta = what the stage adds")
    print(f"{'stage':14s} {'4 in flight':>12s} {'delta':>8s} {'one at a time':>14s} {'delta':>8s}")
    p4_prev = p1_prev = 0.0
    for k, name in enumerate(STAGES):
        p4 = min(period(4, k, n) for _ in range(2))
        p1 = min(period(1, k, n) for _ in range(2))
        print(f"{name:14s} {p4:12.1f} {p4 - p4_prev:8.1f} {p1:14.1f} {p1 - p1_prev:8.1f}")
        p4_prev, p1_prev = p4, p1
    engine._check(engine._lib.vello_hip_set_debug_flags(engine._h, 0), "flags")
    print("whole frame   ", f"{period(4, None, n):12.1f}", " " * 8, f"{period(1, None, n):14.1f}")
    print("rc", engine.sync())


if __name__ == "__main__":
    main()
