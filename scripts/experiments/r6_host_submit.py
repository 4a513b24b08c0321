"""Round 6: is the frames-in-flight number bound by the HOST?  (profiles/r06_host_submit.txt)

rocprofv3's time line shows each lane idle for a third of its time between one frame's k_fine and the next frame's first kernel.
bench.py's loop (FramePipeline) submits frame i, then waits for frame i - 3: the lane that just finished has nothing queued until the
host has woken up and pushed twelve launches.  Measured here, d2, same process:
  (1) host time of one vello_hip_render_resident call (perf_counter around it), frames one at a time and four in flight;
  (2) frames/s of bench.py's loop (depth = lanes = 4);
  (3) frames/s with the SAME four lanes and a deeper queue: the host waits for frame i - (depth - 1), depth 5 ... 12, by an event
      recorded on the lane's stream behind each frame (a lane then holds a running frame and queued ones).
"""
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bench  # noqa: E402
import vello_amd  # noqa: E402



def main():
    wl = bench.Workload(os.environ.get("WORKLOAD", "d2"), 0)
    W, H = wl.width, wl.height
    torch.zeros(1, device="cuda").add_(1)  # (torch's null stream first)
    torch.cuda.synchronize()
    engine = vello_amd.Engine(device=0, capacities=wl.caps)
    engine.upload_scene(wl.packed, wl.layout)
    aa = wl.aa
    lanes = int(os.environ.get("LANES", "4"))
    engine.set_frames_in_flight(lanes)
    ring = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(16)]
    torch.cuda.synchronize()

    def render(slot):
        engine.render_resident(W, H, bench.BASE_COLOR, aa, out=ring[slot])

    for _ in range(12):
        render(0)
        engine.sync_frame(0)
    engine.sync()

    # (1) host time of a submit, one at a time
    ts = []
    for _ in range(50):
        t = time.perf_counter()
        render(0)
        ts.append(time.perf_counter() - t)
        engine.sync_frame(0)
    ts.sort()
    print(f"host time of render_resident, one frame at a time: median {ts[25] * 1e6:.1f} us, p90 {ts[45] * 1e6:.1f} us")

    # (2) bench.py's loop
    def loop_bench(n):
        sub = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            t = time.perf_counter()
            render(i % lanes)
            sub.append(time.perf_counter() - t)
            if i >= lanes - 1:
                engine.sync_frame(lanes - 1)
        engine.sync()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        sub.sort()
        return n / el, sub[len(sub) // 2] * 1e6, sub[len(sub) * 9 // 10] * 1e6

    # (3) deeper queue on the same lanes
    def loop_deep(n, depth):
        evs = [None] * n
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            render(i % len(ring))
            ev = torch.cuda.Event()
            ev.record(torch.cuda.ExternalStream(engine.stream()))
            evs[i] = ev
            if i >= depth - 1:
                evs[i - (depth - 1)].synchronize()
                evs[i - (depth - 1)] = None
        engine.sync()
        torch.cuda.synchronize()
        return n / (time.perf_counter() - t0)

    for rep in range(2):
        f, med, p90 = loop_bench(400)
        print(f"bench loop, {lanes} lanes: {f:.1f} frames/s; host time of a submit median {med:.1f} us, p90 {p90:.1f} us")
        for depth in (lanes, lanes + 1, lanes + 2, 2 * lanes, 3 * lanes):
            if depth > len(ring):
                continue
            print(f"  depth {depth:2d} on {lanes} lanes (events): {loop_deep(400, depth):.1f} frames/s")
    rc = engine.sync()
    print("rc", rc, engine.bump())


if __name__ == "__main__":
    main()
