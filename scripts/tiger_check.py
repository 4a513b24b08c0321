import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, "/root/repo")
import numpy as np, torch, vello_amd
from vello_amd import AaConfig
d = np.load("/root/repo/tests/golden/tiger_scene.npz")
packed, layout = d["packed"], vello_amd.Layout(*[int(v) for v in d["layout"]])
eng = vello_amd.Engine(); eng.set_frames_in_flight(4); eng.upload_scene(packed, layout)
ring = [torch.zeros((1024, 1024, 4), dtype=torch.uint8, device="cuda:0") for _ in range(4)]
torch.cuda.synchronize()
for i in range(20): eng.render_resident(1024, 1024, 0xFFFFFFFF, AaConfig.Msaa8, out=ring[i % 4])
eng.sync()
for n in (200, 200, 1000, 4000, 200, 4000):
    t = time.perf_counter()
    for i in range(n): eng.render_resident(1024, 1024, 0xFFFFFFFF, AaConfig.Msaa8, out=ring[i % 4])
    th = time.perf_counter() - t
    eng.sync()
    tt = time.perf_counter() - t
    print(n, "frames: host enqueue %.1f us/frame, total %.1f us/frame -> %.0f frames/s" % (th / n * 1e6, tt / n * 1e6, n / tt))
# with back-pressure (wait for the oldest of 4)
t = time.perf_counter(); n = 2000
for i in range(n):
    eng.render_resident(1024, 1024, 0xFFFFFFFF, AaConfig.Msaa8, out=ring[i % 4])
    if i >= 3: eng.sync_frame(3)
eng.sync(); tt = time.perf_counter() - t
print("with back-pressure (4 in flight):", n / tt)
