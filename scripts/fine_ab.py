"""k_fine under build variants: d2, MSAA16 -- fine's stage time one frame at a time and the frame rate with 4 frames in flight.
   python scripts/fine_ab.py A|<variant>      (ab_tmp/libvello_hip_<variant>.so, see ab_bench.py)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vello_amd, workloads, bench
import vello_amd._lib as L
which = sys.argv[1]
if which != "A":
    L._use_library(os.path.join(ROOT, "ab_tmp", f"libvello_hip_{which}.so"))
from vello_amd import AaConfig
p, l = workloads.paris_like_scene_d2().resolve()
eng = vello_amd.Engine(capacities=bench.D2_CAPS)
eng.upload_scene(p, l)
nif = 4
eng.set_frames_in_flight(nif)
ring = [torch.zeros((1600, 1600, 4), dtype=torch.uint8, device="cuda:0") for _ in range(nif)]
torch.cuda.synchronize()
for i in range(12):
    eng.render_resident(1600, 1600, 0xFFFFFFFF, AaConfig.Msaa16, out=ring[i % nif])
assert eng.sync() == 0
out = []
for rep in range(2):
    t = time.perf_counter()
    n = 120
    for i in range(n):
        eng.render_resident(1600, 1600, 0xFFFFFFFF, AaConfig.Msaa16, out=ring[i % nif])
    assert eng.sync() == 0
    fps = n / (time.perf_counter() - t)
    eng.set_profiling(["fine"])
    eng.stage_ms()
    for i in range(40):
        eng.render_resident(1600, 1600, 0xFFFFFFFF, AaConfig.Msaa16, out=ring[0]); eng.sync_frame(0)
    ms = eng.stage_ms()["fine"]
    eng.set_profiling([])
    out.append("%.0f fps, fine %.1f us" % (fps, 1e3 * ms[0] / ms[1]))
print(which, " | ".join(out))
