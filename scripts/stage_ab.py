"""One stage under build variants: d2 and r1mix at 1600^2 MSAA16 -- the stage's time one frame at a time (HIP events) and the
frame rate with 4 frames in flight.    python scripts/stage_ab.py A|<variant> <stage>     (ab_tmp/libvello_hip_<variant>.so)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vello_amd, workloads, bench
import vello_amd._lib as L
which, stage = sys.argv[1], sys.argv[2]
if which != "A":
    L._use_library(os.path.join(ROOT, "ab_tmp", f"libvello_hip_{which}.so"))
from vello_amd import AaConfig
out = []
for name, scene, caps in (("d2", workloads.paris_like_scene_d2, bench.D2_CAPS), ("r1mix", workloads.paris_like_scene, None)):
    p, l = scene().resolve()
    eng = vello_amd.Engine(capacities=caps) if caps else vello_amd.Engine()
    eng.set_auto_grow(True)
    eng.upload_scene(p, l)
    nif = 4
    eng.set_frames_in_flight(nif)
    ring = [torch.zeros((1600, 1600, 4), dtype=torch.uint8, device="cuda:0") for _ in range(nif)]
    torch.cuda.synchronize()
    for i in range(12):
        eng.render_resident(1600, 1600, 0xFFFFFFFF, AaConfig.Msaa16, out=ring[i % nif])
    assert eng.sync() == 0
    res = []
    for rep in range(2):
        t = time.perf_counter()
        n = 100
        for i in range(n):
            eng.render_resident(1600, 1600, 0xFFFFFFFF, AaConfig.Msaa16, out=ring[i % nif])
        assert eng.sync() == 0
        fps = n / (time.perf_counter() - t)
        eng.set_profiling([stage])
        eng.stage_ms()
        for i in range(30):
            eng.render_resident(1600, 1600, 0xFFFFFFFF, AaConfig.Msaa16, out=ring[0]); eng.sync_frame(0)
        ms = eng.stage_ms()[stage]
        eng.set_profiling([])
        res.append("%.0f fps %.1f us" % (fps, 1e3 * ms[0] / ms[1]))
    out.append(name + " " + " / ".join(res))
    del eng
print(which, stage, " | ".join(out))
