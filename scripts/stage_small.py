"""One stage under build variants on the smaller workloads (tiger 1024^2 MSAA8, mmark-50k 2048^2 MSAA16, image_sampling, blend_grid,
circle): the stage's time one frame at a time (HIP events around it) and the frame one at a time (enqueue + wait).
    python scripts/stage_small.py A|<variant> <stage>     (ab_tmp/libvello_hip_<variant>.so)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vello_amd, workloads
import vello_amd._lib as L
which, stage = sys.argv[1], sys.argv[2]
if which != "A":
    L._use_library(os.path.join(ROOT, "ab_tmp", f"libvello_hip_{which}.so"))
from vello_amd import AaConfig

def cases():
    d = np.load(os.path.join(ROOT, "tests", "golden", "tiger_scene.npz"))
    yield "tiger", d["packed"], vello_amd.Layout(*[int(v) for v in d["layout"]]), None, 1024, 1024, AaConfig.Msaa8
    p, l = workloads.mmark_scene().resolve()
    yield "mmark", p, l, None, 2048, 2048, AaConfig.Msaa16
    s, w, h = workloads.image_sampling_scene(); r = vello_amd.Resolver().resolve(s)
    yield "image_sampling", r.packed, r.layout, r, w, h, AaConfig.Msaa16
    s, w, h = workloads.blend_grid_scene(); r = vello_amd.Resolver().resolve(s)
    yield "blend_grid", r.packed, r.layout, r, w, h, AaConfig.Msaa16
    p, l = workloads.circle_scene().resolve()
    yield "circle", p, l, None, 256, 256, AaConfig.Area

out = []
for name, p, l, r, w, h, aa in cases():
    eng = vello_amd.Engine()
    eng.set_auto_grow(True)
    if r is not None:
        eng.upload_resolved(r)
    else:
        eng.upload_scene(p, l)
    t = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    for i in range(10):
        eng.render_resident(w, h, 0xFFFFFFFF, aa, out=t); eng.sync_frame(0)
    res = []
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(200):
            eng.render_resident(w, h, 0xFFFFFFFF, aa, out=t); eng.sync_frame(0)
        frame = (time.perf_counter() - t0) / 200 * 1e6
        eng.set_profiling([stage]); eng.stage_ms()
        for i in range(50):
            eng.render_resident(w, h, 0xFFFFFFFF, aa, out=t); eng.sync_frame(0)
        ms = eng.stage_ms()[stage]
        eng.set_profiling([])
        res.append("%.1f us (frame %.0f)" % (1e3 * ms[0] / ms[1], frame))
    out.append(name + " " + " / ".join(res))
    del eng
print(which, stage, " | ".join(out))
