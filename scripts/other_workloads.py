"""Throughput of the engine on the other BASELINE configs and two reference catalogue scenes (not the bench metric).
   python scripts/other_workloads.py   -> one JSON line per workload"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if os.environ.get("VELLO_AB_LIB"):  # same-box A/B: ab_tmp/libvello_hip_<X>.so instead of the in-tree build
    import vello_amd._lib as _L
    _L._use_library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ab_tmp", "libvello_hip_%s.so" % os.environ["VELLO_AB_LIB"]))
import vello_amd, workloads
from vello_amd import AaConfig

def run(name, packed, layout, w, h, aa, resolved=None, nif=4, n=200):
    eng = vello_amd.Engine()
    eng.set_frames_in_flight(nif)
    if resolved is not None:
        eng.upload_resolved(resolved)
    else:
        eng.upload_scene(packed, layout)
    ring = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0") for _ in range(nif)]
    torch.cuda.synchronize()
    for i in range(20):
        eng.render_resident(w, h, 0xFFFFFFFF, aa, out=ring[i % nif])
    assert eng.sync() == 0
    t = time.perf_counter()
    for i in range(n):
        eng.render_resident(w, h, 0xFFFFFFFF, aa, out=ring[i % nif])
    assert eng.sync() == 0
    fps = n / (time.perf_counter() - t)
    t = time.perf_counter()
    for i in range(50):
        eng.render_resident(w, h, 0xFFFFFFFF, aa, out=ring[0]); eng.sync_frame(0)
    lat = (time.perf_counter() - t) / 50 * 1e3
    print(json.dumps({"workload": name, "size": [w, h], "aa": int(aa), "frames_per_s": round(fps, 1), "serial_latency_ms": round(lat, 4),
                      "bump": eng.bump()}), flush=True)

d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tiger_scene.npz"))
run("C2 tiger 1024^2 MSAA8", d["packed"], vello_amd.Layout(*[int(v) for v in d["layout"]]), 1024, 1024, AaConfig.Msaa8)
p, l = workloads.circle_scene().resolve()
run("C1 circle 256^2 area", p, l, 256, 256, AaConfig.Area)
p, l = workloads.mmark_scene().resolve()
run("C4 mmark-50k 2048^2 MSAA16", p, l, 2048, 2048, AaConfig.Msaa16)
s, w, h = workloads.many_draw_objects_scene(); p, l = s.resolve()
run("many_draw_objects (90k circles) 2000x1500 MSAA16", p, l, w, h, AaConfig.Msaa16)
s, w, h = workloads.blend_grid_scene(); r = vello_amd.Resolver().resolve(s)
run("blend_grid (16 mix modes, gradients) 900^2 MSAA16", r.packed, r.layout, w, h, AaConfig.Msaa16, resolved=r)
# (the brush specialisation of fine at stated sizes; per-stage / per-kernel times and k_fine's phases: scripts/brush_prof.py)
s, w, h = workloads.gradient_extend_scene(); r = vello_amd.Resolver().resolve(s)
run("gradient_extend (linear / radial / sweep x extend modes) %dx%d MSAA16" % (w, h), r.packed, r.layout, w, h, AaConfig.Msaa16, resolved=r)
s, w, h = workloads.image_sampling_scene(); r = vello_amd.Resolver().resolve(s)
run("image_sampling (nearest / bilinear) %dx%d MSAA16" % (w, h), r.packed, r.layout, w, h, AaConfig.Msaa16, resolved=r)
