"""Frames of one small workload rendered one after the other (enqueue, wait), for rocprofv3 --kernel-trace + scripts/frame_timeline.py.
   python scripts/small_scene_trace.py circle|tiger|stroke_styles|image_sampling fused|unfused [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vello_amd, workloads
from vello_amd import AaConfig

which, mode = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
resolved = None
if which == "circle":
    (packed, layout), w, h, aa = workloads.circle_scene().resolve(), 256, 256, AaConfig.Area
elif which == "tiger":
    d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tiger_scene.npz"))
    packed, layout, w, h, aa = d["packed"], vello_amd.Layout(*[int(v) for v in d["layout"]]), 1024, 1024, AaConfig.Msaa8
elif which == "stroke_styles":
    (packed, layout), w, h, aa = workloads.stroke_styles_scene().resolve(), 256, 256, AaConfig.Msaa16
else:
    s, w, h = workloads.image_sampling_scene()
    resolved = vello_amd.Resolver().resolve(s)
    packed, layout, aa = resolved.packed, resolved.layout, AaConfig.Msaa16
eng = vello_amd.Engine()
eng.set_debug_flags(no_fusion=mode == "unfused")
if resolved is not None:
    eng.upload_resolved(resolved)
else:
    eng.upload_scene(packed, layout)
t = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
for i in range(n):
    eng.render_resident(w, h, 0xFFFFFFFF, aa, out=t)
    eng.sync_frame(0)
